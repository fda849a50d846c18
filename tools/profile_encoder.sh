#!/bin/bash
# Run on the GPU box: rocprofv3 kernel stats of tools/time_encoder.py (the fused encoder kernels in isolation).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_encoder; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/time_encoder.py > $O/stats.log 2>&1
f=$(ls $O/stats/*/*_kernel_stats.csv | head -1)
head -12 "$f" | cut -c1-220
