#!/bin/bash
# kernel mix of the pinnsf_res step at the bench shape
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/resprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/time_res.py > $O/log.txt 2>&1
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/p
