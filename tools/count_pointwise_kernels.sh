#!/bin/bash
# kernels per pointwise pre-training step (HOT LOOP A, 128 rows, pinnsf_m, train mode): rocprofv3 kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pwcount; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/train_mode_steps.py --models ${MODEL:-pinnsf_m} --reps 300 --pointwise-only > $O/log.txt 2>&1
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/p
