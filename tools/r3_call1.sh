#!/bin/bash
# round 3, call 1: full GPU suite, default bench, train-mode steps plain and under rocprofv3 --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c1; rm -rf $O; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err
python tools/train_mode_steps.py --with-eval > $O/train_mode.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_stats -- python3 $R/tools/train_mode_steps.py --models pinnsf_m --reps 20 > $O/train_stats.log 2>&1
tail -5 $O/gpu_tests.log
