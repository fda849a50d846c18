#!/bin/bash
# kernel stats of the bench step in eval and train mode
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c2; rm -rf $O; mkdir -p $O
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eval -- python3 $R/bench.py $ARGS > $O/eval.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py $ARGS --train-mode 1 > $O/train.log 2>&1
