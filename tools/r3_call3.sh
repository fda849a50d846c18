#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c3; rm -rf $O; mkdir -p $O
cd $R; python -m pytest tests/test_dropout_gpu.py -q 2>&1 | tail -30 > $O/tests.log; cd /tmp
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train -- python3 $R/bench.py $ARGS --train-mode 1 > $O/train.log 2>&1
python3 $R/bench.py --cpu-seconds 0 --secondary 0 --train-mode 1 > $O/bench_train.json 2> $O/bench_train.err
python3 $R/bench.py --cpu-seconds 0 --secondary 0 > $O/bench_eval.json 2> $O/bench_eval.err
