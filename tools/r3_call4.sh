#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3c4; rm -rf $O; mkdir -p $O
python bench.py --cpu-seconds 0 > $O/bench.json 2> $O/bench.err
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > $O/gpu_tests.log
