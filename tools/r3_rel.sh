#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3rel; mkdir -p $O
python -m pytest tests/test_relfeat_gpu.py tests/test_mlpglue_gpu.py tests/test_simulator_gpu.py tests/test_graph_gpu.py -q 2>&1 | grep -E "passed|failed|Error" > $O/tests.log
for res in 1 0; do for w in 16 8; do
  echo "== RESIDENT=$res WAVES=$w" >> $O/time.log
  PIML_RELFEAT_RESIDENT=$res PIML_RELFEAT_WAVES=$w python tools/time_relfeat.py 2>&1 | grep fwd >> $O/time.log
done; done
echo "== default" >> $O/time.log
python tools/time_relfeat.py 2>&1 | grep fwd >> $O/time.log
