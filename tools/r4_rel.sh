#!/bin/bash
# relfeat: bit-exact tests, then split-pass launch vs one wave per row vs grid form
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rel; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_relfeat_gpu.py tests/test_mlpglue_gpu.py -m gpu -x -q -k "relfeat or relative or split_parts" 2>&1 | tail -12
timeout 300 python tools/time_relfeat.py 2>&1 | grep fwd > $O/split.log
PIML_RELFEAT_SPLIT=0 timeout 300 python tools/time_relfeat.py 2>&1 | grep fwd > $O/scan.log
PIML_RELFEAT_SPLIT=0 PIML_RELFEAT_GRID=1 timeout 300 python tools/time_relfeat.py 2>&1 | grep fwd > $O/grid.log
echo "--- split"; cat $O/split.log; echo "--- one wave per row"; cat $O/scan.log; echo "--- grid"; cat $O/grid.log
