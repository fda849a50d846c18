#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rel; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_relfeat_gpu.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/time_relfeat.py 2>&1 | grep fwd > $O/new.log
PIML_LIB=$R/piml_amd/libpiml_hip_sorted.so timeout 300 python tools/time_relfeat.py 2>&1 | grep fwd > $O/sorted.log
echo "--- lane-order drain"; cat $O/new.log; echo "--- sorted drain"; cat $O/sorted.log
