#!/bin/bash
# MLAPM once-per-pair backward: parity tests, then timings for the split choices
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4mlapm; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_pairwise_gpu.py -m gpu -x -q -k "mlapm" 2>&1 | tail -12
for s in 1 2 4; do PIML_MLAPM_BWD_SPLIT=$s timeout 300 python tools/time_mlapm_bwd.py 2>&1 | grep MLAPM | tee -a $O/times.log; done
