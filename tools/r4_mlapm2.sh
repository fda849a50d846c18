#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
for m in 256; do for s in 2 4; do PIML_MLAPM_BWD_SYS_MIN=$m PIML_MLAPM_BWD_SPLIT=$s timeout 300 python tools/time_mlapm_bwd.py 512 1024 1536 2>&1 | grep MLAPM; done; done
timeout 300 python tools/time_mlapm_bwd.py 2048 3000 4096 2>&1 | grep MLAPM
timeout 300 python tools/time_pairwise.py 2>&1 | grep MLAPM
