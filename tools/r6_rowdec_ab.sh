#!/bin/bash
# usage (GPU box): tools/r6_rowdec_ab.sh <variant> ... -- kernel stats of tools/time_rowdec.py under the shipped library and each variant, alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in base "$@"; do
    if [ $v = base ]; then unset PIML_LIB; else export PIML_LIB=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_$v.so; fi
    echo "== $v"
    bash tools/prof_script.sh tools/time_rowdec.py 2>&1 | grep rowdec
done
done
