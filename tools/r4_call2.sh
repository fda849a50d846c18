#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c2; rm -rf $O; mkdir -p $O
cd $R
PIML_LIB=$R/piml_amd/libpiml_hip_stamps.so timeout 300 python tools/f3_stamps.py > $O/stamps.log 2>&1
PIML_ENC_FUSED_BWD=1 timeout 300 python tools/time_encoder.py > $O/time_fused.log 2>&1
PIML_ENC_FUSED_BWD=1 PIML_LIB=$R/piml_amd/libpiml_hip_nosb.so timeout 300 python tools/time_encoder.py > $O/time_nosb.log 2>&1
cat $O/stamps.log; grep "forward + backward" $O/time_fused.log $O/time_nosb.log
