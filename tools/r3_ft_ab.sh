#!/bin/bash
# training loops (pinnsf_m, pinnsf_bm) with the current library against piml_amd/exp/lib_prev.so, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ft; rm -rf $O; mkdir -p $O
cp piml_amd/libpiml_hip.so /tmp/lib_orig.so
for r in 1 2; do
  python tools/train_mode_steps.py --models pinnsf_m,pinnsf_bm --reps 200 2>/dev/null | grep -E "128 rows|122 agents" | sed 's/^/new  /' >> $O/ab.log
  cp piml_amd/exp/lib_prev.so piml_amd/libpiml_hip.so
  python tools/train_mode_steps.py --models pinnsf_m,pinnsf_bm --reps 200 2>/dev/null | grep -E "128 rows|122 agents" | sed 's/^/prev /' >> $O/ab.log
  cp /tmp/lib_orig.so piml_amd/libpiml_hip.so
done
