#!/bin/bash
# round 6: what each part of the two-crew backward costs -- stamps of variant builds that leave work out (results wrong on purpose)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_f5skip; mkdir -p $O
for v in stamps "$@"; do
  echo "== variant $v" | tee -a $O/skip.log
  PIML_LIB=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_f5$v.so timeout 300 python tools/f5_stamps.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -9 | tee -a $O/skip.log
done
