#!/bin/bash
# round 6: bitwise test of the shipped library + stamps of the named variants
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_f5b; mkdir -p $O
timeout 900 python -m pytest tests/test_sums_gpu.py -x -q -k "two_crew or float64" 2>&1 | tail -3 | tee $O/test.log
for v in "$@"; do
  echo "== variant $v" | tee -a $O/skip.log
  PIML_LIB=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_f5$v.so timeout 300 python tools/f5_stamps.py 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -10 | tee -a $O/skip.log
done
PIML_ENC_SUMS_BWD=2 timeout 300 python bench.py --cpu-seconds 0 --secondary 0 --verify 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('bench', round(d['ms_per_step'], 5), {x['name']: round(x['us'], 1) for x in d['roofline'].get('kernels', [])})"
