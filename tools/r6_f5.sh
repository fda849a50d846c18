#!/bin/bash
# round 6: the two-crew backward of the sums path -- bitwise A/B against the one-wave kernel, stamps, bench A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_f5; mkdir -p $O
timeout 900 python -m pytest tests/test_sums_gpu.py -x -q -k "two_crew or float64" 2>&1 | tail -15 | tee $O/test.log
PIML_LIB=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_f5stamps.so timeout 300 python tools/f5_stamps.py 2>&1 | grep -v -i "warn" | tail -12 | tee $O/stamps.log
for rep in 1 2; do for form in 2 1; do
  PIML_ENC_SUMS_BWD=$form timeout 300 python bench.py --cpu-seconds 0 --secondary 0 --verify 0 2>/dev/null > /tmp/ab.json
  python3 - $form <<'PY' | tee -a $O/ab.log
import sys, json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = {x['name']: round(x['us'], 1) for x in d['roofline'].get('kernels', [])}
print('form', sys.argv[1], round(d['ms_per_step'], 5), k)
PY
done; done
