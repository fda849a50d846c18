#!/bin/bash
# A/B of variant libraries on the default bench step (same box, alternating): tools/r4_ab2.sh base rot ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip.so
  PIML_LIB=$lib timeout 300 python bench.py --cpu-seconds 0 --secondary 0 --verify 0 2>/dev/null > /tmp/ab.json
  python3 - $v <<'PY'
import sys, json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = {x['name']: round(x['us'], 1) for x in d['roofline'].get('kernels', [])}
print(sys.argv[1].ljust(8), round(d['ms_per_step'], 5), k.get("enc_fwd_x3_kernel"), k.get("dec_fwd_head_kernel"), k.get("enc_bwd_fused_x3_kernel"), d.get('verified_max_rel_err'))
PY
done
done
