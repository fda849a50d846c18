#!/bin/bash
# round 6: A/B of variant libraries on bench legs (same box, alternating): tools/r6_ab.sh "<bench flags>" base v1 v2 ...
cd $GRAFT_REPO_ROOT
FLAGS="$1"; shift
for rep in 1 2; do
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip_$v.so; [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/piml_amd/libpiml_hip.so
  PIML_LIB=$lib timeout 300 python bench.py --cpu-seconds 0 --secondary 0 --verify 0 $FLAGS 2>/dev/null > /tmp/ab.json
  python3 - $v <<'PY'
import sys, json
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
k = {x['name'].replace('_kernel', ''): round(x['us'], 1) for x in d['roofline'].get('kernels', [])}
print(sys.argv[1].ljust(8), round(d['ms_per_step'], 5), k)
PY
done
done
