#!/bin/bash
# A/B of experimental builds of the one-pass encoder backward: bench step with each PIML_LIB, kernel times from the stage trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ab; rm -rf $O; mkdir -p $O
cd $R
export PIML_ENC_FUSED_BWD=1
for v in "$@"; do
  lib=$R/piml_amd/libpiml_hip_$v.so; [ "$v" = base ] && lib=$R/piml_amd/libpiml_hip.so
  [ -z "$NOTEST" ] && PIML_LIB=$lib timeout 600 python -m pytest tests/test_encoder_gpu.py -m gpu -q -k "one_pass" 2>&1 | tail -1 > $O/test_$v.log
  PIML_LIB=$lib timeout 300 python bench.py --cpu-seconds 0 --secondary 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python - "$v" <<'PY'
import json,os,sys
v=sys.argv[1]; O=os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4ab')
try:
    d=json.loads(open(f'{O}/bench_{v}.json').read().strip().splitlines()[-1])
    k={x['name']:round(x['us'],1) for x in d['roofline'].get('kernels',[])}
    print(v, (open(f'{O}/test_{v}.log').read().strip() if os.path.exists(f'{O}/test_{v}.log') else ''), 'step', round(d['ms_per_step']*1e3,1), 'fused', k.get('enc_bwd_fused_x3_kernel', k.get('enc_bwd_dx_x3_kernel')), k)
except Exception as e:
    print(v, 'unreadable', e)
PY
done
