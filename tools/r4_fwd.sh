#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4fwd; rm -rf $O; mkdir -p $O
cd $R
PIML_ENC_FWD_PERS=2 timeout 1500 python -m pytest tests/test_encoder_gpu.py tests/test_dropout_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 2 1 0; do
  PIML_ENC_FWD_PERS=$v timeout 300 python bench.py --cpu-seconds 0 --secondary 0 > $O/bench_$v.json 2> $O/bench_$v.err
  python - $v <<'PY'
import json,os,sys
v=sys.argv[1]
d=json.loads(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],f'gpurun_out/r4fwd/bench_{v}.json')).read().strip().splitlines()[-1])
print('pers', v, 'step', round(d['ms_per_step']*1e3,1), {x['name']:round(x['us'],1) for x in d['roofline'].get('kernels',[])})
PY
done
