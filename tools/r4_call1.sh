#!/bin/bash
# round 4, call 1: the one-pass encoder backward -- parity tests, encoder timing with and without it, default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c1; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "one_pass" 2>&1 | tail -40 > $O/one_pass.log
tail -5 $O/one_pass.log
timeout 300 python tools/time_encoder.py > $O/time_fused.log 2>&1
PIML_ENC_FUSED_BWD=0 timeout 300 python tools/time_encoder.py > $O/time_two.log 2>&1
timeout 600 python bench.py --cpu-seconds 0 > $O/bench.json 2> $O/bench.err
PIML_ENC_FUSED_BWD=0 timeout 600 python bench.py --cpu-seconds 0 > $O/bench_two.json 2> $O/bench_two.err
cat $O/time_fused.log $O/time_two.log
python - <<'PY'
import json,os
for f in ('bench.json','bench_two.json'):
    try:
        d=json.loads(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4c1',f)).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], [(k['name'],k['us']) for k in d['roofline'].get('kernels',[])])
    except Exception as e:
        print(f, 'unreadable', e)
PY
