#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4c7; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_encoder_gpu.py -m gpu -x -q -k "one_pass" 2>&1 | tail -30 > $O/one_pass.log
tail -4 $O/one_pass.log
PIML_LIB=$R/piml_amd/libpiml_hip_stamps.so timeout 300 python tools/f3_stamps.py > $O/stamps.log 2>&1
cat $O/stamps.log
PIML_ENC_FUSED_BWD=1 timeout 600 python bench.py --cpu-seconds 0 --secondary 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
for f in ('bench.json',):
    try:
        d=json.loads(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4c7',f)).read().strip().splitlines()[-1])
        print(f, d['ms_per_step'], [(k['name'],round(k['us'],1)) for k in d['roofline'].get('kernels',[])])
    except Exception as e:
        print(f, 'unreadable', e)
PY
