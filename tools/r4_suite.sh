#!/bin/bash
# full GPU suite + default bench
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4suite; rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/gpu_tests.log
tail -6 $O/gpu_tests.log
timeout 600 python bench.py --cpu-seconds 0 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4suite/bench.json')).read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['roofline']['frac'], [(k['name'],round(k['us'],1)) for k in d['roofline'].get('kernels',[])])
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY
