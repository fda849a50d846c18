#!/bin/bash
# quick GPU check: the tests named in $1 (a pytest -k / path expression), then the default bench line digested
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4q; rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python -m pytest $1 -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
tail -15 $O/tests.log
timeout 600 python bench.py --cpu-seconds 0 ${2:-} > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ['GRAFT_REPO_ROOT'],'gpurun_out/r4q/bench.json')).read().strip().splitlines()[-1])
print('step', d['ms_per_step'], d['roofline']['frac'], 'verified', d.get('verified_max_rel_err'))
print([(k['name'],round(k['us'],1)) for k in d['roofline'].get('kernels',[])])
print({k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d.get('secondary',{}).items()})
PY
