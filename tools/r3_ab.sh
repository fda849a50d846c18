#!/bin/bash
# A/B of one environment switch on the default bench step: $1 = VAR, values 1 (default build behaviour) and 0
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ab; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|Fatal|Error" | tail -3 > $O/tests.log
for v in default 0 default 0; do
  if [ $v = default ]; then python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('default', b['ms_per_step'], [(k['name'][:14], round(k['us'],1)) for k in b['roofline']['kernels']])" >> $O/ab.log
  else env $1=0 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1=0', b['ms_per_step'], [(k['name'][:14], round(k['us'],1)) for k in b['roofline']['kernels']])" >> $O/ab.log; fi
done
