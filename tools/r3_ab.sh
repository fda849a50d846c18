#!/bin/bash
# full GPU suite, then A/B of one environment switch on the default bench step: $1 = VAR (values: unset, 0), $2 = bench args
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3ab; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -30 > $O/tests.log
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', round(b['ms_per_step'],4), b.get('verify_max_rel_err'), [(k['name'][4:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
B="python bench.py --cpu-seconds 0 --secondary 0 $2"
for r in 1 2 3; do
  $B 2>/dev/null | line default >> $O/ab.log
  env $1=0 $B 2>/dev/null | line $1=0 >> $O/ab.log
done
