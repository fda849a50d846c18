#!/bin/bash
# bench step under a list of environment settings ("VAR=val VAR2=val" per argument), alternating, 2 rounds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3env; rm -rf $O; mkdir -p $O
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', round(b['ms_per_step'],4), [(k['name'][4:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
for r in 1 2; do
  for s in "$@"; do
    env $s python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line "$s" >> $O/ab.log
  done
done
