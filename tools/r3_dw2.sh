#!/bin/bash
# layer-split weight gradients: parity tests first, then the bench step under the three settings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3dw2; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -40 > $O/tests.log
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', b['ms_per_step'], [(k['name'][:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
for r in 1 2; do
  python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line default >> $O/ab.log
  PIML_H1_RECOMPUTE=0 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line h1stored >> $O/ab.log
  PIML_ENC_DW2=0 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line dw2off >> $O/ab.log
done
for sh in 300 420 480; do
  PIML_DW2_L0_SHARE=$sh python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line share$sh >> $O/ab.log
done
