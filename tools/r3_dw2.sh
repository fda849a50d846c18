#!/bin/bash
# layer-split producer/consumer weight gradients: parity tests with it switched on, then the bench step under the settings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3dw2; rm -rf $O; mkdir -p $O
PIML_ENC_DW2=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -40 > $O/tests.log
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', b['ms_per_step'], [(k['name'][:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
for r in 1 2; do
  python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line wide >> $O/ab.log
  PIML_ENC_DW2=1 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line dw2 >> $O/ab.log
  PIML_ENC_DW2=1 PIML_H1_RECOMPUTE=0 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line dw2_h1stored >> $O/ab.log
done
for sh in 300 420 500; do
  PIML_ENC_DW2=1 PIML_DW2_L0_SHARE=$sh python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line share$sh >> $O/ab.log
done
PIML_ENC_DW2=1 python bench.py --cpu-seconds 0 --secondary 0 --train-mode 1 2>/dev/null | line dw2_train >> $O/ab.log
