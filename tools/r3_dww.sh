#!/bin/bash
# wide staging loads in the slab weight-gradient kernel: parity tests first, then the bench step under the settings
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3dww; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -40 > $O/tests.log
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', b['ms_per_step'], [(k['name'][:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
for r in 1 2; do
  python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line wide >> $O/ab.log
  PIML_ENC_DW_WIDE=0 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line narrow >> $O/ab.log
  PIML_ENC_DW2=1 python bench.py --cpu-seconds 0 --secondary 0 2>/dev/null | line dw2 >> $O/ab.log
done
python bench.py --cpu-seconds 0 --secondary 0 --train-mode 1 2>/dev/null | line wide_train >> $O/ab.log
