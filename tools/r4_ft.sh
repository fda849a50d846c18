#!/bin/bash
# fine-tuning step: tests of the new pieces, kernels per step under rocprofv3, step times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ft; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_relfeat_gpu.py tests/test_losses_gpu.py tests/test_simulator_gpu.py -m gpu -x -q 2>&1 | tail -5
cd /tmp
for MODEL in pinnsf_m pinnsf_bm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p_$MODEL -- python3 $R/tools/train_mode_steps.py --models $MODEL --reps 200 --finetune-only > $O/log_$MODEL.txt 2>&1
  cp $(ls $O/p_$MODEL/*/*kernel_stats.csv | head -1) $O/kernel_stats_$MODEL.csv; rm -rf $O/p_$MODEL
done
cd $R
timeout 600 python tools/train_mode_steps.py --models pinnsf_m pinnsf_bm --reps 200 2>&1 | tail -12
python3 - <<'PY'
import csv, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r4ft')
for m in ('pinnsf_m', 'pinnsf_bm'):
    rows = list(csv.DictReader(open(f'{O}/kernel_stats_{m}.csv')))
    calls = sum(int(r['Calls']) for r in rows)
    mine = sum(int(r['Calls']) for r in rows if 'piml' in r['Name'] or 'dec_fwd' in r['Name'])
    steps = max(int(r['Calls']) for r in rows if 'rollout_losses_kernel' in r['Name'])
    print(m, 'kernels per step', round(calls / steps, 1), 'of them this repository\'s', round(mine / steps, 1), 'steps', steps)
PY
