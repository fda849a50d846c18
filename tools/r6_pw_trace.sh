#!/bin/bash
# ordered kernel list (durations, gaps) of one replayed pointwise pre-training step (HOT LOOP A, 128 rows, pinnsf_m; P = dropout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6pw; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/train_mode_steps.py --models ${MODEL:-pinnsf_m} --reps 100 --pointwise-only --dropout ${P:-0.5} > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, re
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r6pw')
f = sorted(glob.glob(O + '/p/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name'].lower()]
a, b = marks[-3], marks[-2]
prev = int(rows[a]['End_Timestamp'])
out = []
for r in rows[a + 1:b + 1]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append(f"{(e - s) / 1e3:6.1f} us  gap {(s - prev) / 1e3:6.1f}  {n[:120]}")
    prev = e
out.append(f'# {b - a} kernels, step {(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3:.1f} us')
open(O + '/step.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
PY
rm -rf $O/p; grep pointwise $O/log.txt
