#!/bin/bash
# ordered kernel list of one replayed fine-tuning step (FT_MODEL=pinnsf_m | pinnsf_bm, 4 x 5 x 122) with durations and gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5ft; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/p -- python3 $R/tools/time_finetune.py ${FT_STEPS:-50} ${FT_MODEL:-pinnsf_m} ${FT_TIMES:-1} > $O/log.txt 2>&1
python3 - <<'PY'
import csv, glob, os, re
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r5ft')
f = sorted(glob.glob(O + '/p/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'adam' in r['Kernel_Name'].lower()]
a, b = marks[-3], marks[-2]
prev = int(rows[a]['End_Timestamp'])
out = []
for r in rows[a + 1:b + 1]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append(f"{(e - s) / 1e3:6.1f} us  gap {(s - prev) / 1e3:6.1f}  {n[:150]}")
    prev = e
out.append(f'# {b - a} kernels, step {(int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3:.1f} us')
open(O + '/step.txt', 'w').write('\n'.join(out) + '\n')
print(out[-1])
PY
rm -rf $O/p; tail -2 $O/log.txt
