"""Ordered kernel list of ONE replayed fine-tuning step from a rocprofv3 --kernel-trace CSV of
tools/time_finetune.py (development aid):  python tools/ft_trace.py <dir> <kernels per step marker count>"""
import csv, glob, re, sys
d = sys.argv[1]
f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'Adam' in r['Kernel_Name'] or 'adam' in r['Kernel_Name']]
if len(marks) < 3:
    marks = [i for i, r in enumerate(rows) if 'train_step_fwd' in r['Kernel_Name']][::5]
a, b = marks[-3], marks[-2]
for r in rows[a + 1:b + 1]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    print(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:6.1f} {n[:140]}")
print('#', b - a, 'kernels')
