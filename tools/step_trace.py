"""Ordered kernel list of ONE replayed bench step from a rocprofv3 --kernel-trace CSV:
   python tools/step_trace.py <dir with *_kernel_trace.csv> [> out.txt]
Takes the last complete step (from one relfeat_fwd launch to the next) and prints
start offset, duration, stream/queue id and a shortened kernel name."""
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r'at::native::', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n[:150]


def main(d):
    f = sorted(glob.glob(d + '/**/*kernel_trace.csv', recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if 'relfeat_fwd_kernel' in r['Kernel_Name']]
    pairs = [(a, b) for a, b in zip(marks, marks[1:]) if b - a > 10]
    a, b = pairs[-2]
    t0 = int(rows[a]['Start_Timestamp'])
    busy = 0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        busy += e - s
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} q{r.get('Queue_Id', '?'):>3} {short(r['Kernel_Name'])}")
    print(f'# {b - a} kernels, busy {busy / 1e3:.1f} us, span {(int(rows[b]["Start_Timestamp"]) - t0) / 1e3:.1f} us')


if __name__ == '__main__':
    main(sys.argv[1])
