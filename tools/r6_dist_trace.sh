#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6dist; rm -rf $O; mkdir -p $O
for EX in p2p bucket; do
rocprofv3 --kernel-trace --output-format csv -d $O/p_$EX -- python3 $R/bench.py --gpus 1 --force-dist 1 --exchange $EX --steps 30 --warmup 5 --cpu-seconds 0 --secondary 0 --spinup-ms 0 --verify 0 --exchange-compare 0 > $O/log_$EX.txt 2>&1
python3 - $EX <<'PY'
import csv, glob, os, re, sys
ex = sys.argv[1]
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r6dist')
f = sorted(glob.glob(O + f'/p_{ex}/**/*kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'enc_bwd_sums2_kernel' in r['Kernel_Name']]
m = marks[len(marks) // 2]                      # a replay in the middle of the timed region: from the step's first launch ...
firsts = [i for i, r in enumerate(rows) if 'relfeat_fwd_kernel' in r['Kernel_Name'] or 'p2p_allgather' in r['Kernel_Name']]
a = max(i for i in firsts if i < m)
while a - 1 >= 0 and ('p2p_allgather' in rows[a - 1]['Kernel_Name'] or 'copyBuffer' in rows[a - 1]['Kernel_Name']):
    a -= 1
b = min(i for i in marks if i > m)
b = max(i for i in firsts if i < b)
while b - 1 >= 0 and ('p2p_allgather' in rows[b - 1]['Kernel_Name'] or 'copyBuffer' in rows[b - 1]['Kernel_Name']):
    b -= 1
prev = int(rows[a - 1]['End_Timestamp'])
print('==', ex)
for r in rows[a:b]:
    n = re.sub(r'at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(e - s) / 1e3:6.1f} us  gap {(s - prev) / 1e3:6.1f}  {n[:110]}")
    prev = e
print(f'# {b - a} kernels, step {(int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3:.1f} us')
PY
grep ms_per_step $O/log_$EX.txt | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
done
