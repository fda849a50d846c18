#!/bin/bash
# rocprofv3 kernel stats of the default bench command -> gpurun_out/r5_stats/ (kernel_stats.csv + the trace's per-launch timestamps)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_stats; rm -rf $O; mkdir -p $O
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0 $BENCH_EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
f=$(find $O/stats -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv
t=$(find $O/stats -name '*kernel_trace.csv' | head -1)
python3 - "$t" <<'PY' > $O/timeline.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 40 launches: name, duration, gap to the previous end
prev = None
out = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    out.append((r['Kernel_Name'][:60], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
idx = [i for i, o in enumerate(out) if 'relfeat_fwd_kernel' in o[0]]
lo = idx[len(idx) // 2] if len(idx) > 8 else 0          # the middle of the run: the timed region's replays
for o in out[lo:lo + 60]:
    print(f'{o[0]:60s} dur {o[1]:8.2f} us   gap-before {o[2]:8.2f} us')
PY
head -14 $O/kernel_stats.csv | cut -c1-150
cat $O/timeline.txt | head -40
