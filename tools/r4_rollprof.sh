#!/bin/bash
# kernel mix of the inference rollout: tools/r4_rollprof.sh [N,M] (default the bench scene); PIML_POOLED_INFERENCE=0 for the message path
cd /tmp && export TMPDIR=/tmp
SZ=${1:-4096,2000}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4rollprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/time_rollout.py --sizes $SZ --no-mlapm $2 $3 > $O/log.txt 2>&1
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/p
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:9]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:8.1f} us', r["Percentage"])
PY
grep "steps/s" $O/log.txt
