#!/bin/bash
# kernel mix of one model's forward + backward step at the bench shape: tools/r4_modelprof.sh PINNSF_bottleneck_multitask
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4modelprof; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $R/tools/time_models.py --models ${1:-PINNSF_bottleneck_multitask} > $O/log.txt 2>&1
cp $(ls $O/p/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/p
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:16]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), f'{float(r["AverageNs"])/1e3:8.1f} us', r["Percentage"])
PY
grep "ms/step" $O/log.txt
