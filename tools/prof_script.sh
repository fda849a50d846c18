#!/bin/bash
# usage (on the GPU box): tools/prof_script.sh <script.py> [args] -- rocprofv3 kernel stats of one python script, top kernels printed
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_script; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/"$@" > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:${TOP:-16}]:
        print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"]) / 1e3:9.2f} us')
PY
