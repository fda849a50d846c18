#!/bin/bash
# usage (on the GPU box): tools/hbm_script.sh <script.py> [args] -- HBM bytes per launch of every kernel of a python script: FETCH_SIZE and
# WRITE_SIZE in separate --pmc passes (gfx950: FETCH_SIZE x 2, MI355X_MICROARCH.md), next to the kernels' durations of a third pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/hbm_script; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/"$@" > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/"$@" > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/"$@" > $O/stats.log 2>&1
python3 - <<PY
import csv, glob, collections
val = {}
for d, c in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[r["Kernel_Name"]] += float(r["Counter_Value"]); n[r["Kernel_Name"]] += 1
    val[c] = {k: v / n[k] for k, v in acc.items()}
dur = {}
for f in glob.glob("$O/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"]] = float(r["AverageNs"]) / 1e3
print("| kernel | avg us | HBM MB / launch (FETCH_SIZE x 2 + WRITE_SIZE, KB units) | GB/s |")
print("|---|---|---|---|")
for k, us in sorted(dur.items(), key=lambda kv: -kv[1])[:${TOP:-14}]:
    mb = (2 * val["FETCH_SIZE"].get(k, 0.0) + val["WRITE_SIZE"].get(k, 0.0)) * 1024 / 1e6
    print(f"| \`{k[:70]}\` | {us:.1f} | {mb:.1f} | {mb / us * 1e3:.0f} |")
PY
