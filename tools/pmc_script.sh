#!/bin/bash
# usage (on the GPU box): tools/pmc_script.sh <kernel-substring> <script.py> [args] -- SQ counters of one kernel (own pass, no --stats)
cd /tmp && export TMPDIR=/tmp
K=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_script; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python3 $R/"$@" > $O/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/"$@" > $O/sq2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        for r in csv.DictReader(open(f)):
            if "$K" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
tail -3 $O/sq.log
