#!/bin/bash
# PMC pass over the one-pass encoder backward only (bench step, fused on): where its wave cycles go
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4pmc; rm -rf $O; mkdir -p $O
export PIML_ENC_FUSED_BWD=1
ARGS="--steps 30 --warmup 5 --cpu-seconds 0 --spinup-ms 0 --secondary 0"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/sq1 -- python3 $R/bench.py $ARGS > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM --output-format csv -d $O/sq2 -- python3 $R/bench.py $ARGS > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
python3 - <<'PY'
import collections, csv, glob, os
O = os.path.join(os.environ['GRAFT_REPO_ROOT'], 'gpurun_out/r4pmc')
for sub in ('sq1', 'sq2'):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{O}/{sub}/*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'enc_bwd' in k:
                agg[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, cs in agg.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f'    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})')
for f in glob.glob(f'{O}/stats/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], r['AverageNs'])
PY
