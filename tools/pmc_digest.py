#!/usr/bin/env python3
"""Digest of gpurun_out/pmc_<tag>_*/: mean counter values per launch for kernels whose name contains a filter."""
import collections, csv, glob, sys
tag, filt = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not filt or any(x in k for x in filt):
            agg[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f'    {c:28s} {sum(v) / len(v):16.0f}   (n={len(v)})')
