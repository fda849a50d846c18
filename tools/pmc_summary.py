import csv, glob, collections, sys
tag, kern = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f'{k:24s} {sum(v) / len(v):16.0f}')
for f in sorted(glob.glob(f'gpurun_out/pmc_{tag}_1/*/*_kernel_trace.csv')):
    d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if kern in r['Kernel_Name']]
    print('kernel ns (profiled):', d)
