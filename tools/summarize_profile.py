#!/usr/bin/env python3
"""Turn gpurun_out/profile_round/ (tools/profile_round.sh) into the committed profiles/ files:
   profiles/rNN_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of bench.py
   profiles/rNN_relfeat_traffic.json     HBM bytes per launch of the relfeat kernels from the
                                         FETCH_SIZE / WRITE_SIZE PMC passes (separate runs)
   profiles/rNN_summary.md               human-readable digest
usage: tools/summarize_profile.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'profile_round')
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)


def one(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)))[0]


stats = one('stats/*/*_kernel_stats.csv')
shutil.copy(stats, os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))


def pmc(sub, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(one(f'{sub}/*/*_counter_collection.csv'))):
        if r['Counter_Name'] == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


fetch, write = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
hit, miss = pmc('l2', 'TCC_HIT_sum'), pmc('l2', 'TCC_MISS_sum')
traffic = {}
for kern in fetch:
    if 'relfeat' not in kern:
        continue
    f = sum(fetch[kern]) / len(fetch[kern])
    w = sum(write[kern]) / len(write[kern])
    name = 'relfeat_fwd_kernel' if 'fwd' in kern else 'relfeat_bwd_kernel'
    h, m = sum(hit[kern]) / len(hit[kern]), sum(miss[kern]) / len(miss[kern])
    traffic[name] = {
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md (HBM section): on
        # gfx950 FETCH_SIZE counts a wide coalesced read stream at exactly 1/2 of its bytes, so the
        # read side is doubled (an upper bound for the narrow 8-B loads of this kernel).
        'fetch_size_kib': f, 'write_size_kib': w,
        'hbm_bytes_per_launch': f * 1024 * 2 + w * 1024,
        'hbm_bytes_per_launch_uncorrected': (f + w) * 1024,
        'l2_hit_rate': h / max(h + m, 1.0), 'launches_sampled': len(fetch[kern]),
    }
bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
traffic['config'] = bench['config']
json.dump(traffic, open(os.path.join(dst, f'{tag}_relfeat_traffic.json'), 'w'), indent=1)

tot = sum(float(r['TotalDurationNs']) for r in rows)
calls = {r['Name']: int(r['Calls']) for r in rows}
steps = calls[[k for k in calls if 'relfeat_fwd' in k][0]]
with open(os.path.join(dst, f'{tag}_summary.md'), 'w') as f:
    f.write(f'# {tag}: rocprofv3 digest of `python bench.py --steps 50 --warmup 10 --cpu-seconds 0` (1x MI355X)\n\n')
    f.write(f'bench line of the same build (un-profiled run): ms_per_step = {bench["ms_per_step"]:.4f}, '
            f'value = {bench["value"]:.4e} {bench["unit"]}, roofline = {json.dumps(bench["roofline"])}\n\n')
    f.write(f'GPU time summed over kernels: {tot / 1e6 / steps:.3f} ms per step ({steps} steps incl. warm-up/capture)\n\n')
    f.write('| kernel | calls | avg us | % |\n|---|---|---|---|\n')
    for r in rows[:20]:
        f.write(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |\n")
    f.write('\nHIP kernels of this repository:\n\n')
    for r in rows:
        if 'piml::' in r['Name']:
            f.write(f"- `{r['Name'][:70]}`: calls {r['Calls']}, avg {float(r['AverageNs']) / 1e3:.2f} us "
                    f"(min {float(r['MinNs']) / 1e3:.2f}, max {float(r['MaxNs']) / 1e3:.2f})\n")
    f.write('\nHBM traffic (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes):\n\n')
    for k, v in traffic.items():
        if k != 'config':
            f.write(f"- {k}: FETCH_SIZE {v['fetch_size_kib']:.1f} KiB, WRITE_SIZE {v['write_size_kib']:.1f} KiB per launch -> "
                    f"{v['hbm_bytes_per_launch'] / 1e6:.3f} MB (read side x2 per the gfx950 correction), L2 hit rate {v['l2_hit_rate']:.3f}\n")
    # the glue kernels around the MLP's GEMMs: HBM bytes per launch next to their algorithmic bytes would show
    # wasted re-reads; they stream L2 / MALL-resident activations, so the HBM side is mostly the write
    f.write('\nGlue kernels (same PMC passes; averages over all launches of a kernel, i.e. over its layer shapes):\n\n')
    for kern in sorted(fetch):
        if 'piml::' not in kern or 'relfeat' in kern or 'mlapm' in kern:
            continue
        fk, wk = sum(fetch[kern]) / len(fetch[kern]), sum(write[kern]) / max(len(write[kern]), 1)
        h, m = sum(hit[kern]) / max(len(hit[kern]), 1), sum(miss[kern]) / max(len(miss[kern]), 1)
        f.write(f"- `{kern[:60]}`: FETCH_SIZE {fk:.0f} KiB, WRITE_SIZE {wk:.0f} KiB per launch -> "
                f"{(2 * fk + wk) * 1024 / 1e6:.2f} MB, L2 hit rate {h / max(h + m, 1.0):.3f}\n")
print(open(os.path.join(dst, f'{tag}_summary.md')).read())
