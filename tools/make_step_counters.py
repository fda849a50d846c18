#!/usr/bin/env python3
"""gpurun_out/profile_round/ (tools/profile_round.sh) -> the committed profile files of a round:
   profiles/<tag>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of bench.py
   profiles/<tag>_step_counters.json       per-kernel time, HBM bytes (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes,
                                           gfx950 correction of MI355X_MICROARCH.md), SIMD / matrix-pipe occupancy;
                                           read by bench.py for `roofline.kernels` / `roofline.traffic`
   profiles/<tag>_summary.md               human-readable digest
usage: tools/make_step_counters.py r02"""
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
N, M, KP, KO = 4096, 2000, 6, 10
ROWS = N * (KP + KO)
F32_MFMA_PEAK = 157.3e12
BF16_MFMA_PEAK = 2.5e15      # dense (MI355X_MICROARCH.md)
HBM_PEAK = 8.0e12
SIMDS = 1024


def latest(pattern):
    return sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)[-1]


stats = latest('stats/*/*_kernel_stats.csv')
shutil.copy(stats, os.path.join(dst, f'{tag}_bench_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats)))


def pmc(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(latest(f'{sub}/*/*_counter_collection.csv'))):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}


fetch, write, sq = pmc('fetch'), pmc('write'), pmc('sq')
bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])

# kernels of the captured step (everything bench.py launches per step; MLAPM = secondary figures, excluded)
STEP = ('enc_', 'dec_', 'head_', 'relfeat_', 'self_features', 'pinnsf_')
FLOPS = {   # algorithmic FLOPs per launch at cfg3 (2 x MACs), encoder: both branches
    'enc_fwd_kernel': 2 * ROWS * (6 * 128 + 2 * 128 * 128),
    'enc_bwd_dx_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    'enc_bwd_dw_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    # split-product forms: the same algorithmic f32 FLOPs; executed bf16 FLOPs = 6 x those of the two 128 x 128 layers
    'enc_fwd_x3_kernel': 2 * ROWS * (6 * 128 + 2 * 128 * 128),
    'enc_bwd_dx_x3_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    'enc_bwd_dw_x3_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    'enc_bwd_dw_x3w_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    'enc_bwd_dw2_x3_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128),      # (+ the recomputed layer 1: f32 instructions, not counted)
    # round 4, one-pass backward: the dX chain (2 layers) + dW3 / dW2 / dW1 (+ the recomputed layers 1 and 2, not counted)
    'enc_bwd_fused_x3_kernel': 2 * ROWS * (2 * 128 * 128 + 6 * 128) + 2 * ROWS * (2 * 128 * 128 + 6 * 128),
    'dec_fwd_kernel': 2 * 2 * N * (128 * 64 + 64 * 64 + 64 * 2),
    'dec_bwd_dx_kernel': 2 * 2 * N * (128 * 64 + 64 * 64 + 64 * 2),
    'dec_bwd_dw_kernel': 2 * 2 * N * (128 * 64 + 64 * 64 + 64 * 2),
    'head_fwd_kernel': 2 * N * KP * (128 * 64 + 64),
    # decoder tails of both branches + the collision head on the pedestrian rows, one launch
    'dec_fwd_head_kernel': 2 * 2 * N * (128 * 64 + 64 * 64 + 64 * 2) + 2 * N * KP * (128 * 64 + 64),
    # round 5, training on the agents' sums of h2 (PIML_POOL_TRAIN): the forward stops after layer 2, the backward has neither the
    # W3^T chain layer nor dW3 (chain layer B + dW2 + dW1), decoder tails on the sums + the head on h2 rows (folded first layers)
    'enc_fwd_sum_x3_kernel': 2 * ROWS * (6 * 128 + 128 * 128),
    'dec_fwd_head_sum_kernel': 2 * 2 * N * (128 * 64 + 64 * 64 + 64 * 2) + 2 * N * KP * (128 * 64 + 64),
}
SUMS_BWD_FLOPS = 2 * ROWS * (2 * 128 * 128 + 2 * 6 * 128)      # enc_bwd_fused_x3_kernel<..., SUMS = true>
FLOPS['enc_bwd_sums2_kernel'] = SUMS_BWD_FLOPS                  # round 6: the same work as two crews of four waves (encoder_bwd5.hip)


def short(name):
    n = name.split('(')[0]
    n = n.replace('void ', '').replace('piml::', '')
    return n.split('<')[0]


steps_profiled = None
for r in rows:
    if 'enc_fwd_kernel' in r['Name'] or 'enc_fwd_x3_kernel' in r['Name'] or 'enc_fwd_sum_x3_kernel' in r['Name']:
        steps_profiled = int(r['Calls'])
kernels, step_hbm, step_us = [], 0.0, 0.0
for r in rows:
    name = r['Name']
    if not any(s in name for s in STEP):
        continue
    key = short(name)
    us = float(r['AverageNs']) / 1e3
    per_step = int(r['Calls']) / steps_profiled
    hbm = 2 * fetch.get(name, {}).get('FETCH_SIZE', 0.0) * 1024 + write.get(name, {}).get('WRITE_SIZE', 0.0) * 1024
    c = sq.get(name, {})
    e = {'name': key, 'us': round(us, 2), 'launches_per_step': round(per_step, 2), 'hbm_bytes': round(hbm)}
    sums_bwd = (key == 'enc_bwd_fused_x3_kernel' and name.rstrip().endswith('true>(piml::F3Args)')) or key == 'enc_bwd_sums2_kernel'
    if key in FLOPS and ('_x3_' in key or key == 'enc_bwd_sums2_kernel'):
        # f32 arithmetic carried by six bf16 matrix instructions per k-block: priced against BOTH ceilings, the larger
        # fraction names the bound.  Executed bf16 FLOPs = 6 x the 128 x 128 layers' (the K <= 8 layer stays f32 / VALU):
        # two per row in the message forward and in the sums backward, four in the message backward, one in the sums forward.
        layers = 1 if key == 'enc_fwd_sum_x3_kernel' else (2 if (sums_bwd or 'fused' not in key) else 4)
        bf16_flops = 6 * 2 * ROWS * 128 * 128 * layers
        if sums_bwd:
            FLOPS[key] = SUMS_BWD_FLOPS
        mfma_frac = bf16_flops / (us * 1e-6) / BF16_MFMA_PEAK
        hbm_frac = hbm / (us * 1e-6) / HBM_PEAK
        e.update(bound='hbm' if hbm_frac >= mfma_frac else 'mfma', frac=round(max(hbm_frac, mfma_frac), 3),
                 hbm_frac=round(hbm_frac, 3), achieved_gbs=round(hbm / us / 1e3, 1),
                 mfma_frac=round(mfma_frac, 3), executed_bf16_tflops=round(bf16_flops / us / 1e6, 1), peak_bf16_tflops=2500.0,
                 flops=FLOPS[key], f32_equivalent_tflops=round(FLOPS[key] / us / 1e6, 1), executed_bf16_flops=bf16_flops)
        if c.get('SQ_BUSY_CU_CYCLES'):
            e['mfma_pipe_busy_of_cu_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']), 3)
    elif key in FLOPS:
        e.update(bound='mfma', flops=FLOPS[key], achieved_tflops=round(FLOPS[key] / us / 1e6, 1),
                 frac=round(FLOPS[key] / (us * 1e-6) / F32_MFMA_PEAK, 3), peak_tflops=157.3)
        if c.get('SQ_BUSY_CU_CYCLES'):
            e['mfma_pipe_busy_of_cu_busy'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * c['SQ_BUSY_CU_CYCLES']), 3)
    elif c.get('SQ_BUSY_CU_CYCLES'):
        # SQ_ACTIVE_INST_VALU counts quad-cycles; 4 SIMDs per CU.  A kernel that moves its bytes at more than half of the HBM
        # peak is bandwidth-bound whatever its issue rate (the slot sums inside relfeat_bwd_reduce_kernel)
        hbm_frac = hbm / (us * 1e-6) / HBM_PEAK
        valu = round(4 * c['SQ_ACTIVE_INST_VALU'] / (4 * c['SQ_BUSY_CU_CYCLES']), 3)
        if hbm_frac >= 0.5:
            e.update(bound='hbm', frac=round(hbm_frac, 3), hbm_frac=round(hbm_frac, 3), achieved_gbs=round(hbm / us / 1e3, 1), valu_busy_frac=valu)
        else:
            e.update(bound='valu' if 'relfeat_fwd' in key else 'latency', frac=valu, hbm_frac=round(hbm_frac, 3))
    kernels.append(e)
    step_hbm += hbm * per_step
    step_us += us * per_step
# MLAPM (secondary figures): VALU-issue share of the SIMD cycles from its own PMC pass
mlapm = None
try:
    mp = pmc('mlapm')
    mlapm = {}
    for kname, c in mp.items():
        for tagk in ('fwd', 'bwd'):
            # backward: the once-per-pair kernel (round 4: mlapm_bwd_sys_kernel), or the two-role kernel it replaced
            if (f'mlapm_{tagk}_kernel' in kname or (tagk == 'bwd' and 'mlapm_bwd_sys_kernel' in kname)) and c.get('SQ_BUSY_CU_CYCLES'):
                mlapm[f'{tagk}_valu_busy_frac'] = round(4 * c['SQ_ACTIVE_INST_VALU'] / (4 * c['SQ_BUSY_CU_CYCLES']), 3)
                mlapm[f'{tagk}_valu_insts_per_wave'] = round(c.get('SQ_INSTS_VALU', 0.0) / max(c.get('SQ_WAVES', 1.0), 1.0), 1)
except (IndexError, FileNotFoundError):
    pass
kernels.sort(key=lambda e: -e['us'] * e['launches_per_step'])
rel = next(e for e in kernels if e['name'] == 'relfeat_fwd_kernel')
out = {
    'source': 'rocprofv3 on an MI355X, tools/profile_round.sh; bench.py --steps 50 --warmup 10 --cpu-seconds 0',
    'config': {'agents_total': N, 'obstacle_points': M},
    'bench_line': {k: bench[k] for k in ('value', 'ms_per_step')},
    'step_hbm_bytes': round(step_hbm), 'step_kernel_us_sum': round(step_us, 1),
    'relfeat_fwd_kernel': {'hbm_bytes_per_launch': rel['hbm_bytes'], 'valu_busy_frac': rel.get('frac')},
    'mlapm': mlapm,
    'other_kernels': [e for e in kernels if e['name'] != 'relfeat_fwd_kernel'][:6],
    'all_step_kernels': kernels,
}
json.dump(out, open(os.path.join(dst, f'{tag}_step_counters.json'), 'w'), indent=1)
with open(os.path.join(dst, f'{tag}_summary.md'), 'w') as f:
    f.write(f'# {tag}: rocprofv3 digest of `python bench.py --steps 50 --warmup 10 --cpu-seconds 0` (1x MI355X)\n\n')
    f.write(f'bench line of the same build (un-profiled run): ms_per_step = {bench["ms_per_step"]:.4f}, value = {bench["value"]:.4e} pairs/s, '
            f'roofline.frac = {bench["roofline"]["frac"]:.3f}\n\n')
    f.write(f'Kernels of the captured step: {step_us:.1f} us of GPU time per step under the profiler, {step_hbm / 1e6:.1f} MB of HBM traffic per step '
            f'(FETCH_SIZE x 2 + WRITE_SIZE, separate PMC passes).\n\n')
    f.write('| kernel | launches/step | avg us | bound | fraction of the bounding unit | HBM MB/launch |\n|---|---|---|---|---|---|\n')
    for e in kernels:
        fr = e.get('frac')
        what = {'mfma': f'{e.get("achieved_tflops")} TF/s = {fr} of 157.3 TF f32 MFMA', 'valu': f'{fr} of the SIMD issue cycles (VALU)',
                'hbm': f'HBM {e.get("achieved_gbs")} GB/s = {fr} of 8 TB/s',
                'latency': f'VALU {fr}, HBM {e.get("hbm_frac")} (launch / latency bound)'}.get(e.get('bound'), '')
        if '_x3_' in e['name'] or e['name'] == 'enc_bwd_sums2_kernel':
            what = (f'HBM {e["achieved_gbs"]} GB/s = {e["hbm_frac"]} of 8 TB/s; bf16 matrix pipe {e["executed_bf16_tflops"]} TF/s executed = '
                    f'{e["mfma_frac"]} of 2.5 PF ({e["f32_equivalent_tflops"]} TF/s of f32 work)')
        f.write(f'| `{e["name"]}` | {e["launches_per_step"]} | {e["us"]} | {e.get("bound", "")} | {what} | {e["hbm_bytes"] / 1e6:.2f} |\n')
    f.write('\nAll kernels (rocprofv3 --stats):\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n')
    for r in rows[:25]:
        f.write(f'| `{r["Name"][:90]}` | {r["Calls"]} | {float(r["AverageNs"]) / 1e3:.1f} | {float(r["Percentage"]):.1f} |\n')
print(json.dumps(out['other_kernels'][:3], indent=1))
