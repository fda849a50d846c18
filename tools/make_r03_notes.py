#!/usr/bin/env python3
"""gpurun_out/profile_r03/ (tools/profile_r03.sh) -> the committed round-3 profile files next to those of
tools/make_step_counters.py r03:
   profiles/r03_train_mode_kernel_stats.csv        rocprofv3 --stats of `bench.py --train-mode 1` (model.train(), dropout 0.5)
   profiles/r03_training_loops_kernel_stats.csv    ... of tools/train_mode_steps.py --models pinnsf_m (HOT LOOP A + C, dropout 0.5)
   profiles/r03_training_loops_bm_kernel_stats.csv ... --models pinnsf_bm
   profiles/r03_bench_driver_cmd.json              the line of `python3 bench.py --gpus 1 --steps 20 --warmup 5` (+ train mode)
   profiles/r03_other_kernels.md                   digest: training loops, models, rollouts, MLAPM / collision kernels, relfeat sizes"""
import csv
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'profile_r03')
dst = os.path.join(ROOT, 'profiles')
for a, b in (('train_step_kernel_stats.csv', 'r03_train_mode_kernel_stats.csv'), ('loops_kernel_stats.csv', 'r03_training_loops_kernel_stats.csv'),
             ('loops_bm_kernel_stats.csv', 'r03_training_loops_bm_kernel_stats.csv')):
    shutil.copy(os.path.join(src, a), os.path.join(dst, b))
lines = {}
for mode, f in (('eval()', 'bench_driver_cmd.json'), ('train() dropout 0.5', 'bench_driver_cmd_train.json')):
    lines[mode] = json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])
json.dump(lines, open(os.path.join(dst, 'r03_bench_driver_cmd.json'), 'w'), indent=1)


def log(name, keep=None):
    out = []
    for ln in open(os.path.join(src, name)):
        ln = ln.rstrip()
        if not ln or 'amdgpu.ids' in ln or ln.startswith('#Trainable'):
            continue
        if keep is None or any(k in ln for k in keep):
            out.append(ln)
    return '\n'.join(out)


def top(csvname, n=14):
    rows = list(csv.DictReader(open(os.path.join(dst, csvname))))
    out = ['| kernel | calls | avg us | % |', '|---|---|---|---|']
    for r in rows[:n]:
        out.append(f"| `{r['Name'][:100]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['Percentage']):.1f} |")
    lib = [r for r in rows if r['Name'].startswith('Cijk_')]
    out.append(f"\nlibrary GEMM kernels (`Cijk_*`) in this trace: {len(lib)} kinds, {sum(int(r['Calls']) for r in lib)} calls")
    return '\n'.join(out)


ev, tr = lines['eval()'], lines['train() dropout 0.5']
with open(os.path.join(dst, 'r03_other_kernels.md'), 'w') as f:
    f.write('# r03: training configuration, training loops, models, rollouts, the other kernels (1x MI355X)\n\n')
    f.write('Collected by `tools/profile_r03.sh`, digested by `tools/make_r03_notes.py`.\n\n')
    f.write('## The driver\'s command line\n\n`python3 bench.py --gpus 1 --steps 20 --warmup 5`: '
            f'**{ev["ms_per_step"]:.4f} ms/step**, value {ev["value"]:.4e} pairs/s, roofline.frac {ev["roofline"]["frac"]:.3f} (eval mode, the default); '
            f'`--train-mode 1` (model.train(), dropout 0.5): **{tr["ms_per_step"]:.4f} ms/step**, frac {tr["roofline"]["frac"]:.3f}.  '
            'Both lines: `profiles/r03_bench_driver_cmd.json`.\n\n')
    sec = ev.get('secondary') or {}
    f.write('Secondary steps of the eval line (ms/step): ' + ', '.join(f'{k} {v["ms_per_step"]:.4f}' for k, v in sec.items()
                                                                        if isinstance(v, dict) and 'ms_per_step' in v) + '\n\n')
    f.write('Live per-kernel times of that line (`roofline.kernels[].us`): ' +
            ', '.join(f'{k["name"]} {k["us"]:.1f}' for k in ev['roofline']['kernels']) + '\n\n')
    f.write('## The bench step in the reference\'s training configuration (rocprofv3 --stats, `bench.py --train-mode 1`)\n\n')
    f.write(top('r03_train_mode_kernel_stats.csv', 10) + '\n\n')
    f.write('## The two training loops at dropout 0.5 (`tools/train_mode_steps.py`)\n\n```\n' + log('train_mode_steps.log', ['step']) + '\n```\n\n')
    f.write('Kernel mix of the `pinnsf_m` loops (pointwise pre-training at 128 / 1024 / 4096 rows + fine-tuning at 4 x 5 x 122 and 4 x 5 x 976):\n\n')
    f.write(top('r03_training_loops_kernel_stats.csv', 16) + '\n\n')
    f.write('Kernel mix of the `pinnsf_bm` loops:\n\n' + top('r03_training_loops_bm_kernel_stats.csv', 16) + '\n\n')
    f.write('## Forward + backward step at cfg3 by model (`tools/time_models.py`)\n\n```\n' + log('time_models.log', ['ms/step']) + '\n```\n\n')
    f.write('## Inference rollout (`tools/time_rollout.py`)\n\n```\n' + log('time_rollout.log', ['steps/s']) + '\n```\n\n')
    f.write('## relfeat forward / backward by size (`tools/time_relfeat.py`)\n\n```\n' + log('time_relfeat.log', ['fwd', 'bwd']) + '\n```\n\n')
    f.write('## MLAPM / collision kernels (`tools/time_pairwise.py`)\n\n```\n' + log('time_pairwise.log', ['MLAPM', 'collision']) + '\n```\n')
print(open(os.path.join(dst, 'r03_other_kernels.md')).read()[:1500])
