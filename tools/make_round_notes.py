#!/usr/bin/env python3
"""gpurun_out/profile_full/ (tools/profile_full.sh) -> the committed profile files of a round next to those of
tools/make_step_counters.py <tag> (usage: tools/make_round_notes.py r04; the names below with that tag):
   profiles/<tag>_train_mode_kernel_stats.csv        rocprofv3 --stats of `bench.py --train-mode 1` (model.train(), dropout 0.5)
   profiles/<tag>_training_loops_kernel_stats.csv    ... of tools/train_mode_steps.py --models pinnsf_m (HOT LOOP A + C, dropout 0.5)
   profiles/<tag>_training_loops_bm_kernel_stats.csv ... --models pinnsf_bm
   profiles/<tag>_bench_driver_cmd.json              the line of `python3 bench.py --gpus 1 --steps 20 --warmup 5` (+ train mode)
   profiles/<tag>_other_kernels.md                   digest: training loops, models, rollouts, MLAPM / collision kernels, relfeat sizes"""
import csv
import json
import os
import shutil
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else 'r04'

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'profile_full')
dst = os.path.join(ROOT, 'profiles')
for a, b in (('train_step_kernel_stats.csv', f'{TAG}_train_mode_kernel_stats.csv'), ('loops_kernel_stats.csv', f'{TAG}_training_loops_kernel_stats.csv'),
             ('loops_bm_kernel_stats.csv', f'{TAG}_training_loops_bm_kernel_stats.csv'),
             ('finetune_step_pinnsf_m_kernel_stats.csv', f'{TAG}_finetune_step_kernel_stats.csv'),
             ('finetune_step_pinnsf_bm_kernel_stats.csv', f'{TAG}_finetune_step_bm_kernel_stats.csv'),
             ('pinnsf_res_kernel_stats.csv', f'{TAG}_pinnsf_res_kernel_stats.csv')):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
lines = {}
for mode, f in (('eval()', 'bench_driver_cmd.json'), ('train() dropout 0.5', 'bench_driver_cmd_train.json')):
    lines[mode] = json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])
json.dump(lines, open(os.path.join(dst, f'{TAG}_bench_driver_cmd.json'), 'w'), indent=1)


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
with open(os.path.join(dst, f'{TAG}_other_kernels.md'), 'w') as f:
    f.write(f'# {TAG}: training configuration, training loops, models, rollouts, the other kernels (1x MI355X)\n\n')
    f.write('Collected by `tools/profile_full.sh`, digested by `tools/make_round_notes.py`.\n\n')
    f.write('## The driver\'s command line\n\n`python3 bench.py --gpus 1 --steps 20 --warmup 5`: '
            f'**{ev["ms_per_step"]:.4f} ms/step**, value {ev["value"]:.4e} pairs/s, roofline.frac {ev["roofline"]["frac"]:.3f} (eval mode, the default); '
            f'`--train-mode 1` (model.train(), dropout 0.5): **{tr["ms_per_step"]:.4f} ms/step**, frac {tr["roofline"]["frac"]:.3f}.  '
            f'Both lines: `profiles/{TAG}_bench_driver_cmd.json`.\n\n')
    sec = ev.get('secondary') or {}
    f.write('Secondary steps of the eval line (ms/step): ' + ', '.join(f'{k} {v["ms_per_step"]:.4f}' for k, v in sec.items()
                                                                        if isinstance(v, dict) and 'ms_per_step' in v) + '\n\n')
    f.write('Live per-kernel times of that line (`roofline.kernels[].us`): ' +
            ', '.join(f'{k["name"]} {k["us"]:.1f}' for k in ev['roofline']['kernels']) + '\n\n')
    f.write('## The bench step in the reference\'s training configuration (rocprofv3 --stats, `bench.py --train-mode 1`)\n\n')
    f.write(top(f'{TAG}_train_mode_kernel_stats.csv', 10) + '\n\n')
    f.write('## The two training loops at dropout 0.5 (`tools/train_mode_steps.py`)\n\n```\n' + log('train_mode_steps.log', ['step']) + '\n```\n\n')
    f.write('Kernel mix of the `pinnsf_m` loops (pointwise pre-training at 128 / 1024 / 4096 rows + fine-tuning at 4 x 5 x 122 and 4 x 5 x 976):\n\n')
    f.write(top(f'{TAG}_training_loops_kernel_stats.csv', 16) + '\n\n')
    f.write('Kernel mix of the `pinnsf_bm` loops:\n\n' + top(f'{TAG}_training_loops_bm_kernel_stats.csv', 16) + '\n\n')
    def per_step(csvname, marker):
        path = os.path.join(dst, csvname)
        if not os.path.exists(path):
            return None
        rows = list(csv.DictReader(open(path)))
        steps = max([int(r['Calls']) for r in rows if marker in r['Name']] or [0])
        if not steps:
            return None
        mine = sum(int(r['Calls']) for r in rows if 'piml' in r['Name'] or 'dec_fwd' in r['Name'])
        return sum(int(r['Calls']) for r in rows) / steps, mine / steps, steps
    f.write('## Kernels per captured fine-tuning step (rocprofv3 --stats, 4 windows x 5 frames x 122 agents, train mode)\n\n')
    for name, csvname in (('pinnsf_m', f'{TAG}_finetune_step_kernel_stats.csv'), ('pinnsf_bm', f'{TAG}_finetune_step_bm_kernel_stats.csv')):
        ps = per_step(csvname, 'rollout_losses_kernel')
        if ps:
            f.write(f'- `{name}`: {ps[0]:.1f} kernels per step, {ps[1]:.1f} of them this repository\'s ({ps[2]} steps profiled; `profiles/{csvname}`)\n')
    # (round 6) ordered kernel lists of the captured steps, the loop with and without train()'s lookahead, the row decoder A/B
    for name in ('finetune_step_pinnsf_m.txt', 'finetune_step_pinnsf_bm.txt', 'pointwise_step_pinnsf_m.txt', 'pointwise_step_pinnsf_bm.txt'):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f'{TAG}_{name}'))
            last = open(os.path.join(src, name)).read().strip().splitlines()[-1]
            f.write(f'- ordered list (durations, gaps) `profiles/{TAG}_{name}`: {last.lstrip("# ")} (under the profiler)\n')
    if os.path.exists(os.path.join(src, 'time_finetune.log')):
        f.write('\nThe fine-tuning step timed as `train_batch()` in a loop and as `train()` runs it (`tools/time_finetune.py`):\n\n```\n' + log('time_finetune.log') + '\n```\n')
    if os.path.exists(os.path.join(src, 'time_rowdec.log')):
        f.write('\n## The row decoder at 24 576 + 40 960 rows, f32 instruction (`*_big` / `*_lds`) against split bf16 products (`*_x3`) (`tools/time_rowdec.py`)\n\n```\n' + log('time_rowdec.log') + '\n```\n')
    if os.path.exists(os.path.join(dst, f'{TAG}_pinnsf_res_kernel_stats.csv')):
        f.write('\n## Kernel mix of the `pinnsf_res` step at cfg3 (`tools/time_res.py`)\n\n' + top(f'{TAG}_pinnsf_res_kernel_stats.csv', 18) + '\n\n')
    f.write('## Forward + backward step at cfg3 by model (`tools/time_models.py`)\n\n```\n' + log('time_models.log', ['ms/step']) + '\n```\n\n')
    f.write('## Inference rollout (`tools/time_rollout.py`)\n\n```\n' + log('time_rollout.log', ['steps/s']) + '\n```\n\n')
    f.write('## relfeat forward / backward by size (`tools/time_relfeat.py`)\n\n```\n' + log('time_relfeat.log', ['fwd', 'bwd']) + '\n```\n\n')
    f.write('## MLAPM / collision kernels (`tools/time_pairwise.py`)\n\n```\n' + log('time_pairwise.log', ['MLAPM', 'collision']) + '\n```\n')
print(open(os.path.join(dst, f'{TAG}_other_kernels.md')).read()[:1500])
