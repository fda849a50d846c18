"""Which source line launches each aten kernel of one fine-tuning step (HOT LOOP C at the golden GC batch).  One EAGER step under
torch.profiler with stacks: per (aten operator, innermost frame inside piml_amd/) the number of GPU kernels.
python tools/ft_aten_sources.py [dropout] [pinnsf_m | pinnsf_bm]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from test_simulator_gpu import sim_args, load_data  # noqa: E402


def main():
    from piml_amd.models.simulators import BaseSimulator
    p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
    model = sys.argv[2] if len(sys.argv) > 2 else 'pinnsf_m'
    data = load_data(g, 'train_' + model)
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args(model=model, dropout=p, learning_rate=1e-3, hip_graph=False))
    sim.model.train(True)
    for _ in range(3):
        sim.train_batch(data)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        sim.train_batch(data)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
    ops_ = [e for e in evs if e.name.startswith('aten::') and e.kernels]
    leaf = [e for e in ops_ if not any(c.name.startswith('aten::') and c.kernels for c in e.cpu_children)]
    cnt = collections.Counter()
    for e in leaf:
        where = 'backward / engine'
        for fr in (e.stack or []):
            if 'piml_amd/' in fr and 'torch/' not in fr:
                where = fr.split('piml_amd/')[-1]
                break
        cnt[(e.name, where, tuple(k.name[:50] for k in e.kernels))] += 1
    total = 0
    for (name, where, ks), n in sorted(cnt.items(), key=lambda kv: (kv[0][1], kv[0][0])):
        total += n * len(ks)
        print(f'{n:3d} x {name:28s} {where:70s} {ks}')
    print('aten-launched kernels in the step:', total)


if __name__ == '__main__':
    main()
