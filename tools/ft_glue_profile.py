"""Where the small torch launches of one fine-tuning step come from: one EAGER step under torch.profiler with Python stacks,
GPU-launching aten operators grouped by the innermost piml_amd source line.   python tools/ft_glue_profile.py [pinnsf_m]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402
from test_simulator_gpu import sim_args, load_data  # noqa: E402


def main():
    from piml_amd.models.simulators import BaseSimulator
    model = sys.argv[1] if len(sys.argv) > 1 else 'pinnsf_m'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
    data = load_data(g, 'train_' + ('pinnsf_m' if model == 'pinnsf_m' else 'pinnsf_bm'))
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args(model=model, dropout=0.5, learning_rate=1e-3, hip_graph=False))
    sim.model.train(True)
    for _ in range(3):
        sim.train_batch(data)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        sim.train_batch(data)
        torch.cuda.synchronize()
    by_site = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU or not ev.name.startswith('aten::'):
            continue
        # operators that launch: those with a kernel among their direct children
        if not ev.kernels or any(c.name.startswith('aten::') and c.kernels for c in ev.cpu_children):
            continue
        site = 'autograd / other'
        for fr in ev.stack:
            if 'piml_amd' in fr and 'site-packages' not in fr:
                site = fr.split('piml_amd/')[-1]
                break
        by_site[(site, ev.name)] += len(ev.kernels)
        by_op[ev.name] += len(ev.kernels)
    print('launches by aten operator:', dict(by_op.most_common(20)))
    for (site, name), n in by_site.most_common(60):
        print(f'{n:4d}  {name:28s} {site}')


if __name__ == '__main__':
    main()
