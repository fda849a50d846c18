"""Where the small torch launches of one fine-tuning step come from.  One EAGER step under torch.profiler: GPU kernels per
autograd node of the backward pass (by time containment in the node's `evaluate_function` range), and per aten operator.
python tools/ft_glue_profile.py [pinnsf_m | pinnsf_bm]"""
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
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        sim.train_batch(data)
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU]
    nodes = [e for e in evs if e.name.startswith('autograd::engine::evaluate_function:')]
    ops_ = [e for e in evs if e.name.startswith('aten::') and e.kernels]
    leaf = [e for e in ops_ if not any(c.name.startswith('aten::') and c.kernels for c in e.cpu_children)]
    per_node = collections.Counter()
    fwd = collections.Counter()
    for e in leaf:
        t = e.time_range.start
        owner = None
        for nd in nodes:
            if nd.time_range.start <= t <= nd.time_range.end and nd.thread == e.thread:
                owner = nd.name.split(': ')[-1]
                break
        if owner:
            per_node[(owner, e.name)] += len(e.kernels)
        else:
            fwd[e.name] += len(e.kernels)
    print('backward, launches per (autograd node, aten operator):')
    for (nd, name), n in sorted(per_node.items(), key=lambda kv: -kv[1]):
        print(f'{n:4d}  {nd:40s} {name}')
    print('forward / optimiser, launches per aten operator:', dict(fwd.most_common(30)))


if __name__ == '__main__':
    main()
