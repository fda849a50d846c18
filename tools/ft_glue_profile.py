"""Where the small torch launches of one fine-tuning step come from: one EAGER forward (rollout + losses) under a
TorchDispatchMode that records, for every aten operator that launches on the GPU, the innermost piml_amd source line (the
backward's operators run on autograd's device thread and are not seen here).   python tools/ft_glue_profile.py [pinnsf_m]"""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from test_simulator_gpu import sim_args, load_data  # noqa: E402

VIEWS = ('view', 'reshape', 'slice', 'select', 'unsqueeze', 'squeeze', 'expand', 'transpose', 'permute', 'alias', 'detach',
         'as_strided', 't.default', 'unbind', 'split', 'empty', 'is_', 'size', 'stride', '_unsafe_view', 'lift', 'narrow')


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.sites = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            site = 'other'
            for fr in reversed(traceback.extract_stack()):
                if 'piml_amd' in fr.filename and 'site-packages' not in fr.filename:
                    site = f'{os.path.relpath(fr.filename, ROOT)}:{fr.lineno}'
                    break
            self.sites[(site, name)] += 1
        return func(*args, **(kwargs or {}))


def main():
    from piml_amd.models.simulators import BaseSimulator
    model = sys.argv[1] if len(sys.argv) > 1 else 'pinnsf_m'
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
    data = load_data(g, 'train_' + ('pinnsf_m' if model == 'pinnsf_m' else 'pinnsf_bm'))
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args(model=model, dropout=0.5, learning_rate=1e-3, hip_graph=False))
    sim.model.train(True)
    for _ in range(2):
        sim.train_batch(data)
    torch.cuda.synchronize()
    with Log() as log:
        out, aux = sim._training_rollout(data)
    total = sum(log.sites.values())
    print(f'{total} launching aten operators in the forward of one step')
    for (site, name), n in sorted(log.sites.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f'{n:4d}  {name:36s} {site}')


if __name__ == '__main__':
    main()
