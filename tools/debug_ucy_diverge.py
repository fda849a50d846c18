"""UCY pre-training twice from the same initial weights -- fused-kernel path and library-GEMM path, each taking its own
optimiser steps -- and the distance between the two weight trajectories, batch by batch (linear growth = a bias,
exponential = chaos)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
from conftest import GOLDEN, golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
from piml_amd.models import simulators as SIM
import piml_amd.models.model as MODEL

case = 'ucy'
g = golden('mainflow_' + case)
DATA = os.path.join(GOLDEN, 'data')
argv = T.COMMON + T.CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                                   '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml'), '--epochs', '1']
init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')}
orig = SIM.BaseSimulator.train_batch
snaps = {}
cur = [None]


def hook(self, batch_data):
    out = orig(self, batch_data)
    if not (hasattr(batch_data, 'mask_p_pred') and hasattr(batch_data, 'waypoints')):
        snaps[cur[0]].append({k: v.detach().double().cpu().clone() for k, v in self.model.state_dict().items()})
    return out


SIM.BaseSimulator.train_batch = hook
for fused in (True, False):
    MODEL.FUSED_ENCODER = MODEL.FUSED_NETWORK = fused
    cur[0] = fused
    snaps[fused] = []
    try:
        MAIN.main(argv, init_state=init)
    except Exception as ex:   # noqa
        print('main ended with', type(ex).__name__, ex)
n = min(len(snaps[True]), len(snaps[False]))
print('pointwise batches', n)
for i in list(range(0, 12)) + list(range(12, n, 10)):
    worst, wk = 0.0, ''
    for k in snaps[True][i]:
        a, b = snaps[True][i][k], snaps[False][i][k]
        e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
        if e > worst:
            worst, wk = e, k
    print(f'after batch {i:4d}: max rel weight distance {worst:.2e} ({wk})')
