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
# where does the jump sit?  (a dead hidden unit waking up in one path only = one COLUMN of the next layer's weight,
# or one ROW + bias of its own layer)
for i in range(n):
    k = 'ped_encoder.mlp.2.weight'
    a, b = snaps[True][i][k], snaps[False][i][k]
    d = (a - b).abs() / b.abs().max()
    if float(d.max()) > 5e-4:
        big = (d > 1e-4).nonzero()
        rows, cols = big[:, 0].unique(), big[:, 1].unique()
        print(f'first batch with a > 5e-4 gap in {k}: {i}; elements > 1e-4: {len(big)} in {len(rows)} rows x {len(cols)} columns;'
              f' rows {rows.tolist()[:8]} cols {cols.tolist()[:8]}')
        for kk in ('ped_encoder.mlp.0.weight', 'ped_encoder.mlp.0.bias', 'ped_encoder.mlp.2.bias'):
            dd = (snaps[True][i][kk] - snaps[False][i][kk]).abs() / snaps[False][i][kk].abs().max()
            print(f'   same batch, {kk}: max gap {float(dd.max()):.2e} at {dd.argmax().item()}')
        break
