"""Per-element relative differences between the fused-kernel path and the library-GEMM path on the pointwise fixture
batches (Adam normalises every element by its own magnitude: small-magnitude gradient elements matter)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
import torch.nn.functional as F
from conftest import golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
from piml_amd.models import simulators as SIM
import piml_amd.models.model as MODEL

for case in ('gc', 'ucy'):
    g = golden('mainflow_' + case)
    pw = golden('mainflow_pointwise_' + case)
    args = MAIN.get_args(T.COMMON + T.CASES[case])
    args.ped_feature_dim, args.obs_feature_dim, args.self_feature_dim = 6, 6, 7
    sim = SIM.BaseSimulator(args)
    sim.model.load_state_dict({k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')})
    sim.model.train()
    for bi in (0, 57):
        ped, obs, selff, labels = [torch.tensor(pw[f'b{bi}/{k}'], device='cuda:0') for k in ('ped', 'obs', 'selff', 'labels')]
        print(case, 'batch', bi, 'ped', tuple(ped.shape), 'obs', tuple(obs.shape), 'nan in ped/obs/selff:',
              int(ped.isnan().sum()), int(obs.isnan().sum()), int(selff.isnan().sum()))
        grads = {}
        for fused in (True, False):
            MODEL.FUSED_ENCODER = MODEL.FUSED_NETWORK = fused
            sim.model.zero_grad(set_to_none=True)
            pred = sim.model(ped, obs, selff)
            loss = F.mse_loss(pred[0], labels[:, 4:6], reduction='sum') + sim.l1_reg_loss(pred[1], args.reg_weight, 'sum') + \
                F.binary_cross_entropy(pred[-1], labels[:, 6:], reduction='sum')
            loss.backward()
            grads[fused] = {k: p.grad.double().cpu().numpy().copy() for k, p in sim.model.named_parameters() if p.grad is not None}
        MODEL.FUSED_ENCODER = MODEL.FUSED_NETWORK = True
        for k in grads[True]:
            a, b = grads[True][k], grads[False][k]
            ref = pw[f'b{bi}/grad/{k}'].astype(np.float64) if f'b{bi}/grad/{k}' in pw.files else None
            mx = np.abs(b).max()
            d = np.abs(a - b)
            relel = d / np.maximum(np.abs(b), 1e-30)
            big = (np.abs(b) > 1e-6 * mx)
            msg = f'   {k:44s} max|d|/max {d.max() / max(mx, 1e-30):.1e}  elem-rel: median {np.median(relel[big]) if big.any() else 0:.1e} p99 {np.quantile(relel[big], 0.99) if big.any() else 0:.1e} max {relel[big].max() if big.any() else 0:.1e}'
            if ref is not None:
                rl = np.abs(b - ref) / np.maximum(np.abs(ref), 1e-30)
                rf = np.abs(a - ref) / np.maximum(np.abs(ref), 1e-30)
                bg = np.abs(ref) > 1e-6 * np.abs(ref).max()
                msg += f' | vs reference p99: lib {np.quantile(rl[bg], 0.99):.1e} fused {np.quantile(rf[bg], 0.99):.1e}'
            print(msg)
