"""UCY pre-training, batch by batch: at every pointwise batch the gradients of the fused-kernel path against the
library-GEMM path on the SAME weights (the step is then taken with the library path)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, 'tests')
import numpy as np, torch
import torch.nn.functional as F
from conftest import GOLDEN, golden
import tests.test_main_gpu as T
from piml_amd import main as MAIN
from piml_amd.models import simulators as SIM
import piml_amd.models.model as MODEL

case = sys.argv[1] if len(sys.argv) > 1 else 'ucy'
g = golden('mainflow_' + case)
DATA = os.path.join(GOLDEN, 'data')
argv = T.COMMON + T.CASES[case] + ['--data_config', os.path.join(DATA, f'mainflow_{case}_pretrain.yaml'),
                                   '--ft_data_config', os.path.join(DATA, f'mainflow_{case}_finetune.yaml'), '--epochs', '1']
init = {k[5:]: torch.tensor(g[k]) for k in g.files if k.startswith('init/')}
orig = SIM.BaseSimulator.train_batch
state = {'n': 0, 'worst': []}


def probe(self, batch_data):
    channelled = hasattr(batch_data, 'mask_p_pred') and hasattr(batch_data, 'waypoints')
    if not channelled:
        ped, obs, selff, labels = batch_data
        res = {}
        for fused in (True, False):
            MODEL.FUSED_ENCODER = MODEL.FUSED_NETWORK = fused
            self.model.zero_grad(set_to_none=True)
            pred = self.model(ped, obs, selff)
            loss = F.mse_loss(pred[0], labels[:, 4:6], reduction='sum') + self.l1_reg_loss(pred[1], self.args.reg_weight, 'sum') + \
                F.binary_cross_entropy(pred[-1], labels[:, 6:], reduction='sum')
            loss.backward()
            res[fused] = (float(loss), {k: p.grad.double().clone() for k, p in self.model.named_parameters() if p.grad is not None},
                          pred[0].detach().double().clone())
        self.model.zero_grad(set_to_none=True)
        for k in res[True][1]:
            a, b = res[True][1][k], res[False][1][k]
            za, zb = int(((a != 0) & (b == 0)).sum()), int(((a == 0) & (b != 0)).sum())
            if za or zb:
                key = (k, 'fused!=0,lib==0' if za else 'fused==0,lib!=0')
                state.setdefault('zeros', {}).setdefault(key, [0, 0.0])
                state['zeros'][key][0] += za + zb
                state['zeros'][key][1] = max(state['zeros'][key][1], float((a - b).abs().max()))
        worst, wk = 0.0, ''
        for k in res[True][1]:
            a, b = res[True][1][k], res[False][1][k]
            e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
            if e > worst:
                worst, wk = e, k
        dl = abs(res[True][0] - res[False][0]) / abs(res[False][0])
        da = float((res[True][2] - res[False][2]).abs().max() / res[False][2].abs().max())
        state['worst'].append((worst, state['n'], wk, dl, da, tuple(ped.shape), tuple(obs.shape),
                               int(ped.isnan().sum() + obs.isnan().sum() + selff.isnan().sum())))
        state['n'] += 1
    MODEL.FUSED_ENCODER = MODEL.FUSED_NETWORK = False
    return orig(self, batch_data)


SIM.BaseSimulator.train_batch = probe
try:
    MAIN.main(argv, init_state=init)
except Exception as ex:   # noqa
    print('main ended with', type(ex).__name__, ex)
w = sorted(state['worst'], reverse=True)
print('batches probed', state['n'])
for row in w[:12]:
    print('grad rel err %.2e  batch %d  %s  loss rel %.1e  acc rel %.1e  ped %s obs %s nans %d' % row)
print('median grad rel err %.2e' % np.median([r[0] for r in w]))
for k, v in state.get('zeros', {}).items():
    print('exact-zero mismatch', k, 'elements (summed over batches)', v[0], 'max abs', v[1])
