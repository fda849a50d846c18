#!/usr/bin/env python3
"""Per-tensor differences of the smoke's fused PINSF step against the torch.nn step (eval mode)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piml_amd import ops
from piml_amd.scenes import synthetic_gc_scene
import piml_amd.models.model as MODEL
sc = synthetic_gc_scene(512, 300, seed=0)
keys = ('position', 'velocity', 'acceleration', 'destination', 'obstacles')
t = [torch.tensor(sc[k], device='cuda:0') for k in keys]
args = types.SimpleNamespace(ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128, processor_hidden_size=128,
                             decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5,
                             activation='relu', dataset_name='gc1560')
torch.manual_seed(0)
net = MODEL.PINNSF_multitask(args).to('cuda:0')
state = torch.tensor(np.concatenate([sc[k] for k in keys[:3]], -1), device='cuda:0')
net.train(False)
res = {}
for fused in (False, True):
    MODEL.FUSED_GLUE = fused
    s = state.clone().requires_grad_(True)
    net.zero_grad(set_to_none=True)
    f = ops.relative_features_packed_self(s, t[3].detach(), t[4], torch.tensor(sc['desired_speed'], device='cuda:0'), 0, s.shape[0])
    acc = net(*f)[0]
    acc.sum().backward()
    res[fused] = dict([('acc', acc.detach()), ('state', s.grad)] + [(k, p.grad) for k, p in net.named_parameters() if p.grad is not None])
for k in res[True]:
    a, b = torch.nan_to_num(res[True][k]), torch.nan_to_num(res[False][k])
    d = (a - b).abs()
    print(f'{k:40s} max|b| {float(b.abs().max()):.3e}  max diff {float(d.max()):.3e}  rel {float(d.max() / b.abs().max().clamp_min(1e-9)):.2e}  at {int(d.argmax())}')
