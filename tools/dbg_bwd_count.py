import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_simulator_gpu import sim_args, load_data, make_sim
from piml_amd import ops
g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
data = load_data(g, 'train_pinnsf_m')
cnt = {'mlp': 0, 'step': 0, 'relfeat': 0}
orig = ops._MLPChain.backward
def wrapped(ctx, gg):
    cnt['mlp'] += 1
    return orig(ctx, gg)
ops._MLPChain.backward = staticmethod(wrapped)
o2 = ops._TrainRolloutStep.backward
def w2(ctx, *a):
    cnt['step'] += 1
    print('  step bwd t_next', ctx.geom[3], [x is not None for x in a[:3]])
    return o2(ctx, *a)
ops._TrainRolloutStep.backward = staticmethod(w2)
for fused in (True, False):
    sim = make_sim(g, sim_args(hip_graph=False), 'train_pinnsf_m/sd/')
    sim.fused_train_step = fused
    for k in cnt: cnt[k] = 0
    out = sim.test_multiple_rollouts_for_training(data)
    out[0].backward()
    print('fused', fused, cnt, 'T', data.num_frames)
