"""Graph-replayed fine-tuning step (HOT LOOP C) on the golden GC batch: time per step, as train_batch() in a loop (every step's
scalars read before the next is queued) and as the training loop runs it (train_batch_async: step i + 1 queued before step i is
read); run under rocprofv3 --kernel-trace --stats for the kernel mix (development aid)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from test_simulator_gpu import sim_args, load_data, make_sim

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
MODEL = sys.argv[2] if len(sys.argv) > 2 else 'pinnsf_m'          # python tools/time_finetune.py [steps] [pinnsf_m | pinnsf_bm]
data = load_data(g, 'train_' + MODEL)
if len(sys.argv) > 3 and int(sys.argv[3]) > 1:                     # python tools/time_finetune.py [steps] [model] [agent axis x this]
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from train_mode_steps import tiled
    data = tiled(data, int(sys.argv[3]))
if MODEL == 'pinnsf_m':
    sim = make_sim(g, sim_args(learning_rate=1e-3, hip_graph=True), 'train_pinnsf_m/sd/')
else:
    from piml_amd.models.simulators import BaseSimulator
    torch.manual_seed(666)
    sim = BaseSimulator(sim_args(model=MODEL, dropout=0.0, learning_rate=1e-3, hip_graph=True))
    sim.model.eval()
for _ in range(6):
    sim.train_batch(data)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
for _ in range(n):
    sim.train_batch(data)
torch.cuda.synchronize()
print(f'{MODEL} fine-tune step (graph): {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step')
torch.cuda.synchronize(); t0 = time.perf_counter()
waiting = None
for _ in range(n):
    nxt = sim.train_batch_async(data)
    if waiting is not None:
        waiting()
    waiting = nxt
waiting()
torch.cuda.synchronize()
print(f'{MODEL} fine-tune step (graph, one step of lookahead as in train()): {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step')
