"""The reference's TRAINING configuration -- model.train(), --dropout 0.5 (src/main.py:45, src/models/simulators.py:311)
-- on the two training loops: HOT LOOP A (pointwise pre-training step, simulators.py:327-360) and HOT LOOP C (fine-tuning
step through the differentiable rollout, :659-832), time per optimiser step.  Run plain for the timings, or under
`rocprofv3 --kernel-trace --stats` for the kernel mix (the encoder kernels must be enc_*_x3_kernel, no library GEMM for
pinnsf_m).   python tools/train_mode_steps.py [--dropout 0.5] [--models pinnsf_m,pinnsf_bm] [--reps 50]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
import numpy as np  # noqa: E402
import torch  # noqa: E402
from test_simulator_gpu import sim_args, load_data  # noqa: E402

DEV = 'cuda:0'


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


def tiled(data, times):
    """the golden batch with its agent axis repeated `times` (copies 40 m apart: they do not interact)"""
    if times == 1:
        return data
    import types
    n = data.position.shape[-2]
    out = types.SimpleNamespace(**data.__dict__)
    for k, v in data.__dict__.items():
        if not torch.is_tensor(v) or k == 'obstacles':
            continue
        ax = [i for i, s in enumerate(v.shape) if s == n]
        if not ax:
            continue
        ax = ax[-1] if k in ('dest_idx', 'mask_p', 'mask_p_pred', 'dest_num', 'abnormal_mask') else ax[0]
        reps = [v] * times
        if k in ('position', 'destination', 'waypoints'):
            reps = [v + torch.tensor([40.0 * i, 0.0], device=v.device) for i in range(times)]
        setattr(out, k, torch.cat(reps, dim=ax).contiguous())
    return out


def timeit(fn, reps):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from piml_amd.models.simulators import BaseSimulator
    p = float(arg('--dropout', '0.5'))
    reps = int(arg('--reps', '50'))
    models = arg('--models', 'pinnsf_m,pinnsf_bm').split(',')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'rollout.npz'), allow_pickle=False)
    gen = torch.Generator().manual_seed(0)
    for model in models:
        for train in ((True, False) if '--with-eval' in sys.argv else (True,)):
            tag = f'{model} {"train() dropout " + str(p) if train else "eval()"}'
            # ---- HOT LOOP A: pointwise rows (batch_size 128 in piml-gcdata.yaml; 1024 and 4096 for scale) ----
            for rows in (() if '--finetune-only' in sys.argv else ((128,) if '--pointwise-only' in sys.argv else (128, 1024, 4096))):
                torch.manual_seed(666)
                sim = BaseSimulator(sim_args(model=model, dropout=p, learning_rate=2e-4, collision_pred_weight=5e-2))
                sim.model.train(train)
                batch = (torch.randn(rows, 6, 6, generator=gen).to(DEV), torch.randn(rows, 10, 6, generator=gen).to(DEV),
                         torch.randn(rows, 7, generator=gen).to(DEV), torch.rand(rows, 12, generator=gen).to(DEV))
                ms = timeit(lambda: sim.train_batch(batch), reps)
                print(f'{tag}: pointwise pre-training step, {rows} rows: {ms:.3f} ms/step (one captured graph)', flush=True)
            # ---- HOT LOOP C: fine-tuning step, the golden GC batch (4 windows x 5 frames x 122 agents) and 8 x its agents ----
            data = load_data(g, 'train_' + ('pinnsf_m' if model == 'pinnsf_m' else 'pinnsf_bm'))
            for times in (() if '--pointwise-only' in sys.argv else ((1,) if '--finetune-only' in sys.argv else (1, 8))):
                torch.manual_seed(666)
                sim = BaseSimulator(sim_args(model=model, dropout=p, learning_rate=1e-3, hip_graph=True))
                sim.model.train(train)
                d = tiled(data, times)
                ms = timeit(lambda: sim.train_batch(d), reps)
                print(f'{tag}: fine-tuning step, {d.position.shape[0]} windows x {d.position.shape[1]} frames x '
                      f'{d.position.shape[2]} agents: {ms:.3f} ms/step (one captured graph)', flush=True)


if __name__ == '__main__':
    main()
