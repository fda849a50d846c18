#!/usr/bin/env python3
"""Headline benchmark: agent-pair force evaluations/s and simulated steps/s of one PINSF
step (forward + backward) on the synthetic 4096-agent GC scene with 2000 obstacle points
(BASELINE.json configs[2]).

One step = relfeat forward (HIP) -> PINNSF_multitask forward (PyTorch-ROCm) -> backward with
upstream gradient ones on the acceleration -> relfeat backward (HIP), with the scene already
resident in HBM.  pairs/step = N * (N + M).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by torch.distributed.run (one rank per GPU, RCCL): agents are block-sharded
4096 per GPU (weak scaling in focal agents: the scene has 4096*N agents), with a per-step
all-gather of the (p,v,a) records and a reduce-scatter of their gradients.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant HIP kernel (relfeat forward)
with the operand-stream byte model of SURVEY.md 8d; `cpu_baseline` times the CPU oracle
(C restatement, OpenMP) + the same PINNSF on the host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def model_args():
    return types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3,
        processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5, activation='relu',
        dataset_name='gc1560')


def cpu_baseline(scene, n_agents, n_obs, budget_s):
    """The same step on the host: oracle relfeat fwd/bwd (C, all cores) + PINNSF fwd/bwd in
    torch on the CPU.  Bounded to ~budget_s seconds of CPU work."""
    from oracle import oracle as O
    from piml_amd.models.model import PINNSF_multitask
    torch.manual_seed(666)
    model = PINNSF_multitask(model_args()).eval()
    cores = O.num_threads()
    torch.set_num_threads(cores)
    keys = ('position', 'velocity', 'acceleration', 'destination')
    args = [scene[k][None] for k in keys]
    v0 = torch.tensor(scene['desired_speed'])

    def step():
        pf, of, df, pi, oi, _, _ = O.relfeat_fwd(*args, scene['obstacles'], return_index=True)
        pf_t, of_t, df_t = [torch.tensor(x[0]).requires_grad_(True) for x in (pf, of, df)]
        selff = torch.cat((df_t, torch.tensor(scene['velocity']), torch.tensor(scene['acceleration']), v0), -1)
        acc = model(pf_t, of_t, selff)[0]
        acc.backward(torch.ones_like(acc))
        O.relfeat_bwd(pf_t.grad.numpy(), of_t.grad.numpy(), df_t.grad.numpy(), pi[0], oi[0],
                      scene['position'], scene['destination'])
    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 200:
            break
    pairs = n_agents * (n_agents + n_obs)
    return {'value': pairs * n / el, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
            'ms_per_step': el / n * 1e3,
            'sample': f'{n} steps of the same N={n_agents}, M={n_obs} scene: oracle relfeat fwd+bwd '
                      f'(C restatement, OpenMP {cores} threads) + PINNSF_multitask fwd+bwd in torch-CPU'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--agents', type=int, default=4096, help='focal agents per GPU')
    ap.add_argument('--obstacles', type=int, default=2000)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the cpu_baseline leg')
    ap.add_argument('--graph', type=int, default=1, help='replay the step from a captured HIP graph')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run '
                         f'--nproc-per-node {args.gpus}')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)

    from piml_amd import ops
    from piml_amd.models.model import PINNSF_multitask
    from piml_amd.scenes import synthetic_gc_scene, pair_count, algorithmic_bytes
    from piml_amd.sharded import ShardedScene, allreduce_gradients

    n_own, M = args.agents, args.obstacles
    N = n_own * world
    scene = synthetic_gc_scene(N, M, seed=args.seed)
    obstacles = torch.tensor(scene['obstacles'], device=dev)
    M_eff = obstacles.shape[0]
    sh = ShardedScene(N, obstacles) if world > 1 else None
    b0 = rank * n_own
    rows = slice(b0, b0 + n_own)
    state_own = torch.tensor(np.concatenate([scene[k][rows] for k in ('position', 'velocity', 'acceleration')],
                                            axis=-1), device=dev).requires_grad_(True)
    dest_own = torch.tensor(scene['destination'][rows], device=dev)
    v0_own = torch.tensor(scene['desired_speed'][rows], device=dev)

    torch.manual_seed(666)
    model = PINNSF_multitask(model_args()).to(dev).eval()   # eval: dropout off, deterministic
    params = [p for p in model.parameters()]
    ones = torch.ones(n_own, 2, device=dev)

    ev_pairs = []

    def step(timed):
        """One forward + backward pass of the hot path over the scene."""
        state_own.grad = None
        for p in params:
            p.grad = None
        state_full = sh.gather_state(state_own) if sh is not None else state_own
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        pf, of, df = ops.relative_features_packed(state_full, dest_own, obstacles, b0, n_own)
        if timed:
            e1.record()
            ev_pairs.append((e0, e1))
        self_features = torch.cat((df, state_own[:, 2:4], state_own[:, 4:6], v0_own), dim=-1)
        acc = model(pf, of, self_features)[0]
        acc.backward(ones)
        if sh is not None:
            allreduce_gradients(params, sh.group)
        return acc

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    kernel_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / max(len(ev_pairs), 1)
    pairs_step = N * (N + M_eff)                       # all ranks together
    alg_bytes = n_own * (24 * N + 8 * M_eff) + 488 * n_own   # this rank's launch
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0

    if rank == 0:
        out = {
            'metric': 'agent-pair force evals/sec + simulated steps/sec, 4096-agent GC scene',
            'value': pairs_step * args.steps / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'steps_per_s': args.steps / elapsed,
            'agent_steps_per_s': N * args.steps / elapsed,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'cfg3: synthetic GC scene, forward+backward PINSF step '
                                   '(HIP relfeat fwd/bwd + PINNSF_multitask fwd/bwd in PyTorch-ROCm)',
                       'agents_per_gpu': n_own, 'agents_total': N, 'obstacle_points': M_eff,
                       'pairs_per_step': pairs_step, 'topk_ped': 6, 'topk_obs': 10,
                       'sharding': 'single GPU' if world == 1 else
                       f'agent blocks over {world} ranks, all-gather(p,v,a) + reduce-scatter(grad) per step'},
            'roofline': {'bound': 'hbm', 'kernel': 'relfeat_fwd_kernel', 'achieved': achieved,
                         'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': None, 'kernel_us': kernel_ms * 1e3, 'algorithmic_bytes': alg_bytes,
                         'note': 'operand-stream byte model (24 B/ped pair + 8 B/obstacle pair + 488 B/focal); '
                                 'the sources are LDS/L2 resident, so frac > 1 is possible and HBM traffic '
                                 'is far below the model (see DESIGN.md)'},
        }
        if args.cpu_seconds > 0 and world == 1:
            out['cpu_baseline'] = cpu_baseline(scene, N, M_eff, args.cpu_seconds)
        elif world > 1:
            out['cpu_baseline'] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
