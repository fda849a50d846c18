#!/usr/bin/env python3
"""Headline benchmark: agent-pair force evaluations/s and simulated steps/s of one PINSF
step (forward + backward) on a synthetic GC scene with 2000 obstacle points.

One step = relfeat forward (HIP) -> PINNSF_multitask forward -> backward with upstream gradient ones
on the acceleration -> relfeat backward (HIP), with the scene already resident in HBM.
pairs/step = N * (N + M).

  python bench.py --gpus G --steps K --warmup W
G = 1: BASELINE.json configs[2] -- 4096 agents on one MI355X.
G > 1: BASELINE.json configs[3] -- ONE 16384-agent scene, agent blocks of 16384/G focal agents per rank
("scaling": "strong"), per-step all-gather of the (p,v,a) records and one all-reduce of [d/d(state), weight
gradients] over RCCL.  When WORLD_SIZE is not set the script launches itself under torch.distributed.run as a
child process (before anything touches the GPU) and forwards rank 0's JSON line.  `--scaling weak` keeps
`--agents` focal agents per GPU instead (scene of agents*G agents).  In strong mode rank 0 also times the same
scene on ONE GPU after the timed region (`single_gpu_same_scene`), the baseline a strong-scaling efficiency needs.

Rank 0 prints ONE JSON line.  `roofline.frac` is SURVEY.md 8d's step-level contract figure (operand-stream
bytes of the step / step time / HBM peak); `roofline.kernels` lists the dominant kernels with the fraction of the
unit that really bounds each (VALU issue for the N-body sweep; HBM / the bf16 matrix pipe for the encoder kernels on split
products, f32 MFMA with PIML_ENC_PRODUCTS=f32).  `cpu_baseline` times the CPU
oracle (C restatement, OpenMP) + the same PINNSF on the host cores on a bounded sample, and the feature step in
the reference's own dataflow (oracle/dataflow.py).

The step is replayed from one captured HIP graph (eager fallback).  After the timed region the gradients left
behind by the replayed step are compared with an eager autograd step (`verified_max_rel_err`); a mismatch is fatal.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL / cross-process tensor sharing)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
# HIP-graph replay with memset nodes is only correct with the CLR packet capture off (piml_amd/__init__.py)
os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')

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
    cores_avail = O.num_threads()
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
    # pick the thread count that runs this step fastest on this host (all hardware threads is often
    # not it for 65k-row GEMMs): a short sweep, then the bounded timed sample with the winner
    best, cores = None, cores_avail
    for th in sorted({cores_avail, max(cores_avail // 2, 1), max(cores_avail // 4, 1), min(cores_avail, 16)}, reverse=True):
        torch.set_num_threads(th)
        os.environ['OMP_NUM_THREADS'] = str(th)
        O.lib().oracle_set_threads(th)
        step()
        dt = float('inf')
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            dt = min(dt, time.perf_counter() - t0)
        if best is None or dt < 0.95 * best:      # prefer more threads unless fewer are clearly faster
            best, cores = dt, th
    torch.set_num_threads(cores)
    O.lib().oracle_set_threads(cores)
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
    out = {'value': pairs * n / el, 'unit': 'pairs/s', 'cores': cores, 'kind': 'port',
           'ms_per_step': el / n * 1e3,
           'sample': f'{n} steps of the same N={n_agents}, M={n_obs} scene: oracle relfeat fwd+bwd '
                     f'(C restatement, OpenMP) + PINNSF_multitask fwd+bwd in torch-CPU, {cores} of {cores_avail} '
                     f'hardware threads (fastest of a short sweep)'}
    # second form: the feature step organised the way the reference computes it -- dense N x M
    # relative tensors, full sort, gather (oracle/dataflow.py) -- forward only, a couple of repetitions
    try:
        from oracle import dataflow
        th = min(cores_avail, 32)
        torch.set_num_threads(th)
        t = [torch.tensor(np.nan_to_num(scene[k]) if k in ('velocity', 'acceleration') else scene[k])
             for k in keys] + [torch.tensor(scene['obstacles'])]
        with torch.no_grad():
            dataflow.relative_features(*t)
            reps, t0 = 0, time.perf_counter()
            while reps < 3 and (reps == 0 or time.perf_counter() - t0 < 6.0):
                dataflow.relative_features(*t)
                reps += 1
            dfl = (time.perf_counter() - t0) / reps
        out['reference_dataflow'] = {
            'features_forward_ms': dfl * 1e3, 'pairs_per_s': pairs / dfl, 'threads': th, 'repetitions': reps,
            'note': 'get_relative_features with the reference\'s dataflow (dense N x M tensors, torch.sort, gather) '
                    'restated in oracle/dataflow.py, torch-CPU, forward only -- a lower bound for the reference, which '
                    'also materialises both operands with .repeat and fills headings in a Python loop (it measured '
                    '2.05 s on 8 cores, BASELINE.md)'}
    except MemoryError as ex:      # the dense tensors need ~3 GB
        out['reference_dataflow'] = {'error': str(ex)}
    torch.set_num_threads(cores)
    return out


_T0 = time.perf_counter()


def _phase(msg):
    if os.environ.get('PIML_BENCH_VERBOSE'):
        print(f'[bench +{time.perf_counter() - _T0:7.2f}s] {msg}', file=sys.stderr, flush=True)


def encoder_products_check(dev, _lib):
    """Fused encoder forward + backward at the bench's row counts (4096 x 12 and 4096 x 4 neighbour rows, random weights)
    in both product forms against a float64 evaluation of the same network: max |error| / max |value| per tensor."""
    from piml_amd import ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(3)
    H = 128

    def branch(k):
        x = (torch.randn(4096, k, 6, generator=g) * 2).to(dev).requires_grad_(True)
        w = [(torch.randn(*d, generator=g) * (0.3 if len(d) == 2 else 0.1)).to(dev).requires_grad_(True)
             for d in [(H, 6), (H,), (H, H), (H,), (H, H), (H,)]]
        return dict(x=x, scale=2.0, weights=w, pooled=True)
    brs = [branch(12), branch(4)]
    gps = [torch.randn(4096, H, generator=g).to(dev) for _ in brs]
    names = ('msgs', 'pooled', 'g_x', 'dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3')
    refs = []
    for br, gp in zip(brs, gps):
        x = br['x'].detach().double().requires_grad_(True)
        w = [t.detach().double().requires_grad_(True) for t in br['weights']]
        h = torch.relu(torch.relu(x @ w[0].t() + w[1]) @ w[2].t() + w[3])
        msgs = 2.0 * (h @ w[4].t() + w[5])
        pooled = msgs.sum(-2)
        (pooled * gp.double()).sum().backward()
        refs.append((msgs.detach(), pooled.detach(), x.grad, *[t.grad for t in w]))
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    out = {}
    old = L.piml_encoder_products(-1)
    try:
        for mode, label in ((1, 'split_bf16_products'), (0, 'f32_matrix_instruction')):
            L.piml_encoder_products(mode)
            outs = ops.fused_encoders(brs)
            grads = torch.autograd.grad(sum((p * gp).sum() for (m, p), gp in zip(outs, gps)), leaves)
            worst = dict.fromkeys(names, 0.0)
            for i, ((m, p), ref) in enumerate(zip(outs, refs)):
                for nm, a, b in zip(names, (m.detach(), p.detach(), *grads[7 * i:7 * i + 7]), ref):
                    worst[nm] = max(worst[nm], float((a.double() - b).abs().max() / b.abs().max()))
            out[label] = {k: float(f'{v:.2e}') for k, v in worst.items()}
    finally:
        L.piml_encoder_products(old)
    out['note'] = ('max |error| / max |value| per tensor against float64, both branches at 4096 agents (one-wave kernels); '
                   'north-star bar 1e-5; tests/test_encoder_gpu.py::test_split_bf16_products_are_f32_arithmetic asserts it')
    return out


def secondary_measurements(scene, n, dev, _lib, prof=None, light=False):
    """Back-to-back launches of the other HIP kernels of the path on the first `n` agents of the scene,
    timed with HIP events (informational; not part of `value`).  MLAPM operand-stream model: 16 B per pair + 36 B
    per agent -- a rate of LDS-resident operands, not HBM traffic and not a fraction of anything; the
    unit that bounds the kernel is the vector pipe (rsq / rcp / exp2 + ~60 VALU instructions per pair), reported as the
    VALU-issue share of the SIMD cycles from the committed PMC pass."""
    from piml_amd import ops
    ok = ~np.isnan(scene['position'][:n, 0])
    p, v, v0, d = [torch.tensor(scene[k][:n][ok], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
    m = p.shape[0]
    gc = dict(version='GC', tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)      # src/main_mlapm.py:16

    def timed(fn, reps=50):
        for _ in range(5):
            fn()
        tm = _lib.StreamTimer()
        tm.start()
        for _ in range(reps):
            fn()
        tm.stop()
        return tm.elapsed_ms() * 1e3 / reps
    fwd_us = timed(lambda: ops.mlapm_step(p, v, v0, d, 0.08, 0.3, **gc))
    leaves = [x.clone().requires_grad_(True) for x in (p, v, v0, d)]
    w = torch.ones(m, 2, device=dev)

    def fwd_bwd():
        act = ops.mlapm_step(*leaves, 0.08, 0.3, **gc)
        return torch.autograd.grad(act, leaves, w)
    # forward + analytic backward replayed from ONE captured HIP graph: the figure is the kernels' time, not the
    # ~100 us of autograd / ctypes bookkeeping an eager backward of a 4096-agent scene costs on the host
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        keep = fwd_bwd()
    fb_us = timed(graph.replay, reps=30)
    bwd_us = max(fb_us - fwd_us, 0.0)
    del keep
    bytes_fwd = 16 * m * m + 36 * m
    # the neighbour search in the shape every rank of the 8-GPU run launches (cfg4: 2048 focal rows against 16384 sources)
    # (light: a sharded run -- rank 0 keeps the other ranks waiting while it measures: only the short figures there)
    try:
        if light:
            raise RuntimeError('skipped in sharded runs (see the 1-GPU line)')
        from piml_amd.scenes import synthetic_gc_scene
        big = synthetic_gc_scene(16384, int(np.asarray(scene['obstacles']).reshape(-1, 2).shape[0]), seed=0)
        state = torch.tensor(np.concatenate([big['position'], big['velocity'], big['acceleration']], axis=-1), device=dev)
        obs = torch.tensor(big['obstacles'], device=dev)
        d_rows = torch.tensor(big['destination'][:2048], device=dev)
        outs = ops.relative_features_packed(state, d_rows, obs, 0, 2048, return_index=True)
        shard_us = timed(lambda: ops.relative_features_packed_into(outs, state, d_rows, obs, 0, 2048), reps=100)
        shard = {'focal_rows': 2048, 'sources': 16384, 'obstacle_points': int(obs.shape[0]), 'fwd_us': shard_us,
                 'note': 'relfeat_fwd_kernel for one rank\'s agent block of the cfg4 scene (back-to-back launches, HIP events)'}
        del state, obs, d_rows, outs
    except Exception as ex:   # noqa: BLE001 - informational
        shard = {'error': f'{type(ex).__name__}: {ex}'}
    # the metric's second half: simulated steps per second of the same 4096-agent scene (inference, no gradients)
    sim = {}
    try:
        if light:
            raise RuntimeError('skipped in sharded runs (see the 1-GPU line)')
        import time as _time
        from piml_amd.models.mlapm import MLAPM
        mm = MLAPM(**gc)
        mm.rollout(p, v, v0, d, 0.08, 0.3, 60)
        torch.cuda.synchronize()
        t0 = _time.perf_counter()
        mm.rollout(p, v, v0, d, 0.08, 0.3, 2000)
        torch.cuda.synchronize()
        el = _time.perf_counter() - t0
        sim['mlapm_gc'] = {'agents': m, 'steps_per_s': 2000 / el, 'us_per_step': el / 2000 * 1e6,
                           'note': 'MLAPM.rollout (src/main_mlapm.py:18-36): a frame = ONE launch (piml_mlapm_rollout_step: state read '
                                   'from the trajectory, arrivals leave, device-side frame counter), 8 frames per captured HIP graph; '
                                   'wall clock over 2000 frames incl. the capture and the allocation of the trajectories'}
    except Exception as ex:   # noqa: BLE001 - informational
        sim['mlapm_gc'] = {'error': f'{type(ex).__name__}: {ex}'}
    try:
        if light:
            raise RuntimeError('skipped in sharded runs (see the 1-GPU line)')
        import time as _time
        from piml_amd.scenes import synthetic_rollout_data
        from piml_amd.models.simulators import BaseSimulator
        from piml_amd.main import get_args
        a = get_args(['--dataset_name', 'gc1560', '--model', 'pinnsf_m'])
        a.ped_feature_dim, a.obs_feature_dim, a.self_feature_dim, a.device = 6, 6, 7, str(dev)
        a.exp_name, a.model_name_suffix = 'bench', 'bench'
        T = 200
        data = synthetic_rollout_data(n, int(np.asarray(scene['obstacles']).reshape(-1, 2).shape[0]), T, dev)
        simulator = BaseSimulator(a)
        simulator.model.eval()
        with torch.no_grad():
            simulator.get_multiple_rollouts(data, 0, load_model=False)
            torch.cuda.synchronize()
            t0 = _time.perf_counter()
            simulator.get_multiple_rollouts(data, 0, load_model=False)
            torch.cuda.synchronize()
            el = _time.perf_counter() - t0
        sim['pinnsf_m'] = {'agents': n, 'steps_per_s': T / el, 'us_per_step': el / T * 1e6,
                           'note': 'BaseSimulator.get_multiple_rollouts (src/models/simulators.py:552-657), PINNSF_multitask eval(), '
                                   'random-init weights: relfeat forward + the inference forward of the network (neighbour-axis sum before the encoders\' '
                                   'last layer, PIML_POOL_H2; no collision head) + integrator epilogue per frame, one captured frame replayed; '
                                   'wall clock over 200 frames'}
        del data, simulator
    except Exception as ex:   # noqa: BLE001 - informational
        sim['pinnsf_m'] = {'error': f'{type(ex).__name__}: {ex}'}
    return {'relfeat_shard_shape': shard, 'simulated_steps': sim,
            'mlapm_gc_step': {'agents': m, 'pairs': m * m, 'fwd_us': fwd_us, 'bwd_us': bwd_us,
                              'pairs_per_s_fwd': m * m / fwd_us * 1e6,
                              'operand_stream_gbs_fwd': bytes_fwd / (fwd_us * 1e-6) / 1e9,
                              'bound': 'valu',
                              'valu_busy_frac_fwd': ((prof or {}).get('mlapm') or {}).get('fwd_valu_busy_frac'),
                              'valu_busy_frac_bwd': ((prof or {}).get('mlapm') or {}).get('bwd_valu_busy_frac'),
                              'valu_busy_source': 'SQ_ACTIVE_INST_VALU share of the SIMD cycles of mlapm_fwd_kernel / mlapm_bwd_sys_kernel, '
                                                  'committed rocprofv3 --pmc pass (static)' if (prof or {}).get('mlapm') else None,
                              'note': 'closed-form social force (MLAPM.step, GC variant) forward / analytic backward, '
                                      'present agents of the same scene; forward: back-to-back eager launches; backward (every ordered pair '
                                      'evaluated once: mlapm_bwd_sys_kernel + its partial-row launch): (forward + backward '
                                      'replayed from one captured HIP graph) - forward, HIP events'}}


def p2p_alive(p2p, dev, collective):
    """True when no exchange of the pair has raised its sticky time-out word, on ANY rank (MIN over the ranks when `collective`).  A
    timed-out exchange is dead for good -- every later launch returns at its first instruction -- so a timed region that used it
    measured skipped exchanges: the caller records an error instead of a time and rebuilds the pair before the next leg."""
    ok = 1 if (p2p is None or all(e.ok() for e in p2p)) else 0
    if collective and dist.is_initialized():
        t = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = int(t.item())
    return bool(ok)


def cfg4_projection(args, dev, messages, shards=8, scene_agents=16384):
    """One rank's step of BASELINE.json configs[3] (16384 agents over 8 GPUs) measured on ONE GPU, and what follows from it.
    The rank's compute is exact (2048 focal rows against all 16384 sources, the network on 2048 agents, relfeat backward over
    its rows: Step(emulate_shard=True)); its exchanges run at world 1 over the bytes the rank contributes, in each form
    (RCCL bucket all-reduce / reduce-scatter + all-reduce / P2P stores): their single-rank latency floor, NOT their cost with
    seven peers.  The same scene on this one GPU is the strong-scaling baseline."""
    from piml_amd.scenes import synthetic_gc_scene
    from piml_amd.sharded import p2p_exchanges
    import piml_amd.models.model as MODEL
    out = {'shards': shards, 'scene_agents': scene_agents, 'focal_rows_per_rank': scene_agents // shards}
    scene = synthetic_gc_scene(scene_agents, args.obstacles, seed=args.seed)
    n_own = scene_agents // shards
    k = 50
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29537')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    torch.manual_seed(666)
    n_params = sum(p.numel() for p in MODEL.PINNSF_multitask(model_args()).parameters())
    p2p = p2p_exchanges(0, 1, n_own, n_params, lambda b: [b])
    compute_us = None
    for form in ('bucket', 'rs', 'p2p'):
        try:
            st = Step(scene, scene_agents, n_own, 0, args.obstacles, dev, dist.group.WORLD, True, False, bool(args.graph), exchange=form,
                      messages=messages, p2p=p2p, emulate_shard=True)
            st.capture()
            el = st.time_steps(k, 10)
            out[f'rank_step_us_{form}'] = el / k * 1e6
            if form == 'p2p' and not p2p_alive(p2p, dev, False):
                out[f'rank_step_us_{form}'] = 'P2P exchange timed out (sticky status word set): the replays skipped it, no time recorded'
            if form != 'p2p' and st.graph is not None and compute_us is None:      # the captured compute alone (RCCL forms: the exchanges sit outside the graph)
                for _ in range(10):
                    st.graph.replay()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(k):
                    st.graph.replay()
                torch.cuda.synchronize()
                compute_us = (time.perf_counter() - t0) / k * 1e6
            del st
        except Exception as ex:   # noqa: BLE001 - informational
            out[f'rank_step_us_{form}'] = f'{type(ex).__name__}: {ex}'
    out['rank_compute_us'] = compute_us
    try:
        one = Step(scene, scene_agents, scene_agents, 0, args.obstacles, dev, None, False, False, bool(args.graph), messages=messages)
        one.capture()
        el = one.time_steps(20, 5)
        out['single_gpu_us'] = el / 20 * 1e6
        del one
    except Exception as ex:   # noqa: BLE001 - informational
        out['single_gpu_us'] = f'{type(ex).__name__}: {ex}'
    if isinstance(out.get('single_gpu_us'), float) and compute_us:
        out['speedup_ceiling_compute_only'] = out['single_gpu_us'] / compute_us
        for form in ('bucket', 'rs', 'p2p'):
            v = out.get(f'rank_step_us_{form}')
            if isinstance(v, float):
                out[f'exchange_floor_us_{form}'] = v - compute_us
                out[f'projected_efficiency_{form}'] = out['single_gpu_us'] / (shards * v)
    out['note'] = ('rank 0 of the 8-way sharded 16384-agent scene on ONE GPU: compute exact, exchanges at world 1 over this rank\'s bytes '
                   '(their latency floor; with seven peers they cost more, the scaling curve is the multi-GPU run\'s); projected_efficiency = '
                   'single_gpu_us / (8 x rank_step_us) is therefore an UPPER bound of the strong-scaling efficiency of that form')
    return out


F32_MFMA_PEAK_TFS = 157.3   # dense f32 matrix peak (v_mfma_f32_32x32x2_f32), same guide


class Step:
    """The hot path over one rank's agent block of one scene: scene tensors, model, the captured HIP graph of
    the compute part and (sharded) the eager exchange either side of it."""

    def __init__(self, scene, N, n_own, b0, M, dev, group, use_dist, two_streams, use_graph, exchange='bucket',
                 overlap=False, model_name='PINNSF_multitask', train_mode=False, ops_module=None, messages=False, p2p=None,
                 emulate_shard=False):
        from piml_amd import ops, _lib
        import piml_amd.models.model as MODEL
        from piml_amd.sharded import ShardedScene
        # ops_module: a stand-in with the feature operators' contracts (tests/test_bench_step_dist.py drives the exchange
        # bookkeeping of this class on CPU ranks over gloo with the oracle behind it); the product uses piml_amd.ops
        self.ops, self._lib = (ops_module if ops_module is not None else ops), _lib
        self.N, self.n_own, self.b0, self.dev, self.group, self.use_dist = N, n_own, b0, dev, group, use_dist
        self.obstacles = torch.tensor(scene['obstacles'], device=dev)
        self.M_eff = self.obstacles.shape[0]
        rows = slice(b0, b0 + n_own)
        self.state_own = torch.tensor(np.concatenate([scene[k][rows] for k in ('position', 'velocity', 'acceleration')],
                                                     axis=-1), device=dev).requires_grad_(True)
        self.dest_own = torch.tensor(scene['destination'][rows], device=dev)
        self.v0_own = torch.tensor(scene['desired_speed'][rows], device=dev)
        # exchange='p2p': p2p = (forward, backward) P2PExchange objects (sharded.p2p_exchanges) -- both exchanges are launches
        # of this library with their step counters on the device, so they sit INSIDE the captured graph; `group` may be None
        self.p2p = p2p if exchange == 'p2p' else None
        if exchange == 'p2p' and p2p is None:
            raise ValueError("Step(exchange='p2p') needs p2p=(forward, backward)")
        if exchange == 'p2p' and (overlap or two_streams):
            # co-residency: the exchange kernel's workgroups spin until all of them (and the peers') have arrived -- nothing else may
            # hold the CUs beside it (piml_amd/csrc/p2p.hip)
            raise ValueError("Step(exchange='p2p'): no --overlap 1 / --two-streams with the P2P exchange (its workgroups must be co-resident)")
        self.sh = (ShardedScene(N, self.obstacles, group=group, force_collectives=True, exchange='p2p' if self.p2p else 'rccl',
                                p2p=self.p2p) if use_dist else None)
        torch.manual_seed(666)
        # eval (default): dropout off, the replayed step can be verified against an eager one.  train_mode: model.train()
        # with the reference's --dropout 0.5 (src/main.py:45, src/models/simulators.py:311): every step draws fresh
        # keep-masks on the device (ops.dropout_keep_bits inside the captured graph) and the fused kernels apply them
        self.model = getattr(MODEL, model_name)(model_args()).to(dev).train(bool(train_mode))
        self.model.messages_wanted = bool(messages)
        if train_mode:      # the device-side (seed, call counter) must exist before a capture; the rank is folded into the seed
            getattr(self.ops, 'dropout_seed', lambda s, d: self.ops.dropout_state(d))(666, dev)
        if two_streams:
            self.model.obs_stream = torch.cuda.Stream()
        self.params = [p for p in self.model.parameters()]
        self.ones = torch.ones(n_own, 2, device=dev)
        # Sharded + graph: only the COMPUTE of a step is captured; the two collectives (all-gather of the
        # records before it; ONE all-reduce of [d/d(state), weight gradients] after it) are issued eagerly on
        # the stream either side of the replay.  `state_all` is the graph's static input: the all-gather
        # target and an autograd leaf whose .grad (N, 6) the captured backward fills.
        self.state_all = torch.zeros(N, 6, device=dev).requires_grad_(True) if use_dist else None
        # emulate_shard (one process, a 1-rank group): ONE rank's share of a scene sharded over N / n_own ranks -- n_own focal rows
        # against all N sources, the network on n_own agents -- with the peers' records already in place and every exchange run at
        # world 1 over the bytes THIS rank contributes (its own block forward; its block of d/d(state) + the weight gradients
        # backward): the compute of a rank of the real run exactly, its exchanges at their single-rank latency floor
        self.emulate = bool(emulate_shard) and use_dist
        if self.emulate:
            with torch.no_grad():
                self.state_all.copy_(torch.tensor(np.concatenate([scene[k] for k in ('position', 'velocity', 'acceleration')], axis=-1), device=dev))
        self.grad_own = torch.zeros(n_own, 6, device=dev) if use_dist else None
        self.bucket = [None, None, None, 0]
        # backward exchange: 'bucket' = ONE all-reduce of [d/d(state) (N, 6) | weight gradients] (the state gradient
        # travels N/n_own times wider than needed, but it is one latency-bound collective); 'rs' = reduce-scatter of
        # d/d(state) to the owners + all-reduce of the weight gradients (minimum bytes, two collectives)
        self.exchange = exchange
        # overlap: the captured step is cut in two -- `pre` (weight pack + the part of the neighbour search that reads
        # only this rank's own records + obstacle branch + self features) is replayed while the all-gather of the other
        # ranks' records is in flight, `graph` (remote half of the search, MLP forward / backward, relfeat backward) after it
        self.overlap = bool(overlap) and use_dist
        self.pre, self.gather_work = None, None
        self.graph, self.static_feats, self.mode = None, None, 'eager'
        self.want_graph = use_graph

    # ---- eager pieces ----
    def features(self):
        """all-gather of the owners' records (sharded) + relfeat forward (HIP)."""
        state_full = self.sh.gather_state(self.state_own) if self.sh is not None else self.state_own
        return self.ops.relative_features_packed_self(state_full, self.dest_own, self.obstacles, self.v0_own,
                                                      self.b0, self.n_own, return_index=True)

    def rest(self, pf, of, self_features, *_idx):
        """PINNSF forward, backward through the MLP and relfeat backward (+ collectives)."""
        from piml_amd.sharded import allreduce_gradients
        acc = self.model(pf, of, self_features)[0]
        with self._deferred():     # the weight-gradient slot sums ride in the relfeat backward's launch
            acc.backward(self.ones)
        if self.sh is not None and self.p2p is not None:
            from piml_amd.sharded import allreduce_gradients_p2p
            allreduce_gradients_p2p(self.params, self.p2p[1])
        elif self.sh is not None:
            allreduce_gradients(self.params, self.group)
        return acc

    def _deferred(self):
        import contextlib
        return getattr(self.ops, 'deferred_slot_sums', contextlib.nullcontext)()

    def features_local(self):
        return self.ops.relative_features_packed_self(self.state_all, self.dest_own, self.obstacles, self.v0_own,
                                                      self.b0, self.n_own, return_index=True)

    def features_local_part(self):
        """What needs only this rank's rows of state_all (legal while the all-gather of the other rows is in flight)."""
        return self.ops.relative_features_local_part(self.state_all, self.dest_own, self.obstacles, self.v0_own,
                                                     self.b0, self.n_own)

    def features_remote_part(self, local):
        return self.ops.relative_features_packed_self(self.state_all, self.dest_own, self.obstacles, self.v0_own,
                                                      self.b0, self.n_own, return_index=True, local=local)

    def rest_local(self, pf, of, self_features, *_idx):
        acc = self.model(pf, of, self_features)[0]
        with self._deferred():
            acc.backward(self.ones)
        # captured: ONE concatenation of the (N, 6) state gradient and all weight gradients into a static
        # bucket, so that the backward exchange is a single latency-bound all-reduce (0.9 MB at 16384 agents)
        grads = [p.grad for p in self.params if p.grad is not None]
        if self.exchange == 'bucket':
            self.bucket[:] = [torch.cat([self.state_all.grad.reshape(-1)] + [g.reshape(-1) for g in grads]), grads]
        elif self.exchange == 'p2p':
            # the fused network's gradients are views of a few flat buffers: those travel as they are and are summed IN PLACE (no
            # concatenation, no copy back); anything else through a padded bucket
            from piml_amd.sharded import grad_bases
            bases = grad_bases(self.params)
            if bases is not None and sum(b.numel() for b in bases) + self.n_own * 6 <= self.p2p[1].fpr:
                self.bucket[:] = [bases, None, None, 0]
            else:
                n = sum(g.numel() for g in grads)
                flat = torch.cat([g.reshape(-1) for g in grads] + ([grads[0].new_zeros((-n) % 4)] if n % 4 else []))
                self.bucket[:] = [flat, grads, torch.empty_like(flat), n]
        else:
            self.bucket[:] = [torch.cat([g.reshape(-1) for g in grads]), grads]
        return acc

    def exchange_forward(self):
        from piml_amd.sharded import gather_records_into, gather_records_async
        own_rows = self.state_all.detach()[self.b0:self.b0 + self.n_own] if self.emulate else self.state_all.detach()
        if self.p2p is not None:                  # every rank's block stored into every peer's buffer, copied out in rank order
            self.p2p[0].exchange(bcast_src=self.state_own.detach().view(-1), out_bcast=own_rows.view(-1), sum=False)
        elif self.emulate:
            gather_records_into(own_rows, self.state_own, self.group)
        elif self.pre is not None:                  # started, not awaited: `pre` runs under it (run / capture wait for it)
            self.gather_work = gather_records_async(self.state_all, self.state_own, self.b0, self.group)
        else:
            gather_records_into(self.state_all, self.state_own, self.group)

    def exchange_backward(self):
        from piml_amd.sharded import reduce_scatter_grad, unflatten_gradients
        N, b0, n_own = self.N, self.b0, self.n_own
        g_state = self.state_all.grad[b0:b0 + n_own] if self.emulate else self.state_all.grad      # (emulate: this rank's block only)
        if self.p2p is not None:
            # ONE launch: the partial d/d(state) rows of every owner's block to THAT owner + this rank's weight-gradient bucket to
            # everybody, both added in rank order on arrival (the same sums on every rank: bit-reproducible)
            flat, grads, out, n = self.bucket
            if grads is None:             # in place on the gradients' own buffers
                self.p2p[1].exchange(scatter_src=g_state.reshape(-1), bcast_src=flat, out_scatter=self.grad_own.view(-1),
                                     out_bcast=flat, sum=True)
            else:
                self.p2p[1].exchange(scatter_src=g_state.reshape(-1), bcast_src=flat, out_scatter=self.grad_own.view(-1),
                                     out_bcast=out, sum=True)
                unflatten_gradients(out[:n], grads)
        elif self.exchange == 'bucket':
            dist.all_reduce(self.bucket[0], op=dist.ReduceOp.SUM, group=self.group)
            self.grad_own.copy_(self.bucket[0][:N * 6].view(N, 6)[b0:b0 + n_own])      # this rank's rows of d/d(state)
            unflatten_gradients(self.bucket[0][N * 6:], self.bucket[1])
        else:
            reduce_scatter_grad(g_state, self.group, out=self.grad_own)      # (all-reduce + slice on gloo: CPU tests)
            dist.all_reduce(self.bucket[0], op=dist.ReduceOp.SUM, group=self.group)
            unflatten_gradients(self.bucket[0], self.bucket[1])

    def step_body(self, timer=None):
        """One forward + backward pass of the hot path over the scene, eagerly."""
        if self.emulate:                      # (no autograd all-gather with absent peers: the captured step's own pieces, eagerly)
            with self.model.packed_weights():
                self.exchange_forward()
                self.rest_local(*self.features_local())
                self.exchange_backward()
            return None
        with self.model.packed_weights():     # the weight pack rides as trailing workgroups of the relfeat forward launch (PIML_DEFER_PACK)
            if timer is not None:
                timer.start()
            feats = self.features()
            if timer is not None:
                timer.stop()
            return self.rest(*feats)

    def reset_grads(self):
        self.state_own.grad = None
        if self.state_all is not None:
            self.state_all.grad = None
        for p in self.params:
            p.grad = None

    def barrier(self):
        if self.use_dist and self.group is not None:
            dist.barrier(group=self.group)
        torch.cuda.synchronize()

    # ---- whole-step HIP graph (removes ~60 per-kernel launch gaps); eager fallback ----
    def capture(self):
        import piml_amd
        if not self.want_graph:
            return
        if not piml_amd.hip_graphs_safe():      # DEBUG_CLR_GRAPH_PACKET_CAPTURE could not be set in time: replays are not trusted
            print('[bench] HIP-graph capture disabled (piml_amd.hip_graphs_safe() is False); running eagerly', file=sys.stderr)
            return
        ok = 1
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    self.reset_grads()
                    self.step_body()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.reset_grads()
            # ONE graph holds the step's compute (relfeat forward ... relfeat backward).  ROCm cannot record
            # events inside a captured graph, so the HIP events that time single kernels bracket extra eager
            # launches of them (same inputs, same output buffers) in front of sampled replays.
            graph = torch.cuda.CUDAGraph()
            if self.use_dist and self.p2p is None:
                self.exchange_forward()
                torch.cuda.synchronize()
            # the process group's watchdog thread polls the events of finished collectives: under the default 'global'
            # capture mode such a query from another thread invalidates the capture (seen: hipErrorStreamCaptureUnsupported)
            cap_mode = 'thread_local' if (self.use_dist and self.group is not None) else 'global'
            if self.overlap:
                import contextlib
                pre = torch.cuda.CUDAGraph()
                with contextlib.ExitStack() as packed:
                    with torch.cuda.graph(pre, capture_error_mode=cap_mode):
                        packed.enter_context(self.model.packed_weights())      # the pack launch belongs to `pre`
                        local = self.features_local_part()
                    with torch.cuda.graph(graph, pool=pre.pool(), capture_error_mode=cap_mode):
                        feats = self.features_remote_part(local)
                        self.rest_local(*feats)
                self.pre = pre
            else:
                with torch.cuda.graph(graph, capture_error_mode=cap_mode), self.model.packed_weights():
                    if self.p2p is not None:      # the exchanges are launches of this library: INSIDE the graph
                        self.exchange_forward()
                    feats = self.features_local() if self.use_dist else self.features()
                    (self.rest_local if self.use_dist else self.rest)(*feats)
                    if self.p2p is not None:
                        self.exchange_backward()
            self.static_feats = feats     # the captured step's feature / index buffers stay alive
            if self.pre is not None:
                self.pre.replay()
            graph.replay()
            if self.use_dist and self.p2p is None:
                self.exchange_backward()
            done = torch.cuda.Event()
            done.record()
            t_wait = time.perf_counter()
            while not done.query():       # watchdog: a replay that never completes must not hang the run
                if time.perf_counter() - t_wait > 30.0:
                    print('[bench] FATAL: captured step did not complete within 30 s (deadlocked kernels); '
                          're-run with --graph 0 or --two-streams 0', file=sys.stderr, flush=True)
                    os._exit(3)
                time.sleep(0.001)
            torch.cuda.synchronize()
            self.graph = graph
        except Exception as ex:   # noqa: BLE001 - any capture problem means: run eagerly
            print(f'[bench] HIP-graph capture unavailable ({type(ex).__name__}: {ex}); running eagerly',
                  file=sys.stderr)
            ok, self.graph = 0, None
        if self.use_dist and self.group is not None:   # all ranks must run the same mode
            t = torch.tensor([ok], device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
            if int(t.item()) == 0:
                self.graph = None
        self.mode = 'hipgraph' if self.graph is not None else 'eager'

    def trace_stages(self, samples=5, queue=24):
        """Live time of every launch stage of the step: `queue` replays of the captured step keep the GPU busy, then the same
        step runs EAGERLY with the library's stage trace open (piml_trace_*: a HIP event behind every stage launch).  The
        interval between two marks is the GPU time of the later stage.  Median over `samples`.  Single-GPU steps only."""
        import ctypes
        per = {}
        cal = ctypes.c_char_p(b'event_pair_overhead')
        for _ in range(samples):
            for _ in range(queue):
                self.run()
            self.reset_grads()
            tr = self._lib.StageTrace()
            tr.start()
            self.step_body()
            # two marks with nothing in between: what an interval costs by itself (subtracted below, like the relfeat sample's)
            self._lib.lib().piml_trace_mark(cal, torch.cuda.current_stream().cuda_stream)
            self._lib.lib().piml_trace_mark(cal, torch.cuda.current_stream().cuda_stream)
            marks = tr.stop()
            over = marks[-1][1] if marks and marks[-1][0] == 'event_pair_overhead' else 0.0
            for name, us in marks[:-2]:
                per.setdefault(name, []).append(max(us - over, 0.0))
            per.setdefault('event_pair_overhead', []).append(over)
            torch.cuda.synchronize()
        return {k: median(v) for k, v in per.items()}

    def relaunch_relfeat(self):
        """The relfeat forward kernel of the captured step once more, eagerly, on the same buffers."""
        src = self.state_all if self.use_dist else self.state_own
        self.ops.relative_features_packed_into(self.static_feats, src, self.dest_own, self.obstacles, self.b0, self.n_own)

    def run(self):
        """One step (exchange + compute), the way the timed region runs it."""
        if self.graph is not None:
            if self.p2p is not None:              # ONE graph: exchange, compute, exchange
                self.graph.replay()
                return
            if self.use_dist:
                self.exchange_forward()
            if self.pre is not None:
                self.pre.replay()                 # under the all-gather
                self.gather_work.wait()           # the stream waits for the other ranks' records
            self.graph.replay()
            if self.use_dist:
                self.exchange_backward()
        else:
            self.reset_grads()
            self.step_body()

    SPIN_REPLAYS = 600       # untimed replays in front of a secondary leg's timed steps (captured graphs only; the same count on every
                             # rank): a leg of 50 steps behind an idle GPU otherwise measures the clock ramp -- the in-line train-mode
                             # figure read 0.168 ms where a run of its own (--train-mode 1, 300 ms of spin-up) reads 0.153

    def time_steps(self, steps, warmup):
        for _ in range(max(warmup, self.SPIN_REPLAYS) if self.graph is not None else warmup):
            self.run()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run()
        self.barrier()
        return time.perf_counter() - t0


def self_launch(args):
    """--gpus G > 1 without a launcher: start torch.distributed.run as a CHILD process (this parent never touches
    the GPU), one rank per GPU, and hand its exit code back.  Rank 0 of the child writes the JSON line to the
    inherited stdout."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print('[bench] launching: ' + ' '.join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


# Latency + bytes model of the two backward exchanges over xGMI (one hop, fully connected): a collective costs ALPHA_US
# of launch + link latency, an all-reduce moves 2 (w - 1) / w of its bytes per rank, a reduce-scatter (w - 1) / w, each over
# the w - 1 links of ~50 GB/s effective at these message sizes.  `bucket` is ONE all-reduce of (6 N + P) floats, `rs` a
# reduce-scatter of 6 N floats + an all-reduce of P floats: fewer bytes, one more latency.  (Constants are estimates: no
# multi-GPU node is available to the builder; every sharded run times both forms and reports them under "exchange".)
ALPHA_US, LINK_GBS = 12.0, 50.0


def exchange_cost_us(kind, n_agents, n_params, world):
    if world < 2:
        return 0.0
    f = (world - 1) / world
    per_us = LINK_GBS * 1e3 * (world - 1)                    # bytes per microsecond over all links of a rank
    sb, pb = 6 * n_agents * 4, n_params * 4
    if kind == 'bucket':
        return ALPHA_US + 2 * f * (sb + pb) / per_us
    return 2 * ALPHA_US + (f * sb + 2 * f * pb) / per_us


def choose_exchange(n_agents, n_params, world):
    return min(('bucket', 'rs'), key=lambda k: exchange_cost_us(k, n_agents, n_params, world))


def pad_scene_np(scene, world):
    """The scene padded to a multiple of `world` agents with ABSENT agents (NaN position / destination: the reference's own
    encoding, src/data/data.py:141-143; piml_amd.sharded.pad_scene): they select nobody, nobody selects them, their
    gradient is zero -- the padded scene computes exactly the unpadded one.  Returns (scene, padded agent count)."""
    n = scene['position'].shape[0]
    pad = (-n) % world
    if pad == 0:
        return scene, n
    out = dict(scene)
    for k in ('position', 'destination'):
        out[k] = np.concatenate([scene[k], np.full((pad, 2), np.nan, np.float32)])
    for k in ('velocity', 'acceleration'):
        out[k] = np.concatenate([scene[k], np.zeros((pad, 2), np.float32)])
    out['desired_speed'] = np.concatenate([scene['desired_speed'], np.zeros((pad, scene['desired_speed'].shape[1]), np.float32)])
    return out, n + pad


def median(xs):
    xs = sorted(xs)
    return 0.0 if not xs else (xs[len(xs) // 2] if len(xs) % 2 else 0.5 * (xs[len(xs) // 2 - 1] + xs[len(xs) // 2]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--spinup-ms', type=float, default=300.0,
                    help='untimed replays of the step BEFORE the --warmup steps, until this much wall time has passed: a '
                         'few-millisecond timed region (--steps 20) otherwise measures the clock ramp of a GPU that idled '
                         'through set-up and capture, not the step (0.258 vs 0.246 ms/step measured); 0 disables')
    ap.add_argument('--scaling', choices=('auto', 'strong', 'weak'), default='auto',
                    help='auto: one GPU = cfg3 (--agents agents); several GPUs = strong scaling of ONE --scene-agents '
                         'scene (cfg4).  weak: --agents focal agents per GPU')
    ap.add_argument('--agents', type=int, default=4096, help='focal agents per GPU (one GPU, or --scaling weak)')
    ap.add_argument('--scene-agents', type=int, default=16384, help='agents of the whole scene under strong scaling')
    ap.add_argument('--obstacles', type=int, default=2000)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='0 disables the cpu_baseline leg')
    ap.add_argument('--graph', type=int, default=1, help='replay the step from a captured HIP graph')
    ap.add_argument('--secondary', type=int, default=1, help='also report MLAPM step / relfeat backward kernel figures')
    ap.add_argument('--force-dist', type=int, default=0, help='exercise the sharded (RCCL) code path even with one rank')
    ap.add_argument('--two-streams', type=int, default=1, help='obstacle branch of the MLP on a side stream')
    ap.add_argument('--verify', type=int, default=1, help='after the timed region compare the replayed step with an eager autograd step')
    ap.add_argument('--exchange', choices=('auto', 'bucket', 'rs', 'p2p'), default='auto',
                    help='backward exchange of the sharded step: one all-reduce of [state gradient | weight gradients] '
                         '(bucket) or reduce-scatter(state gradient) + all-reduce(weight gradients) (rs); auto = the cheaper '
                         'one under choose_exchange()\'s latency + bytes model (the other one is timed after the timed region)')
    ap.add_argument('--overlap', type=int, default=0,
                    help='sharded + graph: start the all-gather, run the part of the step that needs only the own block '
                         '(weight pack, local half of the neighbour search, obstacle branch) under it, then the rest.  Off '
                         'by default: with one rank the second graph launch and the second relfeat launch cost 24 us more '
                         'than they hide (0.292 vs 0.268 ms/step); both forms are timed and reported under "exchange"')
    ap.add_argument('--leg-timeout', type=float, default=180.0,
                    help='sharded runs: seconds the informational legs after the timed region may take before the line is '
                         'written without them')
    ap.add_argument('--exchange-compare', type=int, default=1,
                    help='several GPUs: also time the other --exchange variant after the timed region (informational)')
    ap.add_argument('--strong-baseline', type=int, default=1,
                    help='strong scaling on several GPUs: rank 0 also times the whole scene on one GPU (outside the timed region)')
    ap.add_argument('--train-mode', type=int, default=0,
                    help='1: the model of the timed step in train() mode with the reference\'s --dropout 0.5 (fresh keep-masks '
                         'drawn on the device every step); the replayed step is then not compared with an eager one '
                         '(different masks).  Default 0: eval(); the train-mode step is reported under secondary.train_mode_step')
    ap.add_argument('--emulate-shard', type=int, default=8,
                    help='secondary.cfg4_projection: one rank\'s step of the 16384-agent scene sharded this many ways, measured on this one GPU '
                         '(compute exact, exchanges at world 1) + the same scene on one GPU; 0 disables')
    ap.add_argument('--messages', type=int, default=0,
                    help='1: the model also materialises the per-row messages predictions[1:3] (the drop-in default of a bare model call); '
                         '0 (default): model.messages_wanted = False, what the reference\'s training loops need -- they read predictions[0] '
                         'only unless reg_weight > 0 (src/models/simulators.py:331-347, :702-737) -- which lets the eval-mode / p = 0 '
                         'network run on the agents\' sums of h2 (PIML_POOL_TRAIN)')
    ap.add_argument('--mlp', choices=('fused', 'library'), default='fused',
                    help='fused: the PINNSF network on the hand-written matrix-core kernels (encoder_x3.hip / encoder.hip / decoder.hip); '
                         'library: round 1\'s path, rocBLAS / hipBLASLt GEMMs + HIP glue kernels (A/B comparison)')
    ap.add_argument('--tunableop', type=int, default=1,
                    help='1: load the pre-tuned GEMM selections for the MLP (tuned in-process if this stack rejects the file); '
                         '0: library defaults; 2: re-tune and write --tune-out; 3: force the in-process tuning')
    ap.add_argument('--tune-out', type=str, default='gpurun_out/tunableop_retuned.csv', help='result file of --tunableop 2')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))

    # stdout must carry exactly ONE JSON line: anything libraries print to fd 1 (e.g. RCCL's version
    # banner) is sent to stderr, and the result is written to the saved stdout at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run '
                         f'--nproc-per-node {args.gpus} (or unset WORLD_SIZE: bench.py launches itself)')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    use_dist = world > 1 or bool(args.force_dist)
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    scaling = args.scaling if args.scaling != 'auto' else ('strong' if world > 1 else 'weak')
    if scaling == 'strong':      # any agent count: the scene is padded with absent agents up to a multiple of the world size
        N_real = args.scene_agents
        N = N_real + (-N_real) % world
        n_own = N // world
    else:
        N_real = N = args.agents * world
        n_own = args.agents
    M = args.obstacles
    cfg3_shapes = n_own == 4096        # the row counts the committed GEMM selections were tuned (and validated) for

    # PyTorch-ROCm TunableOp: pick the rocBLAS / hipBLASLt solution per GEMM shape of the PINNSF
    # MLP from a result file tuned once on an MI355X (tuning itself takes minutes and is never
    # done here).  A file whose validators do not match this software stack is ignored by torch.
    from piml_amd import tuning
    import piml_amd.models.model as MODEL
    fused_mlp = args.mlp == 'fused'
    MODEL.FUSED_NETWORK = MODEL.FUSED_ENCODER = fused_mlp
    if fused_mlp and args.tunableop != 2:      # no library GEMM left in the step: nothing to select, one stream
        args.tunableop, args.two_streams = 0, 0
    if args.tunableop == 2:      # re-tune the GEMM selections on this stack (eager steps; not a measurement)
        tuning.tune_begin(os.path.abspath(args.tune_out))
        args.graph, args.two_streams, args.verify, args.secondary, args.cpu_seconds = 0, 0, 0, 0, 0.0
        args.steps, args.warmup = min(args.steps, 3), min(args.warmup, 1)
        gemm_tuning = 'tuning'
    else:
        gemm_tuning = 'tunableop-file' if (args.tunableop == 1 and tuning.load()) else 'default'
    autotune = False
    if cfg3_shapes and (args.tunableop == 3 or (args.tunableop == 1 and gemm_tuning == 'default')):
        # the committed selections belong to another software stack (or --tunableop 3 asks for it): tune the
        # step's GEMM shapes right here (a few seconds, two eager steps below) and keep ONE stream -- only the
        # committed selections are validated for running two library GEMMs concurrently
        print('[bench] NOTE: tuning the GEMM selections for this software stack in-process (the committed '
              'piml_amd/tuning file was not accepted or --tunableop 3)', file=sys.stderr, flush=True)
        tuning.tune_begin(os.path.join('/tmp', f'piml_tunableop_autotuned_{os.getpid()}.csv'))
        autotune, gemm_tuning = True, 'autotuned'

    _phase('tunableop setup done')
    from piml_amd import _lib
    from piml_amd.scenes import synthetic_gc_scene

    scene, N_padded = pad_scene_np(synthetic_gc_scene(N_real, M, seed=args.seed), world if scaling == 'strong' else 1)
    assert N_padded == N
    torch.manual_seed(666)
    n_params = sum(p.numel() for p in getattr(MODEL, 'PINNSF_multitask')(model_args()).parameters())
    if args.exchange == 'auto':
        args.exchange = choose_exchange(N, n_params, world) if use_dist else 'bucket'
        exchange_why = {k: round(exchange_cost_us(k, N, n_params, world), 2) for k in ('bucket', 'rs')}
    else:
        exchange_why = None
    # the P2P-store exchange objects (receive buffers, flags, device-side step counters; IPC handles through the process group's
    # object all-gather): made once, used by the main step and by the comparison legs
    p2p = None

    def make_p2p():
        from piml_amd.sharded import p2p_exchanges

        def _all_bytes(b):
            out = [None] * world
            dist.all_gather_object(out, b)
            return out
        return p2p_exchanges(rank, world, n_own, n_params, _all_bytes)
    if use_dist and (args.exchange == 'p2p' or args.exchange_compare):
        try:
            p2p = make_p2p()
        except Exception as ex:   # noqa: BLE001 - only fatal when it is the exchange that was asked for
            if args.exchange == 'p2p':
                raise
            print(f'[bench] P2P exchange unavailable for the comparison leg ({type(ex).__name__}: {ex})', file=sys.stderr)
        if args.exchange != 'p2p':
            # the comparison leg is all ranks or none: a rank that could not open its peers' buffers must not leave the others waiting
            # for its stores (and for the collective behind the leg)
            ok = torch.tensor([1 if p2p is not None else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                p2p = None
    # The obstacle branch of the MLP on a side stream makes two GEMM chains run concurrently inside the
    # captured graph.  Concurrent library GEMMs are only safe with kernels that never wait for
    # co-residency: hipBLASLt's DEFAULT heuristics pick stream-K style kernels for some shapes
    # and two of them on parallel graph branches deadlock (observed: replay never completes).  The
    # pre-tuned selections are validated for the cfg3 row counts only; otherwise the branches stay on one stream.
    two_streams = (bool(args.two_streams) and gemm_tuning == 'tunableop-file' and cfg3_shapes) or args.two_streams == 2
    if args.exchange == 'p2p' and two_streams:
        print('[bench] --exchange p2p: the side stream is off (the exchange kernel\'s workgroups must be co-resident)', file=sys.stderr)
        two_streams = False
    if args.train_mode:
        args.verify = 0
    st = Step(scene, N, n_own, rank * n_own, M, dev, dist.group.WORLD if use_dist else None, use_dist,
              two_streams, bool(args.graph), exchange=args.exchange, overlap=bool(args.overlap),
              train_mode=bool(args.train_mode), messages=bool(args.messages), p2p=p2p)
    M_eff = st.M_eff

    if autotune:
        for _ in range(2):
            st.reset_grads()
            st.step_body()
        torch.cuda.synchronize()
        torch.cuda.tunable.tuning_enable(False)
        st.reset_grads()
    _phase('scene + model on device')
    st.capture()
    mode, graph = st.mode, st.graph
    _phase(f'capture done, mode={mode}')

    kernel_ms_samples, overhead_ms_samples = [], []
    x3_products = bool(_lib.lib().piml_encoder_products(-1))      # split bf16 products (default) or PIML_ENC_PRODUCTS=f32
    cal_timer = _lib.StreamTimer()
    ev_pairs, sample_timers = [], []
    # The TIMED region holds exactly --steps replays of the step and nothing else (round 4: until then it also held one sample
    # per ten steps -- four event records and an extra relfeat launch each, 2.5 - 4 % of a 20-step region, measured by leaving
    # them out: 0.159 / 0.162 / 0.157 ms/step with, 0.1545 / 0.1558 / 0.1534 without on one box).  The live time of the relfeat
    # forward kernel is sampled right BEHIND the region, in the same stream of replays: five further replays, each followed by
    # ONE extra eager launch of the kernel (same inputs, same output buffers) bracketed by two HIP events -- it runs behind
    # other kernels, not behind an idle gap -- and the cost of the two event records (an empty pair) is subtracted.  The region
    # itself is also bracketed by two HIP events on the stream (`roofline.timed_region_event_ms`).
    TIMED_LAUNCHES = 1
    n_samples = 5
    # (not behind the first replay of a burst: that sample measured a whole step's length more than the others -- 0.13-0.21 ms
    # against 0.024-0.027 -- in every run; the queue in front of it is empty at that point)
    sample_at = set(range(n_samples)) if graph is not None else \
        {min(args.steps - 1, int((i + 0.5) * args.steps / max(1, min(5, args.steps // 10)))) for i in range(max(1, min(5, args.steps // 10)))}
    timer_pool = [(_lib.StreamTimer(), _lib.StreamTimer()) for _ in range(n_samples)] if graph is not None else []   # events are made HERE, not in the timed region
    region_timer = _lib.StreamTimer()

    def run_step(i, timed):
        if graph is not None:
            sample = timed == 'sample' and i in sample_at
            eager_exchange = use_dist and st.p2p is None      # (the P2P exchanges are launches INSIDE the captured graph)
            if eager_exchange:
                st.exchange_forward()
            if st.pre is not None:
                st.pre.replay()                   # the own-block part of the step, under the all-gather
                st.gather_work.wait()
            graph.replay()
            if eager_exchange:
                st.exchange_backward()
            if sample:
                tk, tc = timer_pool[len(sample_timers)]
                tk.start()
                st.relaunch_relfeat()
                tk.stop()
                tc.start(); tc.stop()
                sample_timers.append((tk, tc))
        else:
            st.reset_grads()
            if timed:
                tm = _lib.StreamTimer()
                st.step_body(tm)
                ev_pairs.append(tm)
                if i in sample_at:
                    cal_timer.start(); cal_timer.stop()
                    torch.cuda.current_stream().synchronize()
                    overhead_ms_samples.append(cal_timer.elapsed_ms())
            else:
                st.step_body()

    spin_steps = 0
    if args.spinup_ms > 0:              # clocks up (set-up and capture left the GPU idle); same steps, never timed
        t_spin = time.perf_counter()
        more = True
        while more:
            for _ in range(32):
                run_step(0, False)
            torch.cuda.synchronize()    # bounded queue depth
            spin_steps += 32
            more = (time.perf_counter() - t_spin) * 1e3 < args.spinup_ms
            if use_dist:                # every rank leaves the loop after the same chunk (the exchange is collective)
                t = torch.tensor([1 if more else 0], device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                more = bool(t.item())
    for i in range(args.warmup):
        run_step(i, False)
    st.barrier()
    _phase('warmup done')
    t0 = time.perf_counter()
    region_timer.start()
    for i in range(args.steps):
        run_step(i, True)
    region_timer.stop()
    st.barrier()
    elapsed = time.perf_counter() - t0
    _phase('timed region done')
    region_event_ms = region_timer.elapsed_ms()
    if st.p2p is not None and not p2p_alive(st.p2p, dev, use_dist):
        # the headline would be a step with its exchanges skipped: no line rather than a wrong one
        print('[bench] FATAL: the P2P exchange timed out during the run (sticky status word set on some rank): the timed replays '
              'skipped it.  Raise PIML_P2P_SPIN_LIMIT if the ranks start far apart, or use --exchange bucket.', file=sys.stderr, flush=True)
        sys.exit(5)
    if graph is not None:           # the relfeat samples, behind the region (see above); two unsampled replays lead the burst
        for i in range(-2, n_samples):
            run_step(i, 'sample')
        st.barrier()
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- integrity check, outside the timed region: the gradients the replayed step left behind must equal
    # those of the same step run eagerly through plain autograd (sharded: with the autograd all-gather /
    # reduce-scatter of piml_amd.sharded), i.e. no work was skipped or mis-wired by the capture ----
    verify_err = None
    if graph is not None and args.verify:
        got_state = (st.grad_own if use_dist else st.state_own.grad).clone()
        got_params = [None if p.grad is None else p.grad.clone() for p in st.params]
        st.reset_grads()
        if hasattr(torch.autograd.graph, 'set_warn_on_accumulate_grad_stream_mismatch'):
            # the warm-up steps ran on a side stream, this one runs on the default stream: intended
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        st.step_body()
        torch.cuda.synchronize()

        def rel(a, b):
            a, b = torch.nan_to_num(a), torch.nan_to_num(b)
            return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))
        verify_err = rel(got_state, st.state_own.grad)
        for g, p in zip(got_params, st.params):
            if (g is None) != (p.grad is None):
                verify_err = float('inf')
            elif g is not None:
                verify_err = max(verify_err, rel(g, p.grad))
        if use_dist:
            t = torch.tensor([verify_err], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            verify_err = float(t.item())
        if not verify_err < 1e-3:
            print(f'[bench] FATAL: replayed step and eager autograd step disagree (max rel err {verify_err:.3e})',
                  file=sys.stderr, flush=True)
            sys.exit(4)
    _phase(f'verify done ({verify_err})')

    if graph is None:
        kernel_ms_samples = [tm.elapsed_ms() for tm in ev_pairs]
    else:
        kernel_ms_samples = [tk.elapsed_ms() for tk, _ in sample_timers]
        overhead_ms_samples = [tc.elapsed_ms() for _, tc in sample_timers]
    # median over the sampled launches: robust against the occasional preempted / cold sample
    launches_per_sample = TIMED_LAUNCHES if graph is not None else 1
    raw_ms = median(kernel_ms_samples)      # median of the intervals around `launches_per_sample` launches
    _phase('relfeat samples (ms): ' + ', '.join(f'{v:.4f}' for v in kernel_ms_samples))
    overhead_ms = median(overhead_ms_samples)
    kernel_ms = max(raw_ms - overhead_ms, 1e-6) / launches_per_sample
    pairs_step = N_real * (N_real + M_eff)             # all ranks together (absent padding agents are not counted)
    ms_per_step = elapsed / args.steps * 1e3
    # SURVEY.md 8d, the contract figure: operand-stream bytes of ONE step over all ranks (24 B per ped-ped
    # pair, 8 B per ped-obstacle pair, 488 B per focal agent) / step time / (HBM peak x GPUs)
    bytes_step = N_real * (24 * N_real + 8 * M_eff) + 488 * N_real
    achieved = bytes_step / (ms_per_step * 1e-3) / 1e9 / world        # GB/s per GPU
    kernel_bytes = n_own * (24 * N + 8 * M_eff) + 488 * n_own         # this rank's relfeat launch

    prof = None
    pname = next((f for f in ('r06_step_counters.json', 'r05_step_counters.json', 'r04_step_counters.json', 'r03_step_counters.json') if os.path.exists(os.path.join(ROOT, 'profiles', f))), None)
    if world == 1 and pname:
        pj = json.load(open(os.path.join(ROOT, 'profiles', pname)))
        if pj.get('config', {}).get('agents_total') == N and pj['config'].get('obstacle_points') == M_eff:
            prof = pj
    prof_src = f'profiles/{pname} (rocprofv3 --pmc passes of the same command, committed: STATIC, not measured in this run)'

    # every launch stage of the step, timed LIVE (outside the timed region): HIP events between the stage launches of an
    # eager step queued behind replays of the captured one (Step.trace_stages)
    stage_us = {}
    if world == 1 and fused_mlp and graph is not None and not use_dist:
        try:
            stage_us = st.trace_stages()
        except Exception as ex:   # noqa: BLE001 - informational
            print(f'[bench] stage trace unavailable ({type(ex).__name__}: {ex})', file=sys.stderr)
    _phase('stage trace done')
    x3 = '_x3' if x3_products else ''
    dw = 'dw'
    one_pass = False
    if x3_products and fused_mlp:      # the layer-split weight-gradient kernel (encoder_dw2.hip) above the few-rows bound
        from piml_amd import _lib as _plib
        if _plib.lib().piml_encoder_dw2(-1) == 1 and N * (6 + 10) // 32 > _plib.lib().piml_encoder_split_tiles_train(-1):
            dw = 'dw2'
            # ... and on top of it the one-pass backward (encoder_bwd3.hip): dX chain + every weight gradient in the `enc_bwd_dx`
            # stage, the `enc_bwd_dw` stage launches nothing (PIML_ENC_FUSED_BWD / PIML_ENC_FUSED_DW3 switch it back)
            one_pass = _plib.lib().piml_encoder_fused_bwd(-1) >= 1 and os.environ.get('PIML_RELU_MASK', '1') != '0'
    stage_kernel = {'pinnsf_pack': 'pinnsf_pack_kernel', 'relfeat_fwd': 'relfeat_fwd_kernel', 'enc_fwd': f'enc_fwd{x3}_kernel',
                    'dec_fwd_head': 'dec_fwd_head_kernel', 'dec_bwd': 'dec_bwd_kernel', 'enc_bwd_dx': f'enc_bwd_dx{x3}_kernel',
                    'enc_bwd_dw': (f'enc_bwd_{dw}_x3_kernel' if dw == 'dw2' else ('enc_bwd_dw_x3w_kernel' if x3 else 'enc_bwd_dw_kernel')), 'pinnsf_reduce': 'pinnsf_reduce_kernel', 'relfeat_bwd': 'relfeat_bwd_kernel'}
    if 'pinnsf_reduce' not in stage_us and stage_us:      # the slot sums rode in the relfeat backward's launch (ops.deferred_slot_sums)
        stage_kernel['relfeat_bwd'] = 'relfeat_bwd_reduce_kernel'
    sums_path = 'enc_fwd_sum' in stage_us         # PIML_POOL_TRAIN: the network ran on the agents' sums of h2 (--messages 0, eval mode)
    if sums_path:
        stage_kernel.update({'enc_fwd_sum': 'enc_fwd_sum_x3_kernel', 'dec_fwd_head_sum': 'dec_fwd_head_sum_kernel', 'pinnsf_unfold': 'pinnsf_unfold_kernel'})
    if 'dec_fwd_head_sum' in stage_us:            # also PIML_POOL_MSGS (a dropout step that returns no messages): the decoder launch on sums
        stage_kernel['dec_fwd_head_sum'] = 'dec_fwd_head_sum_kernel'
    if one_pass:
        stage_kernel['enc_bwd_dx'] = 'enc_bwd_fused_x3_kernel'
        if os.environ.get('PIML_ENC_FUSED_DW3', '1') != '0':
            stage_kernel['enc_bwd_dw'] = '(no launch: dW3 is phase 2 of enc_bwd_fused_x3_kernel)'
    if sums_path:
        try:        # round 6: the sums path's backward as two crews of four waves (encoder_bwd5.hip) unless PIML_ENC_SUMS_BWD=1
            from piml_amd import _lib as _plib2
            if _plib2.lib().piml_encoder_sums_bwd(0) == 2:
                stage_kernel['enc_bwd_dx'] = 'enc_bwd_sums2_kernel'
        except Exception:   # noqa: BLE001 - an older library: the name stays
            pass
    # kernels whose f32 products run as six bf16 products (priced against the bf16 matrix pipe AND the HBM ceiling)
    split_kernels = {'enc_fwd_x3_kernel', 'enc_fwd_sum_x3_kernel', 'enc_bwd_dx_x3_kernel', 'enc_bwd_dw2_x3_kernel', 'enc_bwd_dw_x3w_kernel', 'enc_bwd_fused_x3_kernel',
                     'enc_bwd_sums2_kernel'}
    static = {e['name']: e for e in (prof or {}).get('all_step_kernels', [])}
    if os.environ.get('PIML_DEC_BWD_SPLIT', '1') != '0':       # decoder backward as (tile, branch) workgroups (the default)
        stage_kernel['dec_bwd'] = 'dec_bwd_split_kernel'
    live_src = ('live: HIP events between the stage launches of an eager step queued behind 24 replays of the captured '
                f'step, median of 5, minus the cost of an empty event interval ({stage_us.get("event_pair_overhead", 0.0):.1f} us) '
                '(piml_trace_*); includes the dispatch gap in front of the kernel, which rocprofv3\'s kernel duration does not')

    if True:      # (every rank assembles the line; rank 0 writes it)
        kernels = [{'name': 'relfeat_fwd_kernel', 'us': kernel_ms * 1e3, 'share_of_step': kernel_ms / ms_per_step,
                    'us_source': f'live: HIP events around {len(kernel_ms_samples)} x {launches_per_sample} eager re-launches, each behind a '
                                 'replay of the step, queued right behind the timed region (which holds the --steps replays and '
                                 'nothing else); median, event-pair overhead subtracted',
                    'us_stage_trace': stage_us.get('relfeat_fwd'),
                    'bound': 'valu', 'frac': (prof or {}).get('relfeat_fwd_kernel', {}).get('valu_busy_frac'),
                    'frac_source': ('SQ_ACTIVE_INST_VALU share of the SIMD cycles, ' + prof_src) if prof else None,
                    'hbm_bytes': (prof or {}).get('relfeat_fwd_kernel', {}).get('hbm_bytes_per_launch'),
                    'hbm_bytes_source': prof_src if prof else None,
                    'operand_stream_bytes': kernel_bytes,
                    'event_interval_us': raw_ms * 1e3, 'event_pair_overhead_us': overhead_ms * 1e3}]
        for stage, us in sorted(stage_us.items(), key=lambda kv: -kv[1]):
            if stage in ('relfeat_fwd', 'event_pair_overhead'):
                continue
            kname = stage_kernel.get(stage, stage)
            if kname.startswith('(no launch') and us < 3.0:
                continue
            e = {'name': kname, 'us': us, 'share_of_step': us / (ms_per_step * 1e3), 'us_source': live_src}
            sk = static.get(kname)
            if sk:       # counters are static; every fraction is recomputed from the LIVE duration
                e['hbm_bytes'] = sk.get('hbm_bytes')
                e['hbm_frac'] = sk['hbm_bytes'] / (us * 1e-6) / (HBM_PEAK_GBS * 1e9) if sk.get('hbm_bytes') else None
                if kname in split_kernels and sk.get('flops'):
                    # executed bf16 flops: 6 x the 128 x 128 layers' (recorded with the counters; older profiles: two layers per row)
                    bf16 = sk.get('executed_bf16_flops') or 6.0 * sk['flops'] * (2 * 128 * 128) / (2 * 128 * 128 + 6 * 128)
                    e['mfma_frac'] = bf16 / (us * 1e-6) / 2.5e15
                    e['bound'] = 'hbm' if (e['hbm_frac'] or 0) >= e['mfma_frac'] else 'mfma'
                    e['frac'] = max(e['hbm_frac'] or 0, e['mfma_frac'])
                elif sk.get('flops'):
                    e['bound'], e['frac'] = 'mfma', sk['flops'] / (us * 1e-6) / (F32_MFMA_PEAK_TFS * 1e12)
                elif sk.get('bound') == 'hbm':       # moves its bytes at more than half of the HBM peak: priced against that ceiling
                    e['bound'], e['frac'], e['valu_busy_frac'] = 'hbm', e['hbm_frac'], sk.get('valu_busy_frac')
                else:
                    e['bound'], e['valu_busy_frac'] = sk.get('bound'), sk.get('frac')
                e['counters_source'] = prof_src
            kernels.append(e)
        # registers / spills of the kernels named above, read from the code objects inside the library this process loaded
        try:
            usage = _lib.kernel_resource_usage()
        except Exception as ex:   # noqa: BLE001 - informational
            usage = {}
        for e in kernels:
            vs = {k: v for k, v in usage.items() if k.split('<')[0] == e['name']}
            if vs:
                e['registers'] = {'variants_in_library': len(vs), 'vgprs_max': max(v['vgprs'] for v in vs.values()),
                                  'agprs_max': max(v['agprs'] or 0 for v in vs.values()),
                                  'vgpr_spill_max': max(v['vgpr_spill'] for v in vs.values()),
                                  'scratch_bytes_max': max(v['scratch_bytes'] for v in vs.values())}
        if not stage_us:      # no live trace (sharded / eager / library MLP): the committed profile's entries, marked as such
            for k in (prof or {}).get('other_kernels', []):
                kernels.append(dict(k, us_source=prof_src))
        out = {
            'metric': 'agent-pair force evals/sec + simulated steps/sec, 4096-agent GC scene',
            'value': pairs_step * args.steps / elapsed, 'unit': 'pairs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'spinup_steps': spin_steps,
            'ms_per_step': ms_per_step, 'steps_per_s': args.steps / elapsed,
            'agent_steps_per_s': N_real * args.steps / elapsed,
            'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'launch_mode': mode, 'verified_max_rel_err': verify_err,
            'model_mode': 'train() dropout 0.5' if args.train_mode else 'eval()',
            'mlp': (('fused matrix-core kernels (piml_amd/csrc/encoder_x3.hip: f32 products as six bf16 products of exact '
                     'three-way splits, f32 accumulation -- closer to float64 than the f32 matrix instruction, '
                     'tests/test_encoder_gpu.py; decoder.hip: f32 matrix instruction)' if x3_products else
                     'fused f32-MFMA kernels (piml_amd/csrc/encoder.hip, decoder.hip)') if fused_mlp else
                    f'library GEMMs ({gemm_tuning}) + HIP glue kernels, {2 if two_streams else 1} stream(s)'),
            'config': {'workload': ('cfg3: synthetic 4096-agent GC scene' if (world == 1 and N == 4096) else
                                    f'cfg4: synthetic {N}-agent GC scene sharded over {world} GPUs' if scaling == 'strong' else
                                    f'synthetic {N}-agent GC scene ({n_own} focal agents per GPU)') +
                                   ', forward+backward PINSF step (HIP relfeat fwd/bwd + PINNSF_multitask fwd/bwd)',
                       'agents_per_gpu': n_own, 'agents_total': N_real, 'obstacle_points': M_eff,
                       'pairs_per_step': pairs_step, 'topk_ped': 6, 'topk_obs': 10,
                       'messages': ('materialised (--messages 1: model(...)[1:3] returned, the message path)' if args.messages else
                                    'on request only (model.messages_wanted = False: every output and gradient the step reads is '
                                    'produced; the neighbour-axis sum runs in front of the encoders\' last layer, which is folded '
                                    'into the decoders\' first layer -- DESIGN.md 5, tests/test_sums_gpu.py; --messages 1 times the other form)' +
                                    (' [train mode: a dropout mask sits between that layer and the sum -- the forward runs its last layer with exchanged operands and leaves the agents\' sums of the MESSAGES, rows stored for the collision head only (PIML_POOL_MSGS); plain backward]' if args.train_mode else '')),
                       'sharding': 'single GPU' if not use_dist else
                       f'agent blocks over {world} ranks, all-gather(p,v,a) + one all-reduce(state grad + weight grads) per step'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'timed_region_event_ms': region_event_ms,
                         'timed_region_event_note': 'HIP events on the launch stream around the --steps replays of the timed region '
                                                    '(max over ranks of the wall clock between the barriers is what `value` uses)',
                         'traffic': (prof or {}).get('step_hbm_bytes'),
                         'algorithmic_bytes': bytes_step,
                         'definition': 'SURVEY.md 8d step-level contract: operand-stream bytes of one step (24 B/ped pair + '
                                       '8 B/obstacle pair + 488 B/focal agent) / ms_per_step / (n_gpus x HBM peak); the sources '
                                       'are LDS/L2 resident, so real HBM traffic (`traffic`, PMC) is far below this model and '
                                       'the step is bound by the MLP kernels (HBM traffic of the saved activations / matrix pipe) and VALU issue, see `kernels`',
                         'kernels': kernels,
                         'kernels_with_vgpr_spills': sorted(k for k, v in usage.items() if v['vgpr_spill']),
                         'kernels_with_vgpr_spills_note': 'every kernel of libpiml_hip.so with a non-zero .vgpr_spill_count in its code '
                                                          'object (kernels[].registers: the same source; vgprs = .vgpr_count, which counts '
                                                          'architectural + accumulation registers together); both are A/B forms the default dispatch does not launch '
                                                          '(PIML_DEC_BWD_SPLIT=0, PIML_ENC_FUSED_BWD=2; profiles/r05_kernel_usage.md)'},
        }
    import threading
    written = threading.Lock()

    def write_line(note=None):
        """rank 0 writes THE json line exactly once (normal end, or the watchdog below)."""
        if not written.acquire(blocking=False):
            return
        if note:
            out['informational_legs'] = note
        if rank == 0:
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + '\n').encode())

    # The legs below re-capture and re-run the step in other forms on every rank: a mismatch between ranks would hang in
    # a collective for ever.  They are informational; the headline line must survive them.
    leg_timer = None
    if use_dist:
        def bail():
            print(f'[bench] informational legs did not finish within {args.leg_timeout:.0f} s: writing the line without '
                  'them and leaving', file=sys.stderr, flush=True)
            write_line(f'timed out after {args.leg_timeout:.0f} s')
            os._exit(0)
        leg_timer = threading.Timer(args.leg_timeout, bail)
        leg_timer.daemon = True
        leg_timer.start()

    # ---- the other backward-exchange variant, all ranks, outside the timed region (informational) ----
    exchange_other = None
    if use_dist and args.exchange_compare and (world > 1 or args.force_dist):
        exchange_other = []
        legs = [(e, False) for e in ('bucket', 'rs', 'p2p') if e != args.exchange and (e != 'p2p' or p2p is not None)]
        if args.exchange != 'p2p':
            legs.append((args.exchange, not args.overlap))
        for exch, ovl in legs:
            tag = {'exchange': exch, 'overlap': ovl}
            try:
                alt = Step(scene, N, n_own, rank * n_own, M, dev, dist.group.WORLD, True, two_streams and exch != 'p2p', bool(args.graph),
                           exchange=exch, overlap=ovl, messages=bool(args.messages), p2p=p2p)
                alt.capture()
                k = max(10, min(args.steps, 50))
                el = alt.time_steps(k, 5)
                t = torch.tensor([el], device=dev, dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                if exch == 'p2p' and not p2p_alive(p2p, dev, True):
                    exchange_other.append(dict(tag, error='P2P exchange timed out (sticky status word set on some rank): the replays '
                                                          'skipped it, no time recorded'))
                    del alt
                    for e in p2p:
                        e.close()
                    p2p = make_p2p()          # (collective: every rank rebuilds -- the MIN above made the verdict common)
                    continue
                exchange_other.append(dict(tag, ms_per_step=float(t.item()) / k * 1e3, steps=k, launch_mode=alt.mode))
                del alt
            except Exception as ex:   # noqa: BLE001 - informational
                exchange_other.append(dict(tag, error=f'{type(ex).__name__}: {ex}'))

    # ---- strong scaling: the same scene on ONE GPU (rank 0, the others wait), outside the timed region ----
    same_scene_1gpu = None
    if world > 1 and scaling == 'strong' and args.strong_baseline:
        if rank == 0:
            try:
                one = Step(scene, N, N, 0, M, dev, None, False, False, bool(args.graph), messages=bool(args.messages))
                one.capture()
                k = max(10, min(args.steps, 50))
                el1 = one.time_steps(k, 5)
                same_scene_1gpu = {'ms_per_step': el1 / k * 1e3, 'steps': k, 'launch_mode': one.mode,
                                   'note': 'the whole scene on one GPU of this node (rank 0, after the timed region): '
                                           'strong-scaling efficiency = this / (n_gpus x ms_per_step)'}
                del one
            except Exception as ex:   # noqa: BLE001 - informational
                same_scene_1gpu = {'error': f'{type(ex).__name__}: {ex}'}
        st.barrier()

    # ---- secondary figures, measured after (outside) the timed region, rank 0 only ----
    secondary = None
    if rank == 0 and args.secondary:
        try:
            secondary = secondary_measurements(scene, n_own, dev, _lib, prof, light=world > 1)
        except Exception as ex:   # noqa: BLE001 - informational figures must never cost the headline line
            secondary = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1 and fused_mlp and not use_dist and args.emulate_shard:
            try:
                secondary['cfg4_projection'] = cfg4_projection(args, dev, bool(args.messages), shards=args.emulate_shard)
            except Exception as ex:   # noqa: BLE001 - informational
                secondary['cfg4_projection'] = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1 and fused_mlp:      # live evidence that the split products ARE f32 arithmetic: both forms against float64
            try:
                secondary['encoder_products_vs_float64'] = encoder_products_check(dev, _lib)
            except Exception as ex:   # noqa: BLE001 - informational
                secondary['encoder_products_vs_float64'] = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1 and x3_products and fused_mlp:      # the same step with the encoder products on v_mfma_f32_32x32x2_f32
            try:
                _lib.lib().piml_encoder_products(0)
                f32s = Step(scene, N, N, 0, M, dev, None, False, False, bool(args.graph))
                f32s.capture()
                k = max(10, min(args.steps, 50))
                el = f32s.time_steps(k, 5)
                secondary['f32_matrix_instruction_step'] = {
                    'ms_per_step': el / k * 1e3, 'steps': k, 'launch_mode': f32s.mode,
                    'note': 'PIML_ENC_PRODUCTS=f32: the encoder layers on v_mfma_f32_32x32x2_f32 (an fmaf chain) instead of '
                            'split bf16 products; both forms are f32 arithmetic (max error against float64: split 1.5e-7 .. 5e-7, '
                            'f32 instruction 2.2e-7 .. 5.4e-7 over outputs and gradients, tools/probe_x3.py)'}
                del f32s
            except Exception as ex:   # noqa: BLE001 - informational
                secondary['f32_matrix_instruction_step'] = {'error': f'{type(ex).__name__}: {ex}'}
            finally:
                _lib.lib().piml_encoder_products(1)
        if world == 1 and fused_mlp:      # the reference's TRAINING configuration: model.train(), --dropout 0.5
            for key, mname in (('train_mode_step', 'PINNSF_multitask'), ('train_mode_pinnsf_bm_step', 'PINNSF_bottleneck_multitask')):
                try:
                    tr = Step(scene, N, N, 0, M, dev, None, False, False, bool(args.graph), model_name=mname, train_mode=True)
                    tr.capture()
                    k = max(10, min(args.steps, 50))
                    el = tr.time_steps(k, 5)
                    secondary[key] = {
                        'ms_per_step': el / k * 1e3, 'steps': k, 'launch_mode': tr.mode, 'dropout': 0.5,
                        'note': f'the same forward + backward step with {mname} in train() mode, dropout 0.5 (the reference\'s '
                                'training configuration, src/main.py:45, src/models/simulators.py:311): the encoder forward kernel draws '
                                'the keep-masks itself (Philox4x32-10, one call per row, device-side draw counter: a fresh mask on every '
                                'replay), applies them in its epilogue and leaves them as bits for the dX chain and the dW staging; '
                                'parity with injected masks and bit-exact masks: tests/test_dropout_gpu.py'}
                    del tr
                except Exception as ex:   # noqa: BLE001 - informational
                    secondary[key] = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1 and fused_mlp and not args.messages:
            # the reference forward's literal output list [predictions, ped_msgs, obs_msgs, pred_collision]
            # (src/models/model.py:1301-1305): the message path, every per-row tensor materialised
            try:
                ms = Step(scene, N, N, 0, M, dev, None, False, False, bool(args.graph), messages=True)
                ms.capture()
                k = max(10, min(args.steps, 50))
                el = ms.time_steps(k, 5)
                secondary['messages_step'] = {
                    'ms_per_step': el / k * 1e3, 'steps': k, 'launch_mode': ms.mode,
                    'note': '--messages 1: the same step with model.messages_wanted = True -- model(...)[1:3] (the per-neighbour messages, '
                            'src/models/model.py:1301-1305) written and their gradient path kept: three encoder layers, the message-path '
                            'decoder launch and the one-pass backward with its W3^T layer and dW3 phase (no fold, no sums path)'}
                del ms
            except Exception as ex:   # noqa: BLE001 - informational
                secondary['messages_step'] = {'error': f'{type(ex).__name__}: {ex}'}
        if world == 1:       # the same step with the shipped experiments' model (src/configs/exp_configs/piml-*.yaml)
            try:
                bm = Step(scene, N, N, 0, M, dev, None, False, False, bool(args.graph), model_name='PINNSF_bottleneck_multitask')
                bm.capture()
                k = max(10, min(args.steps, 50))
                el = bm.time_steps(k, 5)
                secondary['pinnsf_bm_step'] = {
                    'ms_per_step': el / k * 1e3, 'steps': k, 'launch_mode': bm.mode,
                    'note': '`--model pinnsf_bm` (decoder + predictor per NEIGHBOUR row): the same forward + backward step; '
                            'fused encoders (one-pass backward, encoder_bwd3.hip) + fused row decoder (piml_rowdecoder_*) + its '
                            '64->64->1 collision head on matrix cores (head64.hip): no library GEMM; 1.145 ms/step with the '
                            'decoders on library GEMMs (PIML_FUSED_ROW_DECODER=0)'}
                del bm
            except Exception as ex:   # noqa: BLE001 - informational
                secondary['pinnsf_bm_step'] = {'error': f'{type(ex).__name__}: {ex}'}

    if leg_timer is not None:
        leg_timer.cancel()
    if use_dist:
        # proof that the collective library saw `world` ranks: every rank contributes (rank, device index) to one all-gather on
        # the GPU (backend "nccl" = RCCL over xGMI) and the line carries what came back
        seen = torch.empty(world, 2, dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(seen, torch.tensor([[rank, torch.cuda.current_device()]], dtype=torch.int32, device=dev))
        seen = seen.cpu().tolist()
        out['exchange'] = {'ranks_seen': len({r for r, _ in seen}), 'ranks': seen, 'backend': dist.get_backend(),
                           'backward': args.exchange, 'overlap': bool(st.pre is not None),
                           'model_cost_us': exchange_why, 'padded_agents': N - N_real,
                           'other_variants': exchange_other,
                           'note': 'bucket = one all-reduce of [d/d(state) (N,6) | weight gradients]; rs = '
                                   'reduce-scatter(d/d(state)) + all-reduce(weight gradients); forward = one '
                                   'all-gather of the (p,v,a) records, issued eagerly around the captured compute; overlap = the '
                                   'all-gather is started, the own-block part of the step (weight pack, local half of '
                                   'the neighbour search, obstacle branch) replayed under it, then the rest'}
    if same_scene_1gpu is not None:
        out['single_gpu_same_scene'] = same_scene_1gpu
    if secondary is not None:
        # every form of the step against the same SURVEY 8d contract bytes: the headline does not get to choose
        for key, rk in (('train_mode_step', 'frac_train_mode'), ('messages_step', 'frac_messages'), ('pinnsf_bm_step', 'frac_pinnsf_bm'),
                        ('train_mode_pinnsf_bm_step', 'frac_train_mode_pinnsf_bm'), ('f32_matrix_instruction_step', 'frac_f32_matrix_instruction')):
            leg = secondary.get(key)
            if isinstance(leg, dict) and leg.get('ms_per_step'):
                leg['frac'] = bytes_step / (leg['ms_per_step'] * 1e-3) / 1e9 / HBM_PEAK_GBS
                out['roofline'][rk] = leg['frac']
        out['roofline']['frac_forms_note'] = ('frac = the default form of the step (eval(), messages on request only); frac_train_mode = '
                                              'model.train() with dropout 0.5 (the reference\'s training configuration, src/main.py:45); '
                                              'frac_messages = the reference forward\'s full output list materialised (src/models/model.py:1301-1305); '
                                              'frac_pinnsf_bm = the shipped experiments\' model (src/configs/exp_configs/piml-gcdata.yaml:49-50); all '
                                              'against the same algorithmic_bytes (secondary.<form>_step.ms_per_step)')
        out['secondary'] = secondary
        shard = (secondary.get('relfeat_shard_shape') or {}).get('fwd_us')
        if shard is not None and out['roofline']['kernels'] and out['roofline']['kernels'][0]['name'] == 'relfeat_fwd_kernel':
            out['roofline']['kernels'][0]['shard_shape_us'] = shard      # 2048 focal rows x 16384 sources (secondary.relfeat_shard_shape)
    if rank == 0 and args.cpu_seconds > 0 and world == 1:
        try:
            out['cpu_baseline'] = cpu_baseline(scene, N, M_eff, args.cpu_seconds)
        except Exception as ex:   # noqa: BLE001 - e.g. no C compiler for the oracle on this host
            out['cpu_baseline'] = {'error': f'{type(ex).__name__}: {ex}'}
    elif world > 1:
        out['cpu_baseline'] = None
    write_line()
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
