"""CPU-only, world_size 2 / 4 / 8, gloo: agent-block sharding (piml_amd/sharded.py) -- the all-gather of
state records, the reduce-scatter of their gradients and the bucketed all-reduce of the MLP
gradients -- with the CPU oracle standing in for the HIP kernel (injected feature_fn)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from piml_amd.scenes import synthetic_gc_scene

N, M, WORLD = 64, 40, 2


class OracleFeatures(torch.autograd.Function):
    """relative_features_packed's contract, computed by the oracle on the CPU."""

    @staticmethod
    def forward(ctx, state, dest_rows, obstacles, f0, fc):
        from oracle import oracle as O
        s = state.detach().numpy()
        dest_full = np.full((s.shape[0], 2), np.nan, np.float32)
        dest_full[f0:f0 + fc] = dest_rows.detach().numpy()
        out = O.relfeat_fwd(s[None, :, 0:2], s[None, :, 2:4], s[None, :, 4:6], dest_full[None],
                            obstacles.numpy(), return_index=True)
        ctx.meta = (out[3][0, f0:f0 + fc], out[4][0, f0:f0 + fc], s, dest_full, f0, fc)
        return tuple(torch.tensor(np.ascontiguousarray(x[0, f0:f0 + fc])) for x in out[:3])

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_dest):
        from oracle import oracle as O
        pi, oi, s, dest_full, f0, fc = ctx.meta
        n = s.shape[0]

        def full(g, shape):
            z = np.zeros((n,) + shape, np.float32)
            z[f0:f0 + fc] = g.numpy()
            return z
        pif = np.full((n,) + pi.shape[1:], -1, np.int32); pif[f0:f0 + fc] = pi
        oif = np.full((n,) + oi.shape[1:], -1, np.int32); oif[f0:f0 + fc] = oi
        gp, gv, ga, gd = O.relfeat_bwd(full(g_ped, tuple(g_ped.shape[1:])), full(g_obs, tuple(g_obs.shape[1:])),
                                       full(g_dest, (2,)), pif, oif, s[:, 0:2], dest_full)
        return torch.tensor(np.concatenate((gp, gv, ga), -1)), torch.tensor(gd[f0:f0 + fc]), None, None, None


def feature_fn(state_full, dest_rows, obstacles, f0, fc):
    return OracleFeatures.apply(state_full, dest_rows, obstacles, f0, fc)


def model_args():
    return types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=32,
        processor_hidden_size=32, decoder_hidden_size=16, encoder_hidden_layers=2,
        processor_hidden_layers=2, decoder_hidden_layers=2, dropout=0.0, activation='relu',
        dataset_name='gc1560')


def scene_tensors(n_real=N):
    """the first n_real agents of the 64-agent scene (61: a count that no world size divides)"""
    sc = synthetic_gc_scene(N, M, seed=4)
    rng = np.random.default_rng(0)
    acc = (rng.standard_normal((N, 2)) * 0.3).astype(np.float32)
    state = torch.tensor(np.concatenate((sc['position'], sc['velocity'], acc), -1))
    return state[:n_real], torch.tensor(sc['destination'])[:n_real], torch.tensor(sc['desired_speed'])[:n_real], torch.tensor(sc['obstacles'])


def reference_single_process(n_real=N, world=1):
    """the scene in ONE process, padded like the sharded run pads it (an absent agent still is a row of the network: its
    bias-driven output feeds the weight gradients; that the padding changes nothing for the real agents is
    test_pad_scene_keeps_the_result)"""
    from piml_amd.models.model import PINNSF_multitask
    from piml_amd.sharded import pad_scene
    state, dest, v0, obs = scene_tensors(n_real)
    state, dest, v0, _ = pad_scene(state, dest, v0, world)
    n_pad = state.shape[0]
    state = state.clone().requires_grad_(True)
    torch.manual_seed(1)
    model = PINNSF_multitask(model_args()).eval()
    pf, of, df = feature_fn(state, dest, obs, 0, n_pad)
    acc = model(pf, of, torch.cat((df, state[:, 2:4], state[:, 4:6], v0), -1))[0]
    w = torch.linspace(-1, 1, n_pad * 2).reshape(n_pad, 2)
    (acc * w).sum().backward()
    return acc.detach(), state.grad.clone(), [p.grad.clone() if p.grad is not None else None for p in model.parameters()]


def worker(rank, port, q, WORLD=WORLD, n_real=N):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from piml_amd.models.model import PINNSF_multitask
        from piml_amd.sharded import ShardedScene, agent_block, allreduce_gradients, pad_scene
        state, dest, v0, obs = scene_tensors(n_real)
        state, dest, v0, n_back = pad_scene(state, dest, v0, WORLD)         # absent agents up to a multiple of the world size
        N = state.shape[0]
        assert n_back == n_real and N % WORLD == 0
        sh = ShardedScene(N, obs, feature_fn=feature_fn)
        assert (sh.begin, sh.count) == agent_block(N, rank, WORLD) == (rank * N // WORLD, N // WORLD)
        state_own = sh.own(state).clone().requires_grad_(True)
        torch.manual_seed(1)
        model = PINNSF_multitask(model_args()).eval()
        acc = sh.model_step(model, state_own, sh.own(dest), sh.own(v0))[0]
        w = torch.linspace(-1, 1, N * 2).reshape(N, 2)
        (acc * sh.own(w)).sum().backward()
        params = list(model.parameters())
        allreduce_gradients(params, sh.group)
        first = (acc.detach().numpy(), state_own.grad.numpy().copy(),
                 [None if p.grad is None else p.grad.numpy().copy() for p in params])

        # the same step with the exchange pair used around a captured compute graph (bench.py):
        # plain all-gather into a static leaf, local forward + backward, reduce-scatter of leaf.grad
        from piml_amd.sharded import gather_records_into, reduce_scatter_grad
        for p in params:
            p.grad = None
        state_all = torch.zeros(N, 6).requires_grad_(True)
        gather_records_into(state_all, state_own, sh.group)
        assert np.array_equal(state_all.detach().numpy(), state.numpy(), equal_nan=True)
        pf, of, df = feature_fn(state_all, sh.own(dest), obs, sh.begin, sh.count)
        own = sh.own(state_all)
        acc2 = model(pf, of, torch.cat((df, own[:, 2:4], own[:, 4:6], sh.own(v0)), -1))[0]
        (acc2 * sh.own(w)).sum().backward()
        g_own = reduce_scatter_grad(state_all.grad, sh.group)
        allreduce_gradients(params, sh.group)
        assert np.allclose(acc2.detach().numpy(), first[0], rtol=1e-6, atol=1e-7)
        assert np.allclose(g_own.numpy(), first[1], rtol=1e-5, atol=1e-6)
        for p, ref in zip(params, first[2]):
            assert (p.grad is None) == (ref is None)
            if ref is not None:
                assert np.allclose(p.grad.numpy(), ref, rtol=1e-5, atol=1e-6)
        # the async in-place exchange of the overlapped step: own rows in place at once, the others after wait();
        # its autograd form reduce-scatters like the blocking one
        from piml_amd.sharded import gather_records_async, _AllGatherRecordsAsync
        buf = torch.full((N, 6), -7.0)
        work = gather_records_async(buf, state_own, sh.begin, sh.group)
        assert np.array_equal(buf[sh.begin:sh.begin + sh.count].numpy(), state_own.detach().numpy(), equal_nan=True)
        work.wait()
        assert np.array_equal(buf.numpy(), state.numpy(), equal_nan=True)
        holder = []
        s2 = state_own.detach().clone().requires_grad_(True)
        full = _AllGatherRecordsAsync.apply(s2, sh.group, holder)
        holder[0].wait()
        assert np.array_equal(full.detach().numpy(), state.numpy(), equal_nan=True)
        gw = torch.arange(N * 6, dtype=torch.float32).reshape(N, 6) * (rank + 1)
        (torch.nan_to_num(full) * gw).sum().backward()
        want = (torch.arange(N * 6, dtype=torch.float32).reshape(N, 6) * sum(range(1, WORLD + 1)))[sh.begin:sh.begin + sh.count]
        assert np.allclose(torch.nan_to_num(s2.grad).numpy(),
                           torch.where(torch.isnan(state_own.detach()), torch.zeros_like(want), want).numpy())
        q.put((rank,) + first)
    finally:
        dist.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.parametrize('WORLD,n_real', [(2, 64), (4, 64), (8, 61)])
def test_sharded_step_matches_single_process(oracle, WORLD, n_real):
    """every rank's rows of the sharded step against the same scene in one process: world 2 / 4 / 8, and an agent count (61)
    that no world size divides -- padded with absent agents, which nobody selects"""
    acc_ref, gstate_ref, gparams_ref = reference_single_process(n_real, WORLD)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, port, q, WORLD, n_real)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = {}
    for _ in range(WORLD):
        r = q.get(timeout=120)
        results[r[0]] = r[1:]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = (n_real + WORLD - 1) // WORLD
    for rank in range(WORLD):
        acc, gstate, gparams = results[rank]
        real = max(0, min(n, n_real - rank * n))                 # this rank's rows that are agents of the unpadded scene
        rows = slice(rank * n, rank * n + real)
        assert np.allclose(acc, acc_ref[rank * n:(rank + 1) * n].numpy(), rtol=1e-5, atol=1e-6)
        # reduce-scatter of the partial d/d(state): every source's gradient from ALL ranks' focal rows
        assert np.allclose(np.nan_to_num(gstate), np.nan_to_num(gstate_ref[rank * n:(rank + 1) * n].numpy()), rtol=1e-4, atol=1e-5)
        # (an absent agent's row: no gradient through the neighbour search -- positions and accelerations stay zero)
        assert (np.nan_to_num(gstate[real:])[:, [0, 1, 4, 5]] == 0).all()
        for g, ref in zip(gparams, gparams_ref):
            assert (g is None) == (ref is None)
            if g is not None:
                assert np.allclose(g, ref.numpy(), rtol=1e-4, atol=1e-5)


def test_agent_block_requires_divisibility():
    from piml_amd.sharded import agent_block
    assert agent_block(16384, 3, 8) == (6144, 2048)
    with pytest.raises(ValueError):
        agent_block(10, 0, 4)


def test_pad_scene_keeps_the_result(oracle):
    """A scene whose agent count is not a multiple of the world size is padded with absent (NaN) agents; the features
    and gradients of the real agents are unchanged."""
    from piml_amd.sharded import agent_block, pad_scene
    state, dest, v0, obs = scene_tensors()
    state, dest, v0 = state[:61], dest[:61], v0[:61]
    ps, pd, pv, n = pad_scene(state, dest, v0, 4)
    assert n == 61 and ps.shape == (64, 6) and agent_block(64, 3, 4) == (48, 16)
    assert torch.isnan(ps[61:, :2]).all() and (ps[61:, 2:] == 0).all() and torch.isnan(pd[61:]).all()
    a = state.clone().requires_grad_(True)
    b = ps.clone().requires_grad_(True)
    fa = feature_fn(a, dest, obs, 0, 61)
    fb = feature_fn(b, pd, obs, 0, 64)
    for x, y in zip(fa, fb):
        assert torch.equal(torch.nan_to_num(x), torch.nan_to_num(y[:61]))
        assert (torch.nan_to_num(y[61:]) == 0).all()
    sum(x.sum() for x in fa).backward()
    sum(torch.nan_to_num(y).sum() for y in fb).backward()
    assert torch.allclose(torch.nan_to_num(a.grad), torch.nan_to_num(b.grad[:61]), atol=1e-6)
    assert (torch.nan_to_num(b.grad[61:]) == 0).all()
