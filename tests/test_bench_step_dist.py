"""CPU-only, world_size 2 / 4 / 8, gloo: the exchange bookkeeping of bench.Step -- the class the driver's multi-GPU run executes --
with the CPU oracle standing in for the HIP feature operators (injected `ops_module`): the captured-step path's
exchange_forward / features_local / rest_local / exchange_backward in both backward exchanges (`bucket`, `rs`) and the
overlapped ordering (exchange started, own-block part, wait, remote part), each against the same step in ONE process.
Also the scene padding for agent counts that are not a multiple of the world size."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO

N, M, WORLD = 64, 40, 2


def _ops_stub():
    """piml_amd.ops' feature operators for bench.Step, computed by the oracle on the CPU."""
    from test_sharded import OracleFeatures

    class Stub:
        @staticmethod
        def relative_features_packed_self(state, dest_rows, obstacles, v0_rows, b0, n_own, return_index=False, local=None):
            pf, of, df = OracleFeatures.apply(state, dest_rows, obstacles, b0, n_own)
            own = state[b0:b0 + n_own]
            self_features = torch.cat((df, own[:, 2:4], own[:, 4:6], v0_rows), -1)
            if local is not None:
                assert local == ('local', b0, n_own)              # the token of the own-block part comes back
            out = (pf, of, self_features)
            return out + (None, None) if return_index else out

        @staticmethod
        def relative_features_local_part(state, dest_rows, obstacles, v0_rows, b0, n_own):
            # legal while the other ranks' rows are in flight: reads the own block only
            assert torch.isfinite(torch.nan_to_num(state[b0:b0 + n_own])).all()
            return ('local', b0, n_own)
    return Stub


def _scene(n_real=N):
    """the first n_real agents of the 64-agent scene"""
    sys.path.insert(0, REPO)
    from piml_amd.scenes import synthetic_gc_scene
    sc = synthetic_gc_scene(N, M, seed=4)
    sc['acceleration'] = (np.random.default_rng(0).standard_normal((N, 2)) * 0.3).astype(np.float32)
    return {k: (v[:n_real] if getattr(v, 'shape', (0,))[0] == N else v) for k, v in sc.items()}


def _single_process_reference(scene):
    import bench
    N = scene['position'].shape[0]
    st = bench.Step(scene, N, N, 0, M, torch.device('cpu'), None, False, False, False, ops_module=_ops_stub())
    st.reset_grads()
    st.step_body()
    return st.state_own.grad.clone(), [None if p.grad is None else p.grad.clone() for p in st.params]


def worker(rank, port, q, WORLD=WORLD, n_real=N):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        import bench
        scene, N = bench.pad_scene_np(_scene(n_real), WORLD)      # what bench.py does for --gpus G (absent agents up to a multiple)
        n_own = N // WORLD
        results = {}
        for exchange, overlap in (('bucket', False), ('rs', False), ('bucket', True), ('rs', True)):
            st = bench.Step(scene, N, n_own, rank * n_own, M, torch.device('cpu'), dist.group.WORLD, True, False, False,
                            exchange=exchange, overlap=overlap, ops_module=_ops_stub())
            for rep in range(2):                                  # twice: the static buffers are reused like a replay would
                st.reset_grads()
                if overlap:
                    st.pre = object()                             # "a captured own-block graph exists": exchange_forward starts the gather
                    st.exchange_forward()
                    local = st.features_local_part()
                    st.gather_work.wait()
                    feats = st.features_remote_part(local)
                else:
                    st.exchange_forward()
                    feats = st.features_local()
                st.rest_local(*feats)
                st.exchange_backward()
            results[(exchange, overlap)] = (st.grad_own.clone(), [None if p.grad is None else p.grad.clone() for p in st.params])
        # the eager path of the same class (autograd all-gather, bucketed all-reduce)
        st = bench.Step(scene, N, n_own, rank * n_own, M, torch.device('cpu'), dist.group.WORLD, True, False, False,
                        ops_module=_ops_stub())
        st.reset_grads()
        st.step_body()
        results['eager'] = (st.state_own.grad.clone(), [None if p.grad is None else p.grad.clone() for p in st.params])
        q.put((rank, {k: (v[0].numpy(), [None if g is None else g.numpy() for g in v[1]]) for k, v in results.items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('WORLD,n_real', [(2, 64), (4, 64), (8, 61)])
def test_bench_step_exchanges_match_single_process(oracle, WORLD, n_real):
    """bench.Step on 2 / 4 / 8 ranks in every exchange order against the same (padded) scene in one process (61 agents: padded to
    64 for 8 ranks, nobody selects the absent agents)"""
    sys.path.insert(0, REPO)
    import bench
    gstate_ref, gparams_ref = _single_process_reference(bench.pad_scene_np(_scene(n_real), WORLD)[0])
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=worker, args=(r, port, q, WORLD, n_real)) for r in range(WORLD)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(WORLD))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = (n_real + WORLD - 1) // WORLD
    for rank in range(WORLD):
        assert len(got[rank]) == 5
        for key, (gstate, gparams) in got[rank].items():
            real = max(0, min(n, n_real - rank * n))
            ref = torch.nan_to_num(gstate_ref[rank * n:(rank + 1) * n]).numpy()
            assert np.allclose(np.nan_to_num(gstate), ref, rtol=1e-4, atol=1e-5), (rank, key)
            assert (np.nan_to_num(gstate[real:])[:, [0, 1, 4, 5]] == 0).all(), (rank, key)     # nobody selects an absent agent
            for g, r in zip(gparams, gparams_ref):
                assert (g is None) == (r is None), (rank, key)
                if g is not None:
                    assert np.allclose(g, r.numpy(), rtol=1e-4, atol=1e-5), (rank, key)


def test_scene_padding_and_exchange_model():
    sys.path.insert(0, REPO)
    import bench
    sc = _scene()
    sc = {k: (v[:61] if getattr(v, 'shape', (0,))[0] == N else v) for k, v in sc.items()}
    padded, n = bench.pad_scene_np(sc, 4)
    assert n == 64 and padded['position'].shape == (64, 2) and np.isnan(padded['position'][61:]).all()
    assert np.isnan(padded['destination'][61:]).all() and (padded['velocity'][61:] == 0).all()
    assert bench.pad_scene_np(sc, 61)[1] == 61                     # already a multiple: untouched
    # the byte / latency model: one rank has nothing to exchange, small scenes prefer ONE collective, huge ones fewer bytes
    assert bench.exchange_cost_us('bucket', 16384, 134277, 1) == 0.0
    assert bench.choose_exchange(16384, 134277, 8) == 'bucket'
    assert bench.choose_exchange(4_000_000, 134277, 8) == 'rs'
