"""GPU: the RCCL ("nccl" backend) side of agent-block sharding on ONE device -- a 1-rank process group, so that the
all-gather / reduce-scatter / all-reduce branches of piml_amd/sharded.py execute on hardware (the world-size-2
logic is covered on the CPU with gloo in tests/test_sharded.py; real 2/4/8-GPU runs are the driver's SCALE bench)."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist

from piml_amd.scenes import synthetic_gc_scene

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def model_args():
    return types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128,
        processor_hidden_size=128, decoder_hidden_size=64, encoder_hidden_layers=3,
        processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5, activation='relu',
        dataset_name='gc1560')


@pytest.fixture(scope='module')
def nccl_group():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    created = not dist.is_initialized()
    if created:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                                device_id=torch.device(DEV))
    yield dist.group.WORLD
    if created:
        dist.destroy_process_group()


def _scene(N, M, seed):
    sc = synthetic_gc_scene(N, M, seed=seed)
    rng = np.random.default_rng(seed)
    acc = (rng.standard_normal((N, 2)) * 0.3).astype(np.float32)
    state = torch.tensor(np.concatenate((sc['position'], sc['velocity'], acc), -1), device=DEV)
    return state, [torch.tensor(sc[k], device=DEV) for k in ('destination', 'desired_speed', 'obstacles')]


def test_sharded_model_step_one_rank_nccl_matches_unsharded(nccl_group):
    """ShardedScene.model_step through RCCL all_gather_into_tensor (forward) and reduce_scatter_tensor (backward),
    then the bucketed all-reduce of the weight gradients, against the same step without any collective."""
    from piml_amd import ops
    from piml_amd.models.model import PINNSF_multitask
    from piml_amd.sharded import ShardedScene, allreduce_gradients
    assert dist.get_backend(nccl_group) == 'nccl'
    N, M = 2048, 500
    state, (dest, v0, obs) = _scene(N, M, 3)
    torch.manual_seed(1)
    model = PINNSF_multitask(model_args()).to(DEV).eval()
    w = torch.linspace(-1, 1, N * 2, device=DEV).reshape(N, 2)

    def run(sharded):
        for p in model.parameters():
            p.grad = None
        s = state.clone().requires_grad_(True)
        if sharded:
            sh = ShardedScene(N, obs, group=nccl_group, force_collectives=True)
            assert (sh.begin, sh.count, sh.world) == (0, N, 1)
            acc = sh.model_step(model, s, dest, v0)[0]
        else:
            pf, of, df = ops.relative_features_packed(s, dest, obs, 0, N)
            acc = model(pf, of, torch.cat((df, s[:, 2:4], s[:, 4:6], v0), -1))[0]
        (acc * w).sum().backward()
        if sharded:
            allreduce_gradients(list(model.parameters()), nccl_group)
        torch.cuda.synchronize()
        return acc.detach(), s.grad.clone(), [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    a1, g1, p1 = run(True)
    a0, g0, p0 = run(False)
    assert torch.equal(torch.nan_to_num(a1), torch.nan_to_num(a0))
    # relfeat backward accumulates with float atomics: equal up to summation order
    assert (torch.nan_to_num(g1) - torch.nan_to_num(g0)).abs().max() <= 1e-5 * max(1.0, float(torch.nan_to_num(g0).abs().max()))
    for x, y in zip(p1, p0):
        assert (x is None) == (y is None)
        if x is not None:
            assert (x - y).abs().max() <= 1e-5 * max(1.0, float(y.abs().max()))


def test_overlapped_model_step_one_rank_nccl_matches_serial(nccl_group):
    """ShardedScene.model_step_overlapped (async in-place all-gather; weight pack + local half of the neighbour search +
    obstacle branch under it; wait; remote half + network) against model_step: same accelerations bit for bit, same
    gradients up to the summation order of the relfeat backward's atomics."""
    from piml_amd.models.model import PINNSF_multitask
    from piml_amd.sharded import ShardedScene, allreduce_gradients
    N, M = 2048, 500
    state, (dest, v0, obs) = _scene(N, M, 4)
    torch.manual_seed(2)
    model = PINNSF_multitask(model_args()).to(DEV).eval()
    w = torch.linspace(-1, 1, N * 2, device=DEV).reshape(N, 2)
    sh = ShardedScene(N, obs, group=nccl_group, force_collectives=True)

    def run(overlapped):
        for p in model.parameters():
            p.grad = None
        s = state.clone().requires_grad_(True)
        step = sh.model_step_overlapped if overlapped else sh.model_step
        acc = step(model, s, dest, v0)[0]
        (acc * w).sum().backward()
        allreduce_gradients(list(model.parameters()), nccl_group)
        torch.cuda.synchronize()
        return acc.detach(), s.grad.clone(), [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    a1, g1, p1 = run(True)
    a0, g0, p0 = run(False)
    assert torch.equal(torch.nan_to_num(a1), torch.nan_to_num(a0))
    assert (torch.nan_to_num(g1) - torch.nan_to_num(g0)).abs().max() <= 1e-5 * max(1.0, float(torch.nan_to_num(g0).abs().max()))
    for x, y in zip(p1, p0):
        assert (x is None) == (y is None)
        if x is not None:
            assert (x - y).abs().max() <= 1e-5 * max(1.0, float(y.abs().max()))


def test_exchange_pair_around_local_step_one_rank_nccl(nccl_group):
    """The eager exchange pair bench.py issues either side of the captured compute graph: all-gather into a static
    (N, 6) leaf, then reduce_scatter_grad of its .grad -- with RCCL, 1 rank."""
    from piml_amd import ops
    from piml_amd.sharded import gather_records_into, reduce_scatter_grad
    N, M = 1024, 200
    state, (dest, v0, obs) = _scene(N, M, 5)
    state_all = torch.zeros(N, 6, device=DEV).requires_grad_(True)
    gather_records_into(state_all, state, nccl_group)
    assert torch.equal(torch.nan_to_num(state_all.detach()), torch.nan_to_num(state))
    pf, of, df = ops.relative_features_packed(state_all, dest, obs, 0, N)
    (pf.sum() + of.sum() * 0.5 + df.sum() * 0.25).backward()
    g_own = reduce_scatter_grad(state_all.grad, nccl_group)
    torch.cuda.synchronize()
    assert torch.equal(g_own, state_all.grad)


def test_direct_rccl_comm_through_c_abi_one_rank():
    """piml_comm_* / piml_allgather_state / piml_reducescatter_grad / piml_allreduce_sum (include/piml_hip.h): a
    communicator owned by libpiml_hip.so, 1 rank on one GPU -- the collectives degenerate to copies, which is what
    RCCL must deliver."""
    from piml_amd.rccl import DirectComm
    torch.cuda.set_device(0)
    comm = DirectComm()
    assert (comm.world, comm.rank) == (1, 0)
    own = torch.randn(2048, 6, device=DEV)
    full = torch.zeros(2048, 6, device=DEV)
    comm.all_gather_into(full, own)
    g = torch.randn(2048, 6, device=DEV)
    back = comm.reduce_scatter(g)
    buf = torch.randn(134277 + 2048 * 6, device=DEV)
    want = buf.clone()
    comm.all_reduce(buf)
    torch.cuda.synchronize()
    assert torch.equal(full, own) and torch.equal(back, g) and torch.equal(buf, want)
    comm.close()
