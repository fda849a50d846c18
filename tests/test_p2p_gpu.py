"""GPU: the P2P-store all-gather (piml_amd/csrc/p2p.hip, piml_amd/p2p.py) between TWO PROCESSES sharing the one
GPU of the test box -- IPC handles work across processes on the same device, so the store / flag / two-parity protocol is
exercised for real (RCCL refuses two ranks on one device; an 8-GPU node is not available to the builder).  Every step's
gathered records are compared with the concatenation the RCCL all-gather would give; a missing peer must end in the error
flag, not in a hang.  Children are started fresh with multiprocessing `spawn` (nothing re-executes a process that touched the
GPU)."""
import multiprocessing as mp
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(rank, world, fpr, steps, conns, q, uneven):
    sys.path.insert(0, ROOT)
    import torch
    from piml_amd.p2p import P2PExchange
    try:
        torch.cuda.set_device(0)
        ex = P2PExchange(rank, world, fpr)
        for c in conns:
            c.send(ex.handles())
        for peer, c in zip([r for r in range(world) if r != rank], conns):
            ex.connect(peer, c.recv())
        for c in conns:                      # everybody has opened everybody's buffers
            c.send(b'ready')
        for c in conns:
            c.recv()
        worst = 0.0
        out = torch.empty(world * fpr, device='cuda')
        for s in range(steps):
            g = torch.Generator().manual_seed(1000 * s)
            blocks = [torch.randn(fpr, generator=torch.Generator().manual_seed(1000 * s + r)) for r in range(world)]
            own = blocks[rank].cuda()
            if uneven and rank == 1 and s % 3 == 0:          # a late rank: the peers' polls have to wait for it
                torch.cuda._sleep(20_000_000)
            ex.step(own)
            ex.gather_into(out)
            want = torch.cat(blocks).cuda()
            worst = max(worst, float((out - want).abs().max()))
            if not ex.ok():
                q.put((rank, 'timeout at step %d' % s))
                return
        # a step nobody else takes part in: the poll must run out and raise the flag
        if rank == 0:
            for c in conns:
                c.send(b'done')
            ex.step(own, spin_limit=2000)
            lost = not ex.ok()
        else:
            for c in conns:
                c.recv()
            lost = None
        q.put((rank, worst, lost))
        if rank != 0:
            import time
            time.sleep(1.0)                  # keep the buffers mapped until rank 0's lonely step has timed out
        ex.close()
    except Exception as e:   # noqa: BLE001
        q.put((rank, 'error: %s: %s' % (type(e).__name__, e)))


@pytest.mark.parametrize('uneven', [False, True])
def test_p2p_allgather_between_two_processes_on_one_gpu(uneven):
    world, fpr, steps = 2, 2048 * 6, 12                  # a rank's block of the 16384-agent scene on 8 GPUs: 2048 agents x 6 floats = 49 KB
    ctx = mp.get_context('spawn')
    a, b = ctx.Pipe()
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(0, world, fpr, steps, [a], q, uneven)),
             ctx.Process(target=_child, args=(1, world, fpr, steps, [b], q, uneven))]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r = q.get(timeout=240)
            res[r[0]] = r[1:]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank in range(world):
        assert rank in res, f'rank {rank} did not report'
        assert not isinstance(res[rank][0], str), f'rank {rank}: {res[rank][0]}'
        assert res[rank][0] == 0.0, f'rank {rank}: gathered records differ by {res[rank][0]}'
    assert res[0][1] is True, 'a step without the peer must raise the time-out flag'
    print(f'p2p all-gather, 2 processes on one GPU, {steps} steps of {fpr * 4} B per rank, uneven={uneven}: bit-exact; lonely step flagged')


def _child_step(rank, world, conn, q):
    """One rank of the FULL sharded bench step (2 x 2048 focal rows of a 4096-agent scene) on the P2P-store exchange, replayed from
    ONE captured graph (exchange, compute, exchange), against the same step in one process."""
    sys.path.insert(0, ROOT)
    os.environ['PIML_P2P_SPIN_LIMIT'] = '10000000'           # the two processes start seconds apart (imports, compilation caches)
    import torch
    try:
        torch.cuda.set_device(0)
        import bench
        from piml_amd.scenes import synthetic_gc_scene
        from piml_amd.sharded import p2p_exchanges
        dev = torch.device('cuda', 0)
        N, M = 4096, 2000
        n_own = N // world
        scene = synthetic_gc_scene(N, M, seed=0)

        def all_bytes(b):                                     # two ranks: swap over the pipe
            conn.send(b)
            other = conn.recv()
            return [b, other] if rank == 0 else [other, b]
        torch.manual_seed(666)
        import piml_amd.models.model as MODEL
        n_params = sum(p.numel() for p in MODEL.PINNSF_multitask(bench.model_args()).parameters())
        p2p = p2p_exchanges(rank, world, n_own, n_params, all_bytes)
        st = bench.Step(scene, N, n_own, rank * n_own, M, dev, None, True, False, True, exchange='p2p', p2p=p2p)
        conn.send(b'ready'); conn.recv()
        st.capture()
        assert st.mode == 'hipgraph', st.mode
        for _ in range(3):
            st.run()
        torch.cuda.synchronize()
        ok = all(e.ok() for e in p2p)
        got_state = st.grad_own.clone()
        got_params = [None if p.grad is None else p.grad.clone() for p in st.params]
        # the same scene in ONE process, eager autograd
        ref = bench.Step(scene, N, N, 0, M, dev, None, False, False, False)
        ref.model.load_state_dict(st.model.state_dict())
        ref.reset_grads()
        ref.step_body()
        torch.cuda.synchronize()
        rel = lambda a, b: float((torch.nan_to_num(a) - torch.nan_to_num(b)).abs().max() / torch.nan_to_num(b).abs().max().clamp_min(1e-12))
        worst = rel(got_state, ref.state_own.grad[rank * n_own:(rank + 1) * n_own])
        for g, p in zip(got_params, ref.params):
            if (g is None) != (p.grad is None):
                worst = float('inf')
            elif g is not None:
                worst = max(worst, rel(g, p.grad))
        conn.send(b'done'); conn.recv()                       # nobody unmaps a buffer a peer may still write
        q.put((rank, worst, ok))
        for e in p2p:
            e.close()
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, 'error: %s: %s\n%s' % (type(e).__name__, e, traceback.format_exc())))


def test_sharded_step_on_p2p_exchange_two_processes_one_gpu():
    """bench.Step over two ranks sharing the one GPU, forward all-gather and backward reduce both on the P2P-store exchange and
    INSIDE the captured graph: state-gradient rows and every weight gradient against the single-process step."""
    world = 2
    ctx = mp.get_context('spawn')
    a, b = ctx.Pipe()
    q = ctx.Queue()
    procs = [ctx.Process(target=_child_step, args=(0, world, a, q)), ctx.Process(target=_child_step, args=(1, world, b, q))]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r = q.get(timeout=400)
            res[r[0]] = r[1:]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for rank in range(world):
        assert rank in res, f'rank {rank} did not report'
        assert not isinstance(res[rank][0], str), f'rank {rank}: {res[rank][0]}'
        assert res[rank][1] is True, f'rank {rank}: a wait timed out'
        assert res[rank][0] <= 2e-5, f'rank {rank}: sharded P2P step differs from the single-process step by {res[rank][0]:.2e}'
    print(f'sharded step on the P2P exchange, 2 processes on one GPU: max rel err vs one process {max(res[0][0], res[1][0]):.1e}')
