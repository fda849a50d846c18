"""GPU: the P2P-store all-gather (piml_amd/csrc/p2p.hip, piml_amd/p2p.py) between TWO and EIGHT PROCESSES sharing the one
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


def _mesh(ctx, world):
    """one pipe per pair of ranks: conns[rank] = [connection to peer, ...] in peer order (the rank itself left out)"""
    conns = [[None] * world for _ in range(world)]
    for i in range(world):
        for j in range(i + 1, world):
            conns[i][j], conns[j][i] = ctx.Pipe()
    return [[c for c in row if c is not None] for row in conns]


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
            conns[0].recv()                  # (rank 0's word: it is the first peer of every other rank)
            lost = None
        q.put((rank, worst, lost))
        if rank != 0:
            import time
            time.sleep(1.0)                  # keep the buffers mapped until rank 0's lonely step has timed out
        ex.close()
    except Exception as e:   # noqa: BLE001
        q.put((rank, 'error: %s: %s' % (type(e).__name__, e)))


@pytest.mark.parametrize('world,uneven', [(2, False), (2, True), (8, True)])
def test_p2p_allgather_between_processes_on_one_gpu(world, uneven):
    fpr, steps = 2048 * 6, 12                            # a rank's block of the 16384-agent scene on 8 GPUs: 2048 agents x 6 floats = 49 KB
    ctx = mp.get_context('spawn')
    conns = _mesh(ctx, world)
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(r, world, fpr, steps, conns[r], q, uneven)) for r in range(world)]
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
    print(f'p2p all-gather, {world} processes on one GPU, {steps} steps of {fpr * 4} B per rank, uneven={uneven}: bit-exact; lonely step flagged')


def _child_step(rank, world, conns, q):
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

        def all_bytes(b):                                     # everybody's bytes in rank order, over the pairwise pipes
            for c in conns:
                c.send(b)
            others = [c.recv() for c in conns]
            return others[:rank] + [b] + others[rank:]

        def rendezvous(tag):
            for c in conns:
                c.send(tag)
            for c in conns:
                c.recv()
        torch.manual_seed(666)
        import piml_amd.models.model as MODEL
        n_params = sum(p.numel() for p in MODEL.PINNSF_multitask(bench.model_args()).parameters())
        p2p = p2p_exchanges(rank, world, n_own, n_params, all_bytes)
        st = bench.Step(scene, N, n_own, rank * n_own, M, dev, None, True, False, True, exchange='p2p', p2p=p2p)
        rendezvous(b'ready')
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
        rendezvous(b'done')                                   # nobody unmaps a buffer a peer may still write
        q.put((rank, worst, ok))
        for e in p2p:
            e.close()
    except Exception as e:   # noqa: BLE001
        import traceback
        q.put((rank, 'error: %s: %s\n%s' % (type(e).__name__, e, traceback.format_exc())))


@pytest.mark.parametrize('world', [2, 8])
def test_sharded_step_on_p2p_exchange_processes_one_gpu(world, monkeypatch):
    """bench.Step over `world` ranks sharing the one GPU (8: 8 x 512 focal rows of the 4096-agent scene), forward all-gather and
    backward reduce both on the P2P-store exchange and INSIDE the captured graph: state-gradient rows and every weight gradient
    against the single-process step.  Co-residency: every workgroup of p2p_exchange_kernel spins until all of its launch and
    all of its peers' have arrived, so the ranks sharing ONE device cap the split (8 x 8 x 16 = 1024 workgroups at once); on
    a node with a GPU per rank the default (world x 64) stands."""
    if world > 2:
        monkeypatch.setenv('PIML_P2P_MAX_SPLIT', '16')       # (inherited by the spawned children)
    ctx = mp.get_context('spawn')
    conns = _mesh(ctx, world)
    q = ctx.Queue()
    procs = [ctx.Process(target=_child_step, args=(r, world, conns[r], q)) for r in range(world)]
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
        assert res[rank][0] <= 1e-5, f'rank {rank}: sharded P2P step differs from the single-process step by {res[rank][0]:.2e}'
    print(f'sharded step on the P2P exchange, {world} processes on one GPU: max rel err vs one process {max(r[0] for r in res.values()):.1e}')


def test_p2p_exchange_stages_odd_blocks_and_misaligned_views():
    """P2PExchange.exchange with parts the kernel's 16-byte words cannot take as they stand -- an odd agent block (501 rows x 6
    floats = 3006, not a multiple of 4: 4005 agents padded to 4008 over 8 ranks) and a source that starts 8 bytes into its
    storage -- in a one-rank world (the staging is host-side; piml_amd/sharded.py's RCCL path takes the same shapes)."""
    import torch
    from piml_amd.p2p import P2PExchange
    from piml_amd.sharded import p2p_exchanges
    dev = 'cuda:0'
    fwd, bwd = p2p_exchanges(0, 1, 501, 1000, lambda b: [b])
    try:
        own = torch.randn(501, 6, device=dev)
        full = torch.empty(501 * 6, device=dev)
        fwd.exchange(bcast_src=own.reshape(-1), out_bcast=full, sum=False)
        assert fwd.ok() and torch.equal(full.view(501, 6), own)
        big = torch.randn(2 + 3006, device=dev)
        view = big[2:]                                        # 8 bytes into the storage: not 16-byte aligned
        assert view.data_ptr() % 16 != 0
        g_own = torch.empty(3006, device=dev)
        w = torch.randn(1003, device=dev)
        w0 = w.clone()
        bwd.exchange(scatter_src=view, out_scatter=g_own, bcast_src=[w], out_bcast=[w], sum=True)     # in place on an odd-sized part
        assert bwd.ok() and torch.equal(g_own, view) and torch.equal(w, w0)
    finally:
        fwd.close(); bwd.close()
