"""Agent-block sharding of one scene over the GPUs of a node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The reference has no multi-process path (only nn.DataParallel, src/models/simulators.py:64-67).
Focal agents are independent given everyone's state, so rank r owns the contiguous block
[r*n, (r+1)*n) of agents, obstacles are replicated, and a simulated step needs exactly one
exchange per direction:

  forward   all-gather of the owners' (p, v, a) records, 24 B per agent, into one interleaved
            (N, 6) buffer that the relfeat kernel consumes directly (state_ld = 6);
  backward  the kernel's partial d/d(state) covers all N sources -> reduce-scatter (sum) back
            to the owners; MLP weight gradients -> one bucketed all-reduce per optimiser step.

Messages are tiny (49 KB per rank at 16384 agents on 8 GPUs), i.e. latency-bound: a single
RCCL all-gather / reduce-scatter per step, no ring of sends written by hand.
"""
import torch
import torch.distributed as dist

from . import ops


def agent_block(n_total, rank, world):
    """(begin, count) of the agents owned by `rank`; blocks are equal (n_total % world == 0)."""
    if n_total % world:
        raise ValueError(f'n_total={n_total} must be divisible by the world size {world} '
                         '(pad the scene with absent agents: sharded.pad_scene)')
    n = n_total // world
    return rank * n, n


def pad_scene(state, destination, desired_speed, world):
    """Pad a scene to a multiple of `world` agents with ABSENT agents (NaN position / destination, zero velocity and
    acceleration -- the reference's own encoding of an agent that is not in the scene, src/data/data.py:141-143), so
    that equal agent blocks exist for any N.  Absent agents select nobody, are selected by nobody and receive zero
    gradient, so the padded scene computes exactly the unpadded one.  Returns (state (N', 6), destination (N', 2),
    desired_speed (N', 1), N)."""
    n = state.shape[0]
    pad = (-n) % world
    if pad == 0:
        return state, destination, desired_speed, n
    nan = float('nan')
    rows = torch.tensor([[nan, nan, 0.0, 0.0, 0.0, 0.0]], dtype=state.dtype, device=state.device).expand(pad, 6)
    return (torch.cat((state, rows)), torch.cat((destination, torch.full((pad, 2), nan, dtype=destination.dtype,
                                                                        device=destination.device))),
            torch.cat((desired_speed, torch.zeros(pad, desired_speed.shape[1], dtype=desired_speed.dtype,
                                                  device=desired_speed.device))), n)


def _supports_reduce_scatter(group):
    return dist.get_backend(group) != 'gloo'


class _AllGatherRecords(torch.autograd.Function):
    """own (n, w) -> full (world*n, w); backward sums the partial gradients of all ranks and
    returns this rank's rows."""

    @staticmethod
    def forward(ctx, own, group):
        world = dist.get_world_size(group)
        own = own.contiguous()
        full = torch.empty((world * own.shape[0],) + tuple(own.shape[1:]), device=own.device, dtype=own.dtype)
        dist.all_gather_into_tensor(full, own, group=group)
        ctx.group = group
        return full

    @staticmethod
    def backward(ctx, g_full):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        g_full = g_full.contiguous()
        n = g_full.shape[0] // world
        if _supports_reduce_scatter(group):
            g_own = torch.empty((n,) + tuple(g_full.shape[1:]), device=g_full.device, dtype=g_full.dtype)
            dist.reduce_scatter_tensor(g_own, g_full, op=dist.ReduceOp.SUM, group=group)
        else:   # gloo (CPU tests) has no reduce-scatter
            dist.all_reduce(g_full, op=dist.ReduceOp.SUM, group=group)
            g_own = g_full[rank * n:(rank + 1) * n].clone()
        return g_own, None


def all_gather_records(own, group=None):
    return _AllGatherRecords.apply(own, group if group is not None else dist.group.WORLD)


class _AllGatherRecordsAsync(torch.autograd.Function):
    """_AllGatherRecords whose forward returns before the collective has finished: this rank's rows are already in
    place in the returned (N, w) buffer (in-place all-gather: they are the send buffer), the other rows arrive when
    `holder[0].wait()` has been called.  Backward as _AllGatherRecords."""

    @staticmethod
    def forward(ctx, own, group, holder):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        own = own.contiguous()
        n = own.shape[0]
        full = torch.empty((world * n,) + tuple(own.shape[1:]), device=own.device, dtype=own.dtype)
        mine = full[rank * n:(rank + 1) * n]
        mine.copy_(own)
        holder.append(dist.all_gather_into_tensor(full.view(-1), mine.reshape(-1), group=group, async_op=True))
        ctx.group = group
        return full

    @staticmethod
    def backward(ctx, g_full):
        return _AllGatherRecords.backward(ctx, g_full) + (None,)


class _AllGatherRecordsP2P(torch.autograd.Function):
    """_AllGatherRecords on the P2P-store exchange (piml_amd/p2p.py, include/piml_hip.h: piml_p2p_exchange): forward = every
    rank's block stored into every peer's receive buffer and copied out in rank order; backward = the partial gradients of
    every owner's block stored into THAT owner's buffer and added there in rank order.  No collective library, and both
    launches are capturable (the step counter lives on the device)."""

    @staticmethod
    def forward(ctx, own, ex_fwd, ex_bwd):
        own = own.contiguous()
        full = torch.empty((ex_fwd.world * own.shape[0],) + tuple(own.shape[1:]), device=own.device, dtype=own.dtype)
        ex_fwd.exchange(bcast_src=own.reshape(-1), out_bcast=full.view(-1), sum=False)
        ctx.ex_bwd, ctx.own_shape = ex_bwd, tuple(own.shape)
        return full

    @staticmethod
    def backward(ctx, g_full):
        g_own = torch.empty(ctx.own_shape, device=g_full.device, dtype=g_full.dtype)
        ctx.ex_bwd.exchange(scatter_src=g_full.contiguous().view(-1), out_scatter=g_own.view(-1), sum=True)
        return g_own, None, None


def p2p_exchanges(rank, world, n_own, n_params, all_gather_bytes):
    """(forward, backward) P2PExchange pair of one rank of an agent-block sharded step: the forward slot holds a rank's block of
    records (n_own x 6 floats), the backward slot a block of state gradients + the weight-gradient bucket.
    all_gather_bytes(own: bytes) -> every rank's bytes in rank order (torch.distributed.all_gather_object, pipes, ...)."""
    from .p2p import P2PExchange
    pad4 = lambda n: (n + 3) // 4 * 4
    fwd = P2PExchange(rank, world, pad4(n_own * 6))
    # (the gradient buffers travel whole -- grad_bases -- with their padding and scratch fields: room for twice the parameters)
    bwd = P2PExchange(rank, world, pad4(n_own * 6) + 2 * pad4(n_params) + 4096)
    if world > 1:
        fwd.connect_all(all_gather_bytes)
        bwd.connect_all(all_gather_bytes)
    return fwd, bwd


def grad_bases(parameters):
    """The distinct contiguous float32 buffers behind the parameters' .grad tensors (a fused network hands autograd VIEWS of a
    few flat buffers: ops.fused_pinnsf), as flat tensors -- or None when they are not of that kind (more than 8, odd sizes).
    Exchanging these in place needs no concatenation before and no copy back after the exchange."""
    bases, seen = [], set()
    for p in parameters:
        g = p.grad
        if g is None:
            continue
        # (AccumulateGrad keeps the STORAGE of a view it is handed, not the view relation: the buffer is found through the storage)
        st = g.untyped_storage()
        if not (g.is_cuda and g.dtype == torch.float32 and st.nbytes() % 16 == 0 and st.data_ptr() % 16 == 0):
            return None
        if st.data_ptr() not in seen:
            seen.add(st.data_ptr())
            bases.append(torch.empty(0, dtype=torch.float32, device=g.device).set_(st, 0, (st.nbytes() // 4,)))
    return bases if 0 < len(bases) <= 8 else None


def allreduce_gradients_p2p(parameters, ex):
    """allreduce_gradients on the P2P-store exchange: every rank's gradients to every peer, added in rank order (the same sum on
    every rank) -- in place on the gradients' own buffers where they are a few flat ones (grad_bases), else through a bucket."""
    parameters = list(parameters)
    bases = grad_bases(parameters)
    if bases is not None and sum(b.numel() for b in bases) <= ex.fpr:
        ex.exchange(bcast_src=bases, out_bcast=bases, sum=True)
        return
    flat, grads = flatten_gradients(parameters)
    if flat is None:
        return
    pad = (-flat.numel()) % 4
    if pad:
        flat = torch.cat((flat, flat.new_zeros(pad)))
    out = torch.empty_like(flat)
    ex.exchange(bcast_src=flat, out_bcast=out, sum=True)
    unflatten_gradients(out[:out.numel() - pad] if pad else out, grads)


def gather_records_into(full, own, group=None):
    """Plain (non-autograd) all-gather into a caller-owned (N, w) buffer.  With `reduce_scatter_grad`
    this is the exchange pair for steps whose compute part is replayed from a captured HIP graph: the
    two collectives run eagerly on the stream either side of the replay and `full` is the graph's
    static input (a leaf whose .grad the graph's backward fills)."""
    with torch.no_grad():
        dist.all_gather_into_tensor(full.detach(), own.detach().contiguous(), group=group)
    return full


def gather_records_async(full, own, begin, group=None):
    """The exchange of `gather_records_into` without waiting for it: this rank's rows are copied into their place in
    `full` (N, w) on the current stream, then an IN-PLACE all-gather (send buffer = this rank's rows of the receive
    buffer, which the collective therefore never writes) is started on the backend's own stream.  Returns the work
    handle: everything that needs only the rank's own rows of `full` -- relative_features_local_part, the weight packs
    -- can be enqueued before `work.wait()` and runs while the other blocks are in flight."""
    with torch.no_grad():
        f = full.detach()
        n = own.shape[0]
        mine = f[begin:begin + n]
        mine.copy_(own.detach())
        return dist.all_gather_into_tensor(f.view(-1), mine.reshape(-1), group=group, async_op=True)


def reduce_scatter_grad(g_full, group=None, out=None):
    """Sum the ranks' partial (N, w) gradients and return this rank's (n, w) rows."""
    group = group if group is not None else dist.group.WORLD
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = g_full.shape[0] // world
    if out is None:
        out = torch.empty((n,) + tuple(g_full.shape[1:]), device=g_full.device, dtype=g_full.dtype)
    if _supports_reduce_scatter(group):
        dist.reduce_scatter_tensor(out, g_full.contiguous(), op=dist.ReduceOp.SUM, group=group)
    else:   # gloo (CPU tests) has no reduce-scatter
        summed = g_full.clone()
        dist.all_reduce(summed, op=dist.ReduceOp.SUM, group=group)
        out.copy_(summed[rank * n:(rank + 1) * n])
    return out


def flatten_gradients(parameters):
    """One flat bucket holding all parameter gradients (134 k floats for pinnsf_m = 0.5 MB).
    Returns (flat, grads); capturable in a HIP graph (one concatenation kernel)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if not grads:
        return None, grads
    return torch.cat([g.reshape(-1) for g in grads]), grads


def unflatten_gradients(flat, grads):
    """Write the reduced bucket back into the .grad tensors (one multi-tensor copy)."""
    if flat is None:
        return
    # (ops.multi_copy: ONE launch for contiguous GPU tensors; torch._foreach_copy_ issues a copy per tensor, ~2.6 us each)
    ops.multi_copy(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])


def allreduce_gradients(parameters, group=None, average=False):
    """Bucketed all-reduce (sum, or mean with `average`) of the parameter gradients."""
    flat, grads = flatten_gradients(parameters)
    if flat is None:
        return
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    unflatten_gradients(flat, grads)


class ShardedScene:
    """Per-rank view of one scene under agent-block sharding.

    `feature_fn(state_full, destination_rows, obstacles, focal_begin, focal_count, **params)`
    defaults to the HIP operator; tests inject a CPU stand-in to exercise the sharding and
    the collectives with the gloo backend.
    """

    def __init__(self, n_total, obstacles, group=None, feature_fn=None, force_collectives=False, exchange='rccl', p2p=None,
                 **feature_params):
        # force_collectives: issue the all-gather / reduce-scatter even in a 1-rank group (exercises the RCCL
        # code path on a single GPU; tests/test_sharded_gpu.py, bench.py --force-dist)
        # exchange='p2p' with p2p = (forward, backward) P2PExchange objects (p2p_exchanges): the state exchange of both
        # directions on P2P stores instead of RCCL collectives (rank / world are the exchange objects'; no process group needed)
        self.force_collectives = bool(force_collectives)
        if exchange not in ('rccl', 'p2p') or (exchange == 'p2p') != (p2p is not None):
            raise ValueError("ShardedScene: exchange='rccl', or exchange='p2p' together with p2p=(forward, backward)")
        self.exchange, self.p2p = exchange, p2p
        self.group = group if group is not None else (dist.group.WORLD if (dist.is_initialized() and p2p is None) else None)
        self.world = p2p[0].world if p2p is not None else (dist.get_world_size(self.group) if self.group is not None else 1)
        self.rank = p2p[0].rank if p2p is not None else (dist.get_rank(self.group) if self.group is not None else 0)
        self.n_total = n_total
        self.begin, self.count = agent_block(n_total, self.rank, self.world)
        self.obstacles = obstacles
        self.feature_fn = feature_fn if feature_fn is not None else ops.relative_features_packed
        self.feature_params = feature_params

    def own(self, x):
        """Rows of a replicated (N, ...) tensor owned by this rank."""
        return x[self.begin:self.begin + self.count]

    def gather_state(self, state_own):
        """(n, 6) owner records -> (N, 6) everyone's records (autograd: reduce-scatter)."""
        if self.world == 1 and not self.force_collectives:
            return state_own
        if self.p2p is not None:
            return _AllGatherRecordsP2P.apply(state_own, self.p2p[0], self.p2p[1])
        return all_gather_records(state_own, self.group)

    def relative_features(self, state_own, destination_own):
        """Features of the owned focal rows against all N sources."""
        state_full = self.gather_state(state_own)
        return self.feature_fn(state_full, destination_own, self.obstacles, self.begin, self.count,
                               **self.feature_params)

    def model_step_overlapped(self, model, state_own, destination_own, desired_speed_own):
        """`model_step` with the exchange started first and everything that needs only the own block enqueued under it:
        the weight pack, the neighbour search among the block's own agents, the obstacle branch and the self features
        (ops.relative_features_local_part); after the wait, the remote half of the search and the network.  Same
        results as model_step (the split search is bit-identical; the HIP operators only, no injected feature_fn)."""
        if self.p2p is not None:
            # every workgroup of p2p_exchange_kernel spins until all of its launch and all of its peers' have arrived: correct only
            # if they are co-resident, which an idle stream gives and a second stream full of search workgroups does not
            raise ValueError("ShardedScene(exchange='p2p') has no overlapped step: the P2P exchange needs its workgroups co-resident "
                             "(run it on the one stream, as model_step does; overlap the RCCL exchange instead)")
        group = self.group if self.group is not None else dist.group.WORLD
        holder = []
        state_full = _AllGatherRecordsAsync.apply(state_own, group, holder)
        packed = getattr(model, 'packed_weights', None)
        import contextlib
        with (packed() if packed is not None else contextlib.nullcontext()):
            local = ops.relative_features_local_part(state_full, destination_own, self.obstacles, desired_speed_own,
                                                     self.begin, self.count, **self.feature_params)
            holder[0].wait()
            pf, of, self_features = ops.relative_features_packed_self(
                state_full, destination_own, self.obstacles, desired_speed_own, self.begin, self.count, local=local,
                **self.feature_params)
            return model(pf, of, self_features)

    def model_step(self, model, state_own, destination_own, desired_speed_own):
        """features -> PINNSF forward for the owned rows (simulators.py:642-652 + :602).
        Returns the model's output list; output[0] is the owned agents' acceleration."""
        pf, of, df = self.relative_features(state_own, destination_own)
        self_features = torch.cat((df, state_own[..., 2:4], state_own[..., 4:6], desired_speed_own), dim=-1)
        return model(pf, of, self_features)
