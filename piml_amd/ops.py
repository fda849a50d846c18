"""torch-facing operators over the C ABI (include/piml_hip.h).

PyTorch is plumbing here: it owns device memory and the stream, and autograd.Function glues
the forward/backward kernels into the reference's training loops.  Every operator requires
float32 tensors on the GPU and raises otherwise (no CPU fallback).
"""
import math

import torch

from . import _lib

MAX_TOPK = 32
# Opt-in: the relative-feature backward without float atomics (bit-reproducible gradients; the neighbour-list entries are
# sorted by source with torch.sort(stable=True) and gathered in that fixed order).  Default: the atomic scatter.
import contextlib
import functools
import os as _os
DETERMINISTIC_BWD = _os.environ.get('PIML_DETERMINISTIC_BWD', '0') == '1'


@functools.lru_cache(maxsize=None)
def cos_threshold(angle_deg):
    """float32(cos(3.14 * angle / 180)): the reference's view-cone threshold, with its 3.14
    (src/data/data.py:442-443), rounded the way torch compares float32 with a scalar."""
    return float(torch.tensor(math.cos(3.14 * angle_deg / 180), dtype=torch.float32))


def _gpu_f32(name, t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PimlHipError(f'{name}: expected a GPU tensor (piml_amd has no CPU path), got '
                                f'{getattr(t, "device", type(t))}')
    if t.dtype != torch.float32:
        raise TypeError(f'{name}: expected float32, got {t.dtype}')
    return t.contiguous()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None or t.numel() == 0 else t.data_ptr()


def heading_direction(velocity):
    """Pedestrians.get_heading_direction (src/data/data.py:350-395) for (*c, t, N, 2)."""
    v = _gpu_f32('velocity', velocity)
    if v.dim() < 3:
        raise ValueError('velocity must be (*c, t, N, 2)')
    T, N = v.shape[-3], v.shape[-2]
    C = v.numel() // max(T * N * 2, 1)
    out = torch.empty_like(v)
    with torch.cuda.device(v.device):
        _lib.check(_lib.lib().piml_heading_fwd(_ptr(v), C, T, N, _ptr(out), _stream()), 'piml_heading_fwd')
    return out


def _launch_relfeat_fwd(p_ptr, v_ptr, a_ptr, ld, hd, dest_rows, o, lead, C, N, f0, fcnt, kp, ko,
                        cos_p, cos_o, dthr_p, dthr_o, device, outs=None, dest_ld=2, tick=None):
    M = o.shape[0]
    kpe, koe = min(kp, N), min(ko, M)
    if outs is None:
        opt = dict(device=device, dtype=torch.float32)
        ped_feat = torch.empty(*lead, fcnt, kpe, 6, **opt)
        obs_feat = torch.empty(*lead, fcnt, koe, 6, **opt)
        dest_feat = torch.empty(*lead, fcnt, 2, **opt)
        ped_idx = torch.empty(*lead, fcnt, kpe, device=device, dtype=torch.int32)
        obs_idx = torch.empty(*lead, fcnt, koe, device=device, dtype=torch.int32)
    else:
        ped_feat, obs_feat, dest_feat, ped_idx, obs_idx = outs
        want = ((*lead, fcnt, kpe, 6), (*lead, fcnt, koe, 6), (*lead, fcnt, dest_ld), (*lead, fcnt, kpe), (*lead, fcnt, koe))
        for t, shp, dt in zip(outs, want, (torch.float32,) * 3 + (torch.int32,) * 2):
            if tuple(t.shape) != shp or t.dtype != dt or not t.is_contiguous() or t.device != device:
                raise ValueError(f'output buffer mismatch: expected {shp} {dt}, got {tuple(t.shape)} {t.dtype}')
    with torch.cuda.device(device):
        if tick is not None:
            if tick.dtype != torch.int64 or tick.numel() != 1 or tick.device != device:
                raise ValueError('tick: a one-element int64 tensor on the same device expected')
            _lib.check(_lib.lib().piml_relfeat_fwd_tick(
                p_ptr, _ptr(hd), v_ptr, a_ptr, ld, _ptr(dest_rows), _ptr(o), C, N, M, f0, fcnt,
                kp, ko, cos_p, cos_o, dthr_p, dthr_o, _ptr(ped_feat), _ptr(obs_feat), _ptr(dest_feat), dest_ld,
                _ptr(ped_idx), _ptr(obs_idx), _ptr(tick), _stream()), 'piml_relfeat_fwd_tick')
            return ped_feat, obs_feat, dest_feat, ped_idx, obs_idx
        _lib.check(_lib.lib().piml_relfeat_fwd(
            p_ptr, _ptr(hd), v_ptr, a_ptr, ld, _ptr(dest_rows), _ptr(o), C, N, M, f0, fcnt,
            kp, ko, cos_p, cos_o, dthr_p, dthr_o, _ptr(ped_feat), _ptr(obs_feat), _ptr(dest_feat), dest_ld,
            _ptr(ped_idx), _ptr(obs_idx), _stream()), 'piml_relfeat_fwd')
    return ped_feat, obs_feat, dest_feat, ped_idx, obs_idx


def _launch_relfeat_bwd(ctx_geom, g_ped, g_obs, g_dest, ped_idx, obs_idx, p_ptr, ld, dest_rows, device,
                        g_state=None):
    """`g_state` given: the kernel accumulates into it (all its writes are atomic adds)."""
    C, N, f0, fcnt, kpe, koe, lead = ctx_geom

    def dense(g, shape):
        return torch.zeros(shape, device=device, dtype=torch.float32) if g is None else _gpu_f32('grad', g)
    g_ped = dense(g_ped, (*lead, fcnt, kpe, 6))
    g_obs = dense(g_obs, (*lead, fcnt, koe, 6))
    g_dest = dense(g_dest, (*lead, fcnt, 2))
    if g_state is None and not DETERMINISTIC_BWD:
        g_state = torch.zeros(*lead, N, 6, device=device, dtype=torch.float32)
    g_dest_rows = torch.empty(*lead, fcnt, 2, device=device, dtype=torch.float32)
    if DETERMINISTIC_BWD:
        seeded = g_state is not None
        if g_state is None:
            g_state = torch.empty(*lead, N, 6, device=device, dtype=torch.float32)
        idx = ped_idx.reshape(C, fcnt * kpe).long()
        base = torch.arange(C, device=device, dtype=torch.long).unsqueeze(1) * N
        keys = torch.where(idx >= 0, idx + base, torch.full_like(idx, C * N)).reshape(-1)
        sorted_keys, order = torch.sort(keys, stable=True)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().piml_relfeat_bwd_det(
                _ptr(g_ped), _ptr(g_obs), _ptr(g_dest), _ptr(ped_idx), _ptr(obs_idx), _ptr(sorted_keys), _ptr(order),
                p_ptr, ld, _ptr(dest_rows), C, N, f0, fcnt, kpe, koe, int(seeded), _ptr(g_state), _ptr(g_dest_rows),
                _stream()), 'piml_relfeat_bwd_det')
        return g_state, g_dest_rows
    with torch.cuda.device(device):
        _lib.check(_lib.lib().piml_relfeat_bwd(
            _ptr(g_ped), _ptr(g_obs), _ptr(g_dest), _ptr(ped_idx), _ptr(obs_idx), p_ptr, ld,
            _ptr(dest_rows), C, N, f0, fcnt, kpe, koe, _ptr(g_state), _ptr(g_dest_rows), _stream()),
            'piml_relfeat_bwd')
    return g_state, g_dest_rows


class _RelativeFeatures(torch.autograd.Function):
    """Separate (..., N, 2) position / velocity / acceleration / destination tensors."""

    @staticmethod
    def forward(ctx, position, velocity, acceleration, destination, obstacles, heading,
                focal_begin, focal_count, kp, ko, cos_p, cos_o, dthr_p, dthr_o):
        p = _gpu_f32('position', position)
        v = _gpu_f32('velocity', velocity)
        a = _gpu_f32('acceleration', acceleration)
        d = _gpu_f32('destination', destination)
        o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
        if not (p.shape == v.shape == a.shape == d.shape) or p.shape[-1] != 2 or p.dim() < 2:
            raise ValueError(f'position/velocity/acceleration/destination must share a (..., N, 2) shape, got '
                             f'{tuple(p.shape)} {tuple(v.shape)} {tuple(a.shape)} {tuple(d.shape)}')
        N = p.shape[-2]
        lead = tuple(p.shape[:-2])
        C = p.numel() // max(N * 2, 1)
        if focal_count is None:
            focal_begin, focal_count = 0, N
        hd = None if heading is None else _gpu_f32('heading', heading)
        d_rows = d if focal_count == N else d[..., focal_begin:focal_begin + focal_count, :].contiguous()
        out = _launch_relfeat_fwd(_ptr(p), _ptr(v), _ptr(a), 2, hd, d_rows, o, lead, C, N, focal_begin,
                                  focal_count, kp, ko, cos_p, cos_o, dthr_p, dthr_o, p.device)
        ctx.save_for_backward(out[3], out[4], p, d_rows)
        ctx.geom = (C, N, focal_begin, focal_count, out[3].shape[-1], out[4].shape[-1], lead)
        ctx.mark_non_differentiable(out[3], out[4])
        ctx.set_materialize_grads(False)     # no zero tensors for the index outputs / unused features
        return out

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_dest, _gi, _go):
        if g_ped is None and g_obs is None and g_dest is None:
            return (None,) * 14
        ped_idx, obs_idx, p, d_rows = ctx.saved_tensors
        C, N, f0, fcnt, kpe, koe, lead = ctx.geom
        g_state, g_d_rows = _launch_relfeat_bwd(ctx.geom, g_ped, g_obs, g_dest, ped_idx, obs_idx,
                                                _ptr(p), 2, d_rows, p.device)
        if fcnt == N:
            g_destination = g_d_rows
        else:
            g_destination = torch.zeros(*lead, N, 2, device=p.device, dtype=torch.float32)
            g_destination[..., f0:f0 + fcnt, :] = g_d_rows
        return (g_state[..., 0:2], g_state[..., 2:4], g_state[..., 4:6], g_destination) + (None,) * 10


class _RelativeFeaturesSelf(torch.autograd.Function):
    """_RelativeFeatures whose third output is the model's self_features rows [dest - p, v, a, v0] (..., N, 7)
    (piml_relfeat_fwd_self / piml_relfeat_bwd_self): the training rollout's per-frame torch.cat inside the launch."""

    @staticmethod
    def forward(ctx, position, velocity, acceleration, destination, obstacles, desired_speed, kp, ko, cos_p, cos_o, dthr_p, dthr_o):
        p = _gpu_f32('position', position)
        v = _gpu_f32('velocity', velocity)
        a = _gpu_f32('acceleration', acceleration)
        d = _gpu_f32('destination', destination)
        o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
        if not (p.shape == v.shape == a.shape == d.shape) or p.shape[-1] != 2 or p.dim() < 2:
            raise ValueError('position/velocity/acceleration/destination must share a (..., N, 2) shape')
        N = p.shape[-2]
        lead = tuple(p.shape[:-2])
        C = p.numel() // max(N * 2, 1)
        v0 = _gpu_f32('desired_speed', desired_speed)
        if v0.numel() != C * N:
            raise ValueError(f'desired_speed must hold one value per agent, got {tuple(v0.shape)}')
        M = o.shape[0]
        kpe, koe = min(kp, N), min(ko, M)
        opt = dict(device=p.device, dtype=torch.float32)
        pf, of = torch.empty(*lead, N, kpe, 6, **opt), torch.empty(*lead, N, koe, 6, **opt)
        sf = torch.empty(*lead, N, 7, **opt)
        pi = torch.empty(*lead, N, kpe, device=p.device, dtype=torch.int32)
        oi = torch.empty(*lead, N, koe, device=p.device, dtype=torch.int32)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_relfeat_fwd_self(
                _ptr(p), None, _ptr(v), _ptr(a), 2, _ptr(d), _ptr(o), _ptr(v0), C, N, M, 0, N, kp, ko, cos_p, cos_o, dthr_p, dthr_o,
                _ptr(pf), _ptr(of), _ptr(sf), _ptr(pi), _ptr(oi), None, _stream()), 'piml_relfeat_fwd_self')
        ctx.save_for_backward(pi, oi, p, d)
        ctx.geom = (C, N, kpe, koe, lead, tuple(desired_speed.shape))
        ctx.mark_non_differentiable(pi, oi)
        ctx.set_materialize_grads(False)
        return pf, of, sf, pi, oi

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_self, _gi, _go):
        if g_ped is None and g_obs is None and g_self is None:
            return (None,) * 12
        pi, oi, p, d = ctx.saved_tensors
        C, N, kpe, koe, lead, speed_shape = ctx.geom
        opt = dict(device=p.device, dtype=torch.float32)

        def dense(g, shape):
            return torch.zeros(shape, **opt) if g is None else _gpu_f32('grad', g)
        g_ped, g_obs, g_self = dense(g_ped, (*lead, N, kpe, 6)), dense(g_obs, (*lead, N, koe, 6)), dense(g_self, (*lead, N, 7))
        g_state = torch.zeros(*lead, N, 6, **opt)
        g_dest = torch.empty(*lead, N, 2, **opt)
        g_speed = torch.empty(speed_shape, **opt) if ctx.needs_input_grad[5] else None
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_relfeat_bwd_self(
                _ptr(g_ped), _ptr(g_obs), _ptr(g_self), _ptr(pi), _ptr(oi), _ptr(p), 2, _ptr(d), C, N, 0, N, kpe, koe,
                _ptr(g_state), _ptr(g_dest), _ptr(g_speed), _stream()), 'piml_relfeat_bwd_self')
        return (g_state[..., 0:2], g_state[..., 2:4], g_state[..., 4:6], g_dest, None, g_speed) + (None,) * 6


class _RelativeFeaturesPacked(torch.autograd.Function):
    """Interleaved (..., N, 6) = (p, v, a) state records (the all-gathered buffer of
    agent-block sharding) + destinations of the focal rows only."""

    @staticmethod
    def forward(ctx, state, destination_rows, obstacles, focal_begin, focal_count, kp, ko,
                cos_p, cos_o, dthr_p, dthr_o):
        s = _gpu_f32('state', state)
        d_rows = _gpu_f32('destination_rows', destination_rows)
        o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
        if s.shape[-1] != 6 or s.dim() < 2:
            raise ValueError(f'state must be (..., N, 6), got {tuple(s.shape)}')
        N = s.shape[-2]
        lead = tuple(s.shape[:-2])
        if tuple(d_rows.shape) != (*lead, focal_count, 2):
            raise ValueError(f'destination_rows must be {(*lead, focal_count, 2)}, got {tuple(d_rows.shape)}')
        C = s.numel() // max(N * 6, 1)
        base = s.data_ptr()
        out = _launch_relfeat_fwd(base, base + 8, base + 16, 6, None, d_rows, o, lead, C, N, focal_begin,
                                  focal_count, kp, ko, cos_p, cos_o, dthr_p, dthr_o, s.device)
        ctx.save_for_backward(out[3], out[4], s, d_rows)
        ctx.geom = (C, N, focal_begin, focal_count, out[3].shape[-1], out[4].shape[-1], lead)
        ctx.mark_non_differentiable(out[3], out[4])
        ctx.set_materialize_grads(False)     # no zero tensors for the index outputs / unused features
        return out

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_dest, _gi, _go):
        if g_ped is None and g_obs is None and g_dest is None:
            return (None,) * 11
        ped_idx, obs_idx, s, d_rows = ctx.saved_tensors
        g_state, g_d_rows = _launch_relfeat_bwd(ctx.geom, g_ped, g_obs, g_dest, ped_idx, obs_idx,
                                                s.data_ptr(), 6, d_rows, s.device)
        return (g_state, g_d_rows) + (None,) * 9


class _RelativeFeaturesPackedSelf(torch.autograd.Function):
    """_RelativeFeaturesPacked for a 2-D (N, 6) state that returns the model's self_features rows
    (n, 7) = [dest - p, v, a, v0] instead of dest_features, all from ONE launch (piml_relfeat_self_fwd); when a
    gradient will flow, the same launch clears the (N, 6) state-gradient buffer that the ONE backward launch
    (piml_relfeat_self_bwd: scatter + the rows' own terms + the self-feature columns) accumulates into."""

    @staticmethod
    def forward(ctx, state, destination_rows, obstacles, desired_speed, focal_begin, focal_count, kp, ko,
                cos_p, cos_o, dthr_p, dthr_o, local=None):
        s = _gpu_f32('state', state)
        d_rows = _gpu_f32('destination_rows', destination_rows)
        o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
        w = _gpu_f32('desired_speed', desired_speed)
        if s.dim() != 2 or s.shape[-1] != 6:
            raise ValueError(f'state must be (N, 6), got {tuple(s.shape)}')
        N = s.shape[0]
        if tuple(d_rows.shape) != (focal_count, 2) or w.numel() != focal_count:
            raise ValueError('destination_rows (n, 2) and desired_speed (n, 1) expected for the focal rows')
        M = o.shape[0]
        kpe, koe = min(kp, N), min(ko, M)
        opt = dict(device=s.device, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad[:4]) and not DETERMINISTIC_BWD
        if local is None:
            sf = torch.empty(focal_count, 7, **opt)
            outs = (torch.empty(focal_count, kpe, 6, **opt), torch.empty(focal_count, koe, 6, **opt), sf,
                    torch.empty(focal_count, kpe, device=s.device, dtype=torch.int32),
                    torch.empty(focal_count, koe, device=s.device, dtype=torch.int32))
            ctx.g_state = torch.empty(N, 6, **opt) if need_grad else None       # cleared by the launch below
            with torch.cuda.device(s.device):
                _lib.check(_lib.lib().piml_relfeat_self_fwd(
                    _ptr(s), _ptr(d_rows), _ptr(o), _ptr(w), N, M, focal_begin, focal_count, kp, ko, cos_p, cos_o,
                    dthr_p, dthr_o, _ptr(outs[0]), _ptr(outs[1]), _ptr(sf), _ptr(outs[3]), _ptr(outs[4]),
                    _ptr(ctx.g_state), _stream()), 'piml_relfeat_self_fwd')
        else:                 # the LOCAL part ran earlier (relative_features_local_part): finish with the remote sources
            geom = (s.data_ptr(), N, M, focal_begin, focal_count, kp, ko, cos_p, cos_o, dthr_p, dthr_o)
            if local.geom != geom:
                raise ValueError('relative_features_packed_self: `local` was made for another state buffer / geometry')
            outs = (torch.empty(focal_count, kpe, 6, **opt), local.obs_feat, local.self_features, local.ped_idx,
                    local.obs_idx)
            ctx.g_state = local.g_state if need_grad else None
            local.g_state = None
            with torch.cuda.device(s.device):
                _lib.check(_lib.lib().piml_relfeat_self_fwd_part(
                    2, _ptr(s), None, None, None, N, M, focal_begin, focal_count, kp, ko, cos_p, cos_o,
                    dthr_p, dthr_o, _ptr(outs[0]), None, None, _ptr(outs[3]), None, None, _stream()),
                    'piml_relfeat_self_fwd_part(REMOTE)')
        ctx.save_for_backward(outs[3], outs[4], s, d_rows)
        ctx.geom = (1, N, focal_begin, focal_count, kpe, koe, ())
        ctx.speed_shape = tuple(desired_speed.shape)
        ctx.mark_non_differentiable(outs[3], outs[4])
        ctx.set_materialize_grads(False)
        return outs

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_self, _gi, _go):
        if g_ped is None and g_obs is None and g_self is None:
            return (None,) * 13
        ped_idx, obs_idx, s, d_rows = ctx.saved_tensors
        _, N, f0, fcnt, kpe, koe, _ = ctx.geom
        opt = dict(device=s.device, dtype=torch.float32)
        if DETERMINISTIC_BWD:                 # sorted, atomics-free path: seed with the self-feature part, then gather
            g_dest, g_speed = None, None
            if g_self is None:
                g_state = torch.zeros(N, 6, **opt)
            else:
                g_self = g_self.contiguous()
                g_state = torch.empty(N, 6, **opt) if fcnt == N else torch.zeros(N, 6, **opt)
                g_dest = torch.empty(fcnt, 2, **opt)
                g_speed = torch.empty(ctx.speed_shape, **opt) if ctx.needs_input_grad[3] else None
                with torch.cuda.device(s.device):
                    _lib.check(_lib.lib().piml_self_features_bwd(_ptr(g_self), fcnt, _ptr(g_dest),
                                                                 g_state.data_ptr() + 24 * f0, _ptr(g_speed),
                                                                 _stream()), 'piml_self_features_bwd')
            g_state, g_d_rows = _launch_relfeat_bwd(ctx.geom, g_ped, g_obs, g_dest, ped_idx, obs_idx, s.data_ptr(), 6,
                                                    d_rows, s.device, g_state=g_state)
            return (g_state, g_d_rows, None, g_speed) + (None,) * 9

        def dense(g, shape):
            return torch.zeros(shape, **opt) if g is None else _gpu_f32('grad', g)
        g_ped, g_obs, g_self = dense(g_ped, (fcnt, kpe, 6)), dense(g_obs, (fcnt, koe, 6)), dense(g_self, (fcnt, 7))
        g_state, ctx.g_state = ctx.g_state, None        # the forward's cleared buffer, once; a second backward pass
        if g_state is None:                             # (retain_graph) clears a fresh one
            g_state = torch.zeros(N, 6, **opt)
        g_d_rows = torch.empty(fcnt, 2, **opt)
        g_speed = torch.empty(ctx.speed_shape, **opt) if ctx.needs_input_grad[3] else None
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().piml_relfeat_self_bwd(
                _ptr(g_ped), _ptr(g_obs), _ptr(g_self), _ptr(ped_idx), _ptr(obs_idx), _ptr(s), _ptr(d_rows), N, f0, fcnt,
                kpe, koe, _ptr(g_state), _ptr(g_d_rows), _ptr(g_speed), _stream()), 'piml_relfeat_self_bwd')
        return (g_state, g_d_rows, None, g_speed) + (None,) * 9


class LocalFeatures:
    """What relative_features_local_part leaves behind for relative_features_packed_self(local=...)."""

    def __init__(self):
        self.obs_feat = self.self_features = self.ped_idx = self.obs_idx = self.g_state = None
        self.geom = None


def relative_features_local_part(state, destination_rows, obstacles, desired_speed, focal_begin, focal_count,
                                 topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                                 topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, want_grad=True):
    """Agent-block sharding, first half of relative_features_packed_self: everything that needs only the focal block's
    own rows of the packed (N, 6) state -- the neighbour search among the block's own agents, the obstacle branch, the
    self_features rows -- so that it can be enqueued while the all-gather of the other blocks is still in flight (the
    rows outside [focal_begin, focal_begin + focal_count) are not read).  No autograd here: pass the result as
    `local=` to relative_features_packed_self, which finishes with the remote agents and owns the whole gradient."""
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    s = _gpu_f32('state', state.detach())
    d_rows = _gpu_f32('destination_rows', destination_rows.detach())
    o = _gpu_f32('obstacles', obstacles.detach()).reshape(-1, 2)
    w = _gpu_f32('desired_speed', desired_speed.detach())
    if s.dim() != 2 or s.shape[-1] != 6:
        raise ValueError(f'state must be (N, 6), got {tuple(s.shape)}')
    N, M = s.shape[0], o.shape[0]
    focal_begin, focal_count, kp, ko = int(focal_begin), int(focal_count), int(topk_ped), int(topk_obs)
    if tuple(d_rows.shape) != (focal_count, 2) or w.numel() != focal_count:
        raise ValueError('destination_rows (n, 2) and desired_speed (n, 1) expected for the focal rows')
    kpe, koe = min(kp, N), min(ko, M)
    opt = dict(device=s.device, dtype=torch.float32)
    L = LocalFeatures()
    L.obs_feat = torch.empty(focal_count, koe, 6, **opt)
    L.self_features = torch.empty(focal_count, 7, **opt)
    L.ped_idx = torch.empty(focal_count, kpe, device=s.device, dtype=torch.int32)
    L.obs_idx = torch.empty(focal_count, koe, device=s.device, dtype=torch.int32)
    L.g_state = torch.empty(N, 6, **opt) if (want_grad and not DETERMINISTIC_BWD) else None
    cos_p, cos_o = cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs)
    L.geom = (s.data_ptr(), N, M, focal_begin, focal_count, kp, ko, cos_p, cos_o, float(dist_threshold_ped),
              float(dist_threshold_obs))
    with torch.cuda.device(s.device):
        _lib.check(_lib.lib().piml_relfeat_self_fwd_part(
            1, _ptr(s), _ptr(d_rows), _ptr(o), _ptr(w), N, M, focal_begin, focal_count, kp, ko, cos_p, cos_o,
            float(dist_threshold_ped), float(dist_threshold_obs), None, _ptr(L.obs_feat), _ptr(L.self_features),
            _ptr(L.ped_idx), _ptr(L.obs_idx), _ptr(L.g_state), _stream()), 'piml_relfeat_self_fwd_part(LOCAL)')
    return L


def relative_features_packed_self(state, destination_rows, obstacles, desired_speed, focal_begin, focal_count,
                                  topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                                  topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, return_index=False, local=None):
    """(ped_features, obs_features, self_features (n, 7)) for the focal rows of a packed (N, 6) state:
    relative_features_packed + the model's self-feature rows [dest - p, v, a, v0] in one autograd node.
    local: the LocalFeatures of relative_features_local_part on the same buffer and geometry; only the remote half of
    the neighbour search is then left to do (results bit-identical to the one-launch form)."""
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    out = _RelativeFeaturesPackedSelf.apply(state, destination_rows, obstacles, desired_speed, int(focal_begin),
                                            int(focal_count), int(topk_ped), int(topk_obs),
                                            cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs),
                                            float(dist_threshold_ped), float(dist_threshold_obs), local)
    return out if return_index else out[:3]


def relative_features_packed_into(outs, state, destination_rows, obstacles, focal_begin, focal_count,
                                  topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                                  topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4):
    """Forward only, no autograd: recompute into the preallocated `outs` = (ped_features,
    obs_features, dest_features (n, 2) or self_features (n, 7; columns 0-1 are rewritten), ped_idx,
    obs_idx) of an earlier relative_features_packed / relative_features_packed_self call
    (static buffers of a captured step; bench.py relaunches the kernel between HIP events)."""
    s = _gpu_f32('state', state.detach())
    d_rows = _gpu_f32('destination_rows', destination_rows)
    o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
    N = s.shape[-2]
    lead = tuple(s.shape[:-2])
    C = s.numel() // max(N * 6, 1)
    base = s.data_ptr()
    _launch_relfeat_fwd(base, base + 8, base + 16, 6, None, d_rows, o, lead, C, N, int(focal_begin),
                        int(focal_count), int(topk_ped), int(topk_obs), cos_threshold(sight_angle_ped),
                        cos_threshold(sight_angle_obs), float(dist_threshold_ped), float(dist_threshold_obs),
                        s.device, outs=tuple(outs), dest_ld=outs[2].shape[-1])
    return outs


def relative_features_packed(state, destination_rows, obstacles, focal_begin, focal_count,
                             topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                             topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, return_index=False):
    """relative_features for an interleaved (..., N, 6) state buffer; the gradient w.r.t.
    `state` covers all N sources (a rank's partial sum under agent-block sharding)."""
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    out = _RelativeFeaturesPacked.apply(state, destination_rows, obstacles, int(focal_begin), int(focal_count),
                                        int(topk_ped), int(topk_obs), cos_threshold(sight_angle_ped),
                                        cos_threshold(sight_angle_obs), float(dist_threshold_ped),
                                        float(dist_threshold_obs))
    return out if return_index else out[:3]


def relative_features(position, velocity, acceleration, destination, obstacles,
                      topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                      topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4,
                      heading=None, focal_begin=0, focal_count=None, return_index=False):
    """Top-k in-view relative features of every focal agent (HIP).

    Inputs are (..., N, 2) with any leading dims (each leading index is an independent
    slice); `heading` (same shape) is the unit heading, None derives it from `velocity`
    (exact for the per-step call, t == 1).  Differentiable w.r.t. position, velocity,
    acceleration and destination.  Returns (ped_features (..., n, kp, 6), obs_features
    (..., n, ko, 6), dest_features (..., n, 2)) [+ int32 index tensors], n = focal_count.
    """
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    out = _RelativeFeatures.apply(position, velocity, acceleration, destination, obstacles, heading,
                                  focal_begin, focal_count, int(topk_ped), int(topk_obs),
                                  cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs),
                                  float(dist_threshold_ped), float(dist_threshold_obs))
    return out if return_index else out[:3]


def relative_features_self(position, velocity, acceleration, destination, obstacles, desired_speed,
                           topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                           topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, return_index=False):
    """relative_features whose third result is the model's self_features (..., N, 7) = [dest - p, v, a, v0] -- what the
    training rollout concatenates per frame (src/models/simulators.py:778-779) -- written by the same launch.
    desired_speed (..., N, 1) or (..., N).  Differentiable w.r.t. position, velocity, acceleration, destination, speed."""
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    if DETERMINISTIC_BWD:       # the atomics-free backward exists for the plain operator: features + cat, as before
        out = relative_features(position, velocity, acceleration, destination, obstacles, topk_ped, sight_angle_ped,
                                dist_threshold_ped, topk_obs, sight_angle_obs, dist_threshold_obs, return_index=True)
        v0 = desired_speed if desired_speed.dim() == position.dim() else desired_speed.unsqueeze(-1)
        out = (out[0], out[1], torch.cat((out[2], velocity, acceleration, v0), dim=-1), out[3], out[4])
        return out if return_index else out[:3]
    out = _RelativeFeaturesSelf.apply(position, velocity, acceleration, destination, obstacles, desired_speed,
                                      int(topk_ped), int(topk_obs), cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs),
                                      float(dist_threshold_ped), float(dist_threshold_obs))
    return out if return_index else out[:3]


# ------------------------------------------------------------------------------------------
# closed-form social force (MLAPM.step)
# ------------------------------------------------------------------------------------------
MLAPM_VARIANTS = {'raw': 0, 'GC': 1, 'UCY': 2}


class _MlapmStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, position, velocity, desired_speed, destination, variant, tau, A, B, C, D, theta,
                radius, dt, skip_absent):
        p = _gpu_f32('position', position)
        v = _gpu_f32('velocity', velocity)
        d = _gpu_f32('destination', destination)
        v0 = _gpu_f32('desired_speed', desired_speed)
        if p.dim() != 2 or p.shape[-1] != 2 or p.shape != v.shape or p.shape != d.shape:
            raise ValueError('position / velocity / destination must be (N, 2)')
        N = p.shape[0]
        if v0.numel() != N:
            raise ValueError(f'desired_speed must hold N={N} values, got {tuple(v0.shape)}')
        action = torch.empty_like(p)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_mlapm_step_fwd(_ptr(p), _ptr(v), _ptr(v0), _ptr(d), N, variant, tau, A, B,
                                                      C, D, theta, radius, dt, int(skip_absent), _ptr(action), None,
                                                      _stream()),
                       'piml_mlapm_step_fwd')
        ctx.save_for_backward(p, v, v0, d)
        ctx.params = (variant, tau, A, B, C, D, theta, radius, dt)
        ctx.v0_shape = tuple(desired_speed.shape)
        return action

    @staticmethod
    def backward(ctx, g_action):
        p, v, v0, d = ctx.saved_tensors
        g = _gpu_f32('g_action', g_action)
        N = p.shape[0]
        gp, gv, gd = torch.empty_like(p), torch.empty_like(p), torch.empty_like(p)
        gv0 = torch.empty(N, device=p.device, dtype=torch.float32)
        variant, tau, A, B, C, D, theta, radius, dt = ctx.params
        with torch.cuda.device(p.device):
            L = _lib.lib()
            need = int(L.piml_mlapm_bwd_workspace_floats(N, variant))    # 0: the two-role kernel (small scenes, UCY)
            ws = torch.empty(need, device=p.device, dtype=torch.float32) if need else None
            _lib.check(L.piml_mlapm_step_bwd_ws(_ptr(g), _ptr(p), _ptr(v), _ptr(v0), _ptr(d), N, variant, tau,
                                                A, B, C, D, theta, radius, dt, _ptr(gp), _ptr(gv), _ptr(gv0),
                                                _ptr(gd), _ptr(ws) if need else None, need, _stream()), 'piml_mlapm_step_bwd_ws')
        return (gp, gv, gv0.reshape(ctx.v0_shape), gd) + (None,) * 10


def mlapm_step(position, velocity, desired_speed, destination, dt, radius=0.3, version='GC', tau=0.5,
               A=0.0, B=0.0, C=0.0, D=0.0, theta=0.0, skip_absent=False):
    """MLAPM.step (src/models/mlapm.py:10-58) on the GPU; differentiable (analytic backward; with
    skip_absent the gradient is only defined for scenes without NaN agents)."""
    if version not in MLAPM_VARIANTS:
        raise NotImplementedError(version)
    return _MlapmStep.apply(position, velocity, desired_speed, destination, MLAPM_VARIANTS[version],
                            float(tau), float(A), float(B), float(C), float(D), float(theta), float(radius),
                            float(dt), bool(skip_absent))


def mlapm_rollout_step(traj_position, traj_velocity, desired_speed, destination, frame_counter, done_counter, dt, radius=0.3,
                       version='GC', tau=0.5, A=0.0, B=0.0, C=0.0, D=0.0, theta=0.0):
    """One frame of the simulation loop of src/main_mlapm.py:18-36 in one launch (piml_mlapm_rollout_step): reads frame
    `frame_counter - 1` of the (frames, N, 2) trajectories (agents that arrived in it are absent from now on), writes frame
    `frame_counter` and advances the counter on the device.  frame_counter: int64 (1,), done_counter: zeroed int32 (1,).
    No autograd (inference)."""
    if version not in MLAPM_VARIANTS:
        raise NotImplementedError(version)
    tp, tv = _gpu_f32('traj_position', traj_position), _gpu_f32('traj_velocity', traj_velocity)
    v0, d = _gpu_f32('desired_speed', desired_speed), _gpu_f32('destination', destination)
    if tp.dim() != 3 or tp.shape != tv.shape or tp.shape[-1] != 2 or not (tp.is_contiguous() and tv.is_contiguous()):
        raise ValueError('trajectories must be contiguous (frames, N, 2)')
    frames, N = int(tp.shape[0]), int(tp.shape[1])
    if v0.numel() != N or tuple(d.shape) != (N, 2):
        raise ValueError('desired_speed (N, 1) and destination (N, 2) expected')
    if frame_counter.dtype != torch.int64 or done_counter.dtype != torch.int32 or not (frame_counter.is_cuda and done_counter.is_cuda):
        raise ValueError('frame_counter: int64 (1,), done_counter: int32 (1,), both on the GPU')
    with torch.cuda.device(tp.device):
        _lib.check(_lib.lib().piml_mlapm_rollout_step(_ptr(tp), _ptr(tv), _ptr(v0.contiguous()), _ptr(d.contiguous()), frames, N,
                                                      MLAPM_VARIANTS[version], float(tau), float(A), float(B), float(C), float(D),
                                                      float(theta), float(radius), float(dt), _ptr(frame_counter),
                                                      _ptr(done_counter), _stream()), 'piml_mlapm_rollout_step')


# ------------------------------------------------------------------------------------------
# collisions (Pedestrians.collision_detection / calculate_collision_label)
# ------------------------------------------------------------------------------------------
def collision_matrix(position, threshold, minus_identity=True):
    """(..., N, 2) -> (..., N, N): [|p_j - p_i| < threshold] (- I), NaN -> 0 (data.py:549-564)."""
    p = _gpu_f32('position', position)
    N = p.shape[-2]
    S = p.numel() // max(N * 2, 1)
    coll = torch.empty(*p.shape[:-2], N, N, device=p.device, dtype=torch.float32)
    with torch.cuda.device(p.device):
        _lib.check(_lib.lib().piml_collision_matrix(_ptr(p), S, N, float(threshold), int(bool(minus_identity)),
                                                    _ptr(coll), _stream()), 'piml_collision_matrix')
    return coll


def collision_detection(position, threshold, real_position=None):
    """Pedestrians.collision_detection (data.py:537-601) incl. both "friends" rules."""
    if position.dim() not in (3, 4):
        raise ValueError('position must be (t,n,2) / (c,n,2) or (c,t,n,2)')
    coll = collision_matrix(position.detach(), threshold, True)
    N = coll.shape[-1]
    L = _lib.lib()
    with torch.cuda.device(coll.device):
        if real_position is not None:
            assert real_position.dim() == 3, 'Value Error: real_position only supports 3 dimensional inputs (t,n,2)'
            if position.dim() != 3:
                raise ValueError('real_position requires a 3-dimensional position')
            base = collision_matrix(real_position.detach(), threshold, False)
            _lib.check(L.piml_collision_friends(_ptr(coll), _ptr(base), 1, coll.shape[0], base.shape[0], N,
                                                _stream()), 'piml_collision_friends')
        elif position.dim() == 3:
            # the rule zeroes pairs colliding in MORE than 25 slices: with <= 25 slices (the (c,n,2) calls of
            # the fine-tuning loop) it cannot trigger and the pass over the matrix is skipped
            if coll.shape[0] > 25:
                _lib.check(L.piml_collision_friends(_ptr(coll), _ptr(coll), 1, coll.shape[0], coll.shape[0], N,
                                                    _stream()), 'piml_collision_friends')
        else:
            _lib.check(L.piml_collision_friends(_ptr(coll), None, coll.shape[0], coll.shape[1], 0, N,
                                                _stream()), 'piml_collision_friends')
    return coll


_THRESHOLDS = {}


COLLISION_GRID = _os.environ.get('PIML_COLLISION_GRID', '1') != '0'


def collision_counts(position, thresholds):
    """collision_detection(position (S,N,2), thr).sum(-1) for several thresholds in one sweep,
    without the (S,N,N) matrices.  Returns (len(thresholds), S, N)."""
    p = _gpu_f32('position', position.detach())
    if p.dim() != 3:
        raise ValueError('position must be (S, N, 2)')
    S, N = p.shape[0], p.shape[1]
    key = (p.device, tuple(float(t) for t in thresholds))
    thr = _THRESHOLDS.get(key)
    if thr is None:                 # cached: a host-to-device copy is not capturable into a graph
        thr = _THRESHOLDS[key] = torch.tensor(key[1], device=p.device, dtype=torch.float32)
    counts = torch.empty(len(thresholds), S, N, device=p.device, dtype=torch.float32)
    with torch.cuda.device(p.device):
        if S > 25 and len(thresholds) <= 4 and len(thresholds) * N * N <= (1 << 28):
            # many slices (evaluation rollouts): the two-sweep parallel form with an (nthr, N, N) int32 scratch
            totals = torch.zeros(len(thresholds), N, N, device=p.device, dtype=torch.int32)
            if N <= 8192 and COLLISION_GRID:       # per-frame cell grid: O(N x occupancy) pair tests per slice
                _lib.check(_lib.lib().piml_collision_counts_grid(_ptr(p), S, N, _ptr(thr), len(thresholds), _ptr(totals),
                                                                 _ptr(counts), _stream()), 'piml_collision_counts_grid')
            else:
                _lib.check(_lib.lib().piml_collision_counts_scratch(_ptr(p), S, N, _ptr(thr), len(thresholds), _ptr(totals),
                                                                    _ptr(counts), _stream()), 'piml_collision_counts_scratch')
        else:
            _lib.check(_lib.lib().piml_collision_counts(_ptr(p), S, N, _ptr(thr), len(thresholds), _ptr(counts),
                                                        _stream()), 'piml_collision_counts')
    return counts


def collision_counts_frames(frames, thresholds):
    """[collision_counts(f, thresholds) for f in frames] in ONE launch (piml_collision_counts_frames): `frames` = up to 32
    (S, N, 2) tensors of one shape with S <= 25 (the frames of a training rollout, each counted on its own by
    src/models/simulators.py:708-715).  Returns a list of (len(thresholds), S, N) views of one buffer."""
    import ctypes
    ps = [_gpu_f32('position', f.detach()) for f in frames]
    if not ps:
        return []
    S, N = ps[0].shape[0], ps[0].shape[1]
    if any(p.dim() != 3 or tuple(p.shape) != (S, N, 2) for p in ps):
        raise ValueError('collision_counts_frames: (S, N, 2) frames of one shape expected')
    if len(ps) > 32 or S > 25 or not 1 <= len(thresholds) <= 4:
        return [collision_counts(p, thresholds) for p in ps]
    dev = ps[0].device
    key = (dev, tuple(float(t) for t in thresholds))
    thr = _THRESHOLDS.get(key)
    if thr is None:
        thr = _THRESHOLDS[key] = torch.tensor(key[1], device=dev, dtype=torch.float32)
    counts = torch.empty(len(ps), len(thresholds), S, N, device=dev, dtype=torch.float32)
    arr = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().piml_collision_counts_frames(arr, len(ps), S, N, _ptr(thr), len(thresholds), _ptr(counts), _stream()),
                   'piml_collision_counts_frames')
    return list(counts.unbind(0))


_LOSS_TICKETS = {}
_ZEROS_RO = {}


def _zeros_ro(shape, dev):
    """A float32 zero tensor that is only ever READ (a gradient that did not arrive, handed to a kernel as zeros): one persistent
    tensor per (shape, device) instead of a fill launch per use.  Made outside stream capture only -- a fill recorded into one
    graph has not run when another graph reads the tensor; while capturing, an unknown shape gets a fresh torch.zeros."""
    key = (tuple(shape), dev.index if isinstance(dev, torch.device) else dev)
    z = _ZEROS_RO.get(key)
    if z is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros(shape, device=dev, dtype=torch.float32)
        z = torch.zeros(shape, device=dev, dtype=torch.float32)
        if len(_ZEROS_RO) < 64:          # (never evicted: a captured graph may hold the pointer)
            _ZEROS_RO[key] = z
    return z


def multi_copy(dsts, srcs):
    """dst[i].copy_(src[i]) for lists of GPU tensors in ONE launch (piml_multi_copy) when every pair is contiguous, of one
    dtype and shape and on the current stream's device; anything else goes through torch._foreach_copy_."""
    import ctypes
    dsts, srcs = list(dsts), list(srcs)
    ok = len(dsts) == len(srcs) and len(dsts) > 0 and all(
        d.is_cuda and s.is_cuda and d.device == s.device == dsts[0].device and d.dtype == s.dtype and d.shape == s.shape
        and d.is_contiguous() and s.is_contiguous() and not d.requires_grad for d, s in zip(dsts, srcs))
    if not ok:
        torch._foreach_copy_(dsts, srcs)
        return
    n = len(dsts)
    da = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    sa = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    ba = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d in dsts])
    with torch.cuda.device(dsts[0].device):
        _lib.check(_lib.lib().piml_multi_copy(da, sa, ba, n, _stream()), 'piml_multi_copy')


def rollout_prologue(data, t_start):
    """The prologue of the differentiable training rollout in one launch (piml_rollout_prologue; src/models/simulators.py:672-697,
    :707) or None when the batch is not the expected (C, T, N, .) float32 / int64 layout on the GPU (the caller then uses the
    torch operators).  Returns dict(p, v, a, dest (C, N, 2), dest_idx (C, N) int64, new_flag_u8 (C, T, N) uint8, mask_pred
    (C, T, N) int64, gates (T) bool, gates_f (T) float32, speed (C, N, 1), nan_flag () int32)."""
    f32 = [data.position, data.velocity, data.acceleration, data.destination, data.mask_p, data.mask_p_pred, data.self_features]
    if data.position.dim() != 4 or not all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in f32) or \
            data.dest_idx.dtype != torch.int64 or not data.dest_idx.is_contiguous() or not data.dest_idx.is_cuda or \
            data.self_features.shape[-1] != 7 or tuple(data.mask_p.shape) != tuple(data.position.shape[:3]) or \
            tuple(data.mask_p_pred.shape) != tuple(data.position.shape[:3]) or tuple(data.dest_idx.shape) != tuple(data.position.shape[:3]):
        return None
    C, T, N = data.position.shape[:3]
    if C * T * N > (1 << 20) or not 0 <= int(t_start) < T:      # one workgroup walks the arrays: training windows, not clips
        return None
    dev = data.position.device
    opt = dict(device=dev, dtype=torch.float32)
    out = dict(p=torch.empty(C, N, 2, **opt), v=torch.empty(C, N, 2, **opt), a=torch.empty(C, N, 2, **opt),
               dest=torch.empty(C, N, 2, **opt), dest_idx=torch.empty(C, N, device=dev, dtype=torch.int64),
               new_flag_u8=torch.empty(C, T, N, device=dev, dtype=torch.uint8),
               mask_pred=torch.empty(C, T, N, device=dev, dtype=torch.int64), gates=torch.empty(T, device=dev, dtype=torch.bool),
               gates_f=torch.empty(T, **opt), speed=torch.empty(C, N, 1, **opt), nan_flag=torch.empty((), device=dev, dtype=torch.int32))
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().piml_rollout_prologue(
            _ptr(data.position), _ptr(data.velocity), _ptr(data.acceleration), _ptr(data.destination), _ptr(data.dest_idx),
            _ptr(data.mask_p), _ptr(data.mask_p_pred), _ptr(data.self_features), C, T, N, int(t_start),
            _ptr(out['p']), _ptr(out['v']), _ptr(out['a']), _ptr(out['dest']), _ptr(out['dest_idx']), _ptr(out['new_flag_u8']),
            _ptr(out['mask_pred']), _ptr(out['gates']), _ptr(out['gates_f']), _ptr(out['speed']), _ptr(out['nan_flag']), _stream()),
            'piml_rollout_prologue')
    return out


class _RolloutLosses(torch.autograd.Function):
    """piml_rollout_losses / piml_rollout_losses_bwd (include/piml_hip.h): the three loss sums of the fine-tuning rollout and
    their gradient with respect to the predicted positions."""

    @staticmethod
    def forward(ctx, p, labels, mask_pred, gates, collisions, hard_collisions, abnormal_mask, time_decay):
        L = _lib.lib()
        pc = _gpu_f32('p', p.detach())
        C, T, N = pc.shape[0], pc.shape[1], pc.shape[2]
        lab = _gpu_f32('labels', labels.detach())
        if tuple(lab.shape[:3]) != (C, T, N) or lab.shape[-1] < 2 or pc.shape[-1] != 2:
            raise ValueError('p must be (C, T, N, 2), labels (C, T, N, >= 2)')
        mp = mask_pred.detach().contiguous()
        if mp.dtype != torch.int64 or tuple(mp.shape) != (C, T, N):
            raise ValueError('mask_pred must be int64 (C, T, N)')
        g8 = gates.detach().contiguous().view(torch.uint8)
        dev = pc.device
        opt = dict(device=dev, dtype=torch.float32)
        co = None if collisions is None else _gpu_f32('collisions', collisions.detach())
        hc = None if hard_collisions is None else _gpu_f32('hard_collisions', hard_collisions.detach())
        ab = None if abnormal_mask is None else _gpu_f32('abnormal_mask', abnormal_mask.detach().reshape(-1))
        out = torch.empty(3, **opt)
        gm, gc, gh = torch.empty_like(pc), torch.empty_like(pc), torch.empty_like(pc)
        blocks = L.piml_rollout_losses_blocks(C, N)
        partial, ticket = None, None
        if blocks > 1:
            partial = torch.empty(blocks, 6, **opt)
            ticket = _LOSS_TICKETS.get(dev)
            if ticket is None:       # zeroed once; the launch leaves it zero
                ticket = _LOSS_TICKETS[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(L.piml_rollout_losses(_ptr(pc), _ptr(lab), lab.shape[-1], _ptr(mp), _ptr(g8), _ptr(co), _ptr(hc), _ptr(ab),
                                             C, T, N, float(time_decay), _ptr(out), _ptr(gm), _ptr(gc), _ptr(gh),
                                             _ptr(partial), _ptr(ticket), _stream()), 'piml_rollout_losses')
        ctx.save_for_backward(gm, gc, gh)
        ctx.set_materialize_grads(False)
        # three scalars, not one (3,) tensor: indexing that one costs a zero fill, a copy and an accumulation per term backward
        return out[0], out[1], out[2]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g0, g1, g2):
        if (g0 is None and g1 is None and g2 is None) or not ctx.needs_input_grad[0]:
            return (None,) * 8
        gm, gc, gh = ctx.saved_tensors
        gs = [None if g is None else _gpu_f32('g_out', g) for g in (g0, g1, g2)]
        gp = torch.empty_like(gm)
        with torch.cuda.device(gm.device):
            _lib.check(_lib.lib().piml_rollout_losses_bwd(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gm), _ptr(gc), _ptr(gh),
                                                          gm.numel(), _ptr(gp), _stream()), 'piml_rollout_losses_bwd')
        return (gp,) + (None,) * 7


class _RolloutLossesFrames(torch.autograd.Function):
    """piml_rollout_losses_frames / _bwd: the rollout losses on the frames' collision count records as produced (gated inside),
    with the weighted total and the statistics the step logs.  Outputs: total, mse, w_coll * focus(collisions),
    w_hard * focus(hard collisions) (differentiable w.r.t. p) and stats (3,) = [sum of the gated collisions, of the hard ones,
    number of entries with mask_pred == 1] (not differentiable)."""

    @staticmethod
    def forward(ctx, p, labels, mask_pred, gates, focus, abnormal_mask, time_decay, w_coll, w_hard, *frames):
        import ctypes
        L = _lib.lib()
        pc = _gpu_f32('p', p.detach())
        C, T, N = pc.shape[0], pc.shape[1], pc.shape[2]
        lab = _gpu_f32('labels', labels.detach())
        if tuple(lab.shape[:3]) != (C, T, N) or lab.shape[-1] < 2 or pc.shape[-1] != 2 or len(frames) != T or T > 32:
            raise ValueError('p must be (C, T <= 32, N, 2), labels (C, T, N, >= 2), one count record (or None) per frame')
        mp = mask_pred.detach().contiguous()
        if mp.dtype != torch.int64 or tuple(mp.shape) != (C, T, N):
            raise ValueError('mask_pred must be int64 (C, T, N)')
        g8 = gates.detach().contiguous().view(torch.uint8)
        dev = pc.device
        opt = dict(device=dev, dtype=torch.float32)
        fr = [None if f is None else _gpu_f32('count record', f.detach()) for f in frames]
        for f in fr:
            if f is not None and tuple(f.shape) != (2, C, N):
                raise ValueError('count records must be (2, C, N)')
        table = (ctypes.c_void_p * T)(*[None if f is None else f.data_ptr() for f in fr])
        ab = None if abnormal_mask is None else _gpu_f32('abnormal_mask', abnormal_mask.detach().reshape(-1))
        out = torch.empty(12, **opt)
        gm, gc, gh = torch.empty_like(pc), torch.empty_like(pc), torch.empty_like(pc)
        blocks = L.piml_rollout_losses_blocks(C, N)
        partial, ticket = None, None
        if blocks > 1:
            partial = torch.empty(blocks, 6, **opt)
            ticket = _LOSS_TICKETS.get(dev)
            if ticket is None:       # zeroed once; the launch leaves it zero
                ticket = _LOSS_TICKETS[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(L.piml_rollout_losses_frames(_ptr(pc), _ptr(lab), lab.shape[-1], _ptr(mp), _ptr(g8), table, int(bool(focus)),
                                                    _ptr(ab), C, T, N, float(time_decay), float(w_coll), float(w_hard), _ptr(out),
                                                    _ptr(gm), _ptr(gc), _ptr(gh), _ptr(partial), _ptr(ticket), _stream()),
                       'piml_rollout_losses_frames')
        ctx.save_for_backward(gm, gc, gh)
        ctx.weights = (float(w_coll), float(w_hard))
        ctx.nin = 9 + len(frames)
        ctx.set_materialize_grads(False)
        stats = out[3:6]
        ctx.mark_non_differentiable(stats)
        return out[6], out[0], out[7], out[8], stats

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_total, g_mse, g_collw, g_hardw, _gs):
        if all(g is None for g in (g_total, g_mse, g_collw, g_hardw)) or not ctx.needs_input_grad[0]:
            return (None,) * ctx.nin
        gm, gc, gh = ctx.saved_tensors
        w_coll, w_hard = ctx.weights
        gs = [None if g is None else _gpu_f32('g_out', g) for g in (g_mse, g_collw, g_hardw)]
        gt = _gpu_f32('g_out', g_total) if g_total is not None else torch.zeros((), device=gm.device, dtype=torch.float32)
        gp = torch.empty_like(gm)
        with torch.cuda.device(gm.device):
            _lib.check(_lib.lib().piml_rollout_losses_frames_bwd(_ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gt), w_coll, w_hard,
                                                                 _ptr(gm), _ptr(gc), _ptr(gh), gm.numel(), _ptr(gp), _stream()),
                       'piml_rollout_losses_frames_bwd')
        return (gp,) + (None,) * (ctx.nin - 1)


def rollout_losses_frames(p, labels, mask_pred, gates, count_frames, focus, abnormal_mask=None, time_decay=1.0, w_coll=1.0,
                          w_hard=1.0):
    """rollout_losses on the frames' collision count records as ops.collision_counts left them (one (2, C, N) tensor or None
    per frame; gated by `gates` inside), returning (total = mse + w_coll * focus_c + w_hard * focus_h, mse, w_coll * focus_c,
    w_hard * focus_h, stats) -- stats (3,) = [sum of the gated collisions, sum of the gated hard collisions, number of
    mask_pred == 1 entries], the scalars the training step logs (src/models/simulators.py:708-728, 790-819).  focus False: the
    counts only feed the statistics (collision loss switched off)."""
    return _RolloutLossesFrames.apply(p, labels, mask_pred, gates, bool(focus), abnormal_mask, float(time_decay), float(w_coll),
                                      float(w_hard), *count_frames)


class _CollisionPredLoss(torch.autograd.Function):
    """inputs: gates_f (T,), t_start, T, weight, n frames, then the frames' predictions, then their pedestrian features.
    outputs: weighted BCE sum, accuracy (piml_collision_pred_loss)."""

    @staticmethod
    def forward(ctx, gates_f, t_start, T, weight, nframes, *tensors):
        import ctypes
        L = _lib.lib()
        preds = [_gpu_f32('prediction', t.detach()) for t in tensors[:nframes]]
        feats = [_gpu_f32('ped_features', t.detach()) for t in tensors[nframes:]]
        n, k, ld = preds[0].numel(), preds[0].shape[-1], feats[0].shape[-1]
        if any(q.numel() != n for q in preds) or any(f.numel() != n * ld or f.shape[-2] != k for f in feats) or ld < 4:
            raise ValueError('collision_pred_loss: predictions (..., k) and pedestrian features (..., k, >= 4) of one shape per frame')
        gf = _gpu_f32('gates', gates_f.detach()).reshape(-1)
        if gf.numel() != T or not 0 <= t_start or t_start + nframes > T or not 1 <= nframes <= 32:
            raise ValueError('collision_pred_loss: gates (T,), t_start + frames <= T, at most 32 frames')
        dev = preds[0].device
        opt = dict(device=dev, dtype=torch.float32)
        out = torch.empty(2, **opt)
        grad = torch.empty(nframes, n, **opt)
        blocks = L.piml_collision_pred_loss_blocks(n, nframes)
        partial, ticket = None, None
        if blocks > 1:
            partial = torch.empty(blocks, 2, **opt)
            ticket = _LOSS_TICKETS.get(dev)
            if ticket is None:       # zeroed once; the launch leaves it zero
                ticket = _LOSS_TICKETS[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
        ptab = (ctypes.c_void_p * nframes)(*[q.data_ptr() for q in preds])
        ftab = (ctypes.c_void_p * nframes)(*[f.data_ptr() for f in feats])
        with torch.cuda.device(dev):
            _lib.check(L.piml_collision_pred_loss(ptab, ftab, nframes, n, k, ld, _ptr(gf), int(t_start), int(T), float(weight), _ptr(out),
                                                  _ptr(grad), _ptr(partial), _ptr(ticket), _stream()), 'piml_collision_pred_loss')
        ctx.save_for_backward(grad)
        ctx.meta = (nframes, [tuple(t.shape) for t in tensors[:nframes]])
        ctx.set_materialize_grads(False)
        acc = out[1]
        ctx.mark_non_differentiable(acc)
        return out[0], acc

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, _g_acc):
        nframes, shapes = ctx.meta
        none = (None,) * (5 + 2 * nframes)
        if g_loss is None or not any(ctx.needs_input_grad[5:5 + nframes]):
            return none
        grad, = ctx.saved_tensors
        gp = _scaled_grad(g_loss, grad)
        return (None,) * 5 + tuple(gp[f].view(shapes[f]) for f in range(nframes)) + (None,) * nframes


def collision_pred_loss(pred_frames, feature_frames, gates_f, t_start, T, weight):
    """The collision-prediction loss of `pinnsf_bm` over the frames of a training rollout (src/models/simulators.py:731-733, 826-830) as
    one launch each way: pred_frames / feature_frames = the model's last output (C, N, k) and the pedestrian features (C, N, k, 6) of
    rollout frames t_start, t_start + 1, ...; gates_f (T,) 0 / 1 floats.  Returns (collision_pred_weight * BCE sum, accuracy), the
    reference's two scalars; the labels (calculate_collision_label) are evaluated inside."""
    if len(pred_frames) != len(feature_frames) or not pred_frames:
        raise ValueError('collision_pred_loss: one feature tensor per prediction tensor')
    if not pred_frames[0].is_cuda:
        raise _lib.PimlHipError('collision_pred_loss: expected GPU tensors (piml_amd has no CPU path)')
    return _CollisionPredLoss.apply(gates_f, int(t_start), int(T), float(weight), len(pred_frames), *pred_frames, *feature_frames)


_CONST_ONES = {}           # data pointer -> the persistent tensor that holds 1.0 and is never written (register_const_one)


def register_const_one(t):
    """`t` (0-dim float32) holds 1.0 and nobody ever writes it (the `gradient=` of a captured step's backward): a loss node whose
    upstream gradient IS this tensor hands out its stored gradient fields without a scaling launch.  The registry keeps the tensor
    alive -- its address must never come back as somebody else's gradient."""
    _CONST_ONES[t.data_ptr()] = t


def _scaled_grad(g_up, grad):
    """g_up * grad (one launch) -- grad itself when g_up is a registered constant one"""
    if g_up.data_ptr() in _CONST_ONES and g_up.numel() == 1:
        return grad
    g = _gpu_f32('g_out', g_up)
    out = torch.empty_like(grad)
    with torch.cuda.device(grad.device):
        _lib.check(_lib.lib().piml_collision_pred_loss_bwd(_ptr(g), _ptr(grad), grad.numel(), _ptr(out), _stream()), 'piml_collision_pred_loss_bwd')
    return out


class _PointwiseLosses(torch.autograd.Function):
    """inputs: pred (rows, 2), labels (rows, >= 6 [+ k]), reg_weight, msgs | None, coll_pred (rows, k) | None.
    outputs: loss, mse, reg, cp (piml_pointwise_losses)."""

    @staticmethod
    def forward(ctx, pred, labels, reg_weight, msgs, coll):
        L = _lib.lib()
        pc = _gpu_f32('pred', pred.detach())
        lab = _gpu_f32('labels', labels.detach())
        rows = pc.shape[0]
        if pc.dim() != 2 or pc.shape[1] != 2 or lab.dim() != 2 or lab.shape[0] != rows or lab.shape[1] < 6:
            raise ValueError('pointwise_losses: pred (rows, 2), labels (rows, >= 6)')
        mc = None if msgs is None else _gpu_f32('msgs', msgs.detach())
        cc = None if coll is None else _gpu_f32('coll_pred', coll.detach())
        k = 0
        if cc is not None:
            k = cc.numel() // rows
            if cc.numel() != rows * k or lab.shape[1] < 6 + k:
                raise ValueError('pointwise_losses: coll_pred (rows, k) needs labels (rows, >= 6 + k)')
        nmsg = 0 if mc is None else mc.numel()
        dev = pc.device
        opt = dict(device=dev, dtype=torch.float32)
        out = torch.empty(4, **opt)
        grad = torch.empty(2 * rows + nmsg + rows * k, **opt)
        blocks = L.piml_pointwise_losses_blocks(rows, nmsg, k)
        partial, ticket = None, None
        if blocks > 1:
            partial = torch.empty(blocks, 3, **opt)
            ticket = _LOSS_TICKETS.get(dev)
            if ticket is None:       # zeroed once; the launch leaves it zero
                ticket = _LOSS_TICKETS[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            _lib.check(L.piml_pointwise_losses(_ptr(pc), _ptr(lab), lab.shape[1], rows, _ptr(mc), nmsg, float(reg_weight), _ptr(cc), k,
                                               _ptr(out), _ptr(grad), _ptr(partial), _ptr(ticket), _stream()), 'piml_pointwise_losses')
        ctx.save_for_backward(grad)
        ctx.meta = (rows, nmsg, k, tuple(pred.shape), None if msgs is None else tuple(msgs.shape), None if coll is None else tuple(coll.shape))
        ctx.set_materialize_grads(False)
        mse, reg, cp = out[1], out[2], out[3]
        ctx.mark_non_differentiable(mse, reg, cp)        # (the step logs them; the gradient flows through the total)
        return out[0], mse, reg, cp

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_loss, _a, _b, _c):
        if g_loss is None:
            return (None,) * 5
        rows, nmsg, k, s_pred, s_msgs, s_coll = ctx.meta
        grad, = ctx.saved_tensors
        g = _scaled_grad(g_loss, grad)
        need = ctx.needs_input_grad
        return (g[:2 * rows].view(s_pred) if need[0] else None, None, None,
                g[2 * rows:2 * rows + nmsg].view(s_msgs) if (nmsg and need[3]) else None,
                g[2 * rows + nmsg:].view(s_coll) if (k and need[4]) else None)


def pointwise_losses(pred, labels, reg_weight=0.0, msgs=None, coll_pred=None):
    """The loss terms of a pointwise pre-training batch (src/models/simulators.py:333-352, pinnsf_interaction 'sim') as one launch:
    (loss, mse, reg, cp) with mse = F.mse_loss(pred, labels[:, 4:6], 'sum'), reg = sum(reg_weight |msgs|) (msgs given),
    cp = F.binary_cross_entropy(coll_pred, labels[:, 6:], 'sum') (coll_pred given), loss = their sum in the reference's order.  The
    gradient flows through `loss`; the three terms are for the log."""
    if not pred.is_cuda:
        raise _lib.PimlHipError('pointwise_losses: expected GPU tensors (piml_amd has no CPU path)')
    return _PointwiseLosses.apply(pred, labels, float(reg_weight), msgs, coll_pred)


def rollout_losses(p, labels, mask_pred, gates, collisions=None, hard_collisions=None, abnormal_mask=None, time_decay=1.0):
    """(mse, collision-focus loss of `collisions`, of `hard_collisions`) of a training rollout -- BaseSimulator's
    multiple_rollout_mse_loss and multiple_rollout_collision_loss (reduction 'sum') on the masked / gated positions, as
    test_multiple_rollouts_for_training puts them together (src/models/simulators.py:172-249, 790-819) -- as one launch
    forward and one backward.  p (C, T, N, 2) carries the gradient; labels (C, T, N, >= 2) unmasked; mask_pred (C, T, N) int64;
    gates (T) bool.  Returns the three scalars."""
    return _RolloutLosses.apply(p, labels, mask_pred, gates, collisions, hard_collisions, abnormal_mask, float(time_decay))


def collision_label(ped_features):
    """Pedestrians.calculate_collision_label (data.py:514-535): (..., k, >=4) -> (..., k)."""
    f = _gpu_f32('ped_features', ped_features.detach())
    rows = f.numel() // max(f.shape[-1], 1)
    out = torch.empty(f.shape[:-1], device=f.device, dtype=torch.float32)
    with torch.cuda.device(f.device):
        _lib.check(_lib.lib().piml_collision_label(_ptr(f), rows, f.shape[-1], _ptr(out), _stream()),
                   'piml_collision_label')
    return out


_CALC_ACC = {  # utils.py:44-81: (A, B, C, D, theta)
    ('v0', 'gc1560'): (8.75, -2.5, 0., 0., 0.), ('v0', 'gc2344'): (8.75, -2.5, 0., 0., 0.),
    ('v0', 'ucy'): (10.67, -3.33, 0., 0., 0.),
    ('v1', 'gc1560'): (8.75, -2.5, 0., 0., 0.), ('v1', 'gc2344'): (8.75, -2.5, 0., 0., 0.),
    ('v1', 'ucy'): (10.67, -3.33, 0., 0., 0.),
    ('v2', 'gc2344'): (9.00, -2.75, 0.06, -0.3, 10 * 3.1415 / 180),
}


def calc_acceleration(relative_data, equation_version='v0', dataset='gc1560', eps=1e-6):
    """utils.calc_acceleration (src/utils/utils.py:31-100), forward only (it is a label)."""
    if (equation_version, dataset) not in _CALC_ACC:
        raise NotImplementedError((equation_version, dataset))
    A, B, C, D, th = _CALC_ACC[(equation_version, dataset)]
    r = _gpu_f32('relative_data', relative_data.detach())
    if equation_version == 'v2' and r.dim() not in (3, 4):
        raise ValueError
    rows = r.numel() // max(r.shape[-1], 1)
    out = torch.empty(*r.shape[:-1], 2, device=r.device, dtype=torch.float32)
    with torch.cuda.device(r.device):
        _lib.check(_lib.lib().piml_calc_acceleration(_ptr(r), rows, r.shape[-1], int(equation_version[1]), A, B, C,
                                                     D, th, float(eps), _ptr(out), _stream()),
                   'piml_calc_acceleration')
    return out


def relative_features_into(outs, position, velocity, acceleration, destination, obstacles,
                           topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                           topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, tick=None):
    """Forward only, no autograd, no allocation: per-step features of (..., N, 2) state into the
    preallocated `outs` = (ped_features, obs_features, self_features, ped_idx, obs_idx), where
    dest_features land in columns 0..1 of the (..., N, F) self_features buffer (row stride F).
    Used by the captured inference-rollout step; `tick` (one-element int64 tensor) is advanced by one by the same launch."""
    p, v, a, d = [_gpu_f32(n, x) for n, x in (('position', position), ('velocity', velocity),
                                              ('acceleration', acceleration), ('destination', destination))]
    o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
    N = p.shape[-2]
    lead = tuple(p.shape[:-2])
    C = p.numel() // max(N * 2, 1)
    _launch_relfeat_fwd(_ptr(p), _ptr(v), _ptr(a), 2, None, d, o, lead, C, N, 0, N, int(topk_ped), int(topk_obs),
                        cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs), float(dist_threshold_ped),
                        float(dist_threshold_obs), p.device, outs=tuple(outs), dest_ld=outs[2].shape[-1], tick=tick)
    return outs


def rollout_step(st, data, a_next, remove_arrived=True, ksum=None):
    """One launch of the fused integrator epilogue (piml_rollout_step) on the rollout state `st`
    built by BaseSimulator._rollout_state (buffers updated in place).
    ksum = (pred_ped (..., kp, 2), pred_obs (..., ko, 2) or None, tau) with a_next = None: the bottleneck variants' network
    epilogue (pinnsf_epilogue_ksum's arithmetic on the rows of st.selff) runs inside the same launch (piml_rollout_step_ksum)."""
    C = st.p.numel() // max(st.p.shape[-2] * 2, 1)
    N, T = st.p.shape[-2], st.T
    wp = data.waypoints
    args = (_ptr(st.p), _ptr(st.v), _ptr(st.a), _ptr(st.dest), _ptr(st.dest_idx), _ptr(st.hist), st.hist.shape[-1],
            _ptr(a_next), _ptr(st.waypoints), wp.shape[-3], int(wp.dim() > 3), _ptr(st.dest_num),
            _ptr(st.series['position']), _ptr(st.series['velocity']), _ptr(st.series['acceleration']),
            _ptr(st.series['destination']), _ptr(st.series['dest_idx']), _ptr(st.series['self_features']),
            st.selff.shape[-1], _ptr(st.new_flag_u8), _ptr(st.p_res), _ptr(st.v_res), _ptr(st.a_res),
            _ptr(st.mask_new), _ptr(st.selff), _ptr(st.desired_speed), _ptr(st.t), C, T, N,
            float(data.time_unit), int(remove_arrived), _stream())
    with torch.cuda.device(st.p.device):
        if ksum is not None:
            pp, po, tau = ksum
            pp = _gpu_f32('pred_ped', pp)
            po = _gpu_f32('pred_obs', po) if po is not None else None
            if pp.numel() != C * N * pp.shape[-2] * 2 or (po is not None and po.numel() != C * N * po.shape[-2] * 2):
                raise ValueError('rollout_step: per-neighbour predictions (..., k, 2) of the rollout\'s agents expected')
            _lib.check(_lib.lib().piml_rollout_step_ksum(_ptr(pp), pp.shape[-2], _ptr(po), po.shape[-2] if po is not None else 1,
                                                         float(tau), *args), 'piml_rollout_step_ksum')
        else:
            _lib.check(_lib.lib().piml_rollout_step(*args), 'piml_rollout_step')


# ------------------------------------------------------------------------------------------------
# Glue around the PINNSF network's GEMMs (piml_amd/csrc/mlpglue.hip).  The GEMMs themselves stay
# torch.addmm / torch.mm (rocBLAS / hipBLASLt); these operators replace the launch-bound chains of
# small torch kernels around them.
# ------------------------------------------------------------------------------------------------
class _PinnsfEpilogue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, acc_ped, acc_obs, self_features, tau, agent_norm):
        sf = _gpu_f32('self_features', self_features)
        ap = _gpu_f32('acc_ped', acc_ped)
        ao = _gpu_f32('acc_obs', acc_obs) if acc_obs is not None else None
        rows = sf.numel() // 7
        out = torch.empty_like(ap)
        with torch.cuda.device(sf.device):
            if agent_norm:
                C, N = sf.shape[0], sf.shape[1]
                _lib.check(_lib.lib().piml_pinnsf_epilogue_agentnorm_fwd(_ptr(ap), _ptr(ao), _ptr(sf), C, N,
                                                                         float(tau), _ptr(out), _stream()),
                           'piml_pinnsf_epilogue_agentnorm_fwd')
            else:
                _lib.check(_lib.lib().piml_pinnsf_epilogue_fwd(_ptr(ap), _ptr(ao), _ptr(sf), rows, float(tau),
                                                               _ptr(out), _stream()), 'piml_pinnsf_epilogue_fwd')
        ctx.save_for_backward(sf)
        ctx.tau, ctx.has_obs, ctx.agent_norm = float(tau), acc_obs is not None, bool(agent_norm)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:       # structurally reachable but nothing flows here: propagate "no gradient", launch nothing
            return (None,) * 5
        (sf,) = ctx.saved_tensors
        g = g.contiguous()
        g_self = None
        if ctx.needs_input_grad[2]:
            g_self = torch.empty_like(sf)
            with torch.cuda.device(sf.device):
                if ctx.agent_norm:
                    _lib.check(_lib.lib().piml_pinnsf_epilogue_agentnorm_bwd(_ptr(g), _ptr(sf), sf.shape[0], sf.shape[1],
                                                                             ctx.tau, _ptr(g_self), _stream()),
                               'piml_pinnsf_epilogue_agentnorm_bwd')
                else:
                    _lib.check(_lib.lib().piml_pinnsf_epilogue_bwd(_ptr(g), _ptr(sf), sf.numel() // 7, ctx.tau,
                                                                   _ptr(g_self), _stream()),
                               'piml_pinnsf_epilogue_bwd')
        return g, (g if ctx.has_obs else None), g_self, None, None


class _PinnsfEpilogueKsum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred_ped, pred_obs, self_features, tau, agent_norm=False):
        sf = _gpu_f32('self_features', self_features)
        pp = _gpu_f32('pred_ped', pred_ped)
        po = _gpu_f32('pred_obs', pred_obs) if pred_obs is not None else None
        rows = sf.numel() // 7
        kp, ko = pp.shape[-2], (po.shape[-2] if po is not None else 0)
        out = torch.empty(*sf.shape[:-1], 2, device=sf.device, dtype=torch.float32)
        with torch.cuda.device(sf.device):
            if agent_norm:      # (C, N, 7) input with the reference's dim=1 norm (quirk Q2)
                _lib.check(_lib.lib().piml_pinnsf_epilogue_ksum_agentnorm_fwd(_ptr(pp), kp, _ptr(po), ko, _ptr(sf), sf.shape[0], sf.shape[1],
                                                                              float(tau), _ptr(out), _stream()),
                           'piml_pinnsf_epilogue_ksum_agentnorm_fwd')
            else:
                _lib.check(_lib.lib().piml_pinnsf_epilogue_ksum_fwd(_ptr(pp), kp, _ptr(po), ko, _ptr(sf), rows, float(tau),
                                                                    _ptr(out), _stream()), 'piml_pinnsf_epilogue_ksum_fwd')
        ctx.save_for_backward(sf)
        ctx.agent_norm = bool(agent_norm)
        ctx.meta = (float(tau), kp, ko, tuple(pp.shape), tuple(po.shape) if po is not None else None)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        (sf,) = ctx.saved_tensors
        tau, kp, ko, shp_p, shp_o = ctx.meta
        g = g.contiguous()
        opt = dict(device=sf.device, dtype=torch.float32)
        g_self = torch.empty_like(sf) if ctx.needs_input_grad[2] else None
        g_pp = torch.empty(shp_p, **opt) if ctx.needs_input_grad[0] else None
        g_po = torch.empty(shp_o, **opt) if (shp_o is not None and ctx.needs_input_grad[1]) else None
        with torch.cuda.device(sf.device):
            _lib.check(_lib.lib().piml_pinnsf_epilogue_ksum_bwd(_ptr(g), _ptr(sf), sf.numel() // 7, tau, kp, ko,
                                                                None if ctx.agent_norm else _ptr(g_self),
                                                                _ptr(g_pp), _ptr(g_po), _stream()),
                       'piml_pinnsf_epilogue_ksum_bwd')
            if ctx.agent_norm and g_self is not None:
                _lib.check(_lib.lib().piml_pinnsf_epilogue_agentnorm_bwd(_ptr(g), _ptr(sf), sf.shape[0], sf.shape[1], tau, _ptr(g_self),
                                                                         _stream()), 'piml_pinnsf_epilogue_agentnorm_bwd')
        return g_pp, g_po, g_self, None, None


def pinnsf_epilogue_ksum(pred_ped, pred_obs, self_features, tau, agent_norm=False):
    """Bottleneck variants: sum over the neighbour axis of pred_ped (..., kp, 2) [+ pred_obs (..., ko, 2) or None] + the
    desired-force term of self_features (..., 7), per-row |dest| (src/models/model.py:1116-1134) -- one launch per direction
    instead of two reductions, two broadcasts and the epilogue.  agent_norm=True ((C, N, 7) input only): the reference's
    literal dim=1 norm over the agents of each slice (quirk Q2), as pinnsf_epilogue(agent_norm=True)."""
    if agent_norm and self_features.dim() != 3:
        raise ValueError('pinnsf_epilogue_ksum: agent_norm needs (C, N, 7) input')
    if self_features.shape[-1] != 7 or tuple(pred_ped.shape[:-2]) != tuple(self_features.shape[:-1]) or pred_ped.shape[-1] != 2:
        raise ValueError('pinnsf_epilogue_ksum: pred (..., k, 2) and self_features (..., 7) expected')
    if pred_obs is not None and (tuple(pred_obs.shape[:-2]) != tuple(self_features.shape[:-1]) or pred_obs.shape[-1] != 2):
        raise ValueError('pinnsf_epilogue_ksum: pred_obs (..., k, 2) must match self_features')
    return _PinnsfEpilogueKsum.apply(pred_ped, pred_obs, self_features, tau, bool(agent_norm))


def pinnsf_epilogue(acc_ped, acc_obs, self_features, tau, agent_norm=False):
    """acc_ped + acc_obs + (v0 * dest/|dest| - v) / tau on rows of self_features (..., 7)
    (src/models/model.py:1289-1294); acc_obs may be None.  agent_norm=False: the per-row norm |dest|.
    agent_norm=True (3-D input only): the reference's literal dim=1 norm of channelled input, taken over the
    agents of each slice, per component (quirk Q2)."""
    if self_features.shape[-1] != 7 or acc_ped.shape != self_features.shape[:-1] + (2,):
        raise ValueError('pinnsf_epilogue: acc (..., 2) and self_features (..., 7) expected')
    if agent_norm and self_features.dim() != 3:
        raise ValueError('pinnsf_epilogue: agent_norm needs (C, N, 7) input')
    return _PinnsfEpilogue.apply(acc_ped, acc_obs, self_features, tau, bool(agent_norm))


class _SelfFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dest_feat, state_rows, desired_speed):
        d = _gpu_f32('dest_feat', dest_feat)
        s = _gpu_f32('state_rows', state_rows)
        w = _gpu_f32('desired_speed', desired_speed)
        rows = s.shape[0]
        out = torch.empty(rows, 7, device=s.device, dtype=torch.float32)
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().piml_self_features_fwd(_ptr(d), 2, _ptr(s), _ptr(w), rows, _ptr(out), _stream()),
                       'piml_self_features_fwd')
        ctx.speed_shape = tuple(desired_speed.shape)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None, None
        g = g.contiguous()
        rows = g.shape[0]
        need = ctx.needs_input_grad
        opt = dict(device=g.device, dtype=torch.float32)
        g_dest = torch.empty(rows, 2, **opt) if need[0] else None
        g_state = torch.empty(rows, 6, **opt) if need[1] else None
        g_speed = torch.empty(ctx.speed_shape, **opt) if need[2] else None
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().piml_self_features_bwd(_ptr(g), rows, _ptr(g_dest), _ptr(g_state), _ptr(g_speed),
                                                         _stream()), 'piml_self_features_bwd')
        return g_dest, g_state, g_speed


def self_features_packed(dest_feat, state_rows, desired_speed):
    """(n, 7) model input rows [dest - p, v, a, v0] from the focal rows of the packed (p, v, a) state
    (n, 6) -- the torch.cat at src/models/simulators.py:648-650 as one kernel each way."""
    n = state_rows.shape[0]
    if state_rows.dim() != 2 or state_rows.shape[1] != 6 or dest_feat.shape != (n, 2) or desired_speed.numel() != n:
        raise ValueError('self_features_packed: dest_feat (n,2), state_rows (n,6), desired_speed (n,1) expected')
    return _SelfFeatures.apply(dest_feat, state_rows, desired_speed)


def act_bwd_colsum(g, y=None, defer=False):
    """(g_pre, db) with g_pre = g * [y > 0] (g itself when y is None) and db = g_pre.sum(0); g, y (rows, cols).
    Shapes the kernel does not cover (cols > 1024, or > 256 when cols % 4) use torch's GPU reduction.
    defer=True returns (g_pre, db, pending): only the first stage is launched and `pending` =
    (partials, blocks, cols) must be handed to `layer_reduce` (db is complete when pending is None)."""
    g = _gpu_f32('g', g)
    rows, cols = g.shape
    if (cols % 4 == 0 and cols > 1024) or (cols % 4 and cols > 256):
        g_pre = g if y is None else torch.where(y > 0, g, torch.zeros((), device=g.device))
        return (g_pre, g_pre.sum(0), None) if defer else (g_pre, g_pre.sum(0))
    L = _lib.lib()
    nb = L.piml_colsum_blocks(rows, cols)
    db = torch.empty(cols, device=g.device, dtype=torch.float32)
    g_pre = torch.empty_like(g) if y is not None else g
    partials = torch.empty(nb * cols, device=g.device, dtype=torch.float32) if nb > 1 else None
    fn = L.piml_act_bwd_colsum_stage1 if defer else L.piml_act_bwd_colsum
    with torch.cuda.device(g.device):
        _lib.check(fn(_ptr(g), _ptr(y) if y is not None else None, rows, cols,
                      _ptr(g_pre) if y is not None else None, _ptr(partials), _ptr(db), _stream()),
                   'piml_act_bwd_colsum')
    if defer:
        return g_pre, db, ((partials, nb, cols) if nb > 1 else None)
    return g_pre, db


def layer_reduce(parts, pending, db):
    """One launch for the reductions closing a layer's backward: parts (B, ...) -> parts.sum(0) (or None) and
    the deferred second stage of act_bwd_colsum(..., defer=True) into `db`."""
    out = None
    if parts is not None:
        parts = _gpu_f32('parts', parts)
        out = torch.empty(parts.shape[1:], device=parts.device, dtype=torch.float32)
    if parts is None and pending is None:
        return None
    cp, nb, cols = pending if pending is not None else (None, 0, 0)
    dev = parts.device if parts is not None else cp.device
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().piml_layer_reduce(_ptr(parts) if parts is not None else None,
                                                parts.shape[0] if parts is not None else 0,
                                                out.numel() if out is not None else 0,
                                                _ptr(out) if out is not None else None,
                                                _ptr(cp) if cp is not None else None, nb, cols,
                                                _ptr(db) if cp is not None else None, _stream()), 'piml_layer_reduce')
    return out


def _addmm_relu(bias, x, w_t):
    """relu(x @ w_t + bias) with bias and ReLU in the GEMM's epilogue (hipBLASLt) where torch exposes it."""
    fused = getattr(torch, '_addmm_activation', None)
    if fused is not None:
        return fused(bias, x, w_t, use_gelu=False)
    return torch.relu_(torch.addmm(bias, x, w_t))


class _LinearAct(torch.autograd.Function):
    """y = act(x W^T + b), act in {identity, relu}: torch.addmm forward; backward = fused
    (relu mask + bias-gradient column sum) kernel + the two torch.mm GEMMs autograd would issue."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x2 = x.reshape(-1, x.shape[-1])
        if relu:    # bias and ReLU both ride in the GEMM's epilogue (hipBLASLt)
            y = _addmm_relu(bias, x2, weight.t())
        else:
            y = torch.addmm(bias, x2, weight.t())
        out = y.view(*x.shape[:-1], weight.shape[0])
        ctx.save_for_backward(x2, weight, out if relu else None)
        ctx.x_shape = x.shape
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        x2, weight, y = ctx.saved_tensors
        g2 = g.reshape(-1, weight.shape[0])
        g_pre, db = act_bwd_colsum(g2, y.reshape(-1, weight.shape[0]) if y is not None else None)
        gx = g_pre.mm(weight).view(ctx.x_shape) if ctx.needs_input_grad[0] else None
        gw = g_pre.t().mm(x2) if ctx.needs_input_grad[1] else None
        return gx, gw, (db if ctx.needs_input_grad[2] else None), None


def linear_act(x, weight, bias, relu):
    """nn.Linear (+ nn.ReLU) with the fused backward glue; x (..., in) float32 on the GPU."""
    if not x.is_cuda:
        raise _lib.PimlHipError('linear_act: expected a GPU tensor (piml_amd has no CPU path)')
    return _LinearAct.apply(x, weight, bias, bool(relu))


# Chunked weight gradient: dW = G^T X with K = rows (tens of thousands) and M, N <= 128 is a pure split-K
# problem; rocBLAS's split-K kernel + reduction needs 28-38 us for the 128 x 128 layers of the bench step.
# The same product as ONE strided-batched GEMM over 64 row chunks (10-15 us with the tuned selection) + a
# 3 us sum over the chunks is ~1.7x faster.  Only used when the tuned GEMM selections are in effect
# (piml_amd.tuning): the default heuristics for these batched shapes are erratic (measured 16-51 us).
WGRAD_CHUNKS = ((16384, 64), (2048, 16))     # (minimum rows, chunks): 64 chunks for the encoder layers, 16 for the decoder's


def sum_leading(parts):
    """parts (B, ...) -> parts.sum(0) in a fixed order (HIP); numel of one slice % 4 == 0."""
    parts = _gpu_f32('parts', parts)
    out = torch.empty(parts.shape[1:], device=parts.device, dtype=torch.float32)
    with torch.cuda.device(parts.device):
        _lib.check(_lib.lib().piml_sum_leading(_ptr(parts), parts.shape[0], out.numel(), _ptr(out), _stream()),
                   'piml_sum_leading')
    return out


def _weight_grad(g_pre, x, pending=None, db=None):
    """g_pre (R, out)^T @ x (R, in) -> (out, in); also completes a deferred bias-gradient sum (`pending`, `db`
    from act_bwd_colsum(..., defer=True)) -- in the same launch as the chunk reduction when the chunked
    formulation applies."""
    from . import tuning
    R, cout = g_pre.shape
    cin = x.shape[1]
    B = next((b for rmin, b in WGRAD_CHUNKS if R >= rmin), 0)
    if tuning.LOADED and B and R % B == 0 and cout * cin >= 128 and (cout * cin) % 4 == 0 \
            and g_pre.is_contiguous() and x.is_contiguous():
        parts = torch.bmm(g_pre.view(B, R // B, cout).transpose(1, 2), x.view(B, R // B, cin))
        return layer_reduce(parts, pending, db)
    if pending is not None:
        layer_reduce(None, pending, db)
    return g_pre.t().mm(x)


def _chain_backward(relus, acts, out, weights, g_last, need_x, need_wb, first=None):
    """Backward of Linear(+ReLU) layers from the last to the first.  acts[i]: input of layer i (rows, in); out: the
    last layer's output (only read when it has a ReLU); g_last (rows, out_n).  `first` = (g_pre, db, pending) when
    the last layer's mask / bias-gradient stage was already produced elsewhere (ops._EncoderPool).
    Returns (gx (rows, in_0) | None, [gw0, gb0, gw1, gb1, ...])."""
    n = len(relus)
    g_cur = g_last
    grads = [None] * (2 * n)
    for i in reversed(range(n)):
        w = weights[i]
        if first is not None and i == n - 1:
            g_pre, db, pending = first
        else:
            y = None
            if relus[i]:
                y = (out if i == n - 1 else acts[i + 1]).reshape(-1, w.shape[0])
            g_pre, db, pending = act_bwd_colsum(g_cur, y, defer=True)
        if need_wb[2 * i + 1]:
            grads[2 * i + 1] = db
        if need_wb[2 * i]:
            grads[2 * i] = _weight_grad(g_pre, acts[i], pending, db)
        elif pending is not None:
            layer_reduce(None, pending, db)
        if i > 0 or need_x:
            g_cur = g_pre.mm(w)
    return (g_cur if need_x else None), grads


class _MLPChain(torch.autograd.Function):
    """A whole MLP (src/models/model.py:40-65): Linear(+ReLU) layers back to back as ONE autograd node.
    Forward: one hipBLASLt GEMM per layer with bias (+ReLU) in its epilogue.  Backward, per layer from the
    last: fused ReLU-mask + bias-gradient kernel, then the input-gradient and weight-gradient GEMMs.
    (Issuing the weight-gradient GEMMs on a side stream was measured and rejected: inside a captured
    HIP graph every cross-stream edge costs more than the overlap gains -- 0.51 -> 0.69 ms/step -- and a
    fork from an already forked stream crashed hipStreamEndCapture on ROCm 7.0.)"""

    @staticmethod
    def forward(ctx, x, relus, defer_last_bias, *wb):
        x2 = x.reshape(-1, x.shape[-1])
        acts = [x2]
        n = len(relus)
        for i, relu in enumerate(relus):
            w, b = wb[2 * i], wb[2 * i + 1]
            if defer_last_bias and i == n - 1:       # plain GEMM; the caller adds the bias (scale_ksum)
                acts.append(acts[-1].mm(w.t()))
            else:
                acts.append(_addmm_relu(b, acts[-1], w.t()) if relu else torch.addmm(b, acts[-1], w.t()))
        out = acts[-1].view(*x.shape[:-1], acts[-1].shape[-1])
        ctx.save_for_backward(*acts[:-1], out, *wb[0::2])
        ctx.relus, ctx.x_shape = tuple(relus), x.shape
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        n = len(ctx.relus)
        if g is None:
            return (None,) * (3 + 2 * n)
        saved = ctx.saved_tensors
        acts, out, weights = saved[:n], saved[n], saved[n + 1:]
        need = ctx.needs_input_grad
        gx, grads = _chain_backward(ctx.relus, acts, out, weights, g.reshape(-1, weights[-1].shape[0]), need[0],
                                    need[3:])
        return (gx.view(ctx.x_shape) if gx is not None else None, None, None, *grads)


def mlp_chain(x, relus, *weights_and_biases, defer_last_bias=False):
    """x (..., in) through Linear(+ReLU) layers; `relus[i]` says whether layer i is followed by a ReLU;
    weights_and_biases = (w0, b0, w1, b1, ...).  defer_last_bias: the last layer (which must have no ReLU) is run
    as a plain GEMM and its bias is NOT added -- the caller adds it (scale_ksum(..., bias=...)); the gradient of
    that bias is still returned here."""
    if len(weights_and_biases) != 2 * len(relus):
        raise ValueError('mlp_chain: one (weight, bias) pair per layer expected')
    if not relus:
        return x
    if not x.is_cuda:
        raise _lib.PimlHipError('mlp_chain: expected a GPU tensor (piml_amd has no CPU path)')
    if defer_last_bias and relus[-1]:
        raise ValueError('mlp_chain: defer_last_bias needs a last layer without ReLU')
    if not torch.is_grad_enabled():       # inference: no autograd node, just the GEMMs
        h = x.reshape(-1, x.shape[-1])
        for i, relu in enumerate(relus):
            w, b = weights_and_biases[2 * i], weights_and_biases[2 * i + 1]
            if defer_last_bias and i == len(relus) - 1:
                h = h.mm(w.t())
            else:
                h = _addmm_relu(b, h, w.t()) if relu else torch.addmm(b, h, w.t())
        return h.view(*x.shape[:-1], h.shape[-1])
    return _MLPChain.apply(x, tuple(bool(r) for r in relus), bool(defer_last_bias), *weights_and_biases)


class _ScaleKSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, scale, bias, keep_bits):
        e = _gpu_f32('e', e)
        k, cols = e.shape[-2], e.shape[-1]
        agents = e.numel() // (k * cols)
        msgs = torch.empty_like(e)
        pooled = torch.empty(*e.shape[:-2], cols, device=e.device, dtype=torch.float32)
        b = _gpu_f32('bias', bias.detach()) if bias is not None else None
        keep_bits = _check_keep_bits(keep_bits, agents * k, cols, e.device)
        with torch.cuda.device(e.device):
            _lib.check(_lib.lib().piml_scale_ksum_fwd(_ptr(e), _ptr(b), agents, k, cols, float(scale), _ptr(keep_bits),
                                                      _ptr(msgs), _ptr(pooled), _stream()), 'piml_scale_ksum_fwd')
        ctx.geom = (agents, k, cols, float(scale), tuple(e.shape))
        ctx.keep_bits = keep_bits
        ctx.set_materialize_grads(False)      # an unused output arrives as None, not as a zero tensor
        return msgs, pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_msgs, g_pooled):
        agents, k, cols, scale, shape = ctx.geom
        ref = g_pooled if g_pooled is not None else g_msgs
        if ref is None:
            return None, None, None, None
        g_e = torch.empty(shape, device=ref.device, dtype=torch.float32)
        gm = g_msgs.contiguous() if g_msgs is not None else None
        gp = g_pooled.contiguous() if g_pooled is not None else None
        with torch.cuda.device(ref.device):
            _lib.check(_lib.lib().piml_scale_ksum_bwd(_ptr(gp), _ptr(gm), agents, k, cols, scale, _ptr(ctx.keep_bits),
                                                      _ptr(g_e), None, _stream()), 'piml_scale_ksum_bwd')
        return g_e, None, None, None


class _EncoderPool(torch.autograd.Function):
    """encoder MLP (last layer without activation) -> scale -> neighbour-axis sum as ONE autograd node
    (src/models/model.py:1271-1283 with quirk Q3): forward = _MLPChain with the last bias deferred + the scale_ksum
    kernel; backward = the ksum backward kernel, which also leaves the per-block column sums of d/d(e) -- the first
    stage of the last layer's bias gradient, so that layer needs no pass of its own over d/d(e) -- then the chain."""

    @staticmethod
    def forward(ctx, x, relus, scale, keep_bits, *wb):
        x2 = x.reshape(-1, x.shape[-1])
        acts = [x2]
        n = len(relus)
        for i, relu in enumerate(relus):
            w, b = wb[2 * i], wb[2 * i + 1]
            if i == n - 1:
                acts.append(acts[-1].mm(w.t()))
            else:
                acts.append(_addmm_relu(b, acts[-1], w.t()) if relu else torch.addmm(b, acts[-1], w.t()))
        k, cols = x.shape[-2], acts[-1].shape[-1]
        e = acts[-1]
        agents = e.shape[0] // k
        msgs = torch.empty(*x.shape[:-1], cols, device=x.device, dtype=torch.float32)
        pooled = torch.empty(*x.shape[:-2], cols, device=x.device, dtype=torch.float32)
        keep_bits = _check_keep_bits(keep_bits, agents * k, cols, x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().piml_scale_ksum_fwd(_ptr(e), _ptr(wb[-1]), agents, k, cols, float(scale), _ptr(keep_bits),
                                                      _ptr(msgs), _ptr(pooled), _stream()), 'piml_scale_ksum_fwd')
        ctx.save_for_backward(*acts[:-1], *wb[0::2])
        ctx.keep_bits = keep_bits
        ctx.relus, ctx.x_shape, ctx.geom = tuple(relus), x.shape, (agents, k, cols, float(scale))
        ctx.set_materialize_grads(False)
        return msgs, pooled

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_msgs, g_pooled):
        n = len(ctx.relus)
        ref = g_pooled if g_pooled is not None else g_msgs
        if ref is None:
            return (None,) * (4 + 2 * n)
        saved = ctx.saved_tensors
        acts, weights = saved[:n], saved[n:]
        agents, k, cols, scale = ctx.geom
        L = _lib.lib()
        g_e = torch.empty(agents * k, cols, device=ref.device, dtype=torch.float32)
        nb = L.piml_ksum_blocks(agents, cols)
        partials = torch.empty(nb * cols, device=ref.device, dtype=torch.float32)
        gm = g_msgs.contiguous() if g_msgs is not None else None
        gp = g_pooled.contiguous() if g_pooled is not None else None
        with torch.cuda.device(ref.device):
            _lib.check(L.piml_scale_ksum_bwd(_ptr(gp), _ptr(gm), agents, k, cols, scale, _ptr(ctx.keep_bits), _ptr(g_e),
                                             _ptr(partials), _stream()), 'piml_scale_ksum_bwd')
        if nb > 1:
            db, pending = torch.empty(cols, device=ref.device, dtype=torch.float32), (partials, nb, cols)
        else:
            db, pending = partials, None
        need = ctx.needs_input_grad
        gx, grads = _chain_backward(ctx.relus, acts, None, weights, g_e, need[0], need[4:], first=(g_e, db, pending))
        return (gx.view(ctx.x_shape) if gx is not None else None, None, None, None, *grads)


def encoder_pool(x, relus, scale, *weights_and_biases, keep_bits=None):
    """(scale * encoder(x), its sum over the neighbour axis) for x (..., k, in): mlp_chain (last layer without ReLU)
    + scale_ksum as one autograd node.  Needs cols % 4 == 0 and 256 % (cols / 4) == 0 for the last layer's width.
    keep_bits: the processor's train-mode dropout mask (dropout_keep_bits layout), see scale_ksum."""
    if not x.is_cuda:
        raise _lib.PimlHipError('encoder_pool: expected a GPU tensor (piml_amd has no CPU path)')
    cols = weights_and_biases[-2].shape[0]
    if len(weights_and_biases) != 2 * len(relus) or not relus or relus[-1] or x.dim() < 3 or cols % 4 \
            or 256 % (cols // 4):
        raise ValueError('encoder_pool: (..., k, in) input, one (weight, bias) pair per layer, no ReLU after the '
                         'last layer, and a last width w with w % 4 == 0 and 256 % (w / 4) == 0 expected')
    if not torch.is_grad_enabled():
        return scale_ksum(mlp_chain(x, relus, *weights_and_biases, defer_last_bias=True), scale,
                          bias=weights_and_biases[-1], keep_bits=keep_bits)
    return _EncoderPool.apply(_gpu_f32('x', x), tuple(bool(r) for r in relus), float(scale), keep_bits,
                              *weights_and_biases)


def scale_ksum(e, scale=2.0, bias=None, keep_bits=None):
    """(keep * scale * (e + bias), its sum over axis -2) for e (..., k, cols), cols % 4 == 0: the PINNSF processor
    (quirk Q3: Dropout(2 x)) + neighbour-axis pooling (src/models/model.py:1279-1283) in one pass.  `bias` (cols,
    optional) is the deferred bias of the Linear that produced `e` (mlp_chain(..., defer_last_bias=True)); it is a
    constant here -- its gradient is the column sum of d/d(e), which that layer's backward returns.  keep_bits
    (rows, ceil(cols / 32)) int32, optional: the train-mode dropout mask (dropout_keep_bits); the caller folds
    1 / (1 - p) into `scale`."""
    if e.dim() < 2 or e.shape[-1] % 4:
        raise ValueError('scale_ksum: e (..., k, cols) with cols % 4 == 0 expected')
    if bias is not None and tuple(bias.shape) != (e.shape[-1],):
        raise ValueError('scale_ksum: bias (cols,) expected')
    return _ScaleKSum.apply(e, scale, bias, keep_bits)


# ------------------------------------------------------------------------------------------------
# Train-mode dropout of the PINNSF processor (reference: ResDNN.forward = Dropout_p(2 x), src/models/model.py:82-119
# with quirk Q3; model.train() at src/models/simulators.py:311; --dropout 0.5 at src/main.py:45).  The fused kernels
# take the mask as BITS, (rows, ceil(cols / 32)) int32, bit c & 31 of word c >> 5 = feature c of the row is kept; the
# caller folds 1 / (1 - p) into the processor's scale.  piml_dropout_keep_bits draws it with Philox4x32-10 from a
# (seed, call counter) pair that lives on the device and is advanced by the launch itself, so a captured training step
# draws a fresh mask on every replay.
# ------------------------------------------------------------------------------------------------
_DROPOUT_STATE = {}        # device index -> [state tensor (4 x int64: seed, offset, ticket, 0), torch's CUDA seed seen last]
_DROPOUT_RANK_MIX = {}     # device index -> the rank term dropout_seed folded into the seed (re-applied on automatic re-seeds)


def dropout_state(device, seed=None):
    """The device's dropout state [seed, call counter, ticket, 0] (int64 tensor).  It is seeded from torch's CUDA seed
    and re-seeded (counter back to 0) whenever that seed changes (torch.manual_seed), or explicitly with `seed`.
    Inside a stream capture nothing is written from the host."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):          # the seed of THIS device's generator (not of whichever device is current)
        torch_seed = int(torch.cuda.initial_seed())
    ent = _DROPOUT_STATE.get(idx)
    capturing = torch.cuda.is_current_stream_capturing()
    if ent is None and capturing:
        raise _lib.PimlHipError('dropout_state: first use inside a stream capture (call ops.dropout_state(device) before)')
    if ent is None or ((seed is not None or ent[1] != torch_seed) and not capturing):
        # (an automatic re-seed from torch's seed keeps the rank term of an earlier dropout_seed: the ranks of a sharded run
        # must not fall back to identical masks when somebody calls torch.manual_seed again)
        want = int(seed) if seed is not None else (torch_seed + _DROPOUT_RANK_MIX.get(idx, 0)) & ((1 << 64) - 1)
        signed = want - (1 << 64) if want >= (1 << 63) else want
        st = torch.tensor([signed, 0, 0, 0], dtype=torch.int64, device=device)
        if ent is None:
            _DROPOUT_STATE[idx] = [st, torch_seed]
        else:                       # keep the tensor a captured graph may already hold
            ent[0].copy_(st)
            ent[1] = torch_seed
    return _DROPOUT_STATE[idx][0]


def dropout_seed(seed, device=None, rank=None):
    """Reset the device's dropout draws: (seed, call counter 0, ticket 0) -- what torch.manual_seed does to torch's own
    generator, explicitly.  `torch.manual_seed(s)` with the SAME s as before does not rewind the counter by itself (the state
    only notices a seed that changed): loops that re-seed per run (BaseSimulator, the tools) call this.  rank (default: the
    torch.distributed rank when a process group is up) is folded into the seed, so that the ranks of a sharded run do not
    draw identical masks for equal local row indices."""
    device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if rank is None:
        import torch.distributed as dist
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    mix = (0x9E3779B97F4A7C15 * int(rank)) & ((1 << 64) - 1)
    _DROPOUT_RANK_MIX[device.index if device.index is not None else torch.cuda.current_device()] = mix
    mixed = (int(seed) + mix) & ((1 << 64) - 1)
    return dropout_state(device, seed=mixed)


def dropout_keep_bits(rows, cols, p, device, stream_id=0):
    """Fresh keep-mask bits (rows, ceil(cols / 32)) int32 for dropout probability p (one launch on the current stream;
    advances the device's draw counter).  stream_id tells masks of the same draw apart (include/piml_hip.h)."""
    if not 0.0 <= p <= 1.0:
        raise ValueError('dropout probability has to be between 0 and 1')
    device = torch.device(device)
    if device.type != 'cuda':
        raise _lib.PimlHipError('dropout_keep_bits: expected a GPU device (piml_amd has no CPU path)')
    st = dropout_state(device)
    bits = torch.empty(rows, (cols + 31) // 32, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        _lib.check(_lib.lib().piml_dropout_keep_bits(st.data_ptr(), rows, cols, float(p), int(stream_id), _ptr(bits),
                                                     _stream()), 'piml_dropout_keep_bits')
    return bits


def pack_keep_bits(keep):
    """bool / 0-1 tensor (..., cols) -> int32 bits (rows, ceil(cols / 32)) in the piml_dropout_keep_bits layout
    (torch ops; for masks injected by tests or by a host that draws its own)."""
    cols = keep.shape[-1]
    k2 = keep.reshape(-1, cols).to(torch.int64)
    words = (cols + 31) // 32
    pad = words * 32 - cols
    if pad:
        k2 = torch.cat((k2, k2.new_zeros(k2.shape[0], pad)), 1)
    w = (k2.view(-1, words, 32) << torch.arange(32, device=keep.device, dtype=torch.int64)).sum(-1)
    w = torch.where(w >= (1 << 31), w - (1 << 32), w)
    return w.to(torch.int32)


def unpack_keep_bits(bits, cols):
    """int32 bits (rows, words) -> bool (rows, cols)."""
    b = bits.to(torch.int64) & 0xffffffff
    out = (b.unsqueeze(-1) >> torch.arange(32, device=bits.device, dtype=torch.int64)) & 1
    return out.reshape(bits.shape[0], -1)[:, :cols].bool()


def _resolve_keep(keep, rows, cols, device):
    """A branch's `keep_bits` entry -> (bits tensor | None, draw_p | None): None = no dropout; a tensor = given bits;
    ('draw', p) = the forward launch draws the mask with probability p into a fresh buffer."""
    if isinstance(keep, tuple) and len(keep) == 2 and keep[0] == 'draw':
        p = float(keep[1])
        if not 0.0 <= p <= 1.0:
            raise ValueError('dropout probability has to be between 0 and 1')
        return torch.empty(rows, (cols + 31) // 32, dtype=torch.int32, device=device), p
    return _check_keep_bits(keep, rows, cols, device), None


def _check_keep_bits(keep_bits, rows, cols, device):
    if keep_bits is None:
        return None
    if keep_bits.dtype != torch.int32 or tuple(keep_bits.shape) != (rows, (cols + 31) // 32) or keep_bits.device != device \
            or not keep_bits.is_contiguous():
        raise ValueError(f'keep_bits: contiguous int32 ({rows}, {(cols + 31) // 32}) on {device} expected, got '
                         f'{keep_bits.dtype} {tuple(keep_bits.shape)} on {keep_bits.device}')
    return keep_bits


class _TrainRolloutStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, v, a, a_pred, dest, dest_idx, waypoints, dest_num, new_flag, series, t_next, dt, nan_flag,
                zero_nan):
        p, v, a, a_pred, dest = [_gpu_f32(n, x) for n, x in
                                 (('position', p), ('velocity', v), ('acceleration', a), ('a_pred', a_pred),
                                  ('destination', dest))]
        C, N = p.shape[0], p.shape[1]
        D = waypoints.shape[-3]
        per_slice = int(waypoints.dim() == 4)
        T = series[0].shape[1] if series is not None else max(int(t_next), 1)
        opt = dict(device=p.device, dtype=torch.float32)
        outs = [torch.empty(C, N, 2, **opt) for _ in range(4)]
        idx_out = torch.empty(C, N, device=p.device, dtype=torch.int64)
        sp = [None] * 5 if series is None else [_ptr(x) for x in series]
        zero_mask = torch.empty(C, N, device=p.device, dtype=torch.uint8) if zero_nan else None
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_train_step_fwd(
                _ptr(p), _ptr(v), _ptr(a), _ptr(a_pred), _ptr(dest), _ptr(dest_idx), _ptr(waypoints), D, per_slice,
                _ptr(dest_num), _ptr(new_flag) if new_flag is not None else None, *sp, C, T, N, int(t_next),
                float(dt), *[_ptr(o) for o in outs], _ptr(idx_out), _ptr(nan_flag) if nan_flag is not None else None,
                _ptr(zero_mask) if zero_mask is not None else None, _stream()), 'piml_train_step_fwd')
        ctx.new_flag, ctx.zero_mask, ctx.geom = new_flag, zero_mask, (C, T, N, int(t_next), float(dt))
        ctx.mark_non_differentiable(outs[3], idx_out)
        ctx.set_materialize_grads(False)
        return (*outs, idx_out)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gp_o, gv_o, ga_o, _gd, _gi):
        C, T, N, t_next, dt = ctx.geom
        ref = next((g for g in (gp_o, gv_o, ga_o) if g is not None), None)
        if ref is None:
            return (None,) * 14
        # an output nobody differentiates contributes nothing: hand autograd None (not zeros) so that it does
        # not walk into the model call that produced a_pred just to propagate a zero gradient
        live = (gp_o is not None, gp_o is not None or gv_o is not None, gv_o is not None, ga_o is not None)
        need = ctx.needs_input_grad
        gs = [torch.empty(C, N, 2, device=ref.device, dtype=torch.float32) if (need[k] and live[k]) else None
              for k in range(4)]
        cont = [None if g is None else g.contiguous() for g in (gp_o, gv_o, ga_o)]
        with torch.cuda.device(ref.device):
            _lib.check(_lib.lib().piml_train_step_bwd(
                *[_ptr(g) for g in cont], _ptr(ctx.new_flag) if ctx.new_flag is not None else None,
                _ptr(ctx.zero_mask) if ctx.zero_mask is not None else None, C, T, N, t_next,
                dt, *[_ptr(g) for g in gs], _stream()), 'piml_train_step_bwd')
        return (*gs,) + (None,) * 10


def train_rollout_step(position, velocity, acceleration, a_pred, destination, dest_idx, waypoints, dest_num,
                       dt, new_flag=None, series=None, t_next=0, nan_flag=None, zero_nan=False):
    """One frame of the fine-tuning rollout between the model call and the feature recomputation
    (src/models/simulators.py:741-769) as one differentiable node: lagged Euler, waypoint switch, injection of
    the agents entering at frame `t_next` from `series` = (position, velocity, acceleration, destination,
    dest_idx) each (C, T, N, .).  State tensors are (C, N, 2); dest_idx (C, N) int64; waypoints (D, N, 2) or
    (C, D, N, 2); dest_num (N) int64; new_flag (C, T, N) uint8 / bool or None; nan_flag: optional int32
    scalar tensor that is OR-ed with 1 when a_pred contains a NaN.  zero_nan: also apply, to the new
    velocity / acceleration, the NaN -> 0 that the next get_relative_features call performs in place
    (src/data/data.py:483-484), with the matching gradient cut.
    Returns (position', velocity', acceleration', destination', dest_idx')."""
    if position.dim() != 3 or position.shape[-1] != 2:
        raise ValueError('train_rollout_step: (C, N, 2) state expected')
    if not position.is_cuda:
        raise _lib.PimlHipError('train_rollout_step: expected GPU tensors (piml_amd has no CPU path)')
    C, N = position.shape[0], position.shape[1]
    if dest_idx.dtype != torch.int64 or dest_num.dtype != torch.int64:
        raise TypeError('train_rollout_step: dest_idx / dest_num must be int64')
    if tuple(dest_idx.shape) != (C, N) or dest_num.numel() != N:
        raise ValueError('train_rollout_step: dest_idx (C, N) and dest_num (N) expected')
    if waypoints.dtype != torch.float32 or waypoints.shape[-2:] != (N, 2) or \
            (waypoints.dim() == 4 and waypoints.shape[0] != C) or waypoints.dim() not in (3, 4):
        raise ValueError('train_rollout_step: waypoints (D, N, 2) or (C, D, N, 2) float32 expected')
    if new_flag is not None:
        if series is None or len(series) != 5:
            raise ValueError('train_rollout_step: new_flag needs the five ground-truth series')
        T = series[0].shape[1]
        if new_flag.dtype == torch.bool:
            new_flag = new_flag.view(torch.uint8)
        if new_flag.dtype != torch.uint8 or tuple(new_flag.shape) != (C, T, N):
            raise ValueError('train_rollout_step: new_flag must be (C, T, N) bool / uint8')
        new_flag = new_flag.contiguous()
        for x, w, dt_ in zip(series, (2, 2, 2, 2, None), (torch.float32,) * 4 + (torch.int64,)):
            want = (C, T, N) + ((w,) if w else ())
            if tuple(x.shape) != want or x.dtype != dt_ or not x.is_contiguous():
                raise ValueError(f'train_rollout_step: series tensor must be contiguous {want} {dt_}')
    else:
        series = None
    return _TrainRolloutStep.apply(position, velocity, acceleration, a_pred, destination, dest_idx.contiguous(),
                                   waypoints.contiguous(), dest_num.contiguous(), new_flag, series, int(t_next),
                                   float(dt), nan_flag, bool(zero_nan))


class _RolloutFrame(torch.autograd.Function):
    """One frame of the fine-tuning rollout between two model calls as ONE autograd node: _TrainRolloutStep (integrator,
    waypoint switch, injection, NaN -> 0; src/models/simulators.py:741-769) followed by _RelativeFeaturesSelf on its outputs
    (:772-779).  The new state feeds both the next frame's step and this frame's features; as two nodes autograd sums the two
    gradients per tensor (three strided additions per frame) -- here the features' backward accumulates into a (C, N, 6)
    buffer its forward launch cleared and the step's backward adds that buffer to the gradients it is handed
    (piml_train_step_bwd6): two launches per frame and direction, nothing else."""

    @staticmethod
    def forward(ctx, p, v, a, a_pred, dest, dest_idx, waypoints, dest_num, new_flag, series, t_next, dt, nan_flag,
                obstacles, speed, kp, ko, cos_p, cos_o, dthr_p, dthr_o, alias_p=False, stack=None, tail_ped=None, tail_obs=None,
                tail_sf=None, tail_tau=None):
        L = _lib.lib()
        p_arg = p
        p, v, a, dest = [_gpu_f32(n, x) for n, x in (('position', p), ('velocity', v), ('acceleration', a), ('destination', dest))]
        # tail_*: the model's tail rides in the step's launch (piml_train_step_tail_fwd): a_pred is None, the prediction is made there
        tail = tail_ped is not None
        if tail:
            t_ped = _gpu_f32('tail acc_ped', tail_ped)
            t_obs = _gpu_f32('tail acc_obs', tail_obs) if tail_obs is not None else None
            t_sf = _gpu_f32('tail self_features', tail_sf)
            Cn = p.shape[0] * p.shape[1]
            tkp, tko = t_ped.numel() // (Cn * 2), (t_obs.numel() // (Cn * 2) if t_obs is not None else 1)
            if tuple(t_sf.shape) != (p.shape[0], p.shape[1], 7) or t_ped.numel() != Cn * tkp * 2 or \
                    (t_obs is not None and t_obs.numel() != Cn * tko * 2):
                raise ValueError('rollout_frame: tail = (acc_ped (C, N[, k], 2), acc_obs | None, self_features (C, N, 7), tau)')
        else:
            a_pred = _gpu_f32('a_pred', a_pred)
        o = _gpu_f32('obstacles', obstacles).reshape(-1, 2)
        v0 = _gpu_f32('desired_speed', speed)
        C, N = p.shape[0], p.shape[1]
        D = waypoints.shape[-3]
        per_slice = int(waypoints.dim() == 4)
        T = series[0].shape[1] if series is not None else max(int(t_next), 1)
        dev = p.device
        opt = dict(device=dev, dtype=torch.float32)
        outs = [torch.empty(C, N, 2, **opt) for _ in range(4)]
        idx_out = torch.empty(C, N, device=dev, dtype=torch.int64)
        sp = [None] * 5 if series is None else [_ptr(x) for x in series]
        zero_mask = torch.empty(C, N, device=dev, dtype=torch.uint8)
        M = o.shape[0]
        kpe, koe = min(kp, N), min(ko, M)
        pf, of, sf = torch.empty(C, N, kpe, 6, **opt), torch.empty(C, N, koe, 6, **opt), torch.empty(C, N, 7, **opt)
        pi = torch.empty(C, N, kpe, device=dev, dtype=torch.int32)
        oi = torch.empty(C, N, koe, device=dev, dtype=torch.int32)
        need = any(ctx.needs_input_grad[:4]) or (tail and any(ctx.needs_input_grad[23:26]))
        g6 = torch.empty(C, N, 6, **opt) if need else None
        with torch.cuda.device(dev):
            copy_ptr, copy_stride = None, 0
            if stack is not None:         # (buffer (C, T', N, 2) float32 contiguous, frame index): the input position into its frame
                sbuf, st_ = stack
                copy_ptr, copy_stride = sbuf.data_ptr() + int(st_) * N * 2 * 4, sbuf.shape[1] * N * 2
            if tail:
                _lib.check(L.piml_train_step_tail_fwd(
                    _ptr(p), _ptr(v), _ptr(a), _ptr(t_ped), tkp, _ptr(t_obs), tko, _ptr(t_sf), float(tail_tau), _ptr(dest), _ptr(dest_idx),
                    _ptr(waypoints), D, per_slice, _ptr(dest_num), _ptr(new_flag) if new_flag is not None else None, *sp, C, T, N,
                    int(t_next), float(dt), *[_ptr(x) for x in outs], _ptr(idx_out), _ptr(nan_flag) if nan_flag is not None else None,
                    _ptr(zero_mask), copy_ptr, int(copy_stride), _stream()), 'piml_train_step_tail_fwd')
            else:
                _lib.check(L.piml_train_step_fwd_copy(
                    _ptr(p), _ptr(v), _ptr(a), _ptr(a_pred), _ptr(dest), _ptr(dest_idx), _ptr(waypoints), D, per_slice,
                    _ptr(dest_num), _ptr(new_flag) if new_flag is not None else None, *sp, C, T, N, int(t_next),
                    float(dt), *[_ptr(x) for x in outs], _ptr(idx_out), _ptr(nan_flag) if nan_flag is not None else None,
                    _ptr(zero_mask), copy_ptr, int(copy_stride), _stream()), 'piml_train_step_fwd_copy')
            _lib.check(L.piml_relfeat_fwd_self(
                _ptr(outs[0]), None, _ptr(outs[1]), _ptr(outs[2]), 2, _ptr(outs[3]), _ptr(o), _ptr(v0), C, N, M, 0, N, kp, ko,
                cos_p, cos_o, dthr_p, dthr_o, _ptr(pf), _ptr(of), _ptr(sf), _ptr(pi), _ptr(oi), _ptr(g6), _stream()),
                'piml_relfeat_fwd_self')
        if tail:
            ctx.save_for_backward(pi, oi, outs[0], outs[3], t_sf)
            ctx.tail = (tkp, tko, float(tail_tau), tuple(tail_ped.shape), None if tail_obs is None else tuple(tail_obs.shape))
        else:
            ctx.save_for_backward(pi, oi, outs[0], outs[3])
            ctx.tail = None
        ctx.new_flag, ctx.zero_mask, ctx.geom, ctx.g6 = new_flag, zero_mask, (C, T, N, int(t_next), float(dt), kpe, koe), g6
        ctx.mark_non_differentiable(outs[3], idx_out)
        ctx.set_materialize_grads(False)
        if alias_p:
            # the frame's INPUT position once more, as an output: the rollout loss reads the position of every frame and so does the
            # next frame's step -- read through this alias, the tensor has ONE consumer and the loss's gradient arrives HERE, where the
            # step's backward launch adds it (piml_train_step_bwd7), instead of in a strided addition of the autograd engine per frame
            return (*outs, idx_out, pf, of, sf, p_arg)
        return (*outs, idx_out, pf, of, sf)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gp_o, gv_o, ga_o, _gd, _gi, g_pf, g_of, g_sf, g_alias=None):
        C, T, N, t_next, dt, kpe, koe = ctx.geom
        feats = any(g is not None for g in (g_pf, g_of, g_sf))
        if not feats and all(g is None for g in (gp_o, gv_o, ga_o)):
            return (g_alias if ctx.needs_input_grad[0] else None,) + (None,) * 26
        pi, oi, p_out, dest_out = ctx.saved_tensors[:4]
        dev = p_out.device
        opt = dict(device=dev, dtype=torch.float32)
        L = _lib.lib()
        g6 = None
        with torch.cuda.device(dev):
            if feats:
                g6, ctx.g6 = ctx.g6, None                      # the forward's cleared buffer, once; a second pass clears a fresh one
                if g6 is None:
                    g6 = torch.zeros(C, N, 6, **opt)

                def dense(g, shape):
                    return _zeros_ro(shape, dev) if g is None else _gpu_f32('grad', g)
                g_pf, g_of, g_sf = dense(g_pf, (C, N, kpe, 6)), dense(g_of, (C, N, koe, 6)), dense(g_sf, (C, N, 7))
                g_dest = torch.empty(C, N, 2, **opt)           # (the step's destination output carries no gradient)
                _lib.check(L.piml_relfeat_bwd_self(
                    _ptr(g_pf), _ptr(g_of), _ptr(g_sf), _ptr(pi), _ptr(oi), _ptr(p_out), 2, _ptr(dest_out), C, N, 0, N, kpe, koe,
                    _ptr(g6), _ptr(g_dest), None, _stream()), 'piml_relfeat_bwd_self')
            need = ctx.needs_input_grad
            # an input whose gradient is identically zero gets None (not zeros): autograd then does not walk into the model
            # call that produced a_pred just to propagate nothing (g_p = g_p', g_v = g_v' + dt g_p', g_a = dt g_v', g_a_pred = g_a')
            hp, hv, ha = (gp_o is not None or feats), (gv_o is not None or feats), (ga_o is not None or feats)
            # the alias gradient as it stands when its (N, 2) slices are contiguous (a time slice of the loss's (C, T, N, 2) gradient)
            def sliced(g):          # a (C, N, 2) gradient as it stands when its (N, 2) slices are contiguous -> (tensor, slice stride)
                g_ = g if (g.is_cuda and g.dtype == torch.float32 and g.dim() == 3) else _gpu_f32('grad', g)
                if not (g_.stride(2) == 1 and g_.stride(1) == 2 and g_.stride(0) % 2 == 0 and g_.data_ptr() % 8 == 0):
                    g_ = g_.contiguous()
                return g_, (0 if g_.is_contiguous() else g_.stride(0))
            gin, gin_stride = None, 0
            if g_alias is not None and need[0]:
                gin, gin_stride = sliced(g_alias)
                if gin_stride == 0:
                    gin_stride = N * 2
            live = (hp or gin is not None, hp or hv, hv, ha)
            gs = [torch.empty(C, N, 2, **opt) if (need[k] and live[k]) else None for k in range(4)]
            gpo, gpo_stride = (None, 0) if gp_o is None else sliced(gp_o)
            cont = [None if g is None else _gpu_f32('grad', g) for g in (gv_o, ga_o)]
            g_tail = (None, None, None)
            if ctx.tail is not None and ha and any(need[23:26]):
                # the step's backward and the tail's in one launch: d/d(prediction) -> the tail's summands (broadcast over k) and g_self
                tkp, tko, tau, shp_p, shp_o = ctx.tail
                t_sf = ctx.saved_tensors[4]
                g_pred = torch.empty(C, N, 2, **opt)
                g_tp = (torch.empty(shp_p, **opt) if tkp > 1 else g_pred.view(shp_p)) if need[23] else None
                g_to = (torch.empty(shp_o, **opt) if tko > 1 else g_pred.view(shp_o)) if (shp_o is not None and need[24]) else None
                g_tsf = torch.empty(C, N, 7, **opt) if need[25] else None
                _lib.check(L.piml_train_step_tail_bwd(
                    _ptr(gpo), int(gpo_stride), *[_ptr(g) for g in cont], _ptr(g6), _ptr(gin), int(gin_stride),
                    _ptr(ctx.new_flag) if ctx.new_flag is not None else None, _ptr(ctx.zero_mask), C, T, N, t_next, dt,
                    *[_ptr(g) for g in gs[:3]], _ptr(g_pred), _ptr(t_sf), tau, tkp, tko,
                    _ptr(g_tp) if (g_tp is not None and tkp > 1) else None, _ptr(g_to) if (g_to is not None and tko > 1) else None,
                    _ptr(g_tsf), _stream()), 'piml_train_step_tail_bwd')
                g_tail = (g_tp, g_to, g_tsf)
            else:
                _lib.check(L.piml_train_step_bwd7(
                    _ptr(gpo), int(gpo_stride), *[_ptr(g) for g in cont], _ptr(g6), _ptr(gin), int(gin_stride),
                    _ptr(ctx.new_flag) if ctx.new_flag is not None else None,
                    _ptr(ctx.zero_mask), C, T, N, t_next, dt, *[_ptr(g) for g in gs], _stream()), 'piml_train_step_bwd7')
        return (*gs,) + (None,) * 19 + g_tail + (None,)


def rollout_frame(position, velocity, acceleration, a_pred, destination, dest_idx, waypoints, dest_num, dt, new_flag, series,
                  t_next, nan_flag, obstacles, desired_speed, topk_ped=6, sight_angle_ped=90, dist_threshold_ped=4,
                  topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4, alias_position=False, stack=None, tail=None):
    """train_rollout_step(..., zero_nan=True) + relative_features_self on its result as one autograd node (_RolloutFrame).
    tail = (acc_ped (C, N[, k], 2), acc_obs | None, self_features (C, N, 7), tau) with a_pred None: the model's tail under the
    reference's agent-axis norm (ops.pinnsf_epilogue / pinnsf_epilogue_ksum with agent_norm=True) is evaluated inside the step's
    launch, its backward inside the step's backward launch (piml_train_step_tail_fwd / bwd; bitwise the separate launches).
    Returns (position', velocity', acceleration', destination', dest_idx', ped_features, obs_features, self_features); with
    alias_position a ninth element: the INPUT position again, as an output of the node -- a caller whose loss reads the frame's
    position reads this alias, and the loss's gradient is added inside the node's backward launch (no accumulation per frame).
    stack = (buffer (C, T, N, 2) float32 contiguous, t): the step's launch also writes the input position into frame t of the buffer
    (ops.stack_of makes the filled buffer a differentiable function of the aliases: no concatenation behind the loop)."""
    if position.dim() != 3 or position.shape[-1] != 2 or not position.is_cuda:
        raise ValueError('rollout_frame: (C, N, 2) GPU state expected')
    if topk_ped > MAX_TOPK or topk_obs > MAX_TOPK:
        raise ValueError(f'topk must be <= {MAX_TOPK}')
    if tail is not None and (DETERMINISTIC_BWD or a_pred is not None):
        if a_pred is not None:
            raise ValueError('rollout_frame: a_pred or tail, not both')
        a_pred = (pinnsf_epilogue_ksum(tail[0], tail[1], tail[2], tail[3], agent_norm=True) if tail[0].dim() == 4 else
                  pinnsf_epilogue(tail[0], tail[1], tail[2], tail[3], agent_norm=True))
        tail = None
    if DETERMINISTIC_BWD:        # the atomics-free feature backward exists for the plain operator only
        st = train_rollout_step(position, velocity, acceleration, a_pred, destination, dest_idx, waypoints, dest_num, dt,
                                new_flag=new_flag, series=series, t_next=t_next, nan_flag=nan_flag, zero_nan=True)
        out = (*st, *relative_features_self(st[0], st[1], st[2], st[3], obstacles, desired_speed, topk_ped, sight_angle_ped,
                                            dist_threshold_ped, topk_obs, sight_angle_obs, dist_threshold_obs))
        return out + (position,) if alias_position else out
    C, N = position.shape[0], position.shape[1]
    if dest_idx.dtype != torch.int64 or dest_num.dtype != torch.int64 or tuple(dest_idx.shape) != (C, N) or dest_num.numel() != N:
        raise ValueError('rollout_frame: dest_idx (C, N) int64 and dest_num (N) int64 expected')
    if new_flag is not None:
        if new_flag.dtype == torch.bool:
            new_flag = new_flag.view(torch.uint8)
        T = series[0].shape[1]
        if new_flag.dtype != torch.uint8 or tuple(new_flag.shape) != (C, T, N) or not new_flag.is_contiguous():
            raise ValueError('rollout_frame: new_flag must be contiguous (C, T, N) bool / uint8')
        for x, w, dt_ in zip(series, (2, 2, 2, 2, None), (torch.float32,) * 4 + (torch.int64,)):
            if tuple(x.shape) != (C, T, N) + ((w,) if w else ()) or x.dtype != dt_ or not x.is_contiguous():
                raise ValueError('rollout_frame: series tensors must be contiguous (C, T, N[, 2]) float32 / int64')
    else:
        series = None
    return _RolloutFrame.apply(position, velocity, acceleration, a_pred, destination, dest_idx.contiguous(), waypoints.contiguous(),
                               dest_num.contiguous(), new_flag, series, int(t_next), float(dt), nan_flag, obstacles, desired_speed,
                               int(topk_ped), int(topk_obs), cos_threshold(sight_angle_ped), cos_threshold(sight_angle_obs),
                               float(dist_threshold_ped), float(dist_threshold_obs), bool(alias_position),
                               None if (stack is None or DETERMINISTIC_BWD) else stack,
                               *((None, None, None, None) if tail is None else (tail[0], tail[1], tail[2], float(tail[3]))))


class _StackOf(torch.autograd.Function):
    """buffer (C, T, N, 2) whose frames t0 .. t0 + len(frames) - 1 the frame steps have ALREADY filled with `frames` (rollout_frame's
    stack=): the buffer as a differentiable function of those frames -- forward launches nothing, backward hands every frame its
    time slice of the gradient as it stands (the frame's backward launch takes it strided)."""

    @staticmethod
    def forward(ctx, buf, t0, *frames):
        ctx.t0, ctx.n = int(t0), len(frames)
        ctx.set_materialize_grads(False)
        return buf.view_as(buf)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None) + (None,) * ctx.n
        return (None, None) + tuple(g[:, ctx.t0 + i] if ctx.needs_input_grad[2 + i] else None for i in range(ctx.n))


def stack_of(buf, t0, frames):
    return _StackOf.apply(buf, int(t0), *frames)


class _CollisionCorrection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, predictions, ped_features, velocity, thr, dt):
        P = _gpu_f32('predictions', predictions)
        F = _gpu_f32('ped_features', ped_features)
        V = _gpu_f32('velocity', velocity)
        k, stride = F.shape[-2], F.shape[-1]
        rows = P.numel() // 2
        out = torch.empty_like(P)
        with torch.cuda.device(P.device):
            _lib.check(_lib.lib().piml_collision_correction_fwd(_ptr(P), _ptr(F), _ptr(V), rows, k, stride, float(thr),
                                                                float(dt), _ptr(out), _stream()),
                       'piml_collision_correction_fwd')
        ctx.save_for_backward(P, F, V)
        ctx.cfg = (rows, k, stride, float(thr), float(dt))
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        P, F, V = ctx.saved_tensors
        rows, k, stride, thr, dt = ctx.cfg
        need = ctx.needs_input_grad
        gP = torch.empty_like(P) if need[0] else None
        gF = torch.empty_like(F) if need[1] else None
        gV = torch.empty_like(V) if need[2] else None
        with torch.cuda.device(P.device):
            _lib.check(_lib.lib().piml_collision_correction_bwd(_ptr(g.contiguous()), _ptr(P), _ptr(F), _ptr(V), rows, k,
                                                                stride, thr, dt, _ptr(gP), _ptr(gF), _ptr(gV),
                                                                _stream()), 'piml_collision_correction_bwd')
        return gP, gF, gV, None, None


def collision_post_correction(predictions, ped_features, velocity, collision_threshold=0.5, time_unit=0.08):
    """The hand-written collision handling of `--model pinnsf_pbc` (src/models/model.py:1383-1444):
    predictions (..., N, 2), ped_features (..., N, k, >= 4), velocity (..., N, 2) -> corrected predictions.
    Differentiable w.r.t. all three (flags and neighbour selections are piecewise constant)."""
    if predictions.shape[-1] != 2 or velocity.shape != predictions.shape or ped_features.dim() != predictions.dim() + 1 \
            or ped_features.shape[:-2] != predictions.shape[:-1] or ped_features.shape[-1] < 4:
        raise ValueError('collision_post_correction: predictions / velocity (..., N, 2) and ped_features '
                         '(..., N, k, >= 4) expected')
    return _CollisionCorrection.apply(predictions, ped_features, velocity, collision_threshold, time_unit)


# ------------------------------------------------------------------------------------------------
# Fused PINNSF encoder on the f32 matrix cores (piml_amd/csrc/encoder.hip): the three Linear layers of
# ped_encoder / obs_encoder, the processor's `scale * x` and the neighbour-axis sum, forward and backward,
# for one or two branches per launch.  Replaces nine library GEMMs + their glue passes per branch and step.
# ------------------------------------------------------------------------------------------------
ENCODER_HIDDEN = 128
RELU_MASK = _os.environ.get('PIML_RELU_MASK', '1') != '0'     # sign bits of h1 / h2 for the dX chain (piml_encoder_branch.relu_mask)
ENCODER_MAX_IN = 8


def _enc_branch_struct(x2, k, scale, wb, msgs, h1=None, h2=None, g_pooled=None, g_msgs=None, g2=None, g1=None,
                       g_x=None, partials=None, packed=None, grads=None, keep_bits=None, draw_p=None, relu_mask=None,
                       sum_a=None, sum_b=None):
    B = _lib.EncoderBranch()
    B.sum_a, B.sum_b = _ptr(sum_a), _ptr(sum_b)
    B.keep_bits = _ptr(keep_bits)
    if draw_p is not None:            # the forward draws the mask itself (into keep_bits) from the device's dropout state
        B.drop_state, B.drop_p = dropout_state(x2.device).data_ptr(), float(draw_p)
    B.x, B.rows, B.in_dim, B.k = x2.data_ptr(), x2.shape[0], x2.shape[1], int(k)
    B.w1, B.b1, B.w2, B.b2, B.w3, B.b3 = [t.data_ptr() for t in wb]
    B.scale = float(scale)
    B.h1, B.h2, B.msgs = _ptr(h1), _ptr(h2), _ptr(msgs)
    B.g_pooled, B.g_msgs, B.g2, B.g1, B.g_x, B.partials = [_ptr(t) for t in (g_pooled, g_msgs, g2, g1, g_x, partials)]
    B.packed, B.grads = _ptr(packed), _ptr(grads)
    # the sign bits of h1 / h2 ride behind h2's rows (_h2_buffer): 2 extra rows of 128 floats per 32-row tile
    R = x2.shape[0]
    B.relu_mask = (h2.data_ptr() + R * h2.shape[1] * 4) if (RELU_MASK and h2 is not None and h2.shape[0] >= R + 2 * ((R + 31) // 32)) else None
    if relu_mask is not None:          # PIML_POOL_TRAIN: the sign words in a buffer of their own (a branch may have no h2 rows)
        B.relu_mask = relu_mask.data_ptr()
    return B


H1_RECOMPUTE = _os.environ.get('PIML_H1_RECOMPUTE', '1') != '0'      # do not store h1 where the backward can do without (below)


def _h1_needed(rows_per_branch, alone=True):
    """False when the backward of these branches runs without h1 (include/piml_hip.h, piml_encoder_branch): sign bits for
    the dX chain + layer-split weight gradients that recompute it from x.  The forward then does not store it.
    The backward may run on any subset of the branches (fused_encoders: a branch without upstream gradient is left out; the PINNSF
    network: a loss on the messages or the collision head alone), so every branch has to be above the training bound of the
    one-wave kernels on its own (`alone` is kept for the callers' sake and no longer changes the answer)."""
    if not (H1_RECOMPUTE and RELU_MASK):
        return True
    import ctypes
    L = _lib.lib()
    # EVERY branch on its own above the training bound (round 5: also for the network, whose backward normally has a gradient for all
    # branches -- but a loss on the collision head or the messages alone arrives through ONE branch, and that launch must not fall back
    # to the few-rows kernels, which read h1; until the library used one bound for every training pass this case raised an error)
    per_branch = [(r + 31) // 32 for r in rows_per_branch]
    tiles = min(per_branch)
    if not (L.piml_encoder_products(-1) == 1 and L.piml_encoder_dw2(-1) == 1 and tiles > L.piml_encoder_split_tiles_train(-1)):
        return True
    arr = (_lib.EncoderBranch * len(rows_per_branch))()
    for b, r in enumerate(rows_per_branch):
        arr[b].rows = r
    w0 = ctypes.c_int(0)
    total = L.piml_encoder_workgroups(arr, len(rows_per_branch), ctypes.byref(w0))
    w = [total] if len(rows_per_branch) == 1 else [w0.value, total - w0.value]
    return min(w) < 2


def _h2_buffer(R, opt):
    """h2 (R, 128) with room behind it for piml_encoder_branch.relu_mask (256 dwords per 32-row tile)."""
    return torch.empty(R + 2 * ((R + 31) // 32), ENCODER_HIDDEN, **opt)


class _FusedEncoders(torch.autograd.Function):
    """inputs: nbr, scales (tuple), want_pooled (tuple), then per branch x (..., k, in) and w1, b1, w2, b2, w3, b3.
    outputs: per branch msgs (..., k, 128), pooled (..., 128) (a 0-element tensor when not wanted)."""

    @staticmethod
    def forward(ctx, nbr, scales, want_pooled, need_grad, keeps, packs, *tensors):
        L = _lib.lib()
        xs = [tensors[7 * b] for b in range(nbr)]
        wbs = [[_gpu_f32('encoder weight', t.detach()) for t in tensors[7 * b + 1:7 * b + 7]] for b in range(nbr)]
        dev = xs[0].device
        x2s, msgs, h1s, h2s, ks = [], [], [], [], []
        opt = dict(device=dev, dtype=torch.float32)
        for b in range(nbr):
            x = _gpu_f32('encoder input', xs[b])
            x2 = x.reshape(-1, x.shape[-1])
            R = x2.shape[0]
            x2s.append(x2)
            ks.append(x.shape[-2])
            msgs.append(torch.empty(R, ENCODER_HIDDEN, **opt))
            h1s.append(None)
            h2s.append(_h2_buffer(R, opt) if need_grad else None)
        if need_grad and _h1_needed([x2.shape[0] for x2 in x2s]):
            h1s = [torch.empty(x2.shape[0], ENCODER_HIDDEN, **opt) for x2 in x2s]
        keeps, draws = zip(*[_resolve_keep(keeps[b], x2s[b].shape[0], ENCODER_HIDDEN, dev) for b in range(nbr)])
        if packs is not None:               # images packed once for many forward passes (PinnsfPacks, branch order = pack order)
            if packs.sig_enc is None or packs.sig_enc[:nbr] != tuple(_branch_sig(wb) for wb in wbs):
                raise ValueError('fused_encoders: `packs` were filled from other weight tensors, or the weights were modified '
                                 'in place since (pinnsf_prepack first)')
            packed = packs.epack
        else:
            packed = torch.empty(nbr, L.piml_encoder_pack_floats(), **opt)      # weights as MFMA operand fragments
        arr = (_lib.EncoderBranch * nbr)(*[_enc_branch_struct(x2s[b], ks[b], scales[b], wbs[b], msgs[b], h1s[b], h2s[b],
                                                              packed=packed[b], keep_bits=keeps[b], draw_p=draws[b])
                                           for b in range(nbr)])
        outs = []
        with torch.cuda.device(dev):
            if packs is not None:
                _lib.check(L.piml_encoder_fwd_packed(arr, nbr, _stream()), 'piml_encoder_fwd_packed')
            else:
                _lib.check(L.piml_encoder_fwd(arr, nbr, _stream()), 'piml_encoder_fwd')
            for b in range(nbr):
                lead = tuple(xs[b].shape[:-2])
                if want_pooled[b]:
                    agents = x2s[b].shape[0] // ks[b]
                    pooled = torch.empty(agents, ENCODER_HIDDEN, **opt)
                    _lib.check(L.piml_encoder_ksum(_ptr(msgs[b]), agents, ks[b], _ptr(pooled), _stream()),
                               'piml_encoder_ksum')
                    pooled = pooled.view(*lead, ENCODER_HIDDEN)
                else:
                    pooled = torch.empty(0, **opt)
                outs += [msgs[b].view(*lead, ks[b], ENCODER_HIDDEN), pooled]
        ctx.save_for_backward(*x2s, *h1s, *h2s, *[w for wb in wbs for w in wb], packed)
        ctx.meta = (nbr, tuple(scales), tuple(want_pooled), tuple(ks), [tuple(x.shape) for x in xs], need_grad)
        ctx.sink = ParamGradSink._active if need_grad else None
        ctx.params = tensors if need_grad else None       # (the Parameter objects: the sink's keys / the deferral's .grad test)
        ctx.keeps = keeps
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gouts):
        nbr, scales, want_pooled, ks, xshapes, need_grad = ctx.meta
        nin = 6 + 7 * nbr
        if not need_grad or all(g is None for g in gouts):
            return (None,) * nin
        L = _lib.lib()
        saved = ctx.saved_tensors
        x2s, h1s, h2s = saved[:nbr], saved[nbr:2 * nbr], saved[2 * nbr:3 * nbr]
        wbs = [saved[3 * nbr + 6 * b:3 * nbr + 6 * b + 6] for b in range(nbr)]
        packed = saved[-1]
        dev = x2s[0].device
        opt = dict(device=dev, dtype=torch.float32)
        live = [b for b in range(nbr) if gouts[2 * b] is not None or (want_pooled[b] and gouts[2 * b + 1] is not None)]
        grads = [None] * nin
        if not live:
            return tuple(grads)
        part = L.piml_encoder_partial_floats()
        structs, keep = [], []
        for b in live:
            R, in_dim = x2s[b].shape
            gm = gouts[2 * b]
            gp = gouts[2 * b + 1] if want_pooled[b] else None
            gm = _gpu_f32('g_msgs', gm).reshape(R, ENCODER_HIDDEN) if gm is not None else None
            gp = _gpu_f32('g_pooled', gp).reshape(-1, ENCODER_HIDDEN) if gp is not None else None
            g2 = torch.empty(R, ENCODER_HIDDEN, **opt)
            g1 = torch.empty(R, ENCODER_HIDDEN, **opt)
            gx = torch.empty(R, in_dim, **opt) if ctx.needs_input_grad[6 + 7 * b] else None
            keep.append((gm, gp, g2, g1, gx))
            structs.append(_enc_branch_struct(x2s[b], ks[b], scales[b], wbs[b], None, h1s[b], h2s[b], gp, gm, g2, g1, gx,
                                              packed=packed[b], keep_bits=ctx.keeps[b]))
        arr = (_lib.EncoderBranch * len(live))(*structs)
        import ctypes
        w0 = ctypes.c_int(0)
        total = L.piml_encoder_workgroups(arr, len(live), ctypes.byref(w0))
        slots = [w0.value, total - w0.value] if len(live) == 2 else [total]
        parts = [torch.empty(n, part, **opt) for n in slots]
        sink = ParamGradSink.current(ctx.sink)
        flats, accumulate = (None, False)
        if sink is not None:
            flats, accumulate = sink.take([tuple(id(t) for t in ctx.params[7 * b + 1:7 * b + 7]) for b in live], part, opt)
            if flats is None:
                sink = None
        if sink is None:
            flats = [torch.empty(part, **opt) for _ in slots]
        for i in range(len(live)):
            arr[i].partials, arr[i].grads = parts[i].data_ptr(), flats[i].data_ptr()
        # inside ops.deferred_slot_sums(): the slot sums ride in the relfeat backward's launch, merged with the row decoders' of the
        # same pass (the bottleneck variants: three launches become one)
        defer = _defer_slot_sums([ctx.params[7 * b + jx] for b in live for jx in range(1, 7)], sink, dev)
        with torch.cuda.device(dev):
            _lib.check(L.piml_encoder_bwd_acc(arr, len(live), int(accumulate) | (_lib.DEFER_SLOT_SUMS if defer else 0), _stream()),
                       'piml_encoder_bwd')
        if defer:
            _defer_keep(parts, flats)
        H = ENCODER_HIDDEN
        for i, b in enumerate(live):
            flat = flats[i]
            in_dim = x2s[b].shape[1]
            need = ctx.needs_input_grad[6 + 7 * b:6 + 7 * b + 7]
            o = 6 + 7 * b
            if need[0]:
                grads[o] = keep[i][4].view(xshapes[b])
            dW3, dW2 = flat[:H * H].view(H, H), flat[H * H:2 * H * H].view(H, H)
            dW1 = flat[2 * H * H:2 * H * H + H * in_dim].view(H, in_dim)
            db3, db2, db1 = flat[2 * H * H + 8 * H:].view(3, H)
            for j, t in zip(range(1, 7), (dW1, db1, dW2, db2, dW3, db3)):
                if need[j]:
                    if sink is not None:
                        sink.give(ctx.params[7 * b + j], t)
                    else:
                        grads[o + j] = t
        return tuple(grads)


def fused_encoders(branches, packs=None):
    """branches: list (1 or 2 entries) of dicts {x (..., k, in<=8), scale, weights: (w1, b1, w2, b2, w3, b3) with the
    nn.Linear layouts (128, in), (128,), (128, 128), ..., pooled: bool, keep_bits: optional, the processor's
    train-mode dropout: int32 bits (rows, 4) (dropout_keep_bits layout) or ('draw', p) = the forward launch draws the mask
    itself from the device's dropout state (fold 1 / (1 - p) into `scale`; the same kind for every branch)}.
    Returns [(msgs (..., k, 128), pooled (..., 128) | None), ...]: msgs = keep * scale * encoder(x),
    pooled = msgs.sum(-2)  (src/models/model.py:1271-1283).  packs: a PinnsfPacks that pinnsf_prepack filled from these
    very encoder weights, in this branch order (skips the pack launch)."""
    if not 1 <= len(branches) <= 2:
        raise ValueError('fused_encoders: one or two branches')
    flat = []
    for br in branches:
        x, w = br['x'], br['weights']
        if not x.is_cuda:
            raise _lib.PimlHipError('fused_encoders: expected GPU tensors (piml_amd has no CPU path)')
        if x.dim() < 2 or not 1 <= x.shape[-1] <= ENCODER_MAX_IN or len(w) != 6 or \
                tuple(w[0].shape) != (ENCODER_HIDDEN, x.shape[-1]) or tuple(w[2].shape) != (ENCODER_HIDDEN,) * 2 or \
                tuple(w[4].shape) != (ENCODER_HIDDEN,) * 2 or any(tuple(w[i].shape) != (ENCODER_HIDDEN,) for i in (1, 3, 5)):
            raise ValueError('fused_encoders: x (..., k, in <= 8) and Linear(in, 128), Linear(128, 128) x 2 expected')
        if x.numel() == 0:
            raise ValueError('fused_encoders: empty input')
        flat += [x, *w]
    need_grad = torch.is_grad_enabled() and any(t.requires_grad for t in flat)
    keeps = tuple(b.get('keep_bits') for b in branches)
    if len({(k is None, isinstance(k, tuple)) for k in keeps}) > 1:
        raise ValueError('fused_encoders: the same kind of keep_bits (none / given bits / drawn) for every branch')
    out = _FusedEncoders.apply(len(branches), tuple(float(b['scale']) for b in branches),
                               tuple(bool(b.get('pooled', True)) for b in branches), need_grad, keeps, packs, *flat)
    return [(out[2 * i], out[2 * i + 1] if branches[i].get('pooled', True) else None) for i in range(len(branches))]


# ------------------------------------------------------------------------------------------------
# The whole non-bottleneck PINNSF network (`pinnsf`, `pinnsf_m`) as ONE autograd node on the fused f32-MFMA
# kernels: encoders (encoder.hip) -> neighbour-axis sum + decoder + predictor + desired force (decoder.hip).
# ------------------------------------------------------------------------------------------------
def _dec_branch_struct(msgs, agents, k, wb, packed, pooled=None, h1=None, d2=None, g_pre2=None, g_pre1=None,
                       g_pooled=None, partials=None, grads=None, fold=None, dw1_out=None):
    B = _lib.DecoderBranch()
    if fold is not None:              # (w3, b3, scale) of the branch's encoder: PIML_POOL_TRAIN (include/piml_hip.h)
        B.fold_w3, B.fold_b3, B.fold_scale = fold[0].data_ptr(), fold[1].data_ptr(), float(fold[2])
    B.dw1_out = _ptr(dw1_out)
    B.msgs, B.agents, B.k = msgs.data_ptr(), int(agents), int(k)
    B.w1, B.b1, B.w2, B.b2, B.w3, B.b3 = [t.data_ptr() for t in wb]
    B.pooled, B.h1, B.d2, B.g_pre2, B.g_pre1, B.g_pooled, B.partials = \
        [_ptr(t) for t in (pooled, h1, d2, g_pre2, g_pre1, g_pooled, partials)]
    B.packed, B.grads = _ptr(packed), _ptr(grads)
    return B


class PinnsfPacks:
    """Persistent MFMA operand images of one network's weights (encoder, decoder and head fragments) for callers whose
    weights stay put over several forward passes: a rollout (every frame of it), or the frames of one back-propagated
    training step.  `pinnsf_prepack` fills them once; forward passes given `packs=` then skip their own packs (one small
    launch each).  The caller owns the promise that the weights are not modified between the prepack and the last
    backward that uses the images (models.model._PINNSFBase.packed_weights is the context manager that keeps it)."""

    def __init__(self):
        self.epack = self.dpack = self.hpack = None
        self.sig = None          # (data pointer, version) of the packed weights
        self.sig_enc = self.sig_dec = None       # the same per encoder / decoder branch (fused_encoders / fused_row_decoder)
        self.active = False
        self.fold = None         # processor scales the decoders' / head's FOLDED images were packed with (PIML_POOL_TRAIN), or None

    def __deepcopy__(self, memo):          # images are derived data: a copied model packs for itself
        return PinnsfPacks()

    def __reduce__(self):
        return (PinnsfPacks, ())

    def ensure(self, dev):
        if self.epack is None or self.epack.device != dev:
            L = _lib.lib()
            opt = dict(device=dev, dtype=torch.float32)
            self.epack = torch.empty(2, L.piml_encoder_pack_floats(), **opt)
            self.dpack = torch.empty(2, L.piml_decoder_pack_floats(), **opt)
            self.hpack = torch.empty(L.piml_collision_head_pack_floats(), **opt)


def _pack_structs(enc_w, dec_w, head_w, packs, fold=None):
    nbr = len(enc_w)
    earr = (_lib.EncoderBranch * nbr)()
    darr = (_lib.DecoderBranch * nbr)()
    for b in range(nbr):
        earr[b].in_dim = enc_w[b][0].shape[1]
        earr[b].w1, earr[b].b1, earr[b].w2, earr[b].b2, earr[b].w3, earr[b].b3 = [t.data_ptr() for t in enc_w[b]]
        earr[b].packed = packs.epack[b].data_ptr()
        darr[b].w1, darr[b].b1, darr[b].w2, darr[b].b2, darr[b].w3, darr[b].b3 = [t.data_ptr() for t in dec_w[b]]
        darr[b].packed = packs.dpack[b].data_ptr()
        if fold is not None:
            darr[b].fold_w3, darr[b].fold_b3, darr[b].fold_scale = enc_w[b][4].data_ptr(), enc_w[b][5].data_ptr(), float(fold[b])
    head = None
    if head_w is not None:
        head = _lib.CollisionHead()
        head.w1, head.b1, head.w2, head.b2 = [t.data_ptr() for t in head_w]
        head.packed = packs.hpack.data_ptr()
        if fold is not None:
            head.fold_w3, head.fold_b3, head.fold_scale = enc_w[0][4].data_ptr(), enc_w[0][5].data_ptr(), float(fold[0])
    return earr, darr, head


def _branch_sig(wb):
    return tuple((t.data_ptr(), t._version) for t in wb)


def _weights_sig(enc_w, dec_w, head_w):
    """(storage address, in-place modification count) of every packed weight: an optimizer step or load_state_dict between the
    prepack and a forward pass that uses the images changes the count and is refused (stale operand images)."""
    return tuple((t.data_ptr(), t._version) for wb in (*enc_w, *dec_w, *([head_w] if head_w is not None else [])) for t in wb)


DEFER_PACK = _os.environ.get('PIML_DEFER_PACK', '1') != '0'
POOL_MSGS = _os.environ.get('PIML_POOL_MSGS', '1') != '0'      # fused_pinnsf(sums=True) under a dropout mask: the agents' sums of the messages from the forward's registers


def pinnsf_prepack(packs, enc_w, dec_w, head_w=None, defer=None, fold=None):
    """Pack the weights of a fused PINNSF network into `packs` (one launch on the current stream).
    enc_w / dec_w: per branch (w1, b1, w2, b2, w3, b3) encoder and (w1, b1, w2, b2, wp, bp) decoder + predictor tensors;
    head_w: (w1, b1, w2, b2) of the collision head or None.  defer (default: PIML_DEFER_PACK != 0): the launch is left to
    the next relfeat forward on the stream, which runs the pack as its trailing workgroups (PIML_DEFER_PACK of the C ABI);
    every consumer of the packs launches it itself if no relfeat forward came in between.
    fold: per branch the processor scale -- the same launch ALSO packs the decoders' first layers and the head's with the
    encoders' last layer folded in (fused_pinnsf(..., sums=True) / PIML_POOL_TRAIN); None: plain images only."""
    import ctypes
    enc_w = [[_gpu_f32('encoder weight', t.detach()) for t in wb] for wb in enc_w]
    dec_w = [[_gpu_f32('decoder weight', t.detach()) for t in wb] for wb in dec_w]
    head_w = None if head_w is None else [_gpu_f32('head weight', t.detach()) for t in head_w]
    dev = enc_w[0][0].device
    packs.ensure(dev)
    fold = None if fold is None else tuple(float(f) for f in fold)
    earr, darr, head = _pack_structs(enc_w, dec_w, head_w, packs, fold)
    packs.fold = fold
    with torch.cuda.device(dev):
        flags = _lib.DEFER_PACK if (DEFER_PACK if defer is None else defer) else 0
        _lib.check(_lib.lib().piml_pinnsf_pack(earr, darr, len(enc_w), ctypes.byref(head) if head is not None else None,
                                               flags, _stream()), 'piml_pinnsf_pack')
        packs.pending_structs = (earr, darr, head, enc_w, dec_w, head_w) if flags else None      # (alive until the pack has run)
    packs.sig = _weights_sig(enc_w, dec_w, head_w)
    packs.sig_enc = tuple(_branch_sig(wb) for wb in enc_w)
    packs.sig_dec = tuple(_branch_sig(wb) for wb in dec_w)


# side streams inside piml_pinnsf_fwd / bwd: off by default (cross-stream edges cost more than they hide in a HIP graph)
FORK_NETWORK = _os.environ.get('PIML_FORK_NETWORK', '0') == '1'


class ParamGradSink:
    """Weight gradients of the fused PINNSF network summed across the backward passes of ONE optimiser step inside the slot-sum
    launch (PIML_ACCUMULATE) instead of by autograd's per-tensor accumulation: a training rollout runs the network once per
    frame, and every backward pass after the first costs one `grad += new` launch per parameter tensor (~30 launches of 3 us
    at the fine-tuning step).  Use:

        sink = ops.ParamGradSink()                 # persistent: its buffers are what p.grad points into
        with sink.step():                          # zero_grad(set_to_none=True) before, optimizer.step() after
            loss = ...; loss.backward()

    Inside the block `fused_pinnsf`'s backward returns no weight gradients to autograd; at the end every parameter that got
    one has `p.grad` = a view of the sink's flat buffers (added to an existing p.grad, should autograd have produced one for
    the same tensor on another path).  Capturable into a HIP graph (the buffers are static)."""
    _active = None

    def __init__(self):
        self._bufs = {}          # key -> (encoder flats, decoder flats)
        self._seen = None        # keys that have had their first backward pass of the running step
        self._assign = None      # id(param) -> (param, view)
        self._mode = None        # (key, size) -> 'plain' / 'fold': what the buffer accumulates in the running step

    @contextlib.contextmanager
    def step(self):
        if ParamGradSink._active is not None:
            raise RuntimeError('ParamGradSink.step() does not nest')
        self._seen, self._assign, self._mode = set(), {}, {}
        ParamGradSink._active = self
        # the step's backward passes accumulate the FOLDED layers' gradients (sums path); their unfold is linear and overwrites
        # its outputs, so one launch at the end of the step stands for one per pass (piml_pinnsf_unfold_defer)
        dev = torch.cuda.current_device() if torch.cuda.is_available() else None
        if dev is not None:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().piml_pinnsf_unfold_defer(1, None), 'piml_pinnsf_unfold_defer')
        try:
            yield self
        finally:
            if dev is not None:
                with torch.cuda.device(dev):
                    _lib.check(_lib.lib().piml_pinnsf_unfold_defer(0, _stream()), 'piml_pinnsf_unfold_defer')
            ParamGradSink._active = None
            for p, view in self._assign.values():
                p.grad = view if p.grad is None else p.grad + view
            self._seen, self._assign, self._mode = None, None, None

    def take(self, keys, size, opt, mode='plain'):
        """Persistent flat buffers (`size` floats) for the weight sets `keys` (one per branch of the call) in this backward
        pass -> (buffers, accumulate?).  accumulate: they hold the sums of the step's earlier passes.  One launch has one
        flag: when only some of the branches have been through a pass before (the last frame of a rollout may feed the
        loss through one branch only), the others' buffers are cleared here and the launch accumulates into all of them.
        mode: what the buffer's fields mean in this pass -- 'fold' when the decoder's first-layer field accumulates the gradient
        of the FOLDED weight (the sums path; unfolded from the running sum at the step's end), 'plain' otherwise.  The two
        must not meet in one buffer inside one step (their sum is neither gradient): a mix raises."""
        if not keys:
            return None, False
        for k in keys:
            if self._mode.setdefault((k, size), mode) != mode:
                raise _lib.PimlHipError('ParamGradSink: one optimiser step mixes a sums-path pass and a message-path pass of the same '
                                        'network (the folded and the plain first-layer gradients would be added into one buffer); '
                                        'keep model.messages_wanted / dropout / packs.fold the same for every pass of a step')
        seen = [k in self._seen for k in keys]
        bufs = []
        for k, was in zip(keys, seen):
            b = self._bufs.get((k, size))
            if b is None:
                b = self._bufs[(k, size)] = torch.empty(size, **opt)
            if any(seen) and not was:
                b.zero_()
            bufs.append(b)
            self._seen.add(k)
        return bufs, any(seen)

    @staticmethod
    def current(ctx_sink):
        """the sink a backward pass may use: the one its forward ran under, if its step() is still open"""
        return ctx_sink if (ctx_sink is not None and ctx_sink is ParamGradSink._active) else None

    def give(self, param, view):
        self._assign.setdefault(id(param), (param, view))


_DEFER_DEPTH = 0
_DEFER_KEEP = []        # the slot buffers deferred sums will read: alive until those sums have been launched


def _defer_keep(*buffers):
    """Operators that leave their slot sums one by one (fused_encoders, the row decoder): the library merges what is waiting on the
    stream into one launch, so the buffers of the last few deferrals stay alive together; a further deferral launches its
    predecessors (piml_amd/csrc/network.hip: pending_slot_sums_leave), whose buffers may then go."""
    _DEFER_KEEP.append(buffers)
    del _DEFER_KEEP[:-4]
_DEFER_SEEN = set()     # id(parameter) of every parameter that was handed a deferred gradient inside the open block
_DEFER_DEVS = set()     # devices a deferral was left on (the flush at the block's exit visits each of them)


class deferred_slot_sums:
    """`with ops.deferred_slot_sums(): loss.backward()` -- inside, the backward of a fused network (fused_pinnsf) does not launch
    the sums of its weight-gradient slots; the backward of the relative features that follows it in the same pass
    (relative_features_packed_self) runs them as the leading workgroups of its own launch (PIML_DEFER_SLOT_SUMS: the two kernels
    are independent and small, a launch boundary costs ~5 us), and whatever is still waiting at exit is launched there.  The
    weight gradients are complete when the block exits, not before: the network's backward defers only when no parameter of
    it holds a .grad yet (autograd then keeps the buffers it is handed instead of reading them) or a ParamGradSink owns them.
    PIML_DEFER_SLOT_SUMS=0 in the environment turns the block into a no-op."""

    def __enter__(self):
        global _DEFER_DEPTH
        _DEFER_DEPTH += 1
        return self

    def __exit__(self, *exc):
        global _DEFER_DEPTH
        _DEFER_DEPTH -= 1
        if _DEFER_DEPTH == 0:
            _flush_deferred_sums()
            _DEFER_SEEN.clear()
        return False


def _flush_deferred_sums():
    """launch whatever sums are still waiting, on every device a deferral was left on (the library keeps one entry per device
    and flushes the CURRENT device's)"""
    for dev in list(_DEFER_DEVS) or [None]:
        with (torch.cuda.device(dev) if dev is not None else contextlib.nullcontext()):
            _lib.check(_lib.lib().piml_pinnsf_slot_sums_flush(), 'piml_pinnsf_slot_sums_flush')
    _DEFER_DEVS.clear()
    _DEFER_KEEP.clear()


def _defer_slot_sums(params, sink, dev=None):
    """May this backward pass leave its slot sums to the relfeat backward's launch?  Autograd is handed gradient buffers whose
    sums have not run yet, which is only sound while nobody reads them before the block's next launch on the stream:
      * a ParamGradSink owns the buffers (autograd sees no weight gradients at all): yes;
      * otherwise every parameter must be without a .grad (AccumulateGrad then keeps the buffer it is handed -- a view object
        nobody else holds -- instead of adding to it), without tensor / post-accumulate hooks (they would read it), and must not
        have been handed a deferred gradient by an EARLIER node of this block (the engine adds two nodes' gradients for one
        parameter as soon as the second arrives).
    When the answer is no inside an open block, the sums still waiting are launched first: a buffer of an earlier node may be
    about to be read."""
    if _DEFER_DEPTH <= 0 or FORK_NETWORK or _os.environ.get('PIML_DEFER_SLOT_SUMS', '1') == '0':
        return False
    ok = sink is not None
    if not ok:
        ok = all(getattr(p, 'grad', None) is None and id(p) not in _DEFER_SEEN and not getattr(p, '_backward_hooks', None)
                 and not getattr(p, '_post_accumulate_grad_hooks', None) for p in params)
        if ok:
            _DEFER_SEEN.update(id(p) for p in params)
    if ok:
        _DEFER_DEVS.add(dev)
    else:
        _flush_deferred_sums()
    return ok


class _FusedPinnsf(torch.autograd.Function):
    """inputs: need_grad, nbr, scales, tau, fold_epilogue, packs (PinnsfPacks or None), nhead (0 / 1),
    self_features (..., N, 7), then per branch x (..., N, k, in), encoder w1 b1 w2 b2 w3 b3, decoder w1 b1 w2 b2,
    predictor w b (13 tensors), then the collision head's w1 b1 w2 b2 when nhead.
    outputs: acc (..., N, 2) (= predictions when fold_epilogue), msgs (..., N, k, 128) per branch, and with a head
    sigmoid(head(msgs of branch 0)) (..., N, k)."""
    PER = 13
    FIRST = 10         # index of the first branch tensor among the inputs
    SELF = 9           # index of self_features

    @staticmethod
    def forward(ctx, need_grad, nbr, scales, tau, fold_epilogue, packs, nhead, keeps, sums, self_features, *tensors):
        import ctypes
        L = _lib.lib()
        PER = _FusedPinnsf.PER
        xs = [tensors[PER * b] for b in range(nbr)]
        ewb = [[_gpu_f32('encoder weight', t.detach()) for t in tensors[PER * b + 1:PER * b + 7]] for b in range(nbr)]
        dwb = [[_gpu_f32('decoder weight', t.detach()) for t in tensors[PER * b + 7:PER * b + 13]] for b in range(nbr)]
        hwb = [_gpu_f32('head weight', t.detach()) for t in tensors[PER * nbr:PER * nbr + 4]] if nhead else None
        dev = xs[0].device
        opt = dict(device=dev, dtype=torch.float32)
        H = ENCODER_HIDDEN
        lead = tuple(xs[0].shape[:-2])
        agents = 1
        for d in lead:
            agents *= d
        sf = _gpu_f32('self_features', self_features).reshape(agents, 7) if fold_epilogue else None
        x2s, ks, msgs, h1s, h2s = [], [], [], [], []
        msum = False
        for b in range(nbr):
            x = _gpu_f32('encoder input', xs[b])
            if tuple(x.shape[:-2]) != lead:
                raise ValueError('fused_pinnsf: both branches must share the leading (..., N) shape')
            x2s.append(x.reshape(-1, x.shape[-1]))
            ks.append(x.shape[-2])
        if sums:            # PIML_POOL_TRAIN where the library serves the shape (and no sink sums gradients across passes)
            probe = (_lib.EncoderBranch * nbr)()
            for b in range(nbr):
                probe[b].rows, probe[b].in_dim, probe[b].k = x2s[b].shape[0], x2s[b].shape[1], ks[b]
            wanted = sums
            sums = bool(L.piml_pinnsf_pool_train_ok(probe, nbr)) and not FORK_NETWORK and \
                (packs is None or packs.fold == tuple(float(sc) for sc in scales)) and all(k is None for k in keeps)
            # PIML_POOL_MSGS: the caller reads no messages, but a dropout mask (or packs without the folded images, or a gradient
            # on the collision head: see backward) keeps the sum behind the last layer -- the forward kernel still leaves the
            # agents' sums (last layer with exchanged operands) and stores message rows only for the head; plain backward
            if wanted and not sums and need_grad and not FORK_NETWORK and RELU_MASK and POOL_MSGS:
                for b in range(nbr):
                    probe[b].relu_mask = 1          # (any non-NULL value: the probe looks at the configuration only)
                msum = bool(L.piml_pinnsf_pool_msgs_ok(probe, nbr))
        ctx.sums = sums
        sum_a, sum_b, masks = [None] * nbr, [None] * nbr, [None] * nbr
        for b in range(nbr):
            R = x2s[b].shape[0]
            h1s.append(None)
            if msum:
                msgs.append(torch.empty(R, H, **opt) if (nhead and b == 0) else None)
                sum_a[b], sum_b[b] = torch.empty(agents, H, **opt), torch.empty(agents, H, **opt)
                h2s.append(_h2_buffer(R, opt))
            elif sums:
                # the agents' sums of h2 in two parts; the sign words of h1 / h2 (256 dwords per tile); the h2 rows only where the
                # collision head reads them (branch 0)
                msgs.append(None)
                sum_a[b], sum_b[b] = torch.empty(agents, H, **opt), torch.empty(agents, H, **opt)
                masks[b] = torch.empty(2 * ((R + 31) // 32), H, **opt)
                h2s.append(torch.empty(R, H, **opt) if (nhead and b == 0) else None)
            else:
                msgs.append(torch.empty(R, H, **opt))
                h2s.append(_h2_buffer(R, opt) if need_grad else None)
        if not sums and need_grad and _h1_needed([x2.shape[0] for x2 in x2s], alone=False):
            h1s = [torch.empty(x2.shape[0], H, **opt) for x2 in x2s]
        flags = (_lib.FORK if FORK_NETWORK else 0) | (_lib.POOL_TRAIN if sums else 0) | (_lib.POOL_MSGS if msum else 0)
        if packs is not None:
            if packs.sig != _weights_sig(ewb, dwb, hwb):
                raise ValueError('fused_pinnsf: `packs` were filled from other weight tensors, or the weights were modified in '
                                 'place since (optimizer step / load_state_dict inside a packed_weights() block): pinnsf_prepack first')
            epack, dpack, hpack = packs.epack, packs.dpack, packs.hpack
            flags |= _lib.PACKED_VALID
        else:
            epack = torch.empty(nbr, L.piml_encoder_pack_floats(), **opt)
            dpack = torch.empty(nbr, L.piml_decoder_pack_floats(), **opt)
            hpack = torch.empty(L.piml_collision_head_pack_floats(), **opt) if nhead else None
        keeps, draws = zip(*[_resolve_keep(keeps[b], x2s[b].shape[0], H, dev) for b in range(nbr)])
        earr = (_lib.EncoderBranch * nbr)(*[_enc_branch_struct(x2s[b], ks[b], scales[b], ewb[b], msgs[b], h1s[b], h2s[b],
                                                               packed=epack[b], keep_bits=keeps[b], draw_p=draws[b],
                                                               relu_mask=masks[b], sum_a=sum_a[b], sum_b=sum_b[b])
                                            for b in range(nbr)])
        # (sums: the decoder reads the first parts from `pooled` and leaves the completed sums there)
        pooled = sum_a if (sums or msum) else [torch.empty(agents, H, **opt) for _ in range(nbr)]      # always: the decoder kernel reads it
        dh1 = [torch.empty(agents, 64, **opt) if need_grad else None for _ in range(nbr)]
        dd2 = [torch.empty(agents, 64, **opt) if need_grad else None for _ in range(nbr)]
        folds = [(ewb[b][4], ewb[b][5], scales[b]) if sums else None for b in range(nbr)]
        darr = (_lib.DecoderBranch * nbr)(*[_dec_branch_struct(sum_b[b] if (sums or msum) else msgs[b], agents, ks[b], dwb[b], dpack[b], pooled[b],
                                                               dh1[b], dd2[b], fold=folds[b]) for b in range(nbr)])
        acc = torch.empty(agents, 2, **opt)
        head, coll = None, None
        if nhead:
            coll = torch.empty(x2s[0].shape[0], **opt)
            head = _lib.CollisionHead()
            head.msgs, head.rows = (h2s[0] if sums else msgs[0]).data_ptr(), x2s[0].shape[0]
            head.w1, head.b1, head.w2, head.b2 = [t.data_ptr() for t in hwb]
            head.packed, head.out = hpack.data_ptr(), coll.data_ptr()
            if sums:
                head.fold_w3, head.fold_b3, head.fold_scale = ewb[0][4].data_ptr(), ewb[0][5].data_ptr(), float(scales[0])
        with torch.cuda.device(dev):
            _lib.check(L.piml_pinnsf_fwd(earr, darr, nbr, ctypes.byref(head) if head is not None else None, _ptr(sf),
                                         float(tau), _ptr(acc), flags, _stream()), 'piml_pinnsf_fwd')
        if need_grad:
            # (sums: the sign words in h2's place; the second parts, which the backward does not read, in the messages')
            # (msum: the pedestrian message rows where the head read them -- its backward needs them -- else the second parts, which
            # only stand in the decoder structs' `msgs` field)
            saved_msgs = sum_b if sums else ([msgs[b] if msgs[b] is not None else sum_b[b] for b in range(nbr)] if msum else msgs)
            ctx.save_for_backward(*x2s, *h1s, *(masks if sums else h2s), *saved_msgs, *pooled, *dh1, *dd2,
                                  *[w for wb in ewb for w in wb],
                                  *[w for wb in dwb for w in wb], epack, dpack, *([sf] if sf is not None else []),
                                  *(hwb if nhead else []))
        ctx.meta = (nbr, tuple(scales), float(tau), bool(fold_epilogue), tuple(ks), [tuple(x.shape) for x in xs],
                    tuple(self_features.shape), agents, need_grad, int(nhead))
        ctx.keeps = keeps
        ctx.sink = ParamGradSink._active if need_grad else None
        ctx.params = tensors if need_grad else None       # (the Parameter objects themselves: p.grad is set on them / looked at)
        ctx.set_materialize_grads(False)
        out = (acc.view(*lead, 2), *[None if (sums or msum) else msgs[b].view(*lead, ks[b], H) for b in range(nbr)])
        if nhead:
            out = out + (coll.view(*lead, ks[0]),)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_acc, *g_rest):
        nbr, scales, tau, fold, ks, xshapes, sf_shape, agents, need_grad, nhead = ctx.meta
        PER, FIRST = _FusedPinnsf.PER, _FusedPinnsf.FIRST
        nin = FIRST + PER * nbr + 4 * nhead
        grads = [None] * nin
        g_msgs = list(g_rest[:nbr])
        g_coll = g_rest[nbr] if nhead else None
        if not need_grad or (g_acc is None and g_coll is None and all(g is None for g in g_msgs)):
            return tuple(grads)
        L = _lib.lib()
        sv = list(ctx.saved_tensors)
        take = lambda n: [sv.pop(0) for _ in range(n)]
        x2s, h1s, h2s, msgs, pooled, dh1, dd2 = [take(nbr) for _ in range(7)]
        ewb = [take(6) for _ in range(nbr)]
        dwb = [take(6) for _ in range(nbr)]
        epack, dpack = take(2)
        sf = sv.pop(0) if fold else None
        hwb = take(4) if nhead else None
        dev = x2s[0].device
        opt = dict(device=dev, dtype=torch.float32)
        H = ENCODER_HIDDEN
        flags = _lib.FORK if FORK_NETWORK else 0
        if ctx.sums:
            return _FusedPinnsf._backward_sums(ctx, g_acc, g_coll, grads, x2s, h2s, pooled, dh1, dd2, ewb, dwb, epack, dpack, sf)
        if g_coll is not None:           # rare (the reference trains this head for `pinnsf_bm` only): torch ops
            with torch.enable_grad():
                ins = [t.detach().requires_grad_(True) for t in (msgs[0], *hwb)]
                y = torch.sigmoid(torch.relu(ins[0] @ ins[1].t() + ins[2]) @ ins[3].t() + ins[4]).squeeze(-1)
                hg = torch.autograd.grad(y, ins, g_coll.reshape(-1), allow_unused=True)
            o = FIRST + PER * nbr
            for jx in range(4):
                if ctx.needs_input_grad[o + jx]:
                    grads[o + jx] = hg[1 + jx]
            g_msgs[0] = hg[0] if g_msgs[0] is None else g_msgs[0].reshape(-1, H) + hg[0]
        live = [b for b in range(nbr) if g_acc is not None or g_msgs[b] is not None]
        keep = []
        with torch.cuda.device(dev):
            part = L.piml_encoder_partial_floats()
            estructs, g_pooled = [], [None] * nbr
            for b in live:
                R, in_dim = x2s[b].shape
                gm = _gpu_f32('g_msgs', g_msgs[b]).reshape(R, H) if g_msgs[b] is not None else None
                g2, g1 = torch.empty(R, H, **opt), torch.empty(R, H, **opt)
                gx = torch.empty(R, in_dim, **opt) if ctx.needs_input_grad[FIRST + PER * b] else None
                if g_acc is not None:
                    g_pooled[b] = torch.empty(agents, H, **opt)
                keep.append((gm, g2, g1, gx))
                estructs.append(_enc_branch_struct(x2s[b], ks[b], scales[b], ewb[b], None, h1s[b], h2s[b], g_pooled[b], gm,
                                                   g2, g1, gx, packed=epack[b], keep_bits=ctx.keeps[b]))
            earr = (_lib.EncoderBranch * len(live))(*estructs)
            import ctypes
            w0 = ctypes.c_int(0)
            total = L.piml_encoder_workgroups(earr, len(live), ctypes.byref(w0))
            slots = [w0.value, total - w0.value] if len(live) == 2 else [total]
            parts = [torch.empty(n, part, **opt) for n in slots]
            # weight gradients: fresh buffers handed to autograd, or -- inside ParamGradSink.step(), whole network only -- the
            # sink's persistent ones, summed across the backward passes of the step by the slot-sum launch itself
            sink = ParamGradSink.current(ctx.sink) if (g_acc is not None and not FORK_NETWORK) else None
            dflats = None
            if sink is not None:
                ids = lambda lo, hi: tuple(id(t) for t in ctx.params[lo:hi])
                flats, acc_e = sink.take([ids(PER * b + 1, PER * b + 7) for b in live], part, opt)
                dflats, acc_d = sink.take([ids(PER * b + 7, PER * b + 13) for b in range(nbr)], L.piml_decoder_partial_floats(), opt)
                if flats is None or dflats is None or acc_e != acc_d:
                    sink, dflats = None, None
                elif acc_e:
                    flags |= _lib.ACCUMULATE
            if sink is None:
                flats = [torch.empty(part, **opt) for _ in slots]
            for i in range(len(live)):
                earr[i].partials, earr[i].grads = parts[i].data_ptr(), flats[i].data_ptr()
            if g_acc is not None:           # decoder tails + encoders: one forked call
                ga = _gpu_f32('g_acc', g_acc).reshape(agents, 2)
                want_self = fold and ctx.needs_input_grad[_FusedPinnsf.SELF]
                g_self = torch.empty(agents, 7, **opt) if want_self else None
                nwg = L.piml_decoder_workgroups(agents)
                dparts = [torch.empty(nwg, L.piml_decoder_partial_floats(), **opt) for _ in range(nbr)]
                if dflats is None:
                    dflats = [torch.empty(L.piml_decoder_partial_floats(), **opt) for _ in range(nbr)]
                dstructs = []
                for b in range(nbr):
                    gp2, gp1 = torch.empty(agents, 64, **opt), torch.empty(agents, 64, **opt)
                    keep += [gp2, gp1]
                    dstructs.append(_dec_branch_struct(msgs[b], agents, ks[b], dwb[b], dpack[b], pooled[b], dh1[b], dd2[b],
                                                       gp2, gp1, g_pooled[b], dparts[b], dflats[b]))
                darr = (_lib.DecoderBranch * nbr)(*dstructs)
                if len(live) == nbr and _defer_slot_sums([ctx.params[PER * b + jx] for b in range(nbr) for jx in range(1, 13)], sink, dev):
                    flags |= _lib.DEFER_SLOT_SUMS
                    _defer_keep(parts, dparts)     # (append + bounded trim: the library may MERGE this deferral with one still waiting)
                _lib.check(L.piml_pinnsf_bwd(earr, darr, nbr, _ptr(ga), _ptr(sf), float(tau), _ptr(g_self), flags,
                                             _stream()), 'piml_pinnsf_bwd')
                if want_self:
                    grads[_FusedPinnsf.SELF] = g_self.view(sf_shape)
                for b in range(nbr):
                    flat = dflats[b]
                    o = FIRST + PER * b + 7
                    need = ctx.needs_input_grad[o:o + 6]
                    dW1, dW2 = flat[:64 * H].view(64, H), flat[64 * H:64 * H + 4096].view(64, 64)
                    dW3 = flat[64 * H + 4096:64 * H + 4096 + 128].view(2, 64)
                    rest = flat[64 * H + 4096 + 128:]
                    for jx, t in enumerate((dW1, rest[:64], dW2, rest[64:128], dW3, rest[128:130])):
                        if need[jx]:
                            if sink is not None:
                                sink.give(ctx.params[o + jx - FIRST], t)
                            else:
                                grads[o + jx] = t
            else:                            # only the messages carry a gradient: the encoders alone
                if h1s[live[0]] is None and _h1_needed([x2s[b].shape[0] for b in live], alone=False):
                    raise _lib.PimlHipError(
                        'fused_pinnsf: a backward pass through the messages of a subset of the branches only needs the '
                        'layer-1 activations, which the forward did not store; run it with PIML_H1_RECOMPUTE=0')
                _lib.check(L.piml_encoder_bwd(earr, len(live), _stream()), 'piml_encoder_bwd')
            for i, b in enumerate(live):
                flat = flats[i]
                in_dim = x2s[b].shape[1]
                o = FIRST + PER * b
                need = ctx.needs_input_grad[o:o + 7]
                if need[0]:
                    grads[o] = keep[i][3].view(xshapes[b])
                dW3, dW2 = flat[:H * H].view(H, H), flat[H * H:2 * H * H].view(H, H)
                dW1 = flat[2 * H * H:2 * H * H + H * in_dim].view(H, in_dim)
                db3, db2, db1 = flat[2 * H * H + 8 * H:].view(3, H)
                for jx, t in zip(range(1, 7), (dW1, db1, dW2, db2, dW3, db3)):
                    if need[jx]:
                        if sink is not None:
                            sink.give(ctx.params[o + jx - FIRST], t)
                        else:
                            grads[o + jx] = t
        return tuple(grads)


def _backward_sums(ctx, g_acc, g_coll, grads, x2s, masks, pooled, dh1, dd2, ewb, dwb, epack, dpack, sf):
    """Backward of a PIML_POOL_TRAIN forward (fused_pinnsf(..., sums=True)): decoder tails with the folded first layer ->
    one-pass encoder backward on G2 = d/d(sum)[agent] * [h2 > 0] -> slot sums -> the folded layers' gradients unfolded."""
    import ctypes
    nbr, scales, tau, fold, ks, xshapes, sf_shape, agents, need_grad, nhead = ctx.meta
    PER, FIRST = _FusedPinnsf.PER, _FusedPinnsf.FIRST
    if g_coll is not None:
        raise _lib.PimlHipError('fused_pinnsf(sums=True): the collision head\'s output received a gradient; this form serves callers '
                                'that train on predictions[0] only (model.messages_wanted = True selects the message path)')
    if g_acc is None:
        return tuple(grads)
    L = _lib.lib()
    dev = x2s[0].device
    opt = dict(device=dev, dtype=torch.float32)
    H = ENCODER_HIDDEN
    with torch.cuda.device(dev):
        part1 = H * H + 1024 + 2 * H                       # a layer-1 slot: dW2 | dW1 (1024-float field) | db2 | db1
        g_pooled = [torch.empty(agents, H, **opt) for _ in range(nbr)]
        gxs = [torch.empty(x2s[b].shape, **opt) if ctx.needs_input_grad[FIRST + PER * b] else None for b in range(nbr)]
        if len({g is None for g in gxs}) > 1:              # one kernel variant per launch: both inputs' gradients or neither
            gxs = [g if g is not None else torch.empty(x2s[b].shape, **opt) for b, g in enumerate(gxs)]
        earr = (_lib.EncoderBranch * nbr)(*[_enc_branch_struct(x2s[b], ks[b], scales[b], ewb[b], None, None, None, g_pooled[b], None,
                                                               None, None, gxs[b], packed=epack[b], relu_mask=masks[b])
                                            for b in range(nbr)])
        w0 = ctypes.c_int(0)
        total = L.piml_encoder_workgroups(earr, nbr, ctypes.byref(w0))
        slots = [w0.value, total - w0.value] if nbr == 2 else [total]
        parts = [torch.empty(n, part1, **opt) for n in slots]
        # weight gradients: fresh buffers handed to autograd, or -- inside ParamGradSink.step() -- the sink's persistent ones, summed
        # across the backward passes of the step by the slot-sum launch itself (the folded layers' gradients accumulate FOLDED and
        # are unfolded from their running sums: the unfold is linear)
        sink = ParamGradSink.current(ctx.sink)
        flags = _lib.POOL_TRAIN
        flats = dflats = dw1 = None
        if sink is not None:
            ids = lambda lo, hi: tuple(id(t) for t in ctx.params[lo:hi])
            flats, acc_e = sink.take([ids(PER * b + 1, PER * b + 7) for b in range(nbr)], L.piml_encoder_partial_floats(), opt)
            dflats, acc_d = sink.take([ids(PER * b + 7, PER * b + 13) for b in range(nbr)], L.piml_decoder_partial_floats(), opt, mode='fold')
            dw1, _ = sink.take([ids(PER * b + 7, PER * b + 13) for b in range(nbr)], 64 * H, opt, mode='fold')
            if acc_e != acc_d:
                raise _lib.PimlHipError('fused_pinnsf(sums=True): the encoders and decoders of one network went through different numbers '
                                        'of backward passes inside ParamGradSink.step()')
            if acc_e:
                flags |= _lib.ACCUMULATE
        else:
            flats = [torch.empty(L.piml_encoder_partial_floats(), **opt) for _ in range(nbr)]
        for b in range(nbr):
            earr[b].partials, earr[b].grads = parts[b].data_ptr(), flats[b].data_ptr()
        ga = _gpu_f32('g_acc', g_acc).reshape(agents, 2)
        want_self = fold and ctx.needs_input_grad[_FusedPinnsf.SELF]
        g_self = torch.empty(agents, 7, **opt) if want_self else None
        nwg = L.piml_decoder_workgroups(agents)
        dparts = [torch.empty(nwg, L.piml_decoder_partial_floats(), **opt) for _ in range(nbr)]
        if sink is None:
            dflats = [torch.empty(L.piml_decoder_partial_floats(), **opt) for _ in range(nbr)]
            dw1 = [torch.empty(64 * H, **opt) for _ in range(nbr)]
        keep, dstructs = [], []
        for b in range(nbr):
            gp2, gp1 = torch.empty(agents, 64, **opt), torch.empty(agents, 64, **opt)
            keep += [gp2, gp1]
            dstructs.append(_dec_branch_struct(pooled[b], agents, ks[b], dwb[b], dpack[b], pooled[b], dh1[b], dd2[b], gp2, gp1,
                                               g_pooled[b], dparts[b], dflats[b], fold=(ewb[b][4], ewb[b][5], scales[b]), dw1_out=dw1[b]))
        darr = (_lib.DecoderBranch * nbr)(*dstructs)
        if _defer_slot_sums([ctx.params[PER * b + jx] for b in range(nbr) for jx in range(1, 13)], sink, dev):
            flags |= _lib.DEFER_SLOT_SUMS
            _defer_keep(parts, dparts, dflats, flats, dw1)
        _lib.check(L.piml_pinnsf_bwd(earr, darr, nbr, _ptr(ga), _ptr(sf), float(tau), _ptr(g_self), flags, _stream()), 'piml_pinnsf_bwd')
        if want_self:
            grads[_FusedPinnsf.SELF] = g_self.view(sf_shape)
        for b in range(nbr):
            flat = dflats[b]
            o = FIRST + PER * b + 7
            need = ctx.needs_input_grad[o:o + 6]
            dW2 = flat[64 * H:64 * H + 4096].view(64, 64)
            dW3 = flat[64 * H + 4096:64 * H + 4096 + 128].view(2, 64)
            rest = flat[64 * H + 4096 + 128:]
            # (views: AccumulateGrad keeps a gradient it is handed unread only while nobody else holds the tensor object)
            for jx, t in enumerate((dw1[b].view(64, H), rest[:64], dW2, rest[64:128], dW3, rest[128:130])):      # (d/d(b1) = d/d(b1'))
                if need[jx]:
                    if sink is not None:
                        sink.give(ctx.params[o + jx - FIRST], t)
                    else:
                        grads[o + jx] = t
            flat = flats[b]
            in_dim = x2s[b].shape[1]
            o = FIRST + PER * b
            need = ctx.needs_input_grad[o:o + 7]
            if need[0]:
                grads[o] = gxs[b].view(xshapes[b])
            dW3, dW2 = flat[:H * H].view(H, H), flat[H * H:2 * H * H].view(H, H)
            dW1 = flat[2 * H * H:2 * H * H + H * in_dim].view(H, in_dim)
            db3, db2, db1 = flat[2 * H * H + 8 * H:].view(3, H)
            for jx, t in zip(range(1, 7), (dW1, db1, dW2, db2, dW3, db3)):
                if need[jx]:
                    if sink is not None:
                        sink.give(ctx.params[o + jx - FIRST], t)
                    else:
                        grads[o + jx] = t
    return tuple(grads)


_FusedPinnsf._backward_sums = staticmethod(_backward_sums)


def fused_pinnsf(branches, self_features, tau, fold_epilogue=True, head=None, packs=None, sums=False):
    """The non-bottleneck PINNSF network on the fused kernels.  branches: 1 or 2 dicts {x (..., N, k, in <= 8), scale,
    encoder: (w1, b1, w2, b2, w3, b3), decoder: (w1 (64,128), b1, w2 (64,64), b2), predictor: (w (2,64), b),
    keep_bits: optional int32 (rows, 4) train-mode dropout mask of the processor (see fused_encoders)}.
    Returns (acc (..., N, 2), [msgs per branch]): acc = sum over branches of predictor(decoder(sum_k msgs)), plus the
    desired-force term (v0 d/|d| - v) / tau of self_features (..., N, 7) when fold_epilogue.
    head: (w1 (64,128), b1, w2 (1,64), b2) of the `pinnsf_m` collision head; its sigmoid output on the messages of
    branch 0, (..., N, k), is appended to the result (computed beside the decoder tails on a side stream).
    packs: a PinnsfPacks filled by pinnsf_prepack from these very weights (skips the in-call packs).
    sums: the caller reads neither branch's messages and trains on acc only (the reference's loops with reg_weight = 0,
    src/models/simulators.py:331-347, :702-737).  Where no branch carries a dropout mask and the library serves the shape
    (PIML_POOL_TRAIN, include/piml_hip.h) the network then runs on the agents' SUMS of h2: the messages are linear in h2, so
    the neighbour-axis sum moves in front of the encoders' last layer, which is folded into the decoders' first (and the head's);
    the returned messages are None.  Elsewhere the flag changes nothing."""
    if not 1 <= len(branches) <= 2:
        raise ValueError('fused_pinnsf: one or two branches')
    flat = []
    for br in branches:
        x, e, d, p = br['x'], br['encoder'], br['decoder'], br['predictor']
        if not x.is_cuda:
            raise _lib.PimlHipError('fused_pinnsf: expected GPU tensors (piml_amd has no CPU path)')
        H = ENCODER_HIDDEN
        ok = (x.dim() >= 3 and 1 <= x.shape[-1] <= ENCODER_MAX_IN and x.numel() > 0 and
              [tuple(t.shape) for t in e] == [(H, x.shape[-1]), (H,), (H, H), (H,), (H, H), (H,)] and
              [tuple(t.shape) for t in d] == [(64, H), (64,), (64, 64), (64,)] and
              [tuple(t.shape) for t in p] == [(2, 64), (2,)])
        if not ok:
            raise ValueError('fused_pinnsf: unsupported geometry (encoder in<=8 -> 128 x3, decoder 128 -> 64 -> 64, '
                             'predictor 64 -> 2)')
        flat += [x, *e, *d, *p]
    if head is not None:
        if [tuple(t.shape) for t in head] != [(64, ENCODER_HIDDEN), (64,), (1, 64), (1,)]:
            raise ValueError('fused_pinnsf: head must be (w1 (64,128), b1 (64), w2 (1,64), b2 (1))')
        flat += list(head)
    if tuple(self_features.shape[:-1]) != tuple(branches[0]['x'].shape[:-2]) or self_features.shape[-1] != 7:
        raise ValueError('fused_pinnsf: self_features (..., N, 7) must match the features\' leading shape')
    need_grad = torch.is_grad_enabled() and (any(t.requires_grad for t in flat) or self_features.requires_grad)
    keeps = tuple(b.get('keep_bits') for b in branches)
    if len({(k is None, isinstance(k, tuple)) for k in keeps}) > 1:
        raise ValueError('fused_pinnsf: the same kind of keep_bits (none / given bits / drawn) for every branch')
    out = _FusedPinnsf.apply(need_grad, len(branches), tuple(float(b['scale']) for b in branches), float(tau),
                             bool(fold_epilogue), packs, int(head is not None), keeps, bool(sums), self_features, *flat)
    nbr = len(branches)
    if head is not None:
        return out[0], list(out[1:1 + nbr]), out[1 + nbr]
    return out[0], list(out[1:])


def pooled_h2_decoder_weights(enc_w, dec_w, scale, k):
    """The decoder's first layer with the encoder's last layer folded in (PIML_POOL_H2): sum_r scale (W3 h2_r + b3) fed to
    W_d1 x + b_d1 is (scale W_d1 W3) sum_r h2_r + (b_d1 + scale k W_d1 b3).  enc_w = (w1, b1, w2, b2, w3, b3), dec_w = (w1, b1,
    w2, b2, wp, bp); returns dec_w with the first two replaced (float64 products, rounded once)."""
    w3, b3, wd1, bd1 = enc_w[4].detach().double(), enc_w[5].detach().double(), dec_w[0].detach().double(), dec_w[1].detach().double()
    w1c = (float(scale) * (wd1 @ w3)).float().contiguous()
    b1c = (bd1 + float(scale) * int(k) * (wd1 @ b3)).float().contiguous()
    return [w1c, b1c, *dec_w[2:]]


def fused_pinnsf_pooled(branches, self_features, tau, fold_epilogue=True, packs=None):
    """INFERENCE form of fused_pinnsf (no autograd, no dropout, no head, nobody reads the per-row messages): the encoder launch
    stops after layer 2 and leaves the agents' sums of h2, the decoder tails run on them (PIML_POOL_H2 of piml_pinnsf_fwd).
    branches as for fused_pinnsf, but `decoder` is what pooled_h2_decoder_weights made of the decoder + predictor weights
    (w1', b1', w2, b2, wp, bp) and `scale` is not applied again.  packs: a PinnsfPacks prepacked from these very tensors.
    Returns acc (..., N, 2), or None when the library does not serve the configuration (k other than 6 / 10, few rows, f32
    matrix instruction): the caller then takes fused_pinnsf."""
    import ctypes
    if torch.is_grad_enabled() and any(t.requires_grad for b in branches for t in (b['x'], *b['encoder'], *b['decoder'])):
        raise ValueError('fused_pinnsf_pooled is an inference path: call it under torch.no_grad()')
    L = _lib.lib()
    nbr = len(branches)
    H = ENCODER_HIDDEN
    xs = [_gpu_f32('encoder input', b['x']) for b in branches]
    lead = tuple(xs[0].shape[:-2])
    if any(tuple(x.shape[:-2]) != lead for x in xs) or tuple(self_features.shape[:-1]) != lead or self_features.shape[-1] != 7:
        raise ValueError('fused_pinnsf_pooled: the branches and self_features must share the leading (..., N) shape')
    dev = xs[0].device
    opt = dict(device=dev, dtype=torch.float32)
    agents = 1
    for d in lead:
        agents *= d
    ewb = [[_gpu_f32('encoder weight', t.detach()) for t in b['encoder']] for b in branches]
    dwb = [[_gpu_f32('decoder weight', t.detach()) for t in b['decoder']] for b in branches]
    x2s = [x.reshape(-1, x.shape[-1]) for x in xs]
    ks = [x.shape[-2] for x in xs]
    flags = _lib.POOL_H2
    if packs is not None:
        if packs.sig != _weights_sig(ewb, dwb, None):
            raise ValueError('fused_pinnsf_pooled: `packs` were filled from other weight tensors: pinnsf_prepack first')
        epack, dpack = packs.epack, packs.dpack
        flags |= _lib.PACKED_VALID
    else:
        epack = torch.empty(nbr, L.piml_encoder_pack_floats(), **opt)
        dpack = torch.empty(nbr, L.piml_decoder_pack_floats(), **opt)
    part_a = [torch.empty(agents, H, **opt) for _ in range(nbr)]        # sums from the tile an agent's first row lies in
    part_b = [torch.empty(agents, H, **opt) for _ in range(nbr)]        # ... from the next tile (straddling agents)
    earr = (_lib.EncoderBranch * nbr)(*[_enc_branch_struct(x2s[b], ks[b], 1.0, ewb[b], part_a[b], None, part_b[b], packed=epack[b])
                                        for b in range(nbr)])
    if not L.piml_pinnsf_pool_h2_ok(earr, nbr):
        return None
    darr = (_lib.DecoderBranch * nbr)(*[_dec_branch_struct(part_b[b], agents, ks[b], dwb[b], dpack[b], part_a[b], None, None)
                                        for b in range(nbr)])
    sf = _gpu_f32('self_features', self_features).reshape(agents, 7) if fold_epilogue else None
    acc = torch.empty(agents, 2, **opt)
    with torch.cuda.device(dev):
        _lib.check(L.piml_pinnsf_fwd(earr, darr, nbr, None, _ptr(sf), float(tau), _ptr(acc), flags, _stream()), 'piml_pinnsf_fwd')
    return acc.view(*lead, 2)


class _FusedRowDecoder(torch.autograd.Function):
    """inputs: nbr, then per branch emb (..., 128), decoder w1 (64,128) b1 w2 (64,64) b2, predictor w (2,64) b  (7 tensors).
    outputs per branch: pred (..., 2), decoded (..., 64)."""
    PER = 7

    @staticmethod
    def forward(ctx, nbr, packs, *tensors):
        L = _lib.lib()
        PER = _FusedRowDecoder.PER
        embs = [_gpu_f32('embedding', tensors[PER * b]) for b in range(nbr)]
        wbs = [[_gpu_f32('decoder weight', t.detach()) for t in tensors[PER * b + 1:PER * b + 7]] for b in range(nbr)]
        dev = embs[0].device
        opt = dict(device=dev, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad)
        e2 = [e.reshape(-1, ENCODER_HIDDEN) for e in embs]
        rows = [e.shape[0] for e in e2]
        if packs is not None:
            if packs.sig_dec is None or packs.sig_dec[:nbr] != tuple(_branch_sig(wb) for wb in wbs):
                raise ValueError('fused_row_decoder: `packs` were filled from other weight tensors, or the weights were '
                                 'modified in place since (pinnsf_prepack first)')
            dpack = packs.dpack
        else:
            dpack = torch.empty(nbr, L.piml_decoder_pack_floats(), **opt)
        preds = [torch.empty(r, 2, **opt) for r in rows]
        h1 = [torch.empty(r, 64, **opt) for r in rows]
        d2 = [torch.empty(r, 64, **opt) for r in rows]
        structs = []
        for b in range(nbr):
            B = _dec_branch_struct(e2[b], rows[b], 1, wbs[b], dpack[b], None, h1[b], d2[b])
            B.pred = preds[b].data_ptr()
            structs.append(B)
        with torch.cuda.device(dev):
            fwd = L.piml_rowdecoder_fwd_packed if packs is not None else L.piml_rowdecoder_fwd
            _lib.check(fwd((_lib.DecoderBranch * nbr)(*structs), nbr, _stream()), 'piml_rowdecoder_fwd')
        if need_grad:
            ctx.save_for_backward(*e2, *h1, *d2, *[w for wb in wbs for w in wb], dpack)
        ctx.meta = (nbr, rows, [tuple(e.shape) for e in embs], need_grad)
        ctx.sink = ParamGradSink._active if need_grad else None
        ctx.params = tensors if need_grad else None       # (the Parameter objects: the sink's keys / the deferral's .grad test)
        ctx.set_materialize_grads(False)
        out = []
        for b in range(nbr):
            lead = tuple(embs[b].shape[:-1])
            out += [preds[b].view(*lead, 2), d2[b].view(*lead, 64)]
        return tuple(out)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *gs):
        nbr, rows, eshapes, need_grad = ctx.meta
        PER = _FusedRowDecoder.PER
        grads = [None] * (2 + PER * nbr)
        live = [b for b in range(nbr) if gs[2 * b] is not None or gs[2 * b + 1] is not None]
        if not need_grad or not live:
            return tuple(grads)
        L = _lib.lib()
        sv = list(ctx.saved_tensors)
        take = lambda n: [sv.pop(0) for _ in range(n)]
        e2, h1, d2 = take(nbr), take(nbr), take(nbr)
        wbs = [take(6) for _ in range(nbr)]
        dpack = sv.pop(0)
        dev = e2[0].device
        opt = dict(device=dev, dtype=torch.float32)
        H = ENCODER_HIDDEN
        structs, keep, flats, gembs = [], [], [], []
        sink = ParamGradSink.current(ctx.sink)
        sunk, accumulate = (None, False)
        if sink is not None:
            sunk, accumulate = sink.take([tuple(id(t) for t in ctx.params[PER * b + 1:PER * b + 7]) for b in live],
                                         L.piml_decoder_partial_floats(), opt)
            if sunk is None:
                sink = None
        for i, b in enumerate(live):
            R = rows[b]
            gp = _gpu_f32('g_pred', gs[2 * b]).reshape(R, 2) if gs[2 * b] is not None else _zeros_ro((R, 2), dev)
            gd = _gpu_f32('g_decoded', gs[2 * b + 1]).reshape(R, 64) if gs[2 * b + 1] is not None else None
            gp2, gp1, gemb = torch.empty(R, 64, **opt), torch.empty(R, 64, **opt), torch.empty(R, H, **opt)
            parts = torch.empty(L.piml_rowdecoder_slots(R), L.piml_decoder_partial_floats(), **opt)
            flat = sunk[i] if sink is not None else torch.empty(L.piml_decoder_partial_floats(), **opt)
            B = _dec_branch_struct(e2[b], R, 1, wbs[b], dpack[b], None, h1[b], d2[b], gp2, gp1, gemb, parts, flat)
            B.g_pred_rows, B.g_d2 = gp.data_ptr(), _ptr(gd)
            structs.append(B)
            keep += [gp, gd, gp2, gp1, parts]
            flats.append(flat)
            gembs.append(gemb)
        defer = _defer_slot_sums([ctx.params[PER * b + jx] for b in live for jx in range(1, 7)], sink, dev)
        with torch.cuda.device(dev):
            _lib.check(L.piml_rowdecoder_bwd_acc((_lib.DecoderBranch * len(live))(*structs), len(live),
                                                 int(accumulate) | (_lib.DEFER_SLOT_SUMS if defer else 0), _stream()),
                       'piml_rowdecoder_bwd')
        if defer:
            _defer_keep(keep, flats)
        for i, b in enumerate(live):
            o = 2 + PER * b
            if ctx.needs_input_grad[o]:
                grads[o] = gembs[i].view(eshapes[b])
            flat = flats[i]
            dW1, dW2 = flat[:64 * H].view(64, H), flat[64 * H:64 * H + 4096].view(64, 64)
            dW3 = flat[64 * H + 4096:64 * H + 4096 + 128].view(2, 64)
            rest = flat[64 * H + 4096 + 128:]
            for jx, t in enumerate((dW1, rest[:64], dW2, rest[64:128], dW3, rest[128:130])):
                if ctx.needs_input_grad[o + 1 + jx]:
                    if sink is not None:
                        sink.give(ctx.params[PER * b + 1 + jx], t)
                    else:
                        grads[o + 1 + jx] = t
        return tuple(grads)


def fused_row_decoder(branches, packs=None):
    """Decoder + predictor of the bottleneck PINNSF variants applied to every neighbour ROW on the fused MFMA kernels
    (src/models/model.py:1116-1122: `ped_msgs = self.ped_predictor(self.ped_decoder(ped_embeddings))`).
    branches: 1 or 2 dicts {emb (..., 128), decoder: (w1 (64,128), b1, w2 (64,64), b2), predictor: (w (2,64), b)}.
    Returns per branch (pred (..., 2), decoded (..., 64)); the caller sums pred over the neighbour axis."""
    if not 1 <= len(branches) <= 2:
        raise ValueError('fused_row_decoder: one or two branches')
    flat = []
    for br in branches:
        e, d, p = br['emb'], br['decoder'], br['predictor']
        if not e.is_cuda:
            raise _lib.PimlHipError('fused_row_decoder: expected GPU tensors (piml_amd has no CPU path)')
        if e.shape[-1] != ENCODER_HIDDEN or e.numel() == 0 or \
                [tuple(t.shape) for t in d] != [(64, ENCODER_HIDDEN), (64,), (64, 64), (64,)] or \
                [tuple(t.shape) for t in p] != [(2, 64), (2,)]:
            raise ValueError('fused_row_decoder: unsupported geometry (128 -> 64 -> 64, predictor 64 -> 2)')
        flat += [e, *d, *p]
    out = _FusedRowDecoder.apply(len(branches), packs, *flat)
    return [(out[2 * b], out[2 * b + 1]) for b in range(len(branches))]


class _CollisionHead(torch.autograd.Function):
    """sigmoid(MLP(128, [64, 1])(msgs)) per neighbour row: forward on the fused MFMA kernel, backward (rare: the
    reference trains this head for `pinnsf_bm` only) recomputed with torch ops."""

    @staticmethod
    def forward(ctx, msgs, w1, b1, w2, b2):
        m = _gpu_f32('msgs', msgs)
        rows = m.numel() // ENCODER_HIDDEN
        L = _lib.lib()
        out = torch.empty(m.shape[:-1], device=m.device, dtype=torch.float32)
        packed = torch.empty(L.piml_collision_head_pack_floats(), device=m.device, dtype=torch.float32)
        wb = [_gpu_f32('head weight', t.detach()) for t in (w1, b1, w2, b2)]
        with torch.cuda.device(m.device):
            _lib.check(L.piml_collision_head_fwd(_ptr(m), rows, *[_ptr(t) for t in wb], _ptr(packed), _ptr(out), _stream()),
                       'piml_collision_head_fwd')
        ctx.save_for_backward(m, *wb)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        m, w1, b1, w2, b2 = ctx.saved_tensors
        with torch.enable_grad():
            ins = [t.detach().requires_grad_(True) for t in (m, w1, b1, w2, b2)]
            y = torch.sigmoid(torch.relu(ins[0] @ ins[1].t() + ins[2]) @ ins[3].t() + ins[4]).squeeze(-1)
            grads = torch.autograd.grad(y, ins, g, allow_unused=True)
        return tuple(gr if need else None for gr, need in zip(grads, ctx.needs_input_grad))


class _CollisionHead64(torch.autograd.Function):
    """sigmoid(Linear(64, 1)(relu(Linear(64, 64)(x)))) per row on the f32 matrix cores, forward and backward
    (piml_head64_fwd / bwd): the collision head of `pinnsf_bm` on the row decoder's output."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        L = _lib.lib()
        x2 = _gpu_f32('x', x).reshape(-1, 64)
        wb = [_gpu_f32('head weight', t.detach()) for t in (w1, b1, w2, b2)]
        rows = x2.shape[0]
        need_grad = any(ctx.needs_input_grad)
        opt = dict(device=x2.device, dtype=torch.float32)
        out = torch.empty(rows, **opt)
        hidden = torch.empty(rows, 64, **opt) if need_grad else None
        H = _lib.Head64()
        H.x, H.rows = x2.data_ptr(), rows
        H.w1, H.b1, H.w2, H.b2 = [t.data_ptr() for t in wb]
        H.hidden, H.out = _ptr(hidden), out.data_ptr()
        import ctypes
        with torch.cuda.device(x2.device):
            _lib.check(L.piml_head64_fwd(ctypes.byref(H), _stream()), 'piml_head64_fwd')
        if need_grad:
            ctx.save_for_backward(x2, hidden, out, *wb)
        ctx.x_shape = tuple(x.shape)
        ctx.sink = ParamGradSink._active if need_grad else None
        ctx.params = (w1, b1, w2, b2) if need_grad else None     # (the Parameter objects: the sink's keys / the deferral's .grad test)
        ctx.set_materialize_grads(False)
        return out.view(x.shape[:-1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return (None,) * 5
        import ctypes
        L = _lib.lib()
        x2, hidden, out, w1, b1, w2, b2 = ctx.saved_tensors
        rows = x2.shape[0]
        opt = dict(device=x2.device, dtype=torch.float32)
        g = _gpu_f32('g_out', g).reshape(rows)
        gx = torch.empty(rows, 64, **opt) if ctx.needs_input_grad[0] else None
        part = L.piml_head64_partial_floats()
        partials = torch.empty(L.piml_head64_slots(rows), part, **opt)
        sink = ParamGradSink.current(ctx.sink)
        sunk, accumulate = (None, False)
        if sink is not None:
            sunk, accumulate = sink.take([tuple(id(t) for t in ctx.params)], part, opt)
            if sunk is None:
                sink = None
        grads = sunk[0] if sink is not None else torch.empty(part, **opt)
        H = _lib.Head64()
        H.x, H.rows = x2.data_ptr(), rows
        H.w1, H.b1, H.w2, H.b2 = [t.data_ptr() for t in (w1, b1, w2, b2)]
        H.hidden, H.out, H.g_out = hidden.data_ptr(), out.data_ptr(), g.data_ptr()
        H.g_x, H.partials, H.grads = _ptr(gx), partials.data_ptr(), grads.data_ptr()
        # (inside a deferred_slot_sums block the slot sums join the row decoder's and the encoders' in the relfeat backward's launch)
        defer = ctx.params is not None and _defer_slot_sums(list(ctx.params), sink, x2.device)
        with torch.cuda.device(x2.device):
            _lib.check(L.piml_head64_bwd_acc(ctypes.byref(H), int(accumulate) | (_lib.DEFER_SLOT_SUMS if defer else 0), _stream()),
                       'piml_head64_bwd')
        if defer:
            _defer_keep(partials, g, grads)
        need = ctx.needs_input_grad
        views = (grads[:4096].view(64, 64), grads[4096:4160], grads[4160:4224].view(1, 64), grads[4224:4225])
        if sink is not None:
            for j, v in enumerate(views):
                if need[1 + j]:
                    sink.give(ctx.params[j], v)
            return (gx.view(ctx.x_shape) if gx is not None else None, None, None, None, None)
        return (gx.view(ctx.x_shape) if gx is not None else None,) + tuple(v if need[1 + j] else None for j, v in enumerate(views))


class _Corrector(torch.autograd.Function):
    """The corrector of `pinnsf_res` (piml_corrector_fwd / bwd, piml_amd/csrc/corrector.hip; src/models/model.py:1016-1020,
    :1050-1052): enc (..., N, k, 128), scale, keep_bits, then wa ba wb bb (attn_pooling.get_weights) and wc bc wd bd
    (the 128 -> 64 -> 2 tail) -> (..., N, 2)."""

    @staticmethod
    def forward(ctx, enc, scale, keep_bits, *weights):
        import ctypes
        L = _lib.lib()
        e2 = _gpu_f32('enc', enc)
        k = e2.shape[-2]
        rows = e2.numel() // 128
        agents = rows // k
        e2 = e2.reshape(rows, 128)
        wb = [_gpu_f32('corrector weight', t.detach()) for t in weights]
        need_grad = any(ctx.needs_input_grad)
        opt = dict(device=e2.device, dtype=torch.float32)
        hid = torch.empty(rows, 128, **opt) if need_grad else None
        score, attn = torch.empty(rows, **opt), torch.empty(rows, **opt)
        pooled, chid, out = torch.empty(agents, 128, **opt), torch.empty(agents, 64, **opt), torch.empty(agents, 2, **opt)
        if keep_bits is not None and (keep_bits.dtype != torch.int32 or tuple(keep_bits.shape) != (rows, 4) or not keep_bits.is_contiguous()):
            raise ValueError('fused_corrector: keep_bits must be contiguous int32 (rows, 4) (ops.dropout_keep_bits / pack_keep_bits)')
        C = _lib.Corrector()
        C.agents, C.k, C.scale = agents, k, float(scale)
        C.enc, C.keep_bits = e2.data_ptr(), _ptr(keep_bits)
        C.wa, C.ba, C.wb, C.bb, C.wc, C.bc, C.wd, C.bd = [t.data_ptr() for t in wb]
        C.hid, C.score, C.attn, C.pooled, C.chid, C.out = _ptr(hid), score.data_ptr(), attn.data_ptr(), pooled.data_ptr(), chid.data_ptr(), out.data_ptr()
        with torch.cuda.device(e2.device):
            _lib.check(L.piml_corrector_fwd(ctypes.byref(C), _stream()), 'piml_corrector_fwd')
        if need_grad:
            ctx.save_for_backward(e2, hid, score, attn, pooled, chid, *wb)
        ctx.meta = (tuple(enc.shape), agents, k, float(scale))
        ctx.keep = keep_bits
        ctx.sink = ParamGradSink._active if need_grad else None
        ctx.params = weights if ctx.sink is not None else None
        ctx.set_materialize_grads(False)
        return out.view(*enc.shape[:-2], 2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        if g is None:
            return (None,) * 11
        import ctypes
        L = _lib.lib()
        e2, hid, score, attn, pooled, chid, *wb = ctx.saved_tensors
        enc_shape, agents, k, scale = ctx.meta
        rows = agents * k
        opt = dict(device=e2.device, dtype=torch.float32)
        g = _gpu_f32('g_out', g).reshape(agents, 2)
        g_enc = torch.empty(rows, 128, **opt) if ctx.needs_input_grad[0] else None
        g_pooled, g_score, g_chid = torch.empty(agents, 128, **opt), torch.empty(rows, **opt), torch.empty(agents, 64, **opt)
        pa, pb = L.piml_corrector_partial_floats(0), L.piml_corrector_partial_floats(1)
        parts_a = torch.empty(L.piml_corrector_slots(0, agents, k), pa, **opt)
        parts_b = torch.empty(L.piml_corrector_slots(1, agents, k), pb, **opt)
        sink = ParamGradSink.current(ctx.sink)
        sunk, accumulate = (None, False)
        if sink is not None:
            sunk, accumulate = sink.take([tuple(id(t) for t in ctx.params)], pa + pb, opt)
            if sunk is None:
                sink = None
        grads = sunk[0] if sink is not None else torch.empty(pa + pb, **opt)
        C = _lib.Corrector()
        C.agents, C.k, C.scale = agents, k, scale
        C.enc, C.keep_bits = e2.data_ptr(), _ptr(ctx.keep)
        C.wa, C.ba, C.wb, C.bb, C.wc, C.bc, C.wd, C.bd = [t.data_ptr() for t in wb]
        C.hid, C.score, C.attn, C.pooled, C.chid, C.out = hid.data_ptr(), score.data_ptr(), attn.data_ptr(), pooled.data_ptr(), chid.data_ptr(), None
        C.g_out, C.g_pooled, C.g_score, C.g_chid, C.g_enc = g.data_ptr(), g_pooled.data_ptr(), g_score.data_ptr(), g_chid.data_ptr(), _ptr(g_enc)
        C.partials_a, C.partials_b, C.grads = parts_a.data_ptr(), parts_b.data_ptr(), grads.data_ptr()
        with torch.cuda.device(e2.device):
            _lib.check(L.piml_corrector_bwd(ctypes.byref(C), int(accumulate), _stream()), 'piml_corrector_bwd')
        A, B = grads[:pa], grads[pa:]
        views = (A[:16384].view(128, 128), A[16384:16512], A[16512:16640].view(1, 128), A[16640:16641],
                 B[:8192].view(64, 128), B[8192:8256], B[8256:8384].view(2, 64), B[8384:8386])
        need = ctx.needs_input_grad
        ge = g_enc.view(enc_shape) if g_enc is not None else None
        if sink is not None:
            for j, v in enumerate(views):
                if need[3 + j]:
                    sink.give(ctx.params[j], v)
            return (ge, None, None) + (None,) * 8
        return (ge, None, None) + tuple(v if need[3 + j] else None for j, v in enumerate(views))


def fused_corrector(enc, scale, keep_bits, get_weights, tail):
    """`pinnsf_res`'s corrector on the hand-written kernels: enc (..., N, k, 128) float32 = the pedestrian encoder's raw
    output; (scale, keep_bits) = what corrector[0] does to it (ResDNN.fused_spec: keep * scale * enc, keep_bits None in
    eval mode); get_weights = (wa (128,128), ba, wb (1,128), bb) of attn_pooling.get_weights; tail = (wc (64,128), bc,
    wd (2,64), bd) of corrector[2].  Returns the residual acceleration (..., N, 2)."""
    if not enc.is_cuda:
        raise _lib.PimlHipError('fused_corrector: expected GPU tensors (piml_amd has no CPU path)')
    shapes = [tuple(t.shape) for t in (*get_weights, *tail)]
    if enc.dim() < 3 or enc.shape[-1] != 128 or not 1 <= enc.shape[-2] <= 64 or enc.numel() == 0 or \
            shapes != [(128, 128), (128,), (1, 128), (1,), (64, 128), (64,), (2, 64), (2,)]:
        raise ValueError('fused_corrector: unsupported geometry (enc (..., N, k <= 64, 128); MLP(128, [128, 1]); MLP(128, [64, 2]))')
    return _Corrector.apply(enc, float(scale), keep_bits, *get_weights, *tail)


def collision_head64(x, w1, b1, w2, b2):
    """sigmoid(Linear(64, 1)(relu(Linear(64, 64)(x)))) for x (..., 64) -> (...,): `pinnsf_bm`'s collision head on the decoder
    output of every neighbour row (src/models/model.py:1183, 1214-1215), forward and backward on hand-written kernels."""
    if not x.is_cuda:
        raise _lib.PimlHipError('collision_head64: expected GPU tensors (piml_amd has no CPU path)')
    if x.shape[-1] != 64 or tuple(w1.shape) != (64, 64) or tuple(w2.shape) != (1, 64) or tuple(b1.shape) != (64,) \
            or tuple(b2.shape) != (1,) or x.numel() == 0:
        raise ValueError('collision_head64: x (..., 64), Linear(64, 64), Linear(64, 1) expected')
    return _CollisionHead64.apply(x, w1, b1, w2, b2)


def collision_head(msgs, w1, b1, w2, b2):
    """sigmoid(Linear(64, 1)(relu(Linear(128, 64)(msgs)))) for msgs (..., 128) -> (...,)   (model.py:1296-1300)."""
    if not msgs.is_cuda:
        raise _lib.PimlHipError('collision_head: expected GPU tensors (piml_amd has no CPU path)')
    if msgs.shape[-1] != ENCODER_HIDDEN or tuple(w1.shape) != (64, ENCODER_HIDDEN) or tuple(w2.shape) != (1, 64) \
            or tuple(b1.shape) != (64,) or tuple(b2.shape) != (1,):
        raise ValueError('collision_head: msgs (..., 128), Linear(128, 64), Linear(64, 1) expected')
    return _CollisionHead.apply(msgs, w1, b1, w2, b2)
