"""torch-facing operators over the C ABI (include/piml_hip.h).

PyTorch is plumbing here: it owns device memory and the stream, and autograd.Function glues
the forward/backward kernels into the reference's training loops.  Every operator requires
float32 tensors on the GPU and raises otherwise (no CPU fallback).
"""
import math

import torch

from . import _lib

MAX_TOPK = 32


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


class _RelativeFeatures(torch.autograd.Function):
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
        N, M = p.shape[-2], o.shape[0]
        lead = p.shape[:-2]
        C = p.numel() // max(N * 2, 1)
        if focal_count is None:
            focal_begin, focal_count = 0, N
        kpe, koe = min(kp, N), min(ko, M)
        hd = None if heading is None else _gpu_f32('heading', heading)
        ped_feat = torch.empty(*lead, focal_count, kpe, 6, device=p.device, dtype=torch.float32)
        obs_feat = torch.empty(*lead, focal_count, koe, 6, device=p.device, dtype=torch.float32)
        dest_feat = torch.empty(*lead, focal_count, 2, device=p.device, dtype=torch.float32)
        ped_idx = torch.empty(*lead, focal_count, kpe, device=p.device, dtype=torch.int32)
        obs_idx = torch.empty(*lead, focal_count, koe, device=p.device, dtype=torch.int32)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_relfeat_fwd(
                _ptr(p), _ptr(hd), _ptr(v), _ptr(a), _ptr(d), _ptr(o), C, N, M, focal_begin, focal_count,
                kp, ko, cos_p, cos_o, dthr_p, dthr_o, _ptr(ped_feat), _ptr(obs_feat), _ptr(dest_feat),
                _ptr(ped_idx), _ptr(obs_idx), _stream()), 'piml_relfeat_fwd')
        ctx.save_for_backward(ped_idx, obs_idx, p, d)
        ctx.geom = (C, N, focal_begin, focal_count, kpe, koe, tuple(p.shape))
        ctx.mark_non_differentiable(ped_idx, obs_idx)
        return ped_feat, obs_feat, dest_feat, ped_idx, obs_idx

    @staticmethod
    def backward(ctx, g_ped, g_obs, g_dest, _gi, _go):
        ped_idx, obs_idx, p, d = ctx.saved_tensors
        C, N, f0, fcnt, kpe, koe, shape = ctx.geom
        lead = shape[:-2]

        def dense(g, like_shape):
            return torch.zeros(like_shape, device=p.device, dtype=torch.float32) if g is None \
                else _gpu_f32('grad', g)
        g_ped = dense(g_ped, (*lead, fcnt, kpe, 6))
        g_obs = dense(g_obs, (*lead, fcnt, koe, 6))
        g_dest = dense(g_dest, (*lead, fcnt, 2))
        g_state = torch.zeros(*lead, N, 6, device=p.device, dtype=torch.float32)
        g_d_rows = torch.empty(*lead, fcnt, 2, device=p.device, dtype=torch.float32)
        with torch.cuda.device(p.device):
            _lib.check(_lib.lib().piml_relfeat_bwd(
                _ptr(g_ped), _ptr(g_obs), _ptr(g_dest), _ptr(ped_idx), _ptr(obs_idx), _ptr(p), _ptr(d),
                C, N, f0, fcnt, kpe, koe, _ptr(g_state), _ptr(g_d_rows), _stream()), 'piml_relfeat_bwd')
        if fcnt == N:
            g_destination = g_d_rows
        else:
            g_destination = torch.zeros(shape, device=p.device, dtype=torch.float32)
            g_destination[..., f0:f0 + fcnt, :] = g_d_rows
        return (g_state[..., 0:2], g_state[..., 2:4], g_state[..., 4:6], g_destination,
                None, None, None, None, None, None, None, None, None, None)


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
