"""numpy front-end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Loads oracle/libpiml_oracle.so (built by oracle/Makefile from oracle/piml_oracle.c, a
plain-C restatement of the reference's pairwise hot path).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; nothing
under piml_amd/ does.  Parity status: pinned against golden vectors captured from the real
reference (tests/golden/make_golden.py, tests/test_oracle_golden.py).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libpiml_oracle.so')
_lib = None

_f = ctypes.c_float
_i = ctypes.c_int
_p = ctypes.c_void_p
_z = ctypes.c_size_t


def build(force=False):
    src = os.path.join(_HERE, 'piml_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libpiml_oracle.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_num_threads.restype = _i
    return _lib


def num_threads():
    return int(lib().oracle_num_threads())


def _c(x, dtype=np.float32):
    return np.ascontiguousarray(x, dtype=dtype)


def _ptr(x):
    return x.ctypes.data_as(_p)


def cos_threshold(angle_deg):
    """math.cos(3.14 * angle / 180) as the reference computes it (data.py:442-443, quirk Q1),
    rounded to float32 the way torch compares a float32 tensor with a python scalar."""
    return np.float32(math.cos(3.14 * angle_deg / 180))


def heading(velocity):
    """velocity (..., T, N, 2) -> unit heading, temporal zero-fill (data.py:350-395)."""
    v = _c(velocity)
    T, N = v.shape[-3], v.shape[-2]
    C = int(np.prod(v.shape[:-3], dtype=np.int64)) if v.ndim > 3 else 1
    out = np.empty_like(v)
    lib().oracle_heading(_ptr(v), _i(C), _i(T), _i(N), _ptr(out))
    return out


def relfeat_fwd(position, velocity, acceleration, destination, obstacles, topk_ped=6,
                sight_angle_ped=90, dist_threshold_ped=4, topk_obs=10, sight_angle_obs=90,
                dist_threshold_obs=4, return_index=False):
    """Restates Pedestrians.get_relative_features (data.py:466-512) for inputs shaped
    (*c, t, N, 2).  velocity / acceleration NaNs are treated as 0 (the reference zeroes
    them in place first); inputs are not modified."""
    p = _c(position)
    v = np.nan_to_num(_c(velocity), nan=0.0, posinf=np.inf, neginf=-np.inf)
    a = np.nan_to_num(_c(acceleration), nan=0.0, posinf=np.inf, neginf=-np.inf)
    d = _c(destination)
    o = _c(obstacles).reshape(-1, 2)
    lead = p.shape[:-1]
    N = p.shape[-2]
    M = o.shape[0]
    C = int(np.prod(lead[:-1], dtype=np.int64))
    hd = heading(v)
    kpe, koe = min(topk_ped, N), min(topk_obs, M)
    ped_feat = np.zeros(lead + (kpe, 6), np.float32)
    obs_feat = np.zeros(lead + (koe, 6), np.float32)
    dest_feat = np.zeros(lead + (2,), np.float32)
    ped_idx = np.full(lead + (kpe,), -1, np.int32)
    obs_idx = np.full(lead + (koe,), -1, np.int32)
    ped_dist = np.full(lead + (kpe,), np.inf, np.float32)
    obs_dist = np.full(lead + (koe,), np.inf, np.float32)
    lib().oracle_relfeat_fwd(
        _ptr(p), _ptr(hd), _ptr(v), _ptr(a), _ptr(d), _ptr(o), _i(C), _i(N), _i(M),
        _i(topk_ped), _i(topk_obs), _f(cos_threshold(sight_angle_ped)),
        _f(cos_threshold(sight_angle_obs)), _f(dist_threshold_ped), _f(dist_threshold_obs),
        _ptr(ped_feat), _ptr(obs_feat), _ptr(dest_feat), _ptr(ped_idx), _ptr(obs_idx),
        _ptr(ped_dist), _ptr(obs_dist))
    if return_index:
        return ped_feat, obs_feat, dest_feat, ped_idx, obs_idx, ped_dist, obs_dist
    return ped_feat, obs_feat, dest_feat


def relfeat_bwd(g_ped, g_obs, g_dest, ped_idx, obs_idx, position, destination):
    """Gradient of relfeat_fwd w.r.t. (position, velocity, acceleration, destination)."""
    p = _c(position)
    d = _c(destination)
    lead = p.shape[:-1]
    N = p.shape[-2]
    C = int(np.prod(lead[:-1], dtype=np.int64))
    kpe, koe = ped_idx.shape[-1], obs_idx.shape[-1]
    g_state = np.zeros(lead + (6,), np.float32)
    gd = np.zeros(lead + (2,), np.float32)
    lib().oracle_relfeat_bwd(_ptr(_c(g_ped)), _ptr(_c(g_obs)), _ptr(_c(g_dest)),
                             _ptr(_c(ped_idx, np.int32)), _ptr(_c(obs_idx, np.int32)),
                             _ptr(p), _ptr(d), _i(C), _i(N), _i(kpe), _i(koe),
                             _ptr(g_state), _ptr(gd))
    return g_state[..., 0:2], g_state[..., 2:4], g_state[..., 4:6], gd


def collision_detection(position, threshold, real_position=None):
    """Restates Pedestrians.collision_detection (data.py:537-601) for 3-D / 4-D input."""
    p = _c(position)
    N = p.shape[-2]
    S = int(np.prod(p.shape[:-2], dtype=np.int64))
    coll = np.empty(p.shape[:-2] + (N, N), np.float32)
    lib().oracle_collision_pairs(_ptr(p), _i(S), _i(N), _f(threshold), _ptr(coll))
    if real_position is not None:
        rp = _c(real_position)
        assert rp.ndim == 3
        base = np.empty(rp.shape[:-2] + (N, N), np.float32)
        lib().oracle_collision_pairs_raw(_ptr(rp), _i(rp.shape[0]), _i(N), _f(threshold), _ptr(base))
        fr = (base.sum(0) <= 25).astype(np.float32)
        coll *= fr
    elif p.ndim == 3:
        lib().oracle_collision_friends3(_ptr(coll), _ptr(coll.copy()), _i(S), _i(N))
    elif p.ndim == 4:
        lib().oracle_collision_friends4(_ptr(coll), _i(p.shape[0]), _i(p.shape[1]), _i(N))
    return coll


def collision_label(ped_features):
    f = _c(ped_features)
    R = int(np.prod(f.shape[:-1], dtype=np.int64))
    out = np.empty(f.shape[:-1], np.float32)
    lib().oracle_collision_label(_ptr(f), _z(R), _i(f.shape[-1]), _ptr(out))
    return out


MLAPM_VARIANTS = {'raw': 0, 'GC': 1, 'UCY': 2}


def mlapm_step(position, velocity, desired_speed, destination, dt, radius=0.3, version='GC',
               tau=0.5, A=0.0, B=0.0, C=0.0, D=0.0, theta=0.0, return_force=False):
    p, v, d = _c(position), _c(velocity), _c(destination)
    v0 = _c(desired_speed).reshape(-1)
    N = p.shape[0]
    assert v0.shape[0] == N
    act = np.empty((N, 2), np.float32)
    frc = np.empty((N, 2), np.float32)
    lib().oracle_mlapm_step(_ptr(p), _ptr(v), _ptr(v0), _ptr(d), _i(N),
                            _i(MLAPM_VARIANTS[version]), _f(tau), _f(A), _f(B), _f(C), _f(D),
                            _f(theta), _f(radius), _f(dt), _ptr(act), _ptr(frc))
    return (act, frc) if return_force else act


_CALC_ACC = {  # utils.py:44-81
    ('v0', 'gc1560'): (8.75, -2.5, 0, 0, 0), ('v0', 'gc2344'): (8.75, -2.5, 0, 0, 0),
    ('v0', 'ucy'): (10.67, -3.33, 0, 0, 0),
    ('v1', 'gc1560'): (8.75, -2.5, 0, 0, 0), ('v1', 'gc2344'): (8.75, -2.5, 0, 0, 0),
    ('v1', 'ucy'): (10.67, -3.33, 0, 0, 0),
    ('v2', 'gc2344'): (9.00, -2.75, 0.06, -0.3, 10 * 3.1415 / 180),
}


def calc_acceleration(relative_data, equation_version='v0', dataset='gc1560', eps=1e-6):
    r = _c(relative_data)
    R = int(np.prod(r.shape[:-1], dtype=np.int64))
    A, B, Cc, D, th = _CALC_ACC[(equation_version, dataset)]
    out = np.empty(r.shape[:-1] + (2,), np.float32)
    lib().oracle_calc_acceleration(_ptr(r), _z(R), _i(r.shape[-1]), _i(int(equation_version[1])),
                                   _f(A), _f(B), _f(Cc), _f(D), _f(th), _f(eps), _ptr(out))
    return out


def collision_post_correction(predictions, ped_features, velocity, collision_threshold=0.5, time_unit=0.08):
    """The hand-written collision handling that closes PINNSF_polar_bottleneck_collision.forward
    (src/models/model.py:1383-1444), restated in float32 numpy.  predictions (..., N, 2), ped_features
    (..., N, k, >= 4) = (p_j - p_i, v_j - v_i, ...), velocity (..., N, 2) = v_i (self_features[..., 2:4]).
    Returns the corrected predictions.  (Deviation: the reference `.squeeze()`s the gathered neighbour, which
    also drops a size-1 agent / slice axis; here only the neighbour axis is dropped.)"""
    f32 = np.float32
    P = np.asarray(predictions, f32).copy()
    ped = np.asarray(ped_features, f32)
    vi = np.asarray(velocity, f32)
    dt = f32(time_unit)
    R = f32(collision_threshold + 1.34 * 2 * time_unit)                        # :1385
    pji = np.where(np.isnan(ped[..., :2]), f32(0), ped[..., :2]).astype(f32)    # :1387-1390
    norm = np.sqrt(pji[..., 1] * pji[..., 1] + pji[..., 0] * pji[..., 0], dtype=f32) + f32(1e-6)   # :1391-1392
    nji = pji / norm[..., None]                                                  # :1393
    vji = ped[..., 2:4]
    vik = np.broadcast_to(vi[..., None, :], vji.shape)                           # :1396
    vj = vji + vik
    coll = ((R >= norm) & (norm > f32(1e-4))).astype(f32)                        # :1399
    with np.errstate(invalid='ignore'):
        inter = ((vik * pji).sum(-1, dtype=f32) * (vj * (-pji)).sum(-1, dtype=f32)).astype(f32)   # :1404
    inter = np.where(np.isnan(inter), f32(0), inter)
    inter = (inter > 0).astype(f32)                                              # :1405-1408
    enc, chase = coll * inter, coll * (f32(1) - inter)                           # :1409-1410

    def nearest(flag):
        d = norm * flag
        d = np.where(d < f32(1e-4), d + f32(100), d)                             # :1414-1415
        idx = np.argmin(d, axis=-1)                                              # first minimum
        take = lambda x: np.take_along_axis(x, idx[..., None, None].repeat(x.shape[-1], -1), axis=-2)[..., 0, :]
        return take(nji), take(vji)

    # step 2: head-on encounters (:1412-1425)
    n_c, _ = nearest(enc)
    m = (enc.sum(-1, keepdims=True) > 0).astype(f32)
    a_c = -(vi * n_c).sum(-1, keepdims=True, dtype=f32) * n_c / dt * m
    P_ = P * m
    s = (P_ * n_c).sum(-1, keepdims=True, dtype=f32)
    s = s * (s > 0)
    P = P + ((P_ - s * n_c) + a_c)
    # step 3: chasing (:1427-1442)
    n_c, v_c = nearest(chase)
    m = (chase.sum(-1, keepdims=True) > 0).astype(f32)
    q = (v_c * n_c).sum(-1, keepdims=True, dtype=f32)
    a_ = q * (q < 0) * n_c / dt * m
    P_ = P * m
    s = (P_ * n_c).sum(-1, keepdims=True, dtype=f32)
    s = s * (s > 0) * (q < 0)
    P = P + ((P_ - s * n_c) + a_)
    return P.astype(f32)
