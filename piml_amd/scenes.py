"""Synthetic "GC scene" generator (numpy only, deterministic from a PCG64 seed).

Geometry follows the reference's Grand-Central scenario
(`/root/reference/src/data/scenarios.py:313-366`): a 30 m x 35 m hall, a circular
obstacle of radius 2.75 m centred at (13.52, 10.71) sampled with 100 points, and the
hall's wall polyline.  The reference samples the wall every 0.05 m (3994 points); here
the wall is resampled uniformly in arc length so that the total obstacle count hits the
requested ``M``.  Agents are placed uniformly in the hall
(rejecting the disc), destinations are drawn on the entry segments, desired speed is
``max(0.7, 1.34 + sqrt(0.26) z)`` and the initial velocity is ``v0 * unit(dest - p)``
(the reference spawns agents with v = 0, which blinds every agent because a zero heading
fails the view-cone test; moving agents are required to exercise the path).

Everything returned is float32 and C-contiguous.  Absent agents (``nan_frac``) carry
NaN positions / destinations exactly like `RawData.load_trajectory_data`
(`/root/reference/src/data/data.py:141-143`) and zero velocity / acceleration.
"""
import numpy as np

HALL_W = 30.0
HALL_L = 35.0
DISC_C = (13.52, 10.71)
DISC_R = 2.75

_WALL_NODES = np.array([
    [0, 0], [0, 5.63], [-5, 5.63], [-5, 16.01], [0, 16.01], [0, 35],
    [0, 40], [5.93, 40], [5.93, 35], [21.43, 35], [21.43, 40], [30, 40], [30, 35],
    [35, 35], [35, 29.48], [30, 29.48], [30, 25.62], [35, 25.62], [35, 18.99],
    [30, 18.99], [30, 14.79], [35, 14.79], [35, 7.07], [30, 7.07], [30, 0],
    [30, -5], [0, -5], [0, 0]], dtype=np.float64)

# entry segments (x0, y0, x1, y1), scenarios.py:341-349
_ENTRIES = np.array([
    [0.0, 6.63, 0.0, 15.01],
    [1.0, 35.0, 4.93, 35.0],
    [22.43, 35.0, 29.0, 35.0],
    [30.0, 30.48, 30.0, 34.0],
    [30.0, 19.99, 30.0, 24.62],
    [30.0, 8.07, 30.0, 13.79],
    [1.0, 0.0, 29.0, 0.0]], dtype=np.float64)


def gc_obstacles(M):
    """Obstacle points (M, 2) float32.

    M == 0 returns the reference's "no obstacle" placeholder
    (`/root/reference/src/data/data.py:102-103`); M <= 100 samples the disc only;
    larger M adds the wall polyline resampled to M - 100 points.
    """
    if M == 0:
        return np.array([[1e4, 1e4], [1e4 + 1, 1e4 + 1]], dtype=np.float32)
    n_disc = min(M, 100)
    th = np.linspace(0.0, 2.0 * np.pi, n_disc)
    disc = np.stack((DISC_R * np.cos(th) + DISC_C[0], DISC_R * np.sin(th) + DISC_C[1]), axis=1)
    if M <= 100:
        return disc.astype(np.float32)
    n_wall = M - n_disc
    seg = np.diff(_WALL_NODES, axis=0)
    seg_len = np.linalg.norm(seg, axis=1)
    cum = np.concatenate(([0.0], np.cumsum(seg_len)))
    s = np.linspace(0.0, cum[-1], n_wall, endpoint=False)
    k = np.clip(np.searchsorted(cum, s, side='right') - 1, 0, len(seg_len) - 1)
    frac = (s - cum[k]) / seg_len[k]
    wall = _WALL_NODES[k] + seg[k] * frac[:, None]
    return np.concatenate((wall, disc), axis=0).astype(np.float32)


def synthetic_gc_scene(N, M, seed=0, nan_frac=0.02, channels=None):
    """Return a dict of float32 arrays describing one synthetic GC scene.

    keys: position, velocity, acceleration, destination (N,2) [or (C,N,2) when
    ``channels`` is given], desired_speed (N,1)/(C,N,1), obstacles (M',2).
    """
    rng = np.random.default_rng(seed)
    lead = (N,) if channels is None else (channels, N)
    n_tot = int(np.prod(lead))

    pos = np.empty((n_tot, 2), dtype=np.float64)
    filled = 0
    while filled < n_tot:
        cand = rng.random((n_tot - filled + 64, 2)) * (HALL_W, HALL_L)
        ok = np.hypot(cand[:, 0] - DISC_C[0], cand[:, 1] - DISC_C[1]) > DISC_R + 0.3
        cand = cand[ok][:n_tot - filled]
        pos[filled:filled + len(cand)] = cand
        filled += len(cand)

    e = rng.integers(0, len(_ENTRIES), size=n_tot)
    u = rng.random(n_tot)
    ent = _ENTRIES[e]
    dest = np.stack((ent[:, 0] + (ent[:, 2] - ent[:, 0]) * u,
                     ent[:, 1] + (ent[:, 3] - ent[:, 1]) * u), axis=1)
    dest += rng.random((n_tot, 2)) * 0.8

    v0 = np.maximum(0.7, 1.34 + np.sqrt(0.26) * rng.standard_normal(n_tot))
    d = dest - pos
    dn = np.linalg.norm(d, axis=1, keepdims=True)
    vel = v0[:, None] * d / np.maximum(dn, 1e-9)
    acc = np.zeros_like(pos)

    absent = rng.random(n_tot) < nan_frac
    pos[absent] = np.nan
    dest[absent] = np.nan
    vel[absent] = 0.0

    def shp(x, w):
        return np.ascontiguousarray(x.reshape(*lead, w).astype(np.float32))

    return {
        'position': shp(pos, 2), 'velocity': shp(vel, 2), 'acceleration': shp(acc, 2),
        'destination': shp(dest, 2), 'desired_speed': shp(v0, 1),
        'obstacles': gc_obstacles(M),
    }


def pair_count(N, M, kind='pinsf'):
    """Pair evaluations per step: N*(N+M) for the PINSF feature path,
    N*N for MLAPM.step."""
    return N * (N + M) if kind == 'pinsf' else N * N


def algorithmic_bytes(N, M, kind='pinsf'):
    """Operand-stream byte model, the roofline contract figure."""
    if kind == 'pinsf':
        return N * (24 * N + 8 * M) + 488 * N
    return 16 * N * N + 36 * N


def synthetic_rollout_data(N, M, T, dev, seed=0):
    """A RawData-shaped clip of T frames of the synthetic scene for `BaseSimulator.get_multiple_rollouts` (frame 0 is the
    scene, the rest is what the rollout overwrites): the container fields of src/data/data.py's RawData that the rollout of
    src/models/simulators.py:552-657 reads.  Used by bench.py (`secondary.simulated_steps`) and tools/time_rollout.py."""
    import types
    import torch
    from .pedestrians import Pedestrians
    sc = synthetic_gc_scene(N, M, seed=seed)
    t = lambda x: torch.tensor(x, device=dev)
    rep = lambda x: t(x).unsqueeze(0).repeat(T, *([1] * x.ndim)).contiguous()
    d = types.SimpleNamespace()
    d.position, d.velocity, d.acceleration, d.destination = [rep(sc[k]) for k in ('position', 'velocity', 'acceleration', 'destination')]
    d.velocity = torch.nan_to_num(d.velocity)
    d.obstacles = t(sc['obstacles'])
    far = sc['destination'] + (sc['destination'] - np.nan_to_num(sc['position'])) * 100
    d.waypoints = torch.stack((t(sc['destination']), t(far.astype(np.float32))))
    d.dest_num = torch.full((N,), 2, device=dev, dtype=torch.long)
    d.dest_idx = torch.zeros(T, N, device=dev, dtype=torch.long)
    present = (~torch.isnan(d.position[..., 0])).float()
    d.mask_p, d.mask_p_pred = present, present.clone()
    d.num_frames, d.time_unit, d.meta_data = T, 0.08, None
    pf, of, df = Pedestrians().get_relative_features(d.position[:1].clone(), d.velocity[:1].clone(), d.acceleration[:1].clone(),
                                                     d.destination[:1].clone(), d.obstacles, 6, 90, 4, 10, 90, 4)
    d.ped_features, d.obs_features = pf.repeat(T, 1, 1, 1), of.repeat(T, 1, 1, 1)
    d.self_features = torch.cat((df, d.velocity[:1], d.acceleration[:1], t(sc['desired_speed']).unsqueeze(0)), -1).repeat(T, 1, 1)
    return d
