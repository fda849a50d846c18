"""Reference-DATAFLOW restatement of the per-step feature path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Same arithmetic as oracle/piml_oracle.c, but organised the way the reference computes it
(src/data/data.py:397-512): dense (N, M, d) relative tensors materialised with torch ops, a full
`torch.sort` along the source axis, `gather` of the k nearest rows, threshold zeroing.  It exists so that
bench.py's cpu_baseline can report, on the GPU box's own host cores, what the reference's way of doing one
step costs (SURVEY.md section 8d asks for both forms); tests/test_oracle_dataflow.py pins it on the oracle.
Only tests/ and bench.py's cpu_baseline leg may import it.  Single frame (t == 1), like the per-step calls of
the rollout loops: the heading is v / |v| (0.1 in place of a zero norm), src/data/data.py:390-394.
"""
import math

import torch


def _relative(a, b):
    """rel[i, j, :] = b[j, :] - a[i, :] as a materialised (N, M, d) tensor   (data.py:397-414)."""
    n, m = a.shape[0], b.shape[0]
    return (b.unsqueeze(0).expand(n, m, -1) - a.unsqueeze(1).expand(n, m, -1)).contiguous()


def _nearest_in_sight(position, objects, heading, k, angle):
    """(distance, index) of the k nearest objects inside the view cone, by a full sort   (data.py:416-447)."""
    rel = _relative(position, objects)
    rel[rel.isnan()] = float('inf')
    dist = torch.norm(rel, p=2, dim=-1)
    cos = torch.cosine_similarity(rel, heading.unsqueeze(1).expand_as(rel), dim=-1)
    cos[cos.isnan()] = -1
    dist[cos < math.cos(3.14 * angle / 180)] = float('inf')          # the reference's 3.14 (quirk Q1)
    d, idx = torch.sort(dist, dim=-1)
    return d[:, :k], idx[:, :k]


def _gather_rows(features, idx, dist, threshold):
    """rows idx of (N, M, d); rows farther than `threshold` are zeroed   (data.py:449-464)."""
    out = torch.gather(features, 1, idx.unsqueeze(-1).expand(-1, -1, features.shape[-1]))
    out[(dist > threshold).unsqueeze(-1).expand_as(out)] = 0
    return out


def relative_features(position, velocity, acceleration, destination, obstacles, topk_ped=6, sight_angle_ped=90,
                      dist_threshold_ped=4, topk_obs=10, sight_angle_obs=90, dist_threshold_obs=4):
    """(ped_features (N, k_p, 6), obs_features (N, k_o, 6), dest_features (N, 2)) for one frame of (N, 2)
    tensors and (M, 2) obstacles, with the reference's dataflow (data.py:466-512)."""
    velocity = torch.nan_to_num(velocity, nan=0.0)
    acceleration = torch.nan_to_num(acceleration, nan=0.0)
    norm = torch.norm(velocity, p=2, dim=-1, keepdim=True)
    heading = velocity / torch.where(norm == 0, norm + 0.1, norm)
    state = torch.cat((position, velocity, acceleration), dim=-1)
    d, idx = _nearest_in_sight(position, position, heading, min(topk_ped, position.shape[0]), sight_angle_ped)
    ped_features = _gather_rows(_relative(state, state), idx, d, dist_threshold_ped)
    dest_features = torch.nan_to_num(destination - position, nan=0.0)
    obs_state = torch.cat((obstacles, torch.zeros(obstacles.shape[0], 4)), dim=-1)
    d, idx = _nearest_in_sight(position, obstacles, heading, min(topk_obs, obstacles.shape[0]), sight_angle_obs)
    obs_features = _gather_rows(_relative(state, obs_state), idx, d, dist_threshold_obs)
    return ped_features, obs_features, dest_features
