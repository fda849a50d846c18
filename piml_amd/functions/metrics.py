"""Evaluation metrics on the hot path's side (src/functions/metrics.py:16-42): collision counts via
the fused HIP kernel and the masked mean displacement error (MAE = ADE-equivalent)."""
import torch

from .. import ops


def collision_count(position, threshold, real_position=None, reduction=None):
    """Pedestrians.collision_detection reduced over everything (metrics.py:16-26).  'sum' / 'mean'
    use the fused per-agent count kernel (no (t,n,n) matrix) when no `real_position` is given;
    reduction=None returns the matrix like the reference."""
    if reduction is None:
        return ops.collision_detection(position, threshold, real_position)
    if reduction not in ('sum', 'mean'):
        raise NotImplementedError
    if real_position is not None or position.dim() != 3:
        total = ops.collision_detection(position, threshold, real_position).sum()
    else:
        total = ops.collision_counts(position, (threshold,))[0].sum()
    n = position.shape[-2]
    cells = position.numel() // 2 * n
    return (total if reduction == 'sum' else total / cells).item()


def mae_with_time_mask(p_pred, labels, mask_p_pred, reduction='none'):
    """Mean L2 displacement over the masked (frame, agent) entries (metrics.py:29-42)."""
    m = mask_p_pred == 1
    err = torch.norm(torch.where(m.unsqueeze(-1), p_pred - labels, torch.zeros_like(p_pred)), p=2, dim=-1)
    if reduction == 'sum':
        return err.sum().item()
    if reduction == 'mean':
        return (err.sum() / m.sum().clamp(min=1)).item()
    return err
