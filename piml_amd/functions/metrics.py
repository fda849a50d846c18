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


def _per_frame(p, q, mask):
    """(*c, t, n, d), mask (*c, t, n) -> frames flattened: (F, n, d), (F, n, d), bool (F, n)."""
    n = mask.shape[-1]
    return p.reshape(-1, n, p.shape[-1]), q.reshape(-1, n, q.shape[-1]), (mask == 1).reshape(-1, n)


def ot_with_time_mask(p, q, mask, eps=0.1, max_iter=100, reduction=None, dvs=None):
    """Entropic OT (log-domain Sinkhorn) between the predicted and the true positions of the agents
    present in each frame (metrics.py:45-67, 107-203), all frames batched: absent agents are masked
    out of the marginals and every frame stops updating at its own iteration, exactly where the
    reference's per-frame loop breaks (err < 0.1).  Frames with fewer than 2 agents are skipped."""
    x, y, m = _per_frame(p.detach(), q.detach(), mask)
    cnt = m.sum(-1)
    use = cnt > 1
    x, y, m, cnt = x[use], y[use], m[use], cnt[use]
    if x.shape[0] == 0:
        return 0.0 if reduction in ('sum',) else (float('nan') if reduction == 'mean' else [])
    x = torch.where(m.unsqueeze(-1), x, torch.zeros_like(x))
    y = torch.where(m.unsqueeze(-1), y, torch.zeros_like(y))
    C = ((x.unsqueeze(-2) - y.unsqueeze(-3)).abs() ** 2).sum(-1)                    # F, n, n
    pair = m.unsqueeze(-1) & m.unsqueeze(-2)
    neg = torch.full_like(C, float('-inf'))
    logw = torch.log(1.0 / cnt.to(C.dtype) + 1e-8).unsqueeze(-1)                    # log(mu + 1e-8)
    u = torch.zeros_like(x[..., 0])
    v = torch.zeros_like(u)

    def M(u_, v_):
        return torch.where(pair, (-C + u_.unsqueeze(-1) + v_.unsqueeze(-2)) / eps, neg)
    active = torch.ones(x.shape[0], dtype=torch.bool, device=x.device)
    for _ in range(max_iter):
        u_new = eps * (logw - torch.logsumexp(M(u, v), dim=-1)) + u
        u_new = torch.where(m, u_new, torch.zeros_like(u_new))
        v_new = eps * (logw - torch.logsumexp(M(u_new, v).transpose(-2, -1), dim=-1)) + v
        v_new = torch.where(m, v_new, torch.zeros_like(v_new))
        err = (u_new - u).abs().sum(-1)
        a = active.unsqueeze(-1)
        u, v = torch.where(a, u_new, u), torch.where(a, v_new, v)
        active = active & ~(err < 1e-1)
        if not bool(active.any()):
            break
    pi = torch.where(pair, torch.exp(M(u, v)), torch.zeros_like(C))
    cost = (pi * C).sum(dim=(-2, -1))
    if reduction == 'sum':
        return cost.sum().item()
    if reduction == 'mean':
        return cost.mean().item()
    return cost.tolist()


def mmd_with_time_mask(p, q, mask, kernel_mul=2.0, kernel_num=5, fix_sigma=None, reduction=None):
    """Multi-kernel Gaussian MMD between predicted and true positions per frame
    (metrics.py:70-91, 207-273), all frames batched with masks."""
    x, y, m = _per_frame(p.detach(), q.detach(), mask)
    cnt = m.sum(-1)
    use = cnt > 1
    x, y, m, cnt = x[use], y[use], m[use], cnt[use].to(x.dtype)
    if x.shape[0] == 0:
        return 0.0 if reduction == 'sum' else (float('nan') if reduction == 'mean' else [])
    total = torch.cat((torch.where(m.unsqueeze(-1), x, torch.zeros_like(x)),
                       torch.where(m.unsqueeze(-1), y, torch.zeros_like(y))), dim=1)          # F, 2n, d
    mm = torch.cat((m, m), dim=1)
    pair = (mm.unsqueeze(-1) & mm.unsqueeze(-2)).to(x.dtype)
    L2 = ((total.unsqueeze(1) - total.unsqueeze(2)) ** 2).sum(-1) * pair
    ns = 2 * cnt
    bandwidth = fix_sigma if fix_sigma else L2.sum(dim=(-2, -1)) / (ns ** 2 - ns)
    bandwidth = bandwidth / kernel_mul ** (kernel_num // 2)
    K = sum(torch.exp(-L2 / (bandwidth * kernel_mul ** i).reshape(-1, 1, 1)) for i in range(kernel_num)) * pair
    n = x.shape[1]
    n2 = (cnt * cnt).reshape(-1, 1, 1)
    loss = (K[:, :n, :n] / n2).sum(dim=(-2, -1)) - (K[:, :n, n:] / n2).sum(dim=(-2, -1)) \
        - (K[:, n:, :n] / n2).sum(dim=(-2, -1)) + (K[:, n:, n:] / n2).sum(dim=(-2, -1))
    if reduction == 'sum':
        return loss.sum().item()
    if reduction == 'mean':
        return loss.mean().item()
    return loss.tolist()


def fde_with_time_mask(p_pred, labels, mask_p_pred, reduction='mean'):
    """Final displacement error: L2 error at each agent's LAST masked frame of a (t, n, 2) rollout.
    The reference only reports the mean displacement (its "MAE", = ADE); BASELINE.json's config 5
    asks for ADE/FDE, so the FDE of the same masks is provided alongside."""
    m = mask_p_pred == 1                                                     # t, n
    T = m.shape[0]
    ar = torch.arange(T, device=m.device).unsqueeze(1)
    last = torch.where(m, ar, -1).max(0).values                              # n, -1 = never predicted
    has = last >= 0
    idx = last.clamp(min=0)
    cols = torch.arange(m.shape[1], device=m.device)
    err = torch.norm(p_pred[idx, cols] - labels[idx, cols], p=2, dim=-1)
    err = torch.where(has, torch.nan_to_num(err), torch.zeros_like(err))
    if reduction == 'sum':
        return err.sum().item()
    if reduction == 'mean':
        return (err.sum() / has.sum().clamp(min=1)).item()
    return err
