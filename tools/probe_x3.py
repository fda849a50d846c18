"""Split-bf16-product encoder kernels (encoder_x3.hip) against the f32 matrix-core ones (encoder.hip): error of outputs and
gradients against float64 for both, and kernel time of forward + backward.  Run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib

DEV = 'cuda:0'
H = 128


def make_branch(n, k, in_dim, seed, scale=2.0):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(n, k, in_dim, generator=g) * 2).to(DEV)
    dims = [(H, in_dim), (H,), (H, H), (H,), (H, H), (H,)]
    w = [(torch.randn(*d, generator=g) * (0.3 if len(d) == 2 else 0.1)).to(DEV).requires_grad_(True) for d in dims]
    return dict(x=x.requires_grad_(True), scale=scale, weights=w, pooled=True)


def reference(br, gp):
    x = br['x'].detach().double().requires_grad_(True)
    w = [t.detach().double().requires_grad_(True) for t in br['weights']]
    h = torch.relu(x @ w[0].t() + w[1])
    h = torch.relu(h @ w[2].t() + w[3])
    msgs = br['scale'] * (h @ w[4].t() + w[5])
    pooled = msgs.sum(-2)
    (pooled * gp.double()).sum().backward()
    return msgs.detach(), pooled.detach(), x.grad, [t.grad for t in w]


def rel(a, b):
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


def main():
    L = _lib.lib()
    shapes = [(4096, 12, 6), (4096, 4, 6)]
    brs = [make_branch(n, k, d, seed=3 + i) for i, (n, k, d) in enumerate(shapes)]
    gen = torch.Generator().manual_seed(5)
    gps = [torch.randn(n, H, generator=gen).to(DEV) for n, _, _ in shapes]
    refs = [reference(br, gp) for br, gp in zip(brs, gps)]
    leaves = [t for br in brs for t in (br['x'], *br['weights'])]
    if len(sys.argv) > 1 and sys.argv[1] == 'nograd':      # inference forwards only (no h1 / h2 stores): for the profiler
        for mode in (0, 1):
            L.piml_encoder_products(mode)
            with torch.no_grad():
                for rep in range(20):
                    ops.fused_encoders(brs)
            torch.cuda.synchronize()
        return
    for mode in (0, 1):
        L.piml_encoder_products(mode)
        outs = ops.fused_encoders(brs)
        loss = sum((p * gp).sum() for (m, p), gp in zip(outs, gps))
        grads = torch.autograd.grad(loss, leaves)
        worst = {}
        gi = 0
        for br, (m, p), ref in zip(brs, outs, refs):
            worst['msgs'] = max(worst.get('msgs', 0), rel(m.detach(), ref[0]))
            worst['pooled'] = max(worst.get('pooled', 0), rel(p.detach(), ref[1]))
            worst['g_x'] = max(worst.get('g_x', 0), rel(grads[gi], ref[2]))
            for nm, g, r in zip(('dW1', 'db1', 'dW2', 'db2', 'dW3', 'db3'), grads[gi + 1:gi + 7], ref[3]):
                worst[nm] = max(worst.get(nm, 0), rel(g, r))
            gi += 7
        # mean abs error of msgs relative to mean abs value: the average, not the worst element
        mean = sum(float((m.detach().double() - ref[0]).abs().mean() / ref[0].abs().mean()) for (m, p), ref in zip(outs, refs)) / len(refs)
        print(f'products={"bf16x3" if mode else "f32"}: max rel err vs float64 ' + ', '.join(f'{k} {v:.1e}' for k, v in worst.items()) +
              f'; msgs mean rel err {mean:.2e}')
        # time: forward+backward, 20 repetitions
        for rep in range(3):
            outs = ops.fused_encoders(brs)
            loss = sum((p * gp).sum() for (m, p), gp in zip(outs, gps))
            torch.autograd.grad(loss, leaves)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        tf = tb = 0.0
        for rep in range(20):
            e0.record()
            outs = ops.fused_encoders(brs)
            loss = sum((p * gp).sum() for (m, p), gp in zip(outs, gps))
            e1.record()
            torch.autograd.grad(loss, leaves)
            e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
        print(f'   host-inclusive: forward {tf / 20 * 1e3:.1f} us, backward {tb / 20 * 1e3:.1f} us')


if __name__ == '__main__':
    main()
