#!/usr/bin/env python3
"""Times the fused f32-MFMA encoder kernels (piml_encoder_fwd / piml_encoder_bwd) at the bench shape
(4096 agents: 24 576 pedestrian + 40 960 obstacle neighbour rows) with HIP events; prints achieved TFLOP/s
against the 157.3 TF dense f32 matrix peak, next to the library-GEMM chain of round 1 (PIML_FUSED_ENCODER=0)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from piml_amd import ops, _lib  # noqa: E402

DEV = 'cuda:0'
H = 128


def branch(n, k, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, k, 6, generator=g).to(DEV).requires_grad_(True)
    dims = [(H, 6), (H,), (H, H), (H,), (H, H), (H,)]
    w = [(torch.randn(*d, generator=g) * 0.2).to(DEV).requires_grad_(True) for d in dims]
    return dict(x=x, scale=2.0, weights=w, pooled=True)


def timed(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    tm = _lib.StreamTimer()
    tm.start()
    for _ in range(reps):
        fn()
    tm.stop()
    torch.cuda.synchronize()
    return tm.elapsed_ms() * 1e3 / reps


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    brs = [branch(n, 6, 1), branch(n, 10, 2)]
    rows = n * 16
    fwd_flop = 2 * rows * (6 * H + 2 * H * H)
    bwd_flop = 2 * rows * (4 * H * H + 2 * 6 * H)

    def fwd():
        return ops.fused_encoders(brs)
    outs = fwd()
    gp = [torch.randn_like(p) for _, p in outs]

    def fwd_only():
        with torch.no_grad():
            ops.fused_encoders(brs)
    t_inf = timed(fwd_only)
    t_fwd = timed(fwd)

    def fb():
        o = ops.fused_encoders(brs)
        torch.autograd.backward([p for _, p in o], gp)
    t_fb = timed(fb)
    print(f'agents {n}: rows {rows}')
    print(f'  forward (inference, no h1/h2 stores): {t_inf:8.1f} us  {fwd_flop / t_inf * 1e-6:7.1f} TF/s  '
          f'({fwd_flop / t_inf * 1e-6 / 157.3:.2f} of f32 MFMA peak)   [incl. the k-sum launches]')
    print(f'  forward (training):                   {t_fwd:8.1f} us  {fwd_flop / t_fwd * 1e-6:7.1f} TF/s')
    print(f'  forward + backward:                   {t_fb:8.1f} us  -> backward ~{t_fb - t_fwd:6.1f} us  '
          f'{bwd_flop / max(t_fb - t_fwd, 1e-3) * 1e-6:7.1f} TF/s')


if __name__ == '__main__':
    main()
