"""Achievable HBM rates of plain streaming kernels on this GPU (context for the roofline fractions): device-to-device copy,
fill and a read-only reduction, at sizes from one kernel's working set of the step (33 MB) to far beyond the 256 MB MALL."""
import json
import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    out = []
    for mb in (33, 100, 400, 1600):
        n = mb * (1 << 20) // 4
        a = torch.empty(n, device='cuda', dtype=torch.float32).normal_()
        b = torch.empty_like(a)
        t_copy = timed(lambda: b.copy_(a))
        t_fill = timed(lambda: b.fill_(1.0))
        t_read = timed(lambda: a.sum())
        out.append({'MB': mb, 'copy_TBps': round(2 * n * 4 / t_copy / 1e12, 2), 'copy_us': round(t_copy * 1e6, 1),
                    'fill_TBps': round(n * 4 / t_fill / 1e12, 2), 'read_TBps': round(n * 4 / t_read / 1e12, 2),
                    'read_us': round(t_read * 1e6, 1)})
        del a, b
    print(json.dumps(out))


if __name__ == '__main__':
    main()
