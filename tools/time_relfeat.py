"""Device timing of the relfeat kernels, launched back to back through the C ABI so the GPU
(not the Python host) is the bottleneck (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib
from piml_amd.scenes import synthetic_gc_scene, pair_count, algorithmic_bytes

L = _lib.lib()
dev = 'cuda:0'
# (N, M, C, focal rows): the last entry is the launch every rank of the 8-GPU run executes (cfg4: 2048 focal rows against 16384 sources)
for N, M, C, FC in ((122, 100, 1, None), (1024, 100, 1, None), (4096, 2000, 1, None), (16384, 2000, 1, None), (128, 100, 64, None), (16384, 2000, 1, 2048)):
    sc = synthetic_gc_scene(N, M, seed=0, channels=None if C == 1 else C)
    p, v, a, d, o = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    pf, of, df, pi, oi = ops.relative_features(p, v, a, d, o, return_index=True)
    st = torch.cuda.current_stream().cuda_stream
    cp, co = ops.cos_threshold(90), ops.cos_threshold(90)
    Me = o.shape[0]

    fc = FC or N

    def fwd():
        return L.piml_relfeat_fwd(p.data_ptr(), None, v.data_ptr(), a.data_ptr(), 2, d.data_ptr(), o.data_ptr(), C, N, Me, 0, fc,
                                  6, 10, cp, co, 4.0, 4.0, pf.data_ptr(), of.data_ptr(), df.data_ptr(), 2, pi.data_ptr(), oi.data_ptr(), st)
    gs = torch.zeros(*p.shape[:-1], 6, device=dev); gd = torch.empty_like(df)
    gp, go = torch.randn_like(pf), torch.randn_like(of)

    def bwd():
        return L.piml_relfeat_bwd(gp.data_ptr(), go.data_ptr(), df.data_ptr(), pi.data_ptr(), oi.data_ptr(), p.data_ptr(), 2, d.data_ptr(),
                                  C, N, 0, N, pf.shape[-2], of.shape[-2], gs.data_ptr(), gd.data_ptr(), st)
    for name, fn in ((('fwd', fwd), ('bwd', bwd)) if FC is None else (('fwd', fwd),)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 300
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        pairs = C * pair_count(N, Me) * fc // N
        extra = f'{pairs / us * 1e6:.3e} pairs/s, alg {C * algorithmic_bytes(N, Me) / us / 1e3:.0f} GB/s' if name == 'fwd' else ''
        print(f'C={C} N_focal={fc} N_src={N} M={Me} {name}: {us:.2f} us  {extra}')
