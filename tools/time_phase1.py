"""Split relfeat fwd time: pure streaming (negative distance threshold => no candidate ever
passes) vs full (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd import ops, _lib
from piml_amd.scenes import synthetic_gc_scene
L = _lib.lib(); dev = 'cuda:0'
for N, M in ((4096, 2000), (16384, 2000), (4096, 0), (16384, 0)):
    sc = synthetic_gc_scene(N, M, seed=0)
    p, v, a, d, o = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'acceleration', 'destination', 'obstacles')]
    pf, of, df, pi, oi = ops.relative_features(p, v, a, d, o, return_index=True)
    st = torch.cuda.current_stream().cuda_stream
    cp = ops.cos_threshold(90); Me = o.shape[0]
    for thr in (-1.0, 0.5, 4.0):
        def fwd():
            return L.piml_relfeat_fwd(p.data_ptr(), None, v.data_ptr(), a.data_ptr(), 2, d.data_ptr(), o.data_ptr(), 1, N, Me, 0, N,
                                      6, 10, cp, cp, thr, thr, pf.data_ptr(), of.data_ptr(), df.data_ptr(), 2, pi.data_ptr(), oi.data_ptr(), st)
        for _ in range(10): fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fwd()
        e1.record(); torch.cuda.synchronize()
        print(f'N={N} M={Me} dist_thr={thr}: {e0.elapsed_time(e1) * 5:.2f} us')
