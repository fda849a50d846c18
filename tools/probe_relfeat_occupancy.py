"""Does relfeat_fwd gain from more resident waves per SIMD?  The same 4096-agent scene as C = 1, 2, 4 identical slices
(rows x C, same sources per row): at C = 1 the launch is one 16-wave workgroup per CU (4 waves per SIMD); C = 2 doubles
the workgroups (two per CU resident)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piml_amd import ops, _lib
from piml_amd.scenes import synthetic_gc_scene

L = _lib.lib()
dev = 'cuda:0'
N, M = 4096, 2000
sc = synthetic_gc_scene(N, M, seed=0)
for C in (1, 2, 4):
    rep = lambda k: torch.tensor(np.repeat(sc[k][None], C, 0), device=dev).contiguous()
    p, v, a, d = rep('position'), rep('velocity'), rep('acceleration'), rep('destination')
    o = torch.tensor(sc['obstacles'], device=dev)
    pf, of, df, pi, oi = ops.relative_features(p, v, a, d, o, return_index=True)
    st = torch.cuda.current_stream().cuda_stream
    cp = ops.cos_threshold(90)
    Me = o.shape[0]
    for waves in (16, 8):
        os.environ['PIML_RELFEAT_WAVES'] = str(waves)

        def fwd():
            return L.piml_relfeat_fwd(p.data_ptr(), None, v.data_ptr(), a.data_ptr(), 2, d.data_ptr(), o.data_ptr(), C, N, Me, 0, N,
                                      6, 10, cp, cp, 4.0, 4.0, pf.data_ptr(), of.data_ptr(), df.data_ptr(), 2, pi.data_ptr(), oi.data_ptr(), st)
        for _ in range(10):
            fwd()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fwd()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 200
        print(f'C={C} waves/WG={waves}: {us:.2f} us = {us / C:.2f} us per 4096 rows')
