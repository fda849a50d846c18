"""Inference-rollout steps/s on a synthetic cfg3-sized scene (development aid)."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piml_amd.scenes import synthetic_gc_scene
from piml_amd.models.simulators import BaseSimulator


from piml_amd.scenes import synthetic_rollout_data      # noqa: E402 (the clip builder lives with the scenes)


if __name__ == '__main__':
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from test_simulator_gpu import sim_args
    dev = 'cuda:0'
    tuned = '--tuned' in sys.argv     # pre-tuned GEMM selections + obstacle branch of the MLP on a side stream
    if tuned:
        from piml_amd import tuning
        print('tuned GEMM selections loaded:', tuning.load())
    sizes = ((122, 100), (1024, 100), (4096, 2000))
    if '--sizes' in sys.argv:         # e.g. --sizes 2048,100,3000,100
        v = [int(t) for t in sys.argv[sys.argv.index('--sizes') + 1].split(',')]
        sizes = tuple(zip(v[0::2], v[1::2]))
    for N, M in sizes:
        T = 200
        data = synthetic_rollout_data(N, M, T, dev)
        torch.manual_seed(666)
        extra = dict(model=sys.argv[sys.argv.index('--model') + 1]) if '--model' in sys.argv else {}
        sim = BaseSimulator(sim_args(**extra, **(dict(mlp_side_stream_rows=1024) if tuned else {})))
        sim.model.eval()
        with torch.no_grad():
            for graph in (False, True):
                sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print(f'N={N} M={M} graph={graph}: {T / dt:8.0f} steps/s ({dt / T * 1e6:.0f} us/step)')

    if '--no-mlapm' in sys.argv:
        sys.exit(0)
    # the closed-form simulator (MLAPM.rollout, src/main_mlapm.py:18-36): a frame as one launch against the operator sequence
    from piml_amd.models.mlapm import MLAPM
    m = MLAPM(version='GC', tau=0.5, A=7.55, B=-3.0, C=0.2, D=-0.3, theta=56)
    for N in (122, 1024, 4096):
        sc = synthetic_gc_scene(N, 0, seed=0, nan_frac=0.02)
        a = [torch.tensor(sc[k], device=dev) for k in ('position', 'velocity', 'desired_speed', 'destination')]
        T = 600
        for name, kw in (('operator sequence, 1 frame per graph', dict(fused=False)), ('one launch per frame, 1 frame per graph', dict(frames_per_graph=1)),
                         ('one launch per frame, 8 frames per graph', dict(frames_per_graph=8)),
                         ('one launch per frame, 32 frames per graph', dict(frames_per_graph=32))):
            m.rollout(*a, 0.08, 0.3, 80, **kw)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m.rollout(*a, 0.08, 0.3, T, **kw)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            print(f'MLAPM GC rollout N={N} ({name}): {T / dt:8.0f} steps/s ({dt / T * 1e6:.1f} us/step)')
