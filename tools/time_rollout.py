"""Inference-rollout steps/s on a synthetic cfg3-sized scene (development aid)."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from piml_amd.scenes import synthetic_gc_scene
from piml_amd.models.simulators import BaseSimulator


def synthetic_rollout_data(N, M, T, dev, seed=0):
    sc = synthetic_gc_scene(N, M, seed=seed)
    t = lambda x: torch.tensor(x, device=dev)
    rep = lambda x: t(x).unsqueeze(0).repeat(T, *([1] * x.ndim)).contiguous()
    d = types.SimpleNamespace()
    d.position, d.velocity, d.acceleration, d.destination = [rep(sc[k]) for k in ('position', 'velocity', 'acceleration', 'destination')]
    d.velocity = torch.nan_to_num(d.velocity)
    d.obstacles = t(sc['obstacles'])
    far = sc['destination'] + (sc['destination'] - np.nan_to_num(sc['position'])) * 100
    d.waypoints = torch.stack((t(sc['destination']), t(far.astype(np.float32))))
    d.dest_num = torch.full((N,), 2, device=dev, dtype=torch.long)
    d.dest_idx = torch.zeros(T, N, device=dev, dtype=torch.long)
    present = (~torch.isnan(d.position[..., 0])).float()
    d.mask_p, d.mask_p_pred = present, present.clone()
    d.num_frames, d.time_unit, d.meta_data = T, 0.08, None
    from piml_amd.pedestrians import Pedestrians
    pf, of, df = Pedestrians().get_relative_features(d.position[:1].clone(), d.velocity[:1].clone(), d.acceleration[:1].clone(),
                                                     d.destination[:1].clone(), d.obstacles, 6, 90, 4, 10, 90, 4)
    d.ped_features, d.obs_features = pf.repeat(T, 1, 1, 1), of.repeat(T, 1, 1, 1)
    d.self_features = torch.cat((df, d.velocity[:1], d.acceleration[:1], t(sc['desired_speed']).unsqueeze(0)), -1).repeat(T, 1, 1)
    return d


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
        sim = BaseSimulator(sim_args(**(dict(mlp_side_stream_rows=1024) if tuned else {})))
        sim.model.eval()
        with torch.no_grad():
            for graph in (False, True):
                sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sim.get_multiple_rollouts(data, 0, load_model=False, use_graph=graph)
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print(f'N={N} M={M} graph={graph}: {T / dt:8.0f} steps/s ({dt / T * 1e6:.0f} us/step)')
