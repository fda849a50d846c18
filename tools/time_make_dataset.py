"""make_dataset (feature building of a whole clip) timing (development aid)."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from piml_amd.data.data import RawData, TimeIndexedPedData
args = types.SimpleNamespace(device='cuda:0', topk_ped=6, topk_obs=10, sight_angle_ped=90, sight_angle_obs=90,
                             dist_threshold_ped=4, dist_threshold_obs=4, num_history_velocity=1, skip_frames=25, valid_steps=5)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/data/GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35.npy')
t0 = time.perf_counter(); raw = RawData(); raw.load_trajectory_data(path); t_load = time.perf_counter() - t0
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = TimeIndexedPedData(); d.make_dataset(args, raw); d.set_dataset_info(d, raw, list(range(len(d))))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f'run {i}: load .npy {t_load * 1e3:.0f} ms (host), make_dataset T={d.num_frames} N={d.num_pedestrians}: {(t1 - t0) * 1e3:.1f} ms')
from piml_amd import ops
p, v, a, dd, o = raw.position, raw.velocity, raw.acceleration, raw.destination, raw.obstacles
for name, fn in (('heading fill', lambda: ops.heading_direction(v)),
                 ('relfeat T frames', lambda: ops.relative_features(p, v, a, dd, o, heading=ops.heading_direction(v)))):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize(); print(f'{name}: {(time.perf_counter() - t0) * 100:.2f} ms')
