"""Data containers of the reference (src/data/data.py:14-341, 604-864, 958-1160) restated as the
host side that feeds the hot path: `.npy` scene files -> dense (T,N,2) tensors with NaN for absent
agents -> per-frame relative features (HIP, all T frames in one launch) -> pointwise rows /
sliding-window channels.  Class and attribute names follow the reference so that
`BaseSimulator` and the training loop read them unchanged.

Everything numerical on the path (relative features with temporal heading fill, collision
labels) runs on the GPU through `Pedestrians`; the rest is one-off bookkeeping, vectorised here
where the reference loops in Python (desired speed, first/last valid frame).
"""
import numpy as np
import torch

from ..pedestrians import Pedestrians


def _tensor_attrs_to(obj, device):
    for k, v in list(obj.__dict__.items()):
        if isinstance(v, torch.Tensor):
            setattr(obj, k, v.to(device))


class RawData(object):
    """One scene clip (data.py:14-341).  position / velocity / acceleration / destination (T,N,2),
    waypoints (D,N,2), dest_idx (T,N), dest_num (N), obstacles (M,2), mask_p / mask_v / mask_a (T,N)."""

    def __init__(self, position=None, velocity=None, acceleration=None, destination=None, waypoints=None,
                 obstacles=None, mask_p=None, meta_data=None):
        self.position, self.velocity, self.acceleration = position, velocity, acceleration
        self.destination, self.waypoints, self.obstacles = destination, waypoints, obstacles
        self.mask_p, self.meta_data = mask_p, meta_data
        if meta_data is not None:
            self.time_unit = meta_data['time_unit']
        if position is not None:
            self.num_steps, self.num_pedestrians = position.shape[0], position.shape[1]

    def to(self, device):
        _tensor_attrs_to(self, device)

    def load_trajectory_data(self, data_path):
        """`.npy` (version v2.2: meta_data, trajectories, destinations, obstacles) -> dense tensors
        (data.py:83-167).  Velocity / acceleration are forward differences; an agent's last frame has
        no velocity, its last two no acceleration."""
        print(f"Loading from '{data_path}'...")
        data = np.load(data_path, allow_pickle=True)
        assert ('version' in data[0] and data[0]['version'] == 'v2.2'), f"'{data_path}' is out of date."
        meta_data, trajectories, destinations, obstacles = data
        obstacles = torch.tensor(np.asarray(obstacles), dtype=torch.float)
        if obstacles.shape[-1] == 0:        # no obstacle: the reference's far-away placeholder (:102-103)
            obstacles = torch.tensor([[1e4, 1e4], [1e4 + 1, 1e4 + 1]], dtype=torch.float)
        T = max(u[-1][-1] for u in trajectories) + 1
        N = len(trajectories)
        D = max(len(u) for u in destinations)
        pos = np.zeros((T, N, 2), np.float32)
        mask_p = np.zeros((T, N), np.float32)
        mask_v = np.zeros((T, N), np.float32)
        mask_a = np.zeros((T, N), np.float32)
        for i, traj in enumerate(trajectories):
            arr = np.asarray(traj, dtype=np.float64)
            t = arr[:, 2].astype(np.int64)
            pos[t, i] = arr[:, :2].astype(np.float32)
            mask_p[t, i] = mask_v[t, i] = mask_a[t, i] = 1
            last = t[-1]
            mask_v[last, i] = mask_a[last, i] = 0
            if last >= 1:
                mask_a[last - 1, i] = 0
        assert not np.isnan(pos).any(), 'ValueError: Find nan in raw data. Raw data should not contain any nan values! '
        dest = np.zeros((T, N, 2), np.float32)
        way = np.full((D, N, 2), np.nan, np.float32)
        dest_idx = np.zeros((T, N), np.int64)
        for i, relays in enumerate(destinations):
            rel = torch.tensor(relays)                      # same float32 rounding as the reference
            d = rel[:, 0:2].numpy()
            t = rel[:, 2].type(torch.int).numpy()
            way[:d.shape[0], i] = d
            for j in range(d.shape[0] - 1):
                dest[t[j]:t[j + 1], i] = d[j]
                dest_idx[t[j]:t[j + 1], i] = j
            dest[t[-1]:, i] = d[-1]
            dest_idx[t[-1]:, i] = d.shape[0] - 1
        position = torch.tensor(pos)
        mask_p, mask_v, mask_a = torch.tensor(mask_p), torch.tensor(mask_v), torch.tensor(mask_a)
        destination = torch.tensor(dest)
        nan = torch.tensor(float('nan'))
        destination[mask_p == 0] = nan
        position[mask_p == 0] = nan
        velocity = (torch.cat((position[1:], position[-1:]), 0) - position) / meta_data['time_unit']
        velocity[mask_v == 0] = 0
        acceleration = (torch.cat((velocity[1:], velocity[-1:]), 0) - velocity) / meta_data['time_unit']
        acceleration[mask_a == 0] = 0
        assert not velocity.isnan().any(), 'find nan in velocity.'
        assert not acceleration.isnan().any(), 'find nan in acceleration.'

        self.meta_data = meta_data
        self.num_steps, self.num_pedestrians, self.num_destinations = T, N, D
        self.position, self.velocity, self.acceleration, self.destination = position, velocity, acceleration, destination
        self.waypoints, self.dest_idx = torch.tensor(way), torch.tensor(dest_idx)
        self.dest_num = torch.tensor([len(r) for r in destinations])
        self.obstacles, self.mask_p, self.mask_v, self.mask_a = obstacles, mask_p, mask_v, mask_a
        self.destination_flag = torch.zeros(N, dtype=int)
        self.time_unit = meta_data['time_unit']


class TimeIndexedPedData(Pedestrians):
    """Per-frame features of one clip (data.py:604-864)."""

    def __init__(self):
        super().__init__()
        self.num_frames = self.dataset_len = 0
        self.mask_p_pred = self.mask_v_pred = self.mask_a_pred = self.meta_data = None

    def __len__(self):
        return self.num_frames

    def __getitem__(self, index):
        if self.num_frames <= 0:
            raise ValueError("Haven't load any data yet!")
        return [self.ped_features[index], self.obs_features[index], self.self_features[index], self.labels[index]]

    def to(self, device):
        _tensor_attrs_to(self, device)

    @staticmethod
    def move_index_matrix(idx_matrix, direction='forward', n_steps=1, dim=0):
        """Shift a 0/1 (t, n) matrix along `dim` and AND it with itself (data.py:670-697)."""
        rolled = torch.zeros_like(idx_matrix)
        L = idx_matrix.shape[dim]
        if direction == 'backward':
            rolled.narrow(dim, n_steps, L - n_steps).copy_(idx_matrix.narrow(dim, 0, L - n_steps))
        elif direction == 'forward':
            rolled.narrow(dim, 0, L - n_steps).copy_(idx_matrix.narrow(dim, n_steps, L - n_steps))
        else:
            raise NotImplementedError(direction)
        return rolled * idx_matrix

    @staticmethod
    def turn_detection(data):
        """1 for agents whose entry velocity points (within 20 degrees) at their exit point and who
        do not loiter (data.py:699-744); first / last valid frames found without the T-step loop."""
        position, velocity = data.position, data.velocity
        T, N, _ = position.shape
        valid = ~position.isnan().any(-1)                                     # t, n
        ar = torch.arange(T, device=position.device).unsqueeze(1)
        first = torch.where(valid, ar, T).min(0).values.clamp(max=T - 1)
        last = torch.where(valid, ar, -1).max(0).values.clamp(min=0)
        has = valid.any(0)
        cols = torch.arange(N, device=position.device)
        big = torch.full((N, 2), 1e4, device=position.device)
        starts = torch.where(has.unsqueeze(-1), position[first, cols], big)
        ends = torch.where(has.unsqueeze(-1), position[last, cols], big)
        v_starts = torch.where(has.unsqueeze(-1), velocity[first, cols], big)
        dist = torch.norm(ends - starts, p=2, dim=-1) + 1e-6
        norm_v = torch.norm(v_starts, p=2, dim=-1) + 1e-6
        cos_theta = torch.sum((ends - starts) * v_starts, dim=-1) / dist / norm_v
        non_abnormal = (cos_theta >= np.cos(3.1415 * 20 / 180)).to(position.dtype)
        non_abnormal = torch.where(cos_theta > 0, non_abnormal, torch.zeros_like(non_abnormal))
        mean_velocity = torch.norm(velocity, p=2, dim=-1).sum(0) / data.mask_v.sum(0)
        return torch.where(mean_velocity < 1.3 * 0.3, torch.zeros_like(non_abnormal), non_abnormal)

    def make_dataset(self, args, raw_data):
        """RawData -> per-frame features / labels / prediction masks (data.py:746-833).  The
        relative features of ALL frames are one HIP launch (T slices, heading with temporal fill)."""
        raw_data.to(args.device)
        ped_features, obs_features, dest_features = self.get_relative_features(
            raw_data.position, raw_data.velocity, raw_data.acceleration, raw_data.destination, raw_data.obstacles,
            args.topk_ped, args.sight_angle_ped, args.dist_threshold_ped, args.topk_obs, args.sight_angle_obs,
            args.dist_threshold_obs)
        self.abnormal_mask = self.turn_detection(raw_data)
        self.ped_features, self.obs_features = ped_features, obs_features
        T, N = ped_features.shape[0], ped_features.shape[1]
        vel = raw_data.velocity
        k = args.num_history_velocity
        hist = torch.zeros(T, N, k, 2, device=vel.device)
        for i in range(k):
            lag = k - i - 1
            hist[lag:, :, i, :] = vel[:T - lag]
        hist = hist.reshape(T, N, -1)

        # desired speed = mean |v| over the first skip_frames frames after the agent starts moving
        speed = torch.norm(vel, p=2, dim=-1)                                   # t, n
        moving = speed > 0
        ar = torch.arange(T, device=vel.device).unsqueeze(1)
        start = torch.where(moving, ar, T).min(0).values
        start = torch.where(moving.any(0), start, torch.zeros_like(start))
        win = (ar >= start.unsqueeze(0)) & (ar < (start + args.skip_frames).unsqueeze(0))
        desired = (speed * win).sum(0) / win.sum(0).clamp(min=1)
        desired_speed = desired.reshape(1, N, 1).repeat(T, 1, 1)

        self.self_features = torch.cat((dest_features, hist, raw_data.acceleration, desired_speed), dim=-1)
        labels = torch.cat((raw_data.position, raw_data.velocity, raw_data.acceleration), dim=-1)
        self.labels = torch.cat((labels, self.calculate_collision_label(self.ped_features)), dim=-1)

        s = args.skip_frames
        self.mask_a_pred = self.move_index_matrix(raw_data.mask_a, 'backward', s - 1, dim=0)
        self.mask_v_pred = self.move_index_matrix(raw_data.mask_v, 'backward', s - 1, dim=0)
        self.mask_p_pred = self.move_index_matrix(raw_data.mask_p, 'backward', s - 1, dim=0)
        self.mask_a_pred = self.move_index_matrix(self.mask_a_pred, 'forward', 1, dim=0)
        self.meta_data = raw_data.meta_data
        self.topk_obs = args.topk_obs
        self.num_frames = self.dataset_len = T
        self.num_pedestrians = N
        self.ped_feature_dim = self.ped_features.shape[-1]
        self.obs_feature_dim = self.obs_features.shape[-1]
        self.self_feature_dim = self.self_features.shape[-1]

    def set_dataset_info(self, dataset, raw_data, slice_idx):
        """Attach the raw state of frames `slice_idx` (data.py:840-864)."""
        self.meta_data, self.time_unit = raw_data.meta_data, raw_data.time_unit
        self.num_frames = self.dataset_len = dataset.num_frames
        for k in ('position', 'velocity', 'acceleration', 'destination', 'dest_idx', 'mask_p', 'mask_a', 'mask_v'):
            setattr(self, k, getattr(raw_data, k)[slice_idx])
        self.obstacles, self.waypoints, self.dest_num = raw_data.obstacles, raw_data.waypoints, raw_data.dest_num
        for k in ('mask_p_pred', 'mask_v_pred', 'mask_a_pred'):
            setattr(self, k, getattr(dataset, k)[slice_idx])
        for k in ('self_feature_dim', 'ped_feature_dim', 'obs_feature_dim', 'abnormal_mask'):
            setattr(self, k, getattr(dataset, k))

    def to_pointwise_data(self):
        out = PointwisePedData()
        out.load_from_time_indexed_peddata(self)
        return out

    def to_channeled_time_index_data(self, stride=25, mode='slice'):
        out = ChanneledTimeIndexedPedData()
        out.load_from_time_indexed_peddata(self, stride, mode)
        return out


class PointwisePedData(object):
    """Time-flattened training rows of the agents present in a frame (data.py:958-1043); labels are
    the NEXT frame's (position, velocity, acceleration, collision label)."""

    def __init__(self):
        self.dataset_len = 0

    def __len__(self):
        return self.dataset_len

    def __getitem__(self, idx):
        return [self.ped_features[idx], self.obs_features[idx], self.self_features[idx], self.labels[idx]]

    def add(self, other):
        assert self.time_unit == other.time_unit, 'PointwisePedData with different time_unit cannot be merged'
        assert self.ped_features.shape[-1] == other.ped_features.shape[-1]
        for k in ('ped_features', 'obs_features', 'self_features', 'labels'):
            setattr(self, k, torch.cat((getattr(self, k), getattr(other, k)), dim=0))
        self.dataset_len += other.dataset_len

    def load_from_time_indexed_peddata(self, data, slice_idx=None):
        sel = slice(None) if slice_idx is None else slice_idx
        keep = data.mask_a_pred[sel].reshape(-1) > 0
        labels = data.labels[sel]
        labels = torch.cat((labels[1:], torch.zeros_like(labels[:1])), dim=0)          # next-frame targets
        self.labels = labels.reshape(keep.shape[0], -1)[keep]
        self.ped_features = data.ped_features[sel].reshape(-1, *data.ped_features.shape[2:])[keep]
        self.self_features = data.self_features[sel].reshape(-1, *data.self_features.shape[2:])[keep]
        if data.obs_features.shape[-1]:
            self.obs_features = data.obs_features[sel].reshape(-1, *data.obs_features.shape[2:])[keep]
        else:
            self.obs_features = torch.zeros(self.ped_features.shape[0], data.topk_obs, self.ped_features.shape[-1],
                                            device=self.ped_features.device)
        self.dataset_len = self.labels.shape[0]
        self.ped_feature_dim = self.ped_features.shape[-1]
        self.self_feature_dim = self.self_features.shape[-1]
        self.obs_feature_dim = self.obs_features.shape[-1]
        self.time_unit = data.time_unit

    def to(self, device):
        _tensor_attrs_to(self, device)


class ChanneledTimeIndexedPedData(object):
    """Rollout windows of `stride` frames stacked on a leading channel axis (data.py:1046-1160):
    mode 'slice' = every sliding window, 'split' = disjoint windows."""

    SERIES = ('ped_features', 'obs_features', 'self_features', 'labels', 'mask_p', 'mask_v', 'mask_a',
              'mask_a_pred', 'mask_v_pred', 'mask_p_pred', 'position', 'velocity', 'acceleration',
              'destination', 'dest_idx')
    STATIC = ('obstacles', 'dest_num', 'topk_obs', 'meta_data', 'time_unit', 'num_pedestrians',
              'ped_feature_dim', 'obs_feature_dim', 'self_feature_dim', 'abnormal_mask')

    def __init__(self):
        self.num_frames = 0

    def __len__(self):
        return self.num_frames

    def __getitem__(self, index):
        if self.num_frames <= 0:
            raise ValueError("Haven't load any data yet!")
        return [self.ped_features[index, ...], self.obs_features[index, ...], self.self_features[index, ...],
                self.labels[index, ...]]

    @staticmethod
    def transform(matrix, stride, mode='slice'):
        """(t, ...) -> (c, stride, ...)."""
        T = matrix.shape[0]
        if mode == 'slice':
            return matrix.unfold(0, stride, 1)[:T - stride].movedim(-1, 1).contiguous()
        if mode == 'split':
            step = T // stride
            return matrix[:stride * step].reshape(step, stride, *matrix.shape[1:])
        raise NotImplementedError(mode)

    def load_from_time_indexed_peddata(self, data, stride=25, mode='slice'):
        assert data.num_frames > stride, 'ValueError: stride < #total time steps'
        for k in self.SERIES:
            setattr(self, k, self.transform(getattr(data, k), stride, mode))
        self.waypoints = data.waypoints.unsqueeze(0).repeat(self.position.shape[0], 1, 1, 1)      # c, d, n, 2
        self.num_frames = stride
        self.dataset_len = self.ped_features.shape[0]
        self.set_static_info_like(data)

    @staticmethod
    def slice(data, slice_idx):
        out = ChanneledTimeIndexedPedData()
        for k in ChanneledTimeIndexedPedData.SERIES + ('waypoints',):
            setattr(out, k, getattr(data, k)[slice_idx, ...])
        out.num_frames, out.dataset_len = data.num_frames, data.dataset_len
        out.set_static_info_like(data)
        return out

    def set_static_info_like(self, data):
        for k in self.STATIC:
            setattr(self, k, getattr(data, k))

    def to(self, device):
        _tensor_attrs_to(self, device)
