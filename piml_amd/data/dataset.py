"""Dataset builders of the reference (src/data/dataset.py:22-452): a YAML file lists the `.npy`
clips of each split; every clip becomes a TimeIndexedPedData (features via the HIP path), then
pointwise rows (pre-training) or sliding-window channels (rollout fine-tuning)."""
import os
from collections import defaultdict

import yaml

from .data import RawData, TimeIndexedPedData


class BaseDataset(object):
    def __init__(self):
        self.raw_data = None

    def load_data(self, data_path, add_noise_flag=False, root=None):
        """`data_path`: YAML {split: [clip.npy, ...]} (dataset.py:45-53).  Relative clip paths are
        resolved against `root` (default: the YAML's directory; the reference resolves against the
        working directory and its shipped YAMLs mis-spell `GC_dataset`, quirk Q11)."""
        listing = yaml.load(open(data_path, 'r'), Loader=yaml.FullLoader)
        root = root if root is not None else os.path.dirname(os.path.abspath(data_path))
        data = defaultdict(list)
        for key, paths in listing.items():
            for path in paths or []:
                raw = RawData()
                raw.load_trajectory_data(path if os.path.isabs(path) else os.path.join(root, path))
                data[key].append(raw)
        self.raw_data = data

    def _clips(self, args):
        assert self.raw_data, 'Error: Must load raw data before build dataset.'
        units = {d.time_unit for clips in self.raw_data.values() for d in clips}
        assert len(units) == 1, f'Error: mixed time units {units}'
        self.time_unit = units.pop()
        self.args = args
        out = defaultdict(list)
        for key, clips in self.raw_data.items():
            for raw in clips:
                d = TimeIndexedPedData()
                d.make_dataset(args, raw)
                d.set_dataset_info(d, raw, list(range(len(d))))
                out[key].append(d)
        return out

    @staticmethod
    def _publish_dims(args, d):
        args.ped_feature_dim, args.obs_feature_dim, args.self_feature_dim = \
            d.ped_feature_dim, d.obs_feature_dim, d.self_feature_dim

    @staticmethod
    def merge_pointwise_data(data_list):
        merged = data_list[0]
        for d in data_list[1:]:
            merged.add(d)
        return merged


class PointwisePedDataset(BaseDataset):
    """train / valid: merged pointwise rows; test: per-clip TimeIndexedPedData (dataset.py:106-153)."""

    def build_dataset(self, args):
        self.dataset = self._clips(args)
        for key in ('train', 'valid'):
            self.dataset[key] = [d.to_pointwise_data() for d in self.dataset[key]]
        self.train_data = self.merge_pointwise_data(self.dataset['train'])
        self.valid_data = self.merge_pointwise_data(self.dataset['valid'])
        self.train_data.to(args.device)
        self.valid_data.to(args.device)
        print('\ntrain {}, valid {}'.format(len(self.train_data), len(self.valid_data)))
        if 'test' in self.dataset:
            self.test_data = self.dataset['test']
            for d in self.test_data:
                d.to(args.device)
            print(' test {}'.format([len(d) for d in self.test_data]))
        self._publish_dims(args, self.train_data)
        print('Load data successfully!')


class TimeIndexedPedDataset2(BaseDataset):
    """train: sliding windows of `valid_steps` frames; valid / test: whole clips (dataset.py:367-420)."""

    def build_dataset(self, args):
        self.dataset = self._clips(args)
        for clips in self.dataset.values():
            for d in clips:
                d.to(args.device)
        self.train_data = [d.to_channeled_time_index_data(args.valid_steps, 'slice') for d in self.dataset['train']]
        self.valid_data = self.dataset.get('valid', [])
        self.test_data = self.dataset.get('test', [])
        self._publish_dims(args, self.dataset['train'][0])
        print('Load data successfully!')


TimeIndexedPedDataset = TimeIndexedPedDataset2


class TimeIndexedPedDatasetforVis(BaseDataset):
    """Whole clips under the 'vis' key (dataset.py:423-452)."""

    def build_dataset(self, args):
        self.dataset = self._clips(args)
        for clips in self.dataset.values():
            for d in clips:
                d.to(args.device)
        first = next(iter(self.dataset.values()))[0]
        self._publish_dims(args, first)
