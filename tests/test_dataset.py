"""Data containers: RawData loading (CPU) and the feature-building pipeline (GPU) against outputs of
the reference's own classes on the same clip files (tests/golden/dataset.npz; the clip files under
tests/golden/data/ are data fixtures copied from the reference's data/ directory)."""
import os
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bits, golden

CLIPS = {'toy1': ('GC_Dataset_toy1.npy', 1),
         'gc': ('GC_Dataset_ped1-12685_time1000-1060_interp9_xrange5-25_yrange15-35.npy', 25)}


def data_args(device='cpu'):
    return types.SimpleNamespace(device=device, topk_ped=6, topk_obs=10, sight_angle_ped=90, sight_angle_obs=90,
                                 dist_threshold_ped=4, dist_threshold_obs=4, num_history_velocity=1, skip_frames=25,
                                 valid_steps=5)


@pytest.mark.parametrize('tag', sorted(CLIPS))
def test_raw_data_loading_matches_reference(tag):
    from piml_amd.data.data import RawData
    g = golden('dataset')
    fname, keep = CLIPS[tag]
    raw = RawData()
    raw.load_trajectory_data(os.path.join(GOLDEN, 'data', fname))
    for k in ('position', 'velocity', 'acceleration', 'destination', 'mask_p', 'mask_v', 'mask_a'):
        assert np.array_equal(bits(getattr(raw, k)[::keep].numpy()), bits(g[f'{tag}/raw_{k}'])), k
    assert np.array_equal(raw.dest_idx[::keep].numpy(), g[f'{tag}/raw_dest_idx'])
    assert np.array_equal(bits(raw.waypoints.numpy()), bits(g[f'{tag}/raw_waypoints']))
    assert np.array_equal(raw.dest_num.numpy(), g[f'{tag}/raw_dest_num'])
    assert np.array_equal(bits(raw.obstacles.numpy()), bits(g[f'{tag}/raw_obstacles']))


def test_channel_transform_matches_reference_semantics():
    from piml_amd.data.data import ChanneledTimeIndexedPedData as C
    x = torch.arange(20 * 3, dtype=torch.float32).reshape(20, 3)
    s = C.transform(x, 5, 'slice')
    assert s.shape == (15, 5, 3)
    for c in (0, 7, 14):
        assert torch.equal(s[c], x[c:c + 5])
    p = C.transform(x, 6, 'split')
    assert p.shape == (3, 6, 3) and torch.equal(p[2], x[12:18])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', sorted(CLIPS))
def test_make_dataset_matches_reference(tag):
    from piml_amd.data.data import RawData, TimeIndexedPedData
    g = golden('dataset')
    fname, keep = CLIPS[tag]
    args = data_args('cuda:0')
    raw = RawData()
    raw.load_trajectory_data(os.path.join(GOLDEN, 'data', fname))
    d = TimeIndexedPedData()
    d.make_dataset(args, raw)
    d.set_dataset_info(d, raw, list(range(len(d))))
    # relative features and collision labels: bit exact (all T frames in one launch, temporal heading fill)
    assert np.array_equal(bits(d.ped_features[::keep].cpu().numpy()), bits(g[f'{tag}/ped_features']))
    assert np.array_equal(bits(d.obs_features[::keep].cpu().numpy()), bits(g[f'{tag}/obs_features']))
    assert np.array_equal(bits(d.labels[::keep].cpu().numpy()), bits(g[f'{tag}/labels']))
    for k in ('mask_p_pred', 'mask_a_pred', 'mask_v_pred'):
        assert np.array_equal(getattr(d, k)[::keep].cpu().numpy(), g[f'{tag}/{k}']), k
    assert np.array_equal(d.abnormal_mask.cpu().numpy(), g[f'{tag}/abnormal_mask'])
    # desired speed is a mean over <= 25 frames (summation order differs): 1e-6 relative
    sf, ref = d.self_features[::keep].cpu().numpy(), g[f'{tag}/self_features']
    assert np.array_equal(bits(sf[..., :6]), bits(ref[..., :6]))
    assert np.allclose(sf[..., 6], ref[..., 6], rtol=1e-6, atol=1e-7)
    # channelled windows and pointwise rows
    ch = d.to_channeled_time_index_data(args.valid_steps, 'slice')
    assert list(ch.position.shape) == list(g[f'{tag}/ch_slice_shape'])
    assert np.array_equal(bits(ch.position[7].cpu().numpy()), bits(g[f'{tag}/ch_slice_pos_win7']))
    assert np.allclose(ch.ped_features.sum(dim=(1, 2, 3, 4)).cpu().numpy(), g[f'{tag}/ch_slice_pf_sum'], rtol=1e-5, atol=1e-4)
    sp = d.to_channeled_time_index_data(args.valid_steps, 'split')
    assert list(sp.position.shape) == list(g[f'{tag}/ch_split_shape'])
    assert np.array_equal(bits(sp.position[3].cpu().numpy()), bits(g[f'{tag}/ch_split_pos_win3']))
    pw = d.to_pointwise_data()
    assert len(pw) == int(g[f'{tag}/pw_len'])
    assert np.array_equal(bits(pw.labels[:64].cpu().numpy()), bits(g[f'{tag}/pw_labels_head']))
    assert np.allclose(pw.self_features[:64].cpu().numpy(), g[f'{tag}/pw_self_head'], rtol=1e-6, atol=1e-7)
    assert np.allclose(pw.ped_features.sum(dim=(1, 2)).cpu().numpy(), g[f'{tag}/pw_ped_sum'], rtol=1e-5, atol=1e-4)
