"""CPU: the pieces of the train-mode dropout path that need no GPU -- the numpy restatement of the mask generator against
Philox's published known-answer vectors, the bit layout helpers, and the module-level semantics of an injected mask
(reference: ResDNN.forward = Dropout_p(2 x), src/models/model.py:82-119 with quirk Q3)."""
import types

import numpy as np
import torch

import philox_ref


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox_ref.philox4x32_10(*[np.array([c]) for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want


def test_keep_mask_statistics_and_offsets():
    for p in (0.5, 0.1, 0.9, 0.3):
        m = philox_ref.keep_mask(666, 0, 4096, 128, p)
        assert abs(m.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / m.size) + 1e-5
    a, b = philox_ref.keep_mask(666, 0, 64, 128, 0.5), philox_ref.keep_mask(666, 1, 64, 128, 0.5)
    assert (a != b).mean() > 0.4                          # another draw, another mask
    c = philox_ref.keep_mask(666, 0, 64, 128, 0.5, stream=1)
    assert (a != c).mean() > 0.4                          # another stream of the same draw, another mask
    assert philox_ref.keep_mask(1, 0, 8, 128, 1.0).sum() == 0 and philox_ref.keep_mask(1, 0, 8, 128, 0.0).all()
    for p in (0.5, 0.3):                                  # narrower layers take the leading features of the same stream
        assert np.array_equal(philox_ref.keep_mask(5, 3, 16, 100, p), philox_ref.keep_mask(5, 3, 16, 128, p)[:, :100])
    assert philox_ref.keep_bits(5, 3, 16, 200, 0.5).shape == (16, 7)


def test_bit_layout_round_trip():
    from piml_amd import ops
    g = torch.Generator().manual_seed(0)
    for cols in (128, 100, 32, 4):
        keep = torch.rand(37, cols, generator=g) < 0.5
        bits = ops.pack_keep_bits(keep)
        assert bits.dtype == torch.int32 and tuple(bits.shape) == (37, (cols + 31) // 32)
        assert torch.equal(ops.unpack_keep_bits(bits, cols), keep)
    k = philox_ref.keep_mask(9, 2, 50, 128, 0.3)
    assert np.array_equal(ops.pack_keep_bits(torch.from_numpy(k)).numpy(), philox_ref.keep_bits(9, 2, 50, 128, 0.3))


def test_resdnn_injected_mask_is_dropout_of_twice_the_input():
    import piml_amd.models.model as MODEL
    from piml_amd import ops
    r = MODEL.ResDNN(128, [[128] for _ in range(16)], None, 0.5)
    x = torch.randn(20, 6, 128)
    keep = torch.rand(120, 128) < 0.5
    r.keep_bits = ops.pack_keep_bits(keep)
    r.train()
    assert torch.equal(r(x), (2 * x) * keep.view(20, 6, 128) / 0.5)
    r.eval()
    assert torch.equal(r(x), 2 * x)                         # eval: the mask is not applied
    assert r.scales_input() and not MODEL.ResDNN(128, [[128]], None, 0.5).scales_input()
    r.train()
    assert r.dropout_active()
    r.dropout.p = 0.0
    assert not r.dropout_active() and r.fused_spec(120, 'cpu') == (2.0, None)


def test_model_train_mode_with_injected_masks_on_cpu():
    """The plain torch.nn expression (the comparison side of the GPU tests) honours the injected masks."""
    import piml_amd.models.model as MODEL
    from piml_amd import ops
    args = types.SimpleNamespace(
        ped_feature_dim=6, obs_feature_dim=6, self_feature_dim=7, encoder_hidden_size=128, processor_hidden_size=128,
        decoder_hidden_size=64, encoder_hidden_layers=3, processor_hidden_layers=16, decoder_hidden_layers=2, dropout=0.5,
        activation='relu', dataset_name='gc1560')
    torch.manual_seed(0)
    net = MODEL.PINNSF_multitask(args).train()
    n = 9
    ins = [torch.randn(n, 6, 6), torch.randn(n, 10, 6), torch.randn(n, 7)]
    kp, ko = torch.rand(n * 6, 128) < 0.5, torch.rand(n * 10, 128) < 0.5
    net.ped_processor.keep_bits, net.obs_processor.keep_bits = ops.pack_keep_bits(kp), ops.pack_keep_bits(ko)
    out = net(*ins)
    want = 2 * net.ped_encoder(ins[0]) * kp.view(n, 6, 128) / 0.5
    assert torch.allclose(out[1], want)
    assert torch.equal(net(*ins)[0], out[0])                # the injected mask is reused, not redrawn
