"""CPU-only: the C-ABI library loads and exports every symbol include/piml_hip.h declares
(no compute calls without a GPU), and rejects bad arguments with hipErrorInvalidValue."""
import os
import re

import pytest

from conftest import REPO


HEADERS = ('piml_hip.h', 'piml_hip_tuning.h')      # the drop-in boundary | measurement plumbing, diagnostics, A/B switches


def declared_symbols(headers=HEADERS):
    out = set()
    for h in headers:
        text = open(os.path.join(REPO, 'include', h)).read()
        text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
        out |= set(re.findall(r'\b(piml_[a-z0-9_]+)\s*\(', text))
    return sorted(out)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert {'piml_relfeat_fwd', 'piml_relfeat_bwd', 'piml_heading_fwd', 'piml_abi_version'} <= set(syms)
    # the boundary header carries no switch and no measurement helper; the tuning header nothing a host has to call
    stable, tuning = set(declared_symbols(HEADERS[:1])), set(declared_symbols(HEADERS[1:]))
    assert not (stable & tuning)
    assert {'piml_encoder_split_tiles', 'piml_encoder_products', 'piml_trace_begin', 'piml_timer_create', 'piml_probe_arith'} <= tuning
    assert {'piml_relfeat_self_fwd', 'piml_pinnsf_fwd', 'piml_pinnsf_bwd', 'piml_mlapm_step_fwd', 'piml_collision_counts',
            'piml_p2p_exchange', 'piml_allgather_state'} <= stable


def test_library_exports_every_declared_symbol():
    from piml_amd import _lib, build
    build.build()
    L = _lib.lib()
    for name in declared_symbols():
        assert hasattr(L, name), f'{name} declared in include/piml_hip.h but not exported'
    assert L.piml_abi_version() == _lib.ABI_VERSION
    assert set(_lib.SIGNATURES) | {'piml_error_string'} == set(declared_symbols())


def test_argument_validation_without_gpu():
    from piml_amd import _lib
    L = _lib.lib()
    # negative sizes / oversize k are rejected before any HIP call
    assert L.piml_relfeat_fwd(None, None, None, None, 2, None, None, 1, -1, 0, 0, 0, 6, 10, 0., 0., 4., 4.,
                              None, None, None, 2, None, None, None) == 1
    assert L.piml_relfeat_fwd(None, None, None, None, 2, None, None, 1, 8, 0, 0, 8, 99, 10, 0., 0., 4., 4.,
                              None, None, None, 2, None, None, None) == 1
    assert L.piml_relfeat_fwd(None, None, None, None, 2, None, None, 1, 8, 0, 4, 8, 6, 10, 0., 0., 4., 4.,
                              None, None, None, 2, None, None, None) == 1
    # empty problems are a no-op success
    assert L.piml_relfeat_fwd(None, None, None, None, 2, None, None, 0, 8, 0, 0, 8, 6, 10, 0., 0., 4., 4.,
                              None, None, None, 2, None, None, None) == 0
    assert L.piml_heading_fwd(None, 0, 1, 5, None, None) == 0


def test_ops_refuse_cpu_tensors():
    import torch
    from piml_amd import _lib, ops
    x = torch.zeros(1, 4, 2)
    with pytest.raises(_lib.PimlHipError):
        ops.relative_features(x, x, x, x, torch.zeros(2, 2))
    with pytest.raises(_lib.PimlHipError):
        ops.heading_direction(x)


def test_glue_ops_refuse_cpu_tensors_and_validate_arguments():
    """The glue operators around the MLP have no CPU path either; argument errors are host-side."""
    import torch
    from piml_amd import _lib, ops
    x = torch.zeros(8, 6)
    w, b = torch.zeros(4, 6), torch.zeros(4)
    with pytest.raises(_lib.PimlHipError):
        ops.mlp_chain(x, (True,), w, b)
    with pytest.raises(_lib.PimlHipError):
        ops.linear_act(x, w, b, True)
    with pytest.raises(_lib.PimlHipError):
        ops.act_bwd_colsum(x)
    with pytest.raises(_lib.PimlHipError):
        ops.pinnsf_epilogue(torch.zeros(3, 2), None, torch.zeros(3, 7), 0.5)
    with pytest.raises(_lib.PimlHipError):
        ops.train_rollout_step(torch.zeros(1, 3, 2), torch.zeros(1, 3, 2), torch.zeros(1, 3, 2), torch.zeros(1, 3, 2),
                               torch.zeros(1, 3, 2), torch.zeros(1, 3, dtype=torch.int64), torch.zeros(2, 3, 2),
                               torch.full((3,), 2, dtype=torch.int64), 0.08)
    with pytest.raises(ValueError):
        ops.mlp_chain(x, (True, False), w, b)                    # one (weight, bias) pair per layer
    with pytest.raises(ValueError):
        ops.pinnsf_epilogue(torch.zeros(3, 2), None, torch.zeros(3, 6), 0.5)
    with pytest.raises(ValueError):
        ops.pinnsf_epilogue(torch.zeros(3, 2), None, torch.zeros(3, 7), 0.5, agent_norm=True)   # needs (C, N, 7)
    with pytest.raises(ValueError):
        ops.scale_ksum(torch.zeros(4, 6, 6))                     # cols % 4
    assert ops.mlp_chain(x, ()) is x                             # an MLP without layers is the identity


def test_library_argument_checks_of_glue_entries():
    """Null / out-of-range arguments are refused by the C ABI before any launch."""
    from piml_amd import _lib
    L = _lib.lib()
    assert L.piml_colsum_blocks(40960, 128) == 1280 and L.piml_colsum_blocks(1, 128) == 1
    assert L.piml_act_bwd_colsum(None, None, 10, 0, None, None, None, None) == 1       # cols <= 0
    assert L.piml_act_bwd_colsum(None, None, 10, 2048, None, None, None, None) == 1    # cols too wide
    assert L.piml_sum_leading(None, 0, 16, None, None) == 1
    assert L.piml_scale_ksum_fwd(None, None, 4, 6, 6, 2.0, None, None, None, None) == 1      # cols % 4
    assert L.piml_dropout_keep_bits(None, 8, 128, 1.5, 0, None, None) == 1                  # p outside [0, 1]
    assert L.piml_dropout_keep_bits(None, 0, 128, 0.5, 0, None, None) == 0                  # empty: no-op
    assert L.piml_pinnsf_epilogue_fwd(None, None, None, 0, 0.5, None, None) == 0        # empty: no-op
    assert L.piml_pinnsf_epilogue_fwd(None, None, None, 5, 0.5, None, None) == 1
    assert L.piml_train_step_bwd(None, None, None, None, None, 0, 1, 5, 0, 0.08, None, None, None, None, None) == 0


def test_tuning_file_is_consistent():
    """The committed TunableOp result file: validators first, then one line per GEMM shape, no duplicates."""
    from piml_amd import tuning
    lines = [ln for ln in open(tuning.DEFAULT_FILE).read().splitlines() if ln]
    vals = [ln for ln in lines if ln.startswith('Validator,')]
    ops_ = [ln.split(',') for ln in lines if not ln.startswith('Validator,')]
    assert len(vals) == 5 and lines[:5] == vals
    assert all(len(f) == 4 and f[0].startswith('Gemm') for f in ops_)
    keys = [(f[0], f[1]) for f in ops_]
    assert len(keys) == len(set(keys))
    assert tuning.LOADED is False                               # nothing is loaded on import


def test_no_kernel_of_the_default_dispatch_spills():
    """Register / scratch figures straight from the code objects inside libpiml_hip.so (`_lib.kernel_resource_usage`): the only
    kernels with a non-zero spill count are the two A/B forms the default dispatch never launches (profiles/r05_kernel_usage.md:
    `PIML_DEC_BWD_SPLIT=0`, `PIML_ENC_FUSED_BWD=2`)."""
    pytest.importorskip('msgpack')          # (the code objects' metadata notes are msgpack; not a dependency of the package itself)
    from piml_amd import _lib
    usage = _lib.kernel_resource_usage()
    assert len(usage) > 150 and not [k for k in usage if k.startswith('_Z')]
    for name in ('relfeat_fwd_kernel<16, false>', 'enc_fwd_x3_kernel<0, false>', 'enc_fwd_x3_kernel<2, true>', 'enc_fwd_x3_kernel<1, true>',
                 'enc_bwd_fused_x3_kernel<true, false, false, true, false>',
                 'enc_fwd_sum_x3_kernel', 'enc_bwd_fused_x3_kernel<true, false, false, true, true>', 'dec_fwd_head_sum_kernel<true>',
                 'dec_fwd_head_sum_kernel<false>',
                 'pinnsf_unfold_kernel',
                 'dec_fwd_head_kernel<true>', 'dec_bwd_split_kernel', 'relfeat_bwd_reduce_kernel', 'mlapm_bwd_sys_kernel<1>'):
        assert name in usage, name
    not_reached = {'dec_bwd_kernel', 'enc_bwd_fused8_x3_kernel<true, true, true, false>'}
    # (scalar registers parked in lanes of a vector register -- `sgpr_spill`, no memory traffic -- are not counted)
    spilling = {k: v['vgpr_spill'] for k, v in usage.items() if v['vgpr_spill']}
    assert set(spilling) <= not_reached, spilling
