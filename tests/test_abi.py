"""CPU-only: the C-ABI library loads and exports every symbol include/piml_hip.h declares
(no compute calls without a GPU), and rejects bad arguments with hipErrorInvalidValue."""
import os
import re

import pytest

from conftest import REPO


def declared_symbols():
    text = open(os.path.join(REPO, 'include', 'piml_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(piml_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert {'piml_relfeat_fwd', 'piml_relfeat_bwd', 'piml_heading_fwd', 'piml_abi_version'} <= set(syms)


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
