"""ctypes binding of libpiml_hip.so (include/piml_hip.h).  Loading fails loudly: there is no
CPU fallback for the product path."""
import ctypes
import os

# torch must be loaded first: libpiml_hip.so needs libamdhip64.so.7 and has to bind to the
# HIP runtime instance torch already loaded (device pointers and streams come from torch);
# loading it before torch would pull a second runtime from /opt/rocm into the process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libpiml_hip.so')
ABI_VERSION = 3

_lib = None

_i, _f, _p, _z = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> argtypes, in the order of include/piml_hip.h
SIGNATURES = {
    'piml_abi_version': [],
    'piml_heading_fwd': [_p, _i, _i, _i, _p, _p],
    'piml_relfeat_fwd': [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                         _p, _p, _p, _p, _p, _p],
    'piml_mlapm_step_fwd': [_p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p],
    'piml_mlapm_step_bwd': [_p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p],
    'piml_collision_matrix': [_p, _i, _i, _f, _i, _p, _p],
    'piml_collision_friends': [_p, _p, _i, _i, _i, _i, _p],
    'piml_collision_counts': [_p, _i, _i, _p, _i, _p, _p],
    'piml_collision_label': [_p, _z, _i, _p, _p],
    'piml_calc_acceleration': [_p, _z, _i, _i, _f, _f, _f, _f, _f, _f, _p, _p],
    'piml_probe_arith': [_p, _p, _p, _p, _p, _p, _i, _p],
    'piml_relfeat_bwd': [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p],
}


class PimlHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PimlHipError(
                f'{LIB_PATH} is missing: build it with `python -m piml_amd.build` '
                '(hipcc --offload-arch=gfx950). piml_amd has no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = _i
        L.piml_error_string.argtypes = [_i]
        L.piml_error_string.restype = ctypes.c_char_p
        if L.piml_abi_version() != ABI_VERSION:
            raise PimlHipError(f'{LIB_PATH}: ABI {L.piml_abi_version()} != expected {ABI_VERSION}; rebuild')
        _lib = L
    return _lib


def check(err, what):
    if err != 0:
        raise PimlHipError(f'{what} failed: hipError {err} ({lib().piml_error_string(err).decode()})')
