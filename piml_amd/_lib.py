"""ctypes binding of libpiml_hip.so (include/piml_hip.h).  Loading fails loudly: there is no
CPU fallback for the product path."""
import ctypes
import os

# torch must be loaded first: libpiml_hip.so needs libamdhip64.so.7 and has to bind to the
# HIP runtime instance torch already loaded (device pointers and streams come from torch);
# loading it before torch would pull a second runtime from /opt/rocm into the process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# PIML_LIB=<path>: an experimental build of the same ABI beside the shipped library (piml_amd.build.variant; tools/ A/B timings)
LIB_PATH = os.environ.get('PIML_LIB') or os.path.join(_HERE, 'libpiml_hip.so')
ABI_VERSION = 32

_lib = None

_i, _f, _p, _z = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
_ll = ctypes.c_longlong


class EncoderBranch(ctypes.Structure):
    """piml_encoder_branch (include/piml_hip.h)."""
    _fields_ = [('x', _p), ('rows', _ll), ('in_dim', _i), ('k', _i),
                ('w1', _p), ('b1', _p), ('w2', _p), ('b2', _p), ('w3', _p), ('b3', _p),
                ('scale', _f), ('h1', _p), ('h2', _p), ('msgs', _p), ('g_pooled', _p), ('g_msgs', _p),
                ('g2', _p), ('g1', _p), ('g_x', _p), ('partials', _p), ('grads', _p), ('packed', _p), ('relu_mask', _p),
                ('keep_bits', _p), ('drop_state', _p), ('drop_p', _f), ('sum_a', _p), ('sum_b', _p)]


class DecoderBranch(ctypes.Structure):
    """piml_decoder_branch (include/piml_hip.h)."""
    _fields_ = [('msgs', _p), ('agents', _ll), ('k', _i),
                ('w1', _p), ('b1', _p), ('w2', _p), ('b2', _p), ('w3', _p), ('b3', _p),
                ('pooled', _p), ('h1', _p), ('d2', _p), ('g_pre2', _p), ('g_pre1', _p), ('g_pooled', _p),
                ('partials', _p), ('grads', _p), ('packed', _p), ('pred', _p), ('g_pred_rows', _p), ('g_d2', _p),
                ('fold_w3', _p), ('fold_b3', _p), ('fold_scale', _f), ('dw1_out', _p)]


class CollisionHead(ctypes.Structure):
    """piml_collision_head (include/piml_hip.h)."""
    _fields_ = [('msgs', _p), ('rows', _ll), ('w1', _p), ('b1', _p), ('w2', _p), ('b2', _p), ('packed', _p), ('out', _p),
                ('fold_w3', _p), ('fold_b3', _p), ('fold_scale', _f)]


class Head64(ctypes.Structure):
    """piml_head64 (include/piml_hip.h)."""
    _fields_ = [('x', _p), ('rows', _ll), ('w1', _p), ('b1', _p), ('w2', _p), ('b2', _p), ('hidden', _p), ('out', _p),
                ('g_out', _p), ('g_x', _p), ('partials', _p), ('grads', _p)]


class Corrector(ctypes.Structure):
    """piml_corrector (include/piml_hip.h)."""
    _fields_ = [('agents', _ll), ('k', _i), ('scale', _f), ('enc', _p), ('keep_bits', _p),
                ('wa', _p), ('ba', _p), ('wb', _p), ('bb', _p), ('wc', _p), ('bc', _p), ('wd', _p), ('bd', _p),
                ('hid', _p), ('score', _p), ('attn', _p), ('pooled', _p), ('chid', _p), ('out', _p),
                ('g_out', _p), ('g_pooled', _p), ('g_score', _p), ('g_chid', _p), ('g_enc', _p),
                ('partials_a', _p), ('partials_b', _p), ('grads', _p)]


PACKED_VALID, FORK, ACCUMULATE, DEFER_SLOT_SUMS, DEFER_PACK, POOL_H2, POOL_TRAIN, POOL_MSGS = 1, 2, 4, 8, 16, 32, 64, 128          # piml_pinnsf_* flags

# name -> argtypes, in the order of include/piml_hip.h
SIGNATURES = {
    'piml_abi_version': [],
    'piml_heading_fwd': [_p, _i, _i, _i, _p, _p],
    'piml_relfeat_fwd': [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                         _p, _p, _p, _i, _p, _p, _p],
    'piml_mlapm_step_fwd': [_p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _i, _p, _p, _p],
    'piml_mlapm_step_bwd': [_p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p],
    'piml_mlapm_rollout_step': [_p, _p, _p, _p, _ll, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p],
    'piml_mlapm_bwd_workspace_floats': [_i, _i],
    'piml_mlapm_step_bwd_ws': [_p, _p, _p, _p, _p, _i, _i, _f, _f, _f, _f, _f, _f, _f, _f, _p, _p, _p, _p, _p, _ll, _p],
    'piml_collision_matrix': [_p, _i, _i, _f, _i, _p, _p],
    'piml_collision_friends': [_p, _p, _i, _i, _i, _i, _p],
    'piml_collision_counts': [_p, _i, _i, _p, _i, _p, _p],
    'piml_collision_counts_frames': [_p, _i, _i, _i, _p, _i, _p, _p],
    'piml_collision_counts_scratch': [_p, _i, _i, _p, _i, _p, _p, _p],
    'piml_collision_counts_grid': [_p, _i, _i, _p, _i, _p, _p, _p],
    'piml_collision_label': [_p, _z, _i, _p, _p],
    'piml_calc_acceleration': [_p, _z, _i, _i, _f, _f, _f, _f, _f, _f, _p, _p],
    'piml_rollout_step': [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p,
                          _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p],
    'piml_rollout_step_ksum': [_p, _i, _p, _i, _f, _p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p,
                               _p, _p, _p, _p, _p, _i, _i, _i, _f, _i, _p],
    'piml_collision_correction_fwd': [_p, _p, _p, _z, _i, _i, _f, _f, _p, _p],
    'piml_collision_correction_bwd': [_p, _p, _p, _p, _z, _i, _i, _f, _f, _p, _p, _p, _p],
    'piml_train_step_fwd': [_p] * 7 + [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p],
    'piml_train_step_fwd_copy': [_p] * 7 + [_i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _p],
    'piml_train_step_bwd': [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p],
    'piml_train_step_bwd6': [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p],
    'piml_pinnsf_unfold_defer': [_i, _p],
    'piml_train_step_bwd7': [_p, _ll, _p, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p],
    'piml_train_step_tail_fwd': [_p, _p, _p, _p, _i, _p, _i, _p, _f, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f,
                                 _p, _p, _p, _p, _p, _p, _p, _p, _ll, _p],
    'piml_train_step_tail_bwd': [_p, _ll, _p, _p, _p, _p, _ll, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _f, _i, _i, _p, _p, _p, _p],
    'piml_pinnsf_epilogue_fwd': [_p, _p, _p, _z, _f, _p, _p],
    'piml_pinnsf_epilogue_bwd': [_p, _p, _z, _f, _p, _p],
    'piml_pinnsf_epilogue_ksum_fwd': [_p, _i, _p, _i, _p, _z, _f, _p, _p],
    'piml_pinnsf_epilogue_ksum_bwd': [_p, _p, _z, _f, _i, _i, _p, _p, _p, _p],
    'piml_pinnsf_epilogue_agentnorm_fwd': [_p, _p, _p, _i, _i, _f, _p, _p],
    'piml_pinnsf_epilogue_ksum_agentnorm_fwd': [_p, _i, _p, _i, _p, _i, _i, _f, _p, _p],
    'piml_pinnsf_epilogue_agentnorm_bwd': [_p, _p, _i, _i, _f, _p, _p],
    'piml_self_features_fwd': [_p, _i, _p, _p, _z, _p, _p],
    'piml_self_features_bwd': [_p, _z, _p, _p, _p, _p],
    'piml_colsum_blocks': [_z, _i],
    'piml_act_bwd_colsum': [_p, _p, _z, _i, _p, _p, _p, _p],
    'piml_sum_leading': [_p, _i, _z, _p, _p],
    'piml_act_bwd_colsum_stage1': [_p, _p, _z, _i, _p, _p, _p, _p],
    'piml_layer_reduce': [_p, _i, _z, _p, _p, _i, _i, _p, _p],
    'piml_scale_ksum_fwd': [_p, _p, _z, _i, _i, _f, _p, _p, _p, _p],
    'piml_ksum_blocks': [_z, _i],
    'piml_scale_ksum_bwd': [_p, _p, _z, _i, _i, _f, _p, _p, _p, _p],
    'piml_dropout_keep_bits': [_p, _ll, _i, _f, _i, _p, _p],
    'piml_timer_create': [ctypes.POINTER(_p)],
    'piml_timer_record': [_p, _p],
    'piml_timer_elapsed_ms': [_p, _p, ctypes.POINTER(_f)],
    'piml_timer_destroy': [_p],
    'piml_trace_begin': [],
    'piml_trace_mark': [ctypes.c_char_p, _p],
    'piml_trace_end': [ctypes.c_char_p, _i, ctypes.POINTER(_f), _i],
    'piml_probe_arith': [_p, _p, _p, _p, _p, _p, _i, _p],
    'piml_encoder_partial_floats': [],
    'piml_encoder_pack_floats': [],
    'piml_encoder_split_tiles': [_ll],
    'piml_encoder_split_tiles_train': [_ll],
    'piml_multi_copy': [ctypes.POINTER(_p), ctypes.POINTER(_p), ctypes.POINTER(_z), _i, _p],
    'piml_rollout_prologue': [_p] * 8 + [_i] * 4 + [_p] * 11 + [_p],
    'piml_rollout_losses_blocks': [_i, _i],
    'piml_rollout_losses': [_p, _p, _ll, _p, _p, _p, _p, _p, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p],
    'piml_rollout_losses_bwd': [_p, _p, _p, _p, _p, _p, _ll, _p, _p],
    'piml_rollout_losses_frames': [_p, _p, _ll, _p, _p, ctypes.POINTER(_p), _i, _p, _i, _i, _i, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p],
    'piml_rollout_losses_frames_bwd': [_p, _p, _p, _p, _f, _f, _p, _p, _p, _ll, _p, _p],
    'piml_collision_pred_loss_blocks': [_ll, _i],
    'piml_collision_pred_loss': [ctypes.POINTER(_p), ctypes.POINTER(_p), _i, _ll, _i, _i, _p, _i, _i, _f, _p, _p, _p, _p, _p],
    'piml_collision_pred_loss_bwd': [_p, _p, _ll, _p, _p],
    'piml_adam_tickets': [],
    'piml_adam_step': [ctypes.POINTER(_p)] * 5 + [ctypes.POINTER(_ll), _i, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _p, _p],
    'piml_pointwise_losses_blocks': [_ll, _ll, _i],
    'piml_pointwise_losses': [_p, _p, _ll, _ll, _p, _ll, _f, _p, _i, _p, _p, _p, _p, _p],
    'piml_p2p_alloc': [_z, ctypes.POINTER(_p)],
    'piml_p2p_free': [_p],
    'piml_p2p_export': [_p, _p],
    'piml_p2p_open': [_p, ctypes.POINTER(_p)],
    'piml_p2p_close': [_p],
    'piml_p2p_copy': [_p, _p, _z, _p],
    'piml_allgather_state_p2p': [_p, _z, _i, _i, ctypes.POINTER(_p), ctypes.POINTER(_p), ctypes.c_uint, ctypes.c_uint, _p, _p],
    'piml_encoder_products': [_i],
    'piml_rowdecoder_products': [_i],
    'piml_encoder_dw2': [_i],
    'piml_encoder_fused_bwd': [_i],
    'piml_encoder_sums_bwd': [_i],
    'piml_encoder_pack': [ctypes.POINTER(EncoderBranch), _i, _p],
    'piml_encoder_workgroups': [ctypes.POINTER(EncoderBranch), _i, ctypes.POINTER(_i)],
    'piml_encoder_fwd': [ctypes.POINTER(EncoderBranch), _i, _p],
    'piml_encoder_fwd_packed': [ctypes.POINTER(EncoderBranch), _i, _p],
    'piml_encoder_bwd': [ctypes.POINTER(EncoderBranch), _i, _p],
    'piml_encoder_bwd_acc': [ctypes.POINTER(EncoderBranch), _i, _i, _p],
    'piml_encoder_ksum': [_p, _ll, _i, _p, _p],
    'piml_decoder_pack_floats': [],
    'piml_decoder_partial_floats': [],
    'piml_decoder_workgroups': [_ll],
    'piml_decoder_fwd': [ctypes.POINTER(DecoderBranch), _i, _p, _f, _p, _p],
    'piml_decoder_bwd': [ctypes.POINTER(DecoderBranch), _i, _p, _p, _f, _p, _p],
    'piml_rowdecoder_slots': [_ll],
    'piml_rowdecoder_fwd': [ctypes.POINTER(DecoderBranch), _i, _p],
    'piml_rowdecoder_fwd_packed': [ctypes.POINTER(DecoderBranch), _i, _p],
    'piml_rowdecoder_bwd': [ctypes.POINTER(DecoderBranch), _i, _p],
    'piml_rowdecoder_bwd_acc': [ctypes.POINTER(DecoderBranch), _i, _i, _p],
    'piml_collision_head_pack_floats': [],
    'piml_collision_head_fwd': [_p, _ll, _p, _p, _p, _p, _p, _p, _p],
    'piml_head64_partial_floats': [],
    'piml_head64_slots': [_ll],
    'piml_head64_fwd': [ctypes.POINTER(Head64), _p],
    'piml_head64_bwd': [ctypes.POINTER(Head64), _p],
    'piml_head64_bwd_acc': [ctypes.POINTER(Head64), _i, _p],
    'piml_corrector_partial_floats': [_i],
    'piml_corrector_slots': [_i, _ll, _i],
    'piml_corrector_fwd': [ctypes.POINTER(Corrector), _p],
    'piml_corrector_bwd': [ctypes.POINTER(Corrector), _i, _p],
    'piml_pinnsf_pool_h2_ok': [ctypes.POINTER(EncoderBranch), _i],
    'piml_pinnsf_pool_train_ok': [ctypes.POINTER(EncoderBranch), _i],
    'piml_pinnsf_pool_msgs_ok': [ctypes.POINTER(EncoderBranch), _i],
    'piml_pinnsf_streams_init': [],
    'piml_pinnsf_pack_flush': [],
    'piml_pinnsf_slot_sums_flush': [],
    'piml_pinnsf_pack': [ctypes.POINTER(EncoderBranch), ctypes.POINTER(DecoderBranch), _i, ctypes.POINTER(CollisionHead),
                         _i, _p],
    'piml_pinnsf_fwd': [ctypes.POINTER(EncoderBranch), ctypes.POINTER(DecoderBranch), _i, ctypes.POINTER(CollisionHead),
                        _p, _f, _p, _i, _p],
    'piml_pinnsf_bwd': [ctypes.POINTER(EncoderBranch), ctypes.POINTER(DecoderBranch), _i, _p, _p, _f, _p, _i, _p],
    'piml_comm_available': [],
    'piml_comm_unique_id': [_p],
    'piml_comm_init': [ctypes.POINTER(_p), _i, _i, _p],
    'piml_comm_destroy': [_p],
    'piml_allgather_state': [_p, _p, _z, _p, _p],
    'piml_reducescatter_grad': [_p, _p, _p, _z, _p],
    'piml_allreduce_sum': [_p, _p, _z, _p],
    'piml_relfeat_fwd_self': [_p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p],
    'piml_relfeat_bwd_self': [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    'piml_relfeat_fwd_tick': [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f,
                              _p, _p, _p, _i, _p, _p, _p, _p],
    'piml_relfeat_self_fwd': [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p],
    'piml_relfeat_self_fwd_part': [_i, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _p, _p, _p, _p, _p, _p, _p],
    'piml_relfeat_self_bwd': [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    'piml_relfeat_bwd_det': [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    'piml_relfeat_bwd': [_p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p],
}


class P2PMsg(ctypes.Structure):
    """piml_p2p_msg (include/piml_hip.h)."""
    _fields_ = [('scatter_src', _p), ('scatter_floats', _z), ('out_scatter', _p), ('n_bcast', _i),
                ('bcast_src', _p * 8), ('bcast_floats', _z * 8), ('out_bcast', _p * 8), ('sum', _i)]


SIGNATURES['piml_p2p_exchange'] = [ctypes.POINTER(P2PMsg), _i, _i, ctypes.POINTER(_p), ctypes.POINTER(_p), _z, _p, ctypes.c_uint, _p, _p]


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
        L.piml_encoder_split_tiles.restype = _ll
        L.piml_encoder_split_tiles_train.restype = _ll
        L.piml_mlapm_bwd_workspace_floats.restype = _ll
        L.piml_error_string.argtypes = [_i]
        L.piml_error_string.restype = ctypes.c_char_p
        if L.piml_abi_version() != ABI_VERSION:
            raise PimlHipError(f'{LIB_PATH}: ABI {L.piml_abi_version()} != expected {ABI_VERSION}; rebuild')
        _lib = L
    return _lib


def _demangle_lite(sym):
    """`_ZN4piml23enc_fwd_x3_kernelILi0EEEvNS_7EncArgsE` -> `enc_fwd_x3_kernel<0>`: the nested name's last component and
    its literal template arguments (bool / int), which is all this library's kernels use."""
    import re
    i = 2
    if sym[i:i + 1] == 'N':
        i += 1
    name = None
    while i < len(sym) and sym[i].isdigit():
        m = re.match(r'\d+', sym[i:])
        n = int(m.group())
        name = sym[i + m.end():i + m.end() + n]
        i += m.end() + n
    if name is None:
        return sym
    if sym[i:i + 1] == 'I':
        args, i = [], i + 1
        while sym[i:i + 1] == 'L':
            m = re.match(r'L([bi])(n?\d+)E', sym[i:])
            if not m:
                break
            v = int(m.group(2).replace('n', '-'))
            args.append(('true' if v else 'false') if m.group(1) == 'b' else str(v))
            i += m.end()
        if args:
            name += '<' + ', '.join(args) + '>'
    return name


def kernel_resource_usage(path=None):
    """{kernel name: {vgprs, agprs, vgpr_spill, sgpr_spill, scratch_bytes, lds_bytes, symbol}} of every kernel IN the library
    file (default: the one this process loads), read from the code objects' own metadata -- the NT_AMDGPU_METADATA notes of
    the gfx950 ELFs in the file's offload bundles -- not from a build log.  bench.py prints the entries of the step's kernels;
    tests/test_abi.py holds every kernel the default dispatch reaches to vgpr_spill == 0."""
    import struct
    import msgpack
    blob = open(path or LIB_PATH, 'rb').read()
    magic, out, at = b'__CLANG_OFFLOAD_BUNDLE__', {}, 0
    while True:
        at = blob.find(magic, at)
        if at < 0:
            break
        n, = struct.unpack_from('<Q', blob, at + 24)
        off = at + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from('<QQQ', blob, off)
            triple = blob[off + 24:off + 24 + tl]
            off += 24 + tl
            elf = blob[at + o:at + o + size]
            if b'gfx950' not in triple or elf[:4] != b'\x7fELF':
                continue
            shoff, = struct.unpack_from('<Q', elf, 0x28)
            shentsize, shnum = struct.unpack_from('<HH', elf, 0x3A)
            for k in range(shnum):
                sh_type, = struct.unpack_from('<I', elf, shoff + k * shentsize + 4)
                if sh_type != 7:      # SHT_NOTE
                    continue
                q, sz = struct.unpack_from('<QQ', elf, shoff + k * shentsize + 0x18)
                end = q + sz
                while q + 12 <= end:
                    namesz, descsz, typ = struct.unpack_from('<III', elf, q)
                    q += 12 + ((namesz + 3) & ~3)
                    desc = elf[q:q + descsz]
                    q += (descsz + 3) & ~3
                    if typ != 32:     # NT_AMDGPU_METADATA
                        continue
                    for kd in msgpack.unpackb(desc, raw=False).get('amdhsa.kernels', []):
                        out[_demangle_lite(kd['.name'])] = {
                            'vgprs': kd.get('.vgpr_count'), 'agprs': kd.get('.agpr_count'),
                            'vgpr_spill': kd.get('.vgpr_spill_count'), 'sgpr_spill': kd.get('.sgpr_spill_count'),
                            'scratch_bytes': kd.get('.private_segment_fixed_size'),
                            'lds_bytes': kd.get('.group_segment_fixed_size'), 'symbol': kd['.name']}
        at += len(magic)
    return out


def check(err, what):
    if err != 0:
        raise PimlHipError(f'{what} failed: hipError {err} ({lib().piml_error_string(err).decode()})')


class StageTrace:
    """piml_trace_*: live time of every launch stage between `start()` and `stop()` on torch's current stream.
    stop() -> [(stage name, microseconds), ...] in launch order."""
    _START = ctypes.c_char_p(b'start')

    def start(self):
        check(lib().piml_trace_begin(), 'piml_trace_begin')
        lib().piml_trace_mark(self._START, torch.cuda.current_stream().cuda_stream)

    def stop(self):
        names = ctypes.create_string_buffer(4096)
        us = (_f * 64)()
        n = lib().piml_trace_end(names, 4096, us, 64)
        if n < 0:
            check(-n, 'piml_trace_end')
        return list(zip(names.value.decode().split('\n')[:n], [float(us[i]) for i in range(n)]))


class StreamTimer:
    """A pair of HIP events recorded on torch's current stream through the C ABI; works
    inside HIP-graph capture (torch.cuda.Event(external=True) is refused on ROCm)."""

    def __init__(self):
        self._e = [_p(), _p()]
        for e in self._e:
            check(lib().piml_timer_create(ctypes.byref(e)), 'piml_timer_create')

    def start(self):
        check(lib().piml_timer_record(self._e[0], torch.cuda.current_stream().cuda_stream), 'piml_timer_record')

    def stop(self):
        check(lib().piml_timer_record(self._e[1], torch.cuda.current_stream().cuda_stream), 'piml_timer_record')

    def elapsed_ms(self):
        ms = _f()
        check(lib().piml_timer_elapsed_ms(self._e[0], self._e[1], ctypes.byref(ms)), 'piml_timer_elapsed_ms')
        return float(ms.value)

    def __del__(self):
        try:
            for e in self._e:
                lib().piml_timer_destroy(e)
        except Exception:   # interpreter shutdown
            pass
