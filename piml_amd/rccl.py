"""RCCL through the C ABI (include/piml_hip.h: piml_comm_* / piml_allgather_state / piml_reducescatter_grad /
piml_allreduce_sum): the per-step exchange of agent-block sharding as plain calls on the caller's stream, for hosts
that are not PyTorch.  Inside PyTorch `torch.distributed` (backend "nccl" = RCCL) remains
the default transport of piml_amd.sharded; `DirectComm` is the same exchange on a communicator owned by
libpiml_hip.so, created from a torch.distributed group (the 128-byte id travels through the group's store)."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib


class DirectComm:
    def __init__(self, group=None):
        L = _lib.lib()
        if not L.piml_comm_available():
            raise _lib.PimlHipError('librccl is not available in this process')
        if dist.is_initialized():
            group = group if group is not None else dist.group.WORLD
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        buf = (ctypes.c_char * 128)()
        if self.rank == 0:
            _lib.check(L.piml_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)), 'piml_comm_unique_id')
        if self.world > 1:
            box = [bytes(buf.raw) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0), group=group)
            buf.raw = box[0]
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(torch.cuda.current_device()):
            _lib.check(L.piml_comm_init(ctypes.byref(self._comm), self.world, self.rank,
                                        ctypes.cast(buf, ctypes.c_void_p)), 'piml_comm_init')

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def all_gather_into(self, full, own):
        """own (n, w) -> full (world * n, w), float32, contiguous."""
        own = own.detach().contiguous()
        if full.numel() != own.numel() * self.world or full.dtype != torch.float32 or not full.is_contiguous():
            raise ValueError('all_gather_into: full must be contiguous float32 with world * own.numel() elements')
        _lib.check(_lib.lib().piml_allgather_state(self._comm, own.data_ptr(), own.numel(), full.data_ptr(),
                                                   self._stream()), 'piml_allgather_state')
        return full

    def reduce_scatter(self, full, out=None):
        """full (world * n, w) partial sums -> this rank's (n, w) rows summed over the ranks."""
        full = full.detach().contiguous()
        n = full.shape[0] // self.world
        if out is None:
            out = torch.empty((n,) + tuple(full.shape[1:]), device=full.device, dtype=torch.float32)
        _lib.check(_lib.lib().piml_reducescatter_grad(self._comm, full.data_ptr(), out.data_ptr(), out.numel(),
                                                      self._stream()), 'piml_reducescatter_grad')
        return out

    def all_reduce(self, buf):
        _lib.check(_lib.lib().piml_allreduce_sum(self._comm, buf.data_ptr(), buf.numel(), self._stream()),
                   'piml_allreduce_sum')
        return buf

    def close(self):
        if self._comm:
            _lib.lib().piml_comm_destroy(self._comm)
            self._comm = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:      # interpreter shutdown
            pass
