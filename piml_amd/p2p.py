"""P2P-store all-gather of the state records (include/piml_hip.h: piml_p2p_*, piml_allgather_state_p2p).

Host side: one `P2PExchange` per rank.  It owns this rank's receive buffer ([parity 2][sender world][floats_per_rank]) and flag
words, exports them as 64-byte IPC handles, opens the peers' (the handles travel through any byte channel the host has: a
pipe, the torch.distributed store) and runs exchange steps on torch's current stream.  The gathered records of a step are a
device pointer into the receive buffer (`gathered_ptr`), or a torch tensor through `gather_into` (one device copy).

The reference has no multi-process path at all (nn.DataParallel, src/models/simulators.py:64-67); the default exchange of
this package is the RCCL all-gather (piml_amd/sharded.py)."""
import ctypes

import torch

from . import _lib


class P2PExchange:
    def __init__(self, rank, world, floats_per_rank):
        if floats_per_rank % 4:
            raise ValueError('floats_per_rank must be a multiple of 4 (16-byte stores)')
        self.rank, self.world, self.fpr = int(rank), int(world), int(floats_per_rank)
        L = _lib.lib()
        self._recv, self._flags = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(L.piml_p2p_alloc(2 * world * floats_per_rank * 4, ctypes.byref(self._recv)), 'piml_p2p_alloc')
        _lib.check(L.piml_p2p_alloc(2 * world * 4, ctypes.byref(self._flags)), 'piml_p2p_alloc')
        self._peer_recv = (ctypes.c_void_p * world)()
        self._peer_flags = (ctypes.c_void_p * world)()
        self._peer_recv[rank], self._peer_flags[rank] = self._recv, self._flags
        self._opened = []
        self.status = torch.zeros(1, dtype=torch.int32, device='cuda')
        self.seq = 0

    def handles(self):
        """(recv handle, flags handle) as bytes, for the peers."""
        L = _lib.lib()
        out = []
        for ptr in (self._recv, self._flags):
            h = ctypes.create_string_buffer(64)
            _lib.check(L.piml_p2p_export(ptr, h), 'piml_p2p_export')
            out.append(h.raw)
        return tuple(out)

    def connect(self, peer, handles):
        L = _lib.lib()
        for table, raw in zip((self._peer_recv, self._peer_flags), handles):
            p = ctypes.c_void_p()
            _lib.check(L.piml_p2p_open(ctypes.create_string_buffer(raw, 64), ctypes.byref(p)), 'piml_p2p_open')
            table[peer] = p
            self._opened.append(p)

    def step(self, own, spin_limit=0):
        """own: this rank's (floats_per_rank) float32 block on the device.  Enqueues one exchange step on the current stream and
        returns the device pointer of the gathered (world * floats_per_rank) floats; `ok()` tells (synchronising) whether
        every peer arrived."""
        assert own.is_cuda and own.dtype == torch.float32 and own.numel() == self.fpr and own.is_contiguous()
        self.seq += 1
        L = _lib.lib()
        _lib.check(L.piml_allgather_state_p2p(own.data_ptr(), self.fpr, self.rank, self.world, self._peer_recv, self._peer_flags,
                                              self.seq, spin_limit, self.status.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   'piml_allgather_state_p2p')
        return self.gathered_ptr()

    def gathered_ptr(self):
        return self._recv.value + (self.seq & 1) * self.world * self.fpr * 4

    def gather_into(self, out):
        """copy the last step's gathered records into a torch tensor (world * floats_per_rank floats)"""
        assert out.is_cuda and out.dtype == torch.float32 and out.numel() == self.world * self.fpr and out.is_contiguous()
        _lib.check(_lib.lib().piml_p2p_copy(out.data_ptr(), self.gathered_ptr(), out.numel() * 4, torch.cuda.current_stream().cuda_stream),
                   'piml_p2p_copy')
        return out

    def ok(self):
        return int(self.status.item()) == 0

    def close(self):
        L = _lib.lib()
        torch.cuda.synchronize()
        for p in self._opened:
            L.piml_p2p_close(p)
        self._opened = []
        for p in (self._recv, self._flags):
            if p:
                L.piml_p2p_free(p)
        self._recv = self._flags = ctypes.c_void_p()
