"""P2P-store all-gather of the state records (include/piml_hip.h: piml_p2p_*, piml_allgather_state_p2p).

Host side: one `P2PExchange` per rank.  It owns this rank's receive buffer ([parity 2][sender world][floats_per_rank]) and flag
words, exports them as 64-byte IPC handles, opens the peers' (the handles travel through any byte channel the host has: a
pipe, the torch.distributed store) and runs exchange steps on torch's current stream.  The gathered records of a step are a
device pointer into the receive buffer (`gathered_ptr`), or a torch tensor through `gather_into` (one device copy).

The reference has no multi-process path at all (nn.DataParallel, src/models/simulators.py:64-67); the default exchange of
this package is the RCCL all-gather (piml_amd/sharded.py)."""
import ctypes
import os

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
        # the general exchange (`exchange`): step counter and workgroup counters on the device, zeroed once
        self.ctr = torch.zeros(3 + world, dtype=torch.int32, device='cuda')

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

    def exchange(self, scatter_src=None, bcast_src=None, out_scatter=None, out_bcast=None, sum=False, spin_limit=0):
        """One step of the general exchange (piml_p2p_exchange: device-side step counter, capturable into a HIP graph) on the
        current stream.  scatter_src: (world * n_s) floats, receiver r gets block r.  bcast_src: a tensor, or a list of up to 8
        tensors, every receiver gets all of them; the parts together must fit floats_per_rank (the slot size given to the
        constructor).  sum=False: out_scatter (world * n_s) / out_bcast[j] (world * n_j) receive the senders' parts in rank order;
        sum=True: out_scatter (n_s) / out_bcast[j] (n_j) their sums in rank order -- out_bcast[j] may BE bcast_src[j] (in place).
        `ok()` (synchronising) tells whether every peer arrived; after a time-out the exchange stays dead."""
        def flat(t, name, n=None):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError(f'P2PExchange.exchange: {name} must be a contiguous float32 GPU tensor')
            if n is not None and t.numel() != n:
                raise ValueError(f'P2PExchange.exchange: {name} has {t.numel()} elements, expected {n}')
            return t
        as_list = lambda x: [] if x is None else (list(x) if isinstance(x, (list, tuple)) else [x])
        pad4 = lambda n: (n + 3) // 4 * 4
        ok16 = lambda t: t.data_ptr() % 16 == 0
        srcs, outs = as_list(bcast_src), as_list(out_bcast)
        if outs and len(outs) != len(srcs):
            raise ValueError('P2PExchange.exchange: one out_bcast per bcast_src (or none)')
        if len(srcs) > 8:
            raise ValueError('P2PExchange.exchange: at most 8 broadcast parts')
        W = self.world
        mult = 1 if sum else W
        # The kernel moves 16-byte words: every part is a multiple of 4 floats on a 16-byte boundary.  A part that is not (an odd
        # agent block: 501 rows x 6 floats; a row slice of a larger tensor) travels through a padded, aligned staging buffer --
        # `copy_back` holds the (destination, staged view) pairs copied out behind the launch.  (Extra launches, off the fast path.)
        copy_back = []

        def stage_src(t, blocks):          # (blocks, n) floats -> (blocks, pad4(n)), zero-padded
            n = t.numel() // blocks
            if n % 4 == 0 and ok16(t):
                return t, n
            buf = torch.zeros(blocks, pad4(n), dtype=torch.float32, device=t.device)
            buf[:, :n].copy_(t.view(blocks, n))
            return buf, pad4(n)

        def stage_out(t, blocks, n, staged_src=None):      # destination of (blocks, n) floats
            if n % 4 == 0 and ok16(t) and staged_src is None:
                return t
            buf = staged_src if (staged_src is not None and blocks == 1) else torch.empty(blocks, pad4(n), dtype=torch.float32, device=t.device)
            copy_back.append((t, buf.view(blocks, -1)[:, :n]))
            return buf

        ns = 0 if scatter_src is None else flat(scatter_src, 'scatter_src').numel() // W
        if scatter_src is not None and scatter_src.numel() != ns * W:
            raise ValueError('P2PExchange.exchange: scatter_src must hold one block per rank')
        if out_scatter is not None:
            flat(out_scatter, 'out_scatter', ns * mult)
        m = _lib.P2PMsg()
        keep = []                             # staging buffers stay alive until the launch is enqueued (stream-ordered allocator)
        ns_k = 0
        if ns:
            sc, ns_k = stage_src(scatter_src, W)
            keep.append(sc)
            m.scatter_src = sc.data_ptr()
        m.scatter_floats = ns_k
        if out_scatter is not None:
            o = stage_out(out_scatter, mult, ns)
            keep.append(o)
            m.out_scatter = o.data_ptr()
        m.n_bcast, total = len(srcs), ns_k
        for j, t in enumerate(srcs):
            flat(t, f'bcast_src[{j}]')
            n = t.numel()
            in_place = bool(outs) and outs[j] is not None and outs[j].data_ptr() == t.data_ptr()
            src, n_k = stage_src(t, 1)
            keep.append(src)
            m.bcast_src[j], m.bcast_floats[j] = src.data_ptr(), n_k
            if outs and outs[j] is not None:
                flat(outs[j], f'out_bcast[{j}]', n * mult)
                if in_place and not sum:
                    raise ValueError('P2PExchange.exchange: an in-place out_bcast needs sum=True')
                o = stage_out(outs[j], mult, n, staged_src=src if (in_place and src is not t) else None)
                keep.append(o)
                m.out_bcast[j] = o.data_ptr()
            total += n_k
        if total == 0 or total > self.fpr:
            raise ValueError(f'P2PExchange.exchange: parts of {total} floats in all (each padded to a multiple of 4) must fit the slot of {self.fpr}')
        m.sum = 1 if sum else 0
        if not spin_limit:      # PIML_P2P_SPIN_LIMIT: rounds of ~0.5 us a wait may take (default ~0.5 s; hosts whose ranks start far apart raise it)
            spin_limit = int(os.environ.get('PIML_P2P_SPIN_LIMIT', '0'))
        _lib.check(_lib.lib().piml_p2p_exchange(ctypes.byref(m), self.rank, self.world, self._peer_recv, self._peer_flags, self.fpr,
                                                self.ctr.data_ptr(), int(spin_limit), self.status.data_ptr(),
                                                torch.cuda.current_stream().cuda_stream), 'piml_p2p_exchange')
        for dst, view in copy_back:
            dst.view(view.shape).copy_(view)

    def connect_all(self, exchange_bytes):
        """Open every peer's buffers.  exchange_bytes(own: bytes) -> list of every rank's bytes in rank order (e.g. a
        torch.distributed all_gather_object, or pipes between the processes)."""
        mine = b''.join(self.handles())
        for peer, raw in enumerate(exchange_bytes(mine)):
            if peer != self.rank:
                self.connect(peer, (raw[:64], raw[64:128]))

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
