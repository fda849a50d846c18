"""piml_amd -- MI355X (gfx950) implementation of PIML's per-timestep pairwise hot path.

Host side mirrors the reference's operator API (`Pedestrians`, `MLAPM`, the PINNSF models,
`BaseSimulator` rollouts); the arithmetic runs in hand-written HIP kernels behind the C ABI
declared in include/piml_hip.h (libpiml_hip.so).  There is no CPU fallback: every operator
raises if the library is missing or a tensor is not on the GPU.
"""
__all__ = ['scenes']

import os as _os

# HIP-graph replays on this ROCm stack (7.0 / CLR "graph packet capture") mis-order memset nodes against the kernels
# that follow them once another graph or eager work ran in between: torch's multi-block reductions (they zero their
# semaphores with hipMemsetAsync) then return garbage from a replayed graph -- reproduced by tools/probe_graph_memset.py
# and pinned by tests/test_graph_gpu.py.  With the packet capture off the replay is correct at the same speed.
# Must be set before the HIP runtime initialises (first GPU call); an explicit user setting wins.
_os.environ.setdefault('DEBUG_CLR_GRAPH_PACKET_CAPTURE', '0')
