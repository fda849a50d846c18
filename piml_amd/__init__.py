"""piml_amd -- MI355X (gfx950) implementation of PIML's per-timestep pairwise hot path.

Host side mirrors the reference's operator API (`Pedestrians`, `MLAPM`, the PINNSF models,
`BaseSimulator` rollouts); the arithmetic runs in hand-written HIP kernels behind the C ABI
declared in include/piml_hip.h (libpiml_hip.so).  There is no CPU fallback: every operator
raises if the library is missing or a tensor is not on the GPU.
"""
__all__ = ['scenes']

import os as _os
import warnings as _warnings

import torch as _torch

# HIP-graph replays on this ROCm stack (7.0 / CLR "graph packet capture") mis-order memset nodes against the kernels
# that follow them once another graph or eager work ran in between: torch's multi-block reductions (they zero their
# semaphores with hipMemsetAsync) then return garbage from a replayed graph -- reproduced by tools/probe_graph_memset.py
# and pinned by tests/test_graph_gpu.py.  With the packet capture off the replay is correct at the same speed.
# The variable is read when the HIP runtime initialises (first GPU call), so it only helps when it is set before that.
_GRAPH_ENV = 'DEBUG_CLR_GRAPH_PACKET_CAPTURE'


def _graph_env_state(value_at_import, hip_initialised):
    """(safe, reason): can this process trust replayed HIP graphs?  Pure function of what import found (unit-tested)."""
    if value_at_import == '0':
        return True, None
    if value_at_import is not None:
        return False, f'{_GRAPH_ENV}={value_at_import!r} was set explicitly: replayed HIP graphs with memset nodes return garbage on this stack'
    if hip_initialised:
        return False, (f'the HIP runtime was initialised before piml_amd was imported, so {_GRAPH_ENV}=0 can no longer take '
                       'effect: set it in the environment (or import piml_amd) before the first GPU call')
    return True, None


_GRAPHS_SAFE, _GRAPHS_WHY = _graph_env_state(_os.environ.get(_GRAPH_ENV), _torch.cuda.is_initialized())
_os.environ.setdefault(_GRAPH_ENV, '0')
if not _GRAPHS_SAFE:
    _warnings.warn(f'piml_amd: {_GRAPHS_WHY}.  HIP-graph capture is DISABLED in this process (rollouts, fine-tuning steps and '
                   'bench.py run eagerly); PIML_TRUST_HIP_GRAPHS=1 overrides.', RuntimeWarning, stacklevel=2)


def hip_graphs_safe():
    """False when this process must not replay captured HIP graphs (see above).  Every capture site of the package
    (BaseSimulator rollouts / fine-tuning steps, MLAPM.rollout, bench.py) asks here and falls back to eager execution."""
    return _GRAPHS_SAFE or _os.environ.get('PIML_TRUST_HIP_GRAPHS') == '1'
