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


_EXPORTED_BY_LAUNCHER = _os.environ.get(_GRAPH_ENV) == '0'
_GRAPHS_SAFE, _GRAPHS_WHY = _graph_env_state(_os.environ.get(_GRAPH_ENV), _torch.cuda.is_initialized())
_os.environ.setdefault(_GRAPH_ENV, '0')
if not _GRAPHS_SAFE:
    _warnings.warn(f'piml_amd: {_GRAPHS_WHY}.  HIP-graph capture is DISABLED in this process (rollouts, fine-tuning steps and '
                   'bench.py run eagerly); PIML_TRUST_HIP_GRAPHS=1 overrides.', RuntimeWarning, stacklevel=2)


_PROBED = None


def _probe_graph_replay():
    """One small capture / replay self-test of the hazard itself (tools/probe_graph_memset.py): graph A holds a multi-block
    reduction (torch zeroes its semaphore with a memset node), a second graph is captured and replayed, then A must still
    return the right sum.  ~50 ms, once per process, only when nobody can say whether the variable took effect."""
    try:
        dev = _torch.device('cuda', _torch.cuda.current_device())
        x = (_torch.arange(1 << 20, device=dev, dtype=_torch.float32) % 7.0)
        want = int((x > 2.5).sum())

        def capture(fn):
            s = _torch.cuda.Stream()
            s.wait_stream(_torch.cuda.current_stream())
            with _torch.cuda.stream(s):
                for _ in range(2):
                    fn()
            _torch.cuda.current_stream().wait_stream(s)
            _torch.cuda.synchronize()
            g = _torch.cuda.CUDAGraph()
            with _torch.cuda.graph(g):
                out = fn()
            return g, out
        ga, outa = capture(lambda: (x > 2.5).sum())
        y = x * 0.5
        gb, outb = capture(lambda: ((y * 2).sum(), (y > 0.1).sum()))
        for _ in range(3):
            gb.replay()
        for _ in range(3):
            (_torch.rand(1 << 20, device=dev) > 0.3).sum()
        ok = True
        for _ in range(3):
            ga.replay()
            _torch.cuda.synchronize()
            ok = ok and int(outa) == want
            gb.replay()
        return bool(ok)
    except Exception:   # noqa: BLE001 - a probe that cannot run decides nothing
        return None


def hip_graphs_safe():
    """False when this process must not replay captured HIP graphs (see above).  Every capture site of the package
    (BaseSimulator rollouts / fine-tuning steps, MLAPM.rollout, bench.py) asks here and falls back to eager execution.

    `torch.cuda.is_initialized()` only sees torch's own lazy initialisation: torch.cuda.is_available(), another HIP library
    or a profiler preload can bring the runtime up earlier, and then the variable this module sets comes too late without
    anybody noticing.  So unless the launcher exported it (`DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` in the environment before the
    process started -- bench.py, `python -m piml_amd.main` and the tools do that for themselves), the first call on a GPU
    runs the hazard's own reproduction once (`_probe_graph_replay`) and believes that."""
    global _PROBED
    if _os.environ.get('PIML_TRUST_HIP_GRAPHS') == '1':
        return True
    if not _GRAPHS_SAFE:
        return False
    if _EXPORTED_BY_LAUNCHER or not _torch.cuda.is_available():
        return True
    if _PROBED is None and not _torch.cuda.is_current_stream_capturing():
        _PROBED = _probe_graph_replay()
        if _PROBED is False:
            _warnings.warn(f'piml_amd: a replayed HIP graph returned a wrong reduction in this process ({_GRAPH_ENV}=0 came too '
                           'late: the HIP runtime was already up when piml_amd was imported).  HIP-graph capture is DISABLED; '
                           f'export {_GRAPH_ENV}=0 before starting the process.', RuntimeWarning, stacklevel=2)
    return _PROBED is not False
