"""piml_amd -- MI355X (gfx950) implementation of PIML's per-timestep pairwise hot path.

Host side mirrors the reference's operator API (`Pedestrians`, `MLAPM`, the PINNSF models,
`BaseSimulator` rollouts); the arithmetic runs in hand-written HIP kernels behind the C ABI
declared in include/piml_hip.h (libpiml_hip.so).  There is no CPU fallback: every operator
raises if the library is missing or a tensor is not on the GPU.
"""
__all__ = ['scenes']
