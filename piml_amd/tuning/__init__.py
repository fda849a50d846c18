"""Pre-tuned GEMM selections for the PINNSF MLP (PyTorch-ROCm TunableOp result files).

`tunableop_gfx950_cfg3.csv` was produced once on an MI355X with PYTORCH_TUNABLEOP_TUNING=1 over the
GEMM shapes of the 4096-agent step (24 576 / 40 960 / 4 096 rows x 6 / 64 / 128 columns, forward
and backward); tuning takes minutes and is never repeated at run time.  torch ignores a file whose
validators (torch / ROCm / hipBLASLt / rocBLAS versions, GPU architecture) do not match.

Re-tuning on a new software stack (minutes, once):  python bench.py --tunableop 2 --tune-out <file.csv>
runs a few eager steps of the benchmark with tuning enabled (`tune_begin` below) and torch writes the result file
at exit; copy it over `tunableop_gfx950_cfg3.csv` (or pass its path to `load`)."""
import os
import sys

import torch

LOADED = False      # True once torch accepted a result file: the tuned selections are in effect

DEFAULT_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tunableop_gfx950_cfg3.csv')


def load(path=DEFAULT_FILE):
    """Enable TunableOp in look-up-only mode with the given result file.  Returns True when torch
    accepted the file (then, and only then, its selections are in effect)."""
    if not os.path.exists(path):
        return False
    try:
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)
        torch.cuda.tunable.record_untuned_enable(False)
        if hasattr(torch.cuda.tunable, 'write_file_on_exit'):
            torch.cuda.tunable.write_file_on_exit(False)      # look-up only: leave no tunableop*.csv behind
        global LOADED
        LOADED = bool(torch.cuda.tunable.read_file(path))
        return LOADED
    except Exception as ex:   # noqa: BLE001 - a missing / changed API means: stay on the defaults
        print(f'[piml_amd] TunableOp results not loaded ({ex})', file=sys.stderr)
        return False


def tune_begin(path, max_duration_ms=30, max_iterations=100):
    """Enable TunableOp WITH tuning: every GEMM shape met from now on is timed over the available rocBLAS /
    hipBLASLt solutions and the winners are written to `path` when the process exits.  The chunked
    weight-gradient formulation (ops._weight_grad_chunks) is switched on so that its strided-batched shapes are
    tuned as well."""
    global LOADED
    torch.cuda.tunable.enable(True)
    torch.cuda.tunable.tuning_enable(True)
    torch.cuda.tunable.set_max_tuning_duration(int(max_duration_ms))
    torch.cuda.tunable.set_max_tuning_iterations(int(max_iterations))
    torch.cuda.tunable.set_filename(path)
    LOADED = True
    return path
