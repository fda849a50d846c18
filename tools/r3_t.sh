#!/bin/bash
# a subset of the GPU tests with full output: $1 = pytest -k expression
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3t; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s -k "$1" 2>&1 | grep -vE "NCCL|RCCL|rccl" | tail -60 > $O/tests.log
