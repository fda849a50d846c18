#!/bin/bash
# experiment build: a variant library with -D flags, profiled with the bench step (results may be wrong: timing only)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3exp; rm -rf $O; mkdir -p $O
cp piml_amd/libpiml_hip.so /tmp/lib_orig.so
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden $EXP_FLAGS -o piml_amd/libpiml_hip.so piml_amd/csrc/*.hip 2> $O/build.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0 --verify 0 > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cp /tmp/lib_orig.so $GRAFT_REPO_ROOT/piml_amd/libpiml_hip.so
