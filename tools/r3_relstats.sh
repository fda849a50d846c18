#!/bin/bash
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -DPIML_RELFEAT_STATS -o tools/libpiml_hip_stats.so piml_amd/csrc/*.hip 2> gpurun_out/r3_relstats_build.log
python tools/relfeat_stats.py > gpurun_out/r3_relstats.log 2>&1
