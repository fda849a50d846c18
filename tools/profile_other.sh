#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the other HIP kernels' timing tools.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_other; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pw -- python3 $R/tools/time_pairwise.py > $O/pw.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ro -- python3 $R/tools/time_rollout.py > $O/ro.log 2>&1
cp $(ls $O/prof_pw/*/*kernel_stats.csv | head -1) $O/pw_kernel_stats.csv
cp $(ls $O/prof_ro/*/*kernel_stats.csv | head -1) $O/ro_kernel_stats.csv
rm -rf $O/prof_pw $O/prof_ro
grep -h "MLAPM\|collision\|steps/s" $O/pw.log $O/ro.log | tail -20
