#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats of the default bench command, plus separate PMC passes
# (never combined with --stats / sys traces) for HBM traffic and SIMD / matrix-pipe occupancy of the step's kernels.
# Outputs under gpurun_out/profile_round/; digest with tools/make_step_counters.py <tag>.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/profile_round; rm -rf $O; mkdir -p $O
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py $ARGS > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
# the MLAPM kernels (secondary figures of the bench line): SIMD occupancy in a pass of their own
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/mlapm -- python3 $R/tools/mlapm_gc_4096.py > $O/mlapm.log 2>&1
python3 $R/bench.py --cpu-seconds 0 > $O/bench.json 2> $O/bench.err
tail -c 300 $O/stats.log
