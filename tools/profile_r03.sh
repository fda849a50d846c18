#!/bin/bash
# Round 3, on the GPU box (via gpurun): everything profiles/r03_* is made from.
#   1. tools/profile_round.sh: kernel stats + separate PMC passes of the default bench step (eval mode) + MLAPM PMC
#   2. the same step in the reference's TRAINING configuration (model.train(), dropout 0.5): kernel stats
#   3. the two training loops at dropout 0.5 (pointwise pre-training, fine-tuning rollout): kernel stats + timings
#   4. timing tools of the other kernels / models / rollouts
#   5. the driver's own command line
# Digest: tools/make_step_counters.py r03 ; tools/make_r03_notes.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh > /dev/null 2>&1
O=$R/gpurun_out/profile_r03; rm -rf $O; mkdir -p $O
ARGS="--steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 --secondary 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_step -- python3 $R/bench.py $ARGS --train-mode 1 > $O/train_step.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loops -- python3 $R/tools/train_mode_steps.py --models pinnsf_m --reps 20 > $O/loops_prof.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/loops_bm -- python3 $R/tools/train_mode_steps.py --models pinnsf_bm --reps 20 > $O/loops_bm_prof.log 2>&1
cd $R
python3 tools/train_mode_steps.py --with-eval > $O/train_mode_steps.log 2>&1
python3 tools/time_models.py > $O/time_models.log 2>&1
python3 tools/time_rollout.py > $O/time_rollout.log 2>&1
python3 tools/time_pairwise.py > $O/time_pairwise.log 2>&1
python3 tools/time_relfeat.py > $O/time_relfeat.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --train-mode 1 --cpu-seconds 0 --secondary 0 > $O/bench_driver_cmd_train.json 2>/dev/null
for d in train_step loops loops_bm; do cp $(ls $O/$d/*/*kernel_stats.csv | head -1) $O/${d}_kernel_stats.csv; rm -rf $O/$d; done
