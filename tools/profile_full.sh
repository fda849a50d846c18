#!/bin/bash
# On the GPU box (via gpurun): everything profiles/<round>_* is made from (rounds 3 and 4).
#   1. tools/profile_round.sh: kernel stats + separate PMC passes of the default bench step (eval mode) + MLAPM PMC
#   2. the same step in the reference's TRAINING configuration (model.train(), dropout 0.5): kernel stats
#   3. the two training loops at dropout 0.5 (pointwise pre-training, fine-tuning rollout): kernel stats + timings
#   4. timing tools of the other kernels / models / rollouts
#   5. the driver's own command line
#   6. (round 4) kernels per fine-tuning step (pinnsf_m / pinnsf_bm) and the kernel mix of the pinnsf_res step
# Digest: tools/make_step_counters.py r04 ; tools/make_round_notes.py r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh > /dev/null 2>&1
O=$R/gpurun_out/profile_full; rm -rf $O; mkdir -p $O
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
cd /tmp
for MODEL in pinnsf_m pinnsf_bm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ft_$MODEL -- python3 $R/tools/train_mode_steps.py --models $MODEL --reps 200 --finetune-only > $O/ft_$MODEL.log 2>&1
  cp $(ls $O/ft_$MODEL/*/*kernel_stats.csv | head -1) $O/finetune_step_${MODEL}_kernel_stats.csv; rm -rf $O/ft_$MODEL
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/res -- python3 $R/tools/time_res.py > $O/res.log 2>&1
cp $(ls $O/res/*/*kernel_stats.csv | head -1) $O/pinnsf_res_kernel_stats.csv; rm -rf $O/res
# (round 6) the fine-tuning / pointwise steps as ordered kernel lists, the loop timed synchronously and with train()'s lookahead,
# the row decoder's many-rows kernels in both arithmetic forms
cd $R
for MODEL in pinnsf_m pinnsf_bm; do
  FT_MODEL=$MODEL bash tools/r5_ft_trace.sh > /dev/null 2>&1; cp gpurun_out/r5ft/step.txt $O/finetune_step_${MODEL}.txt
  MODEL=$MODEL P=0 bash tools/r6_pw_trace.sh > /dev/null 2>&1; cp gpurun_out/r6pw/step.txt $O/pointwise_step_${MODEL}.txt
  python3 tools/time_finetune.py 100 $MODEL 2>&1 | grep fine-tune >> $O/time_finetune.log
done
bash tools/prof_script.sh tools/time_rowdec.py 2>&1 | grep rowdec > $O/time_rowdec.log
