#!/bin/bash
# round-5 measurement suite: training loops at the reference's sizes, models at cfg3, rollouts; outputs under gpurun_out/r5_suite/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_suite; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python tools/train_mode_steps.py > $O/train_mode_steps.log 2>&1
timeout 600 python tools/train_mode_steps.py --dropout 0.0 > $O/train_mode_steps_p0.log 2>&1
timeout 600 python tools/time_models.py > $O/time_models.log 2>&1
timeout 600 python tools/time_rollout.py > $O/time_rollout.log 2>&1
timeout 300 python tools/time_finetune.py > $O/time_finetune.log 2>&1
for f in train_mode_steps train_mode_steps_p0 time_models time_rollout time_finetune; do echo "== $f"; grep -v -i "warn\|amdgpu.ids\|^$" $O/$f.log | tail -25; done
