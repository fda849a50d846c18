#!/bin/bash
# round 6: the fine-tuning step (HOT LOOP C, 4 x 5 x 122, pinnsf_m, dropout 0): time, kernels per step (ordered list with gaps), tests
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_simulator_gpu.py tests/test_main_gpu.py tests/test_graph_gpu.py -x -q 2>&1 | tail -3
timeout 300 python tools/time_finetune.py 2>&1 | grep "fine-tune"
bash tools/r5_ft_trace.sh 2>&1 | tail -2
cp gpurun_out/r5ft/step.txt gpurun_out/r6_ft_step.txt; cut -c1-110 gpurun_out/r6_ft_step.txt | tail -70
