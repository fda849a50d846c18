#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3dxs; rm -rf $O; mkdir -p $O
python -m pytest tests/test_encoder_gpu.py tests/test_dropout_gpu.py tests/test_simulator_gpu.py tests/test_main_gpu.py -q 2>&1 | grep -E "passed|failed|Error|assert" > $O/tests.log
python tools/train_mode_steps.py --models pinnsf_m,pinnsf_bm 2>&1 | grep "fine-tuning" > $O/ft_x3.log
PIML_ENC_DX_SPLIT=f32 python tools/train_mode_steps.py --models pinnsf_m,pinnsf_bm 2>&1 | grep "fine-tuning" > $O/ft_f32.log
python tools/time_encoder.py > $O/enc_x3.log 2>&1
PIML_ENC_DX_SPLIT=f32 python tools/time_encoder.py > $O/enc_f32.log 2>&1
