python -m pytest tests/test_encoder_gpu.py tests/test_mlpglue_gpu.py tests/test_simulator_gpu.py -x -q -m gpu 2>&1 | grep -n "passed\|failed" 
