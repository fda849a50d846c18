python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -s 2>&1 | grep -n "passed\|failed\|row_decoder\|row decoder\|Error" | head
python tools/time_models.py 2>&1 | grep "ms/step"
