python -m pytest tests/test_encoder_gpu.py -x -q -m gpu -k "split_tile" 2>&1 | grep -n "passed\|failed\|Error" | head -5
python -m pytest tests/ -x -q -m gpu 2>&1 | grep "passed\|failed"
for t in 0 1024 100000; do echo "split<=$t:"; PIML_ENC_SPLIT_TILES=$t python tools/time_rollout.py 2>&1 | grep "graph=True"; done
