python -m pytest tests/ -x -q -m gpu 2>&1 | grep -n "passed\|failed\|Error" | head
