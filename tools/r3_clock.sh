#!/bin/bash
# shader clock while the bench step replays (is the chip at its 2.4 GHz while MFMA + HBM are both busy?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3clock; rm -rf $O; mkdir -p $O
rocm-smi --showclocks --showpower > $O/idle.txt 2>&1
python bench.py --cpu-seconds 0 --secondary 0 --verify 0 --steps 500000 --warmup 100 > $O/bench.json 2>/dev/null &
BP=$!
sleep 50
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|socclk" >> $O/busy.txt; echo --- >> $O/busy.txt; sleep 1; done
amd-smi metric --clock --power > $O/amdsmi.txt 2>&1
wait $BP
