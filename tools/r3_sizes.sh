#!/bin/bash
# the bench step at other scene sizes and through the sharded code path, this build against the earlier kernel choices
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3sizes; rm -rf $O; mkdir -p $O
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', round(b['ms_per_step'],4), b.get('verified_max_rel_err'), b['launch_mode'])"; }
for a in "--agents 1024" "--agents 2048" "--agents 3000" "--agents 8192" "--agents 4096 --force-dist 1" "--agents 4096 --train-mode 1"; do
  python bench.py --cpu-seconds 0 --secondary 0 $a 2>$O/err.log | line "new[$a]" >> $O/ab.log || tail -3 $O/err.log >> $O/ab.log
  PIML_ENC_DW2=0 PIML_H1_RECOMPUTE=0 PIML_DEC_BWD_SPLIT=0 python bench.py --cpu-seconds 0 --secondary 0 $a 2>$O/err.log | line "old[$a]" >> $O/ab.log || tail -3 $O/err.log >> $O/ab.log
done
