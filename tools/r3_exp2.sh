#!/bin/bash
# timing of variant libraries built beforehand (piml_amd/exp/lib_*.so) against the default one, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3exp2; rm -rf $O; mkdir -p $O
cp piml_amd/libpiml_hip.so /tmp/lib_orig.so
line() { python -c "import json,sys; b=json.loads(sys.stdin.read()); print('$1', round(b['ms_per_step'],4), b.get('verify_max_rel_err'), [(k['name'][4:14], round(k['us'],1)) for k in b['roofline']['kernels']])"; }
B="python bench.py --cpu-seconds 0 --secondary 0 --verify 0"
for r in 1 2; do
$B 2>/dev/null | line base >> $O/ab.log
for v in $(ls piml_amd/exp/ | sed 's/lib_//; s/.so//'); do
  cp piml_amd/exp/lib_$v.so piml_amd/libpiml_hip.so
  $B 2>$O/err_$v.log | line $v >> $O/ab.log
done
cp /tmp/lib_orig.so piml_amd/libpiml_hip.so
done
