python -m pytest tests/test_encoder_gpu.py tests/test_mlpglue_gpu.py tests/test_simulator_gpu.py -x -q -m gpu 2>&1 | grep -n "passed\|failed" 
for rep in 1 2; do
  echo "default: $(python bench.py --steps 300 --warmup 30 --cpu-seconds 0 2>/dev/null | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')"
done
cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/r30 && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r30 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --spinup-ms 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r30/*/*_kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:13]:
    print(r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
