#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3bm; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert|rel err" > $O/tests.log
python tools/time_models.py 2>&1 | grep -E "bottleneck" > $O/time.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/bm_step.py <<'PY'
import os, sys
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import torch, bench
from piml_amd.scenes import synthetic_gc_scene
dev = torch.device('cuda:0')
scene = synthetic_gc_scene(4096, 2000, seed=0)
st = bench.Step(scene, 4096, 4096, 0, 2000, dev, None, False, False, True, model_name='PINNSF_bottleneck_multitask', train_mode=True)
st.capture()
for _ in range(40):
    st.run()
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/bm -- python3 /tmp/bm_step.py > /dev/null 2>&1
