"""One rank's step of the 8-way sharded 16384-agent scene on one GPU (bench.cfg4_projection), e.g. under PIML_ENC_SPLIT_TILES=..."""
import json, os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
args = types.SimpleNamespace(obstacles=2000, seed=0, graph=1)
out = bench.cfg4_projection(args, dev, bool(int(os.environ.get('MSG', '0'))), shards=int(os.environ.get('SHARDS', '8')))
out.pop('note', None)
print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))
