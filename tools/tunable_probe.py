import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
t0 = time.time()
torch.cuda.tunable.enable(True); torch.cuda.tunable.tuning_enable(False)
ok = torch.cuda.tunable.read_file(sys.argv[1])
print('read_file', ok, time.time() - t0)
x = torch.randn(40960, 128, device='cuda'); lin = torch.nn.Linear(128, 128).cuda()
for i in range(3):
    t0 = time.time(); y = lin(x); torch.cuda.synchronize(); print('linear fwd', i, time.time() - t0)
x2 = torch.randn(24576, 128, device='cuda')
t0 = time.time(); y = lin(x2); torch.cuda.synchronize(); print('linear fwd shape2', time.time() - t0)
w = torch.randn(128, 128, device='cuda')
t0 = time.time(); z = x @ w; torch.cuda.synchronize(); print('mm NN', time.time() - t0)
t0 = time.time(); z = x.t() @ x; torch.cuda.synchronize(); print('mm TN', time.time() - t0)
