import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from oracle import oracle as O
from piml_amd.models.mlapm import MLAPM
from piml_amd.scenes import synthetic_gc_scene
sc = synthetic_gc_scene(4096, 0, seed=3, nan_frac=0.0)
pr=dict(tau=5 / 6, A=10.67, B=-3.33, C=0.5, theta=20)
dev=lambda x: torch.tensor(x,device='cuda:0')
act = MLAPM(version='UCY', **pr).step(dev(sc['position']), dev(sc['velocity']), dev(sc['desired_speed']), dev(sc['destination']), dt=0.08, radius=0.3).cpu().numpy()
ref = O.mlapm_step(sc['position'], sc['velocity'], sc['desired_speed'], sc['destination'], 0.08, 0.3, version='UCY', **pr)
err=np.linalg.norm(act-ref,axis=-1)/np.maximum(np.linalg.norm(ref,axis=-1),1e-3)
print('max',err.max(),'n>1e-5',(err>1e-5).sum(), np.argsort(err)[-5:], np.sort(err)[-5:])
i=int(np.argmax(err)); print(act[i],ref[i])
