import importlib, sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from oracle import pyoracle as O
ife=importlib.import_module("image-feature-extraction_amd")
import test_gpu_parity as T
A=T._eigen_cases(np.random.default_rng(7))
ctx=ife.Context(0)
ev=ctx.eigenvalues(A); ref=O.eig3(A,0)
scale=np.maximum(np.abs(ref[:,0:1]).astype(np.float64),1e-30)
err=(np.abs(ev.astype(np.float64)-ref)/scale).max(1)
idx=np.argsort(-err)[:12]
np.set_printoptions(precision=9, linewidth=200)
for i in idx:
    print(i, err[i], A[i], ev[i], ref[i], O.eig3(A[i].astype(np.float64)))
print('exact frac', (ev==ref).all(1).mean(), 'n>1e-7', (err>1e-7).sum())
