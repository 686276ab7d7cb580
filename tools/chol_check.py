import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time, numpy as np, torch, scipy.linalg
import hydra_amd
from hydra_amd import linear_solver as ls
rng=np.random.default_rng(0)
for n in (5, 32, 33, 100, 1953, 4000):
    M=rng.standard_normal((n,n)); A=M@M.T+n*np.eye(n); b=rng.standard_normal(n)
    x=ls.solve_dense(A,b)
    ref=scipy.linalg.solve(A,b,assume_a='pos')
    print(n, np.abs(np.asarray(x)-ref).max()/np.abs(ref).max(), flush=True)
n=1953
M=rng.standard_normal((n,n)); A=torch.tensor(M@M.T+n*np.eye(n),device='cuda'); b=torch.tensor(rng.standard_normal(n),device='cuda')
for _ in range(3): ls.solve_dense(A.clone(),b.clone())
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(20): ls.solve_dense(A.clone(),b.clone())
torch.cuda.synchronize(); print('ms per solve', (time.perf_counter()-t)/20*1e3)
