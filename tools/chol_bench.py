"""Time the large-matrix Cholesky (hydra_chol_solve_blocked) and report the factorisation's
FP64 rate: n^3/3 flops.  python tools/chol_bench.py [n ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hydra_amd import _dev


def spd(n, m=512):
    g = torch.Generator(device="cuda").manual_seed(n)
    B = torch.randn(n, m, dtype=torch.float64, device="cuda", generator=g)
    A = B @ B.T
    A.div_(m)
    A.diagonal().add_(1.0)
    return A


for n in [int(a) for a in sys.argv[1:]] or [8192, 16384, 32768]:
    A0 = spd(n)
    b0 = torch.randn(n, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    best_f = best_s = 1e30
    reps = 3 if n <= 40000 else 1
    for rep in range(reps):
        A = A0.clone()
        b = b0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _dev.call("hydra_chol_solve_blocked", _dev.ptr(A), None, n, _dev.ptr(info), _dev.stream_ptr())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        best_f = min(best_f, t1 - t0)
        A = A0.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _dev.call("hydra_chol_solve_blocked", _dev.ptr(A), _dev.ptr(b), n, _dev.ptr(info),
                  _dev.stream_ptr())
        torch.cuda.synchronize()
        best_s = min(best_s, time.perf_counter() - t0)
    x = b
    res = float(torch.linalg.norm(A0 @ x - b0) / torch.linalg.norm(b0))
    fl = n ** 3 / 3
    print(f"n={n}: factor {best_f*1e3:.1f} ms = {fl/best_f/1e12:.1f} TFLOP/s "
          f"({fl/best_f/78.6e12*100:.1f} % of 78.6), factor+solve {best_s*1e3:.1f} ms, "
          f"info={int(info.item())}, rel residual {res:.2e}", flush=True)
    del A0, A
