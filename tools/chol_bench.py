#!/usr/bin/env python
"""Time the blocked dense solve (hydra_chol_solve_blocked) on a diagonally dominant matrix and
check its residual.    python tools/chol_bench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydra_amd import _dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dev = _dev.device()
gen = torch.Generator(device=dev).manual_seed(1)
R = torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen)
A0 = R + R.T
del R
A0.mul_(0.02)
A0.diagonal().add_(0.1 * n ** 0.5 + 2.0)
b0 = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
info = torch.zeros(1, dtype=torch.int32, device=dev)
for rep in range(2):
    A, b = A0.clone(), b0.clone()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _dev.call("hydra_chol_solve_blocked", _dev.ptr(A), _dev.ptr(b), n, _dev.ptr(info), _dev.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
res = float(torch.linalg.norm(A0 @ b - b0) / torch.linalg.norm(b0))
fl = n ** 3 / 3.0
print(f"{os.environ.get('HYDRA_HIP_LIBRARY', 'product')}: n={n}: {ms:.1f} ms = {fl / ms / 1e9:.2f} TFLOP/s = {fl / ms / 1e9 / 78.6:.3f} of peak; info {int(info.item())}, relative residual {res:.2e}")

# the two triangular sweeps alone, from the kept factor (hydra_chol_apply: what a sample costs
# when the operator has not changed)
b = b0.clone()
_dev.call("hydra_chol_apply", _dev.ptr(A), _dev.ptr(b), n, _dev.stream_ptr())
torch.cuda.synchronize()
reps = 5
e0.record()
for _ in range(reps):
    _dev.call("hydra_chol_apply", _dev.ptr(A), _dev.ptr(b), n, _dev.stream_ptr())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
b = b0.clone()
_dev.call("hydra_chol_apply", _dev.ptr(A), _dev.ptr(b), n, _dev.stream_ptr())
res = float(torch.linalg.norm(A0 @ b - b0) / torch.linalg.norm(b0))
print(f"  sweeps alone: {ms:.3f} ms = {8.0 * n * n / ms / 1e9:.2f} TB/s on the 2 x 4 n^2 bytes of L; "
      f"relative residual {res:.2e}")
