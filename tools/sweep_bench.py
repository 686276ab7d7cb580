#!/usr/bin/env python
"""The two triangular sweeps alone (hydra_chol_apply) on a synthetic factor, for kernel traces:
    rocprofv3 --kernel-trace --stats -d out -o sw -- python3 tools/sweep_bench.py 1024 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydra_amd import _dev

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = _dev.device()
gen = torch.Generator(device=dev).manual_seed(3)
L = torch.randn((n, n), dtype=torch.float64, device=dev, generator=gen).mul_(0.5 / n ** 0.5).tril_()
L.diagonal().copy_(1.0 + torch.rand(n, dtype=torch.float64, device=dev, generator=gen))
b0 = torch.randn(n, dtype=torch.float64, device=dev, generator=gen)
b = b0.clone()
_dev.call("hydra_chol_apply", _dev.ptr(L), _dev.ptr(b), n, _dev.stream_ptr())
torch.cuda.synchronize()
res = float(torch.linalg.norm(L @ (L.T @ b) - b0) / torch.linalg.norm(b0))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    _dev.call("hydra_chol_apply", _dev.ptr(L), _dev.ptr(b), n, _dev.stream_ptr())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"n={n}: sweeps {ms:.3f} ms ({ms * 1e3 / (2 * ((n + 255) // 256)):.1f} us per panel launch), "
      f"relative residual {res:.2e}")
