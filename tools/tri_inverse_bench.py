import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from hydra_amd import _dev
for n in (1681, 4096):
    L = torch.tril(torch.randn(n, n, dtype=torch.float64, device="cuda")) / n ** 0.5
    L.diagonal().add_(2.0)
    X = torch.empty_like(L)
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _dev.call("hydra_tri_inverse", _dev.ptr(L), n, n, _dev.ptr(X), n, _dev.stream_ptr())
        e1.record(); torch.cuda.synchronize()
    t0 = e0.elapsed_time(e1)
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    e0.record(); Y = torch.linalg.solve_triangular(L, eye, upper=False); e1.record(); torch.cuda.synchronize()
    e0.record(); Y = torch.linalg.solve_triangular(L, eye, upper=False); e1.record(); torch.cuda.synchronize()
    print(n, "own kernel %.2f ms, library %.2f ms, max diff %.1e" % (t0, e0.elapsed_time(e1), float((X - Y).abs().max())))
