#!/usr/bin/env python
"""Rate of the Cholesky trailing-update kernel (hydra_chol_update_rect, 2-D tile order) against
the row stride of C and of the panel: is it the 128 x 128 tile whose rows lie `ld` doubles apart
(address translation, DRAM pages) that keeps it below the matrix peak?
    python tools/syrk_stride_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydra_amd import _dev

dev = _dev.device()
m = 16384
for ldc, ldp in [(m, 256), (m, m), (65536, 256), (100000, 256), (100000, 100000), (131072, 256)]:
    C = torch.zeros((m, ldc), dtype=torch.float64, device=dev)
    P = torch.randn((m, ldp), dtype=torch.float64, device=dev)
    st = _dev.stream_ptr()
    args = (_dev.ptr(C), ldc, _dev.ptr(P), ldp, _dev.ptr(P), ldp, m, m, m, 256, st)   # diag = m: every tile
    _dev.call("hydra_chol_update_rect", *args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        _dev.call("hydra_chol_update_rect", *args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * m * m * 256
    print(f"m={m} ldc={ldc:7d} ldp={ldp:7d}: {ms:7.3f} ms = {fl / ms / 1e9:6.2f} TFLOP/s = {fl / ms / 1e9 / 78.6:.3f}")
    del C, P
