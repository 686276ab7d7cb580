#!/usr/bin/env python
"""R x and R^H z at config 5's array (350 antennas, 61 075 baselines): the antenna-space kernels
on the matrix cores (apply_V_sb_kernel / adjoint_V_sb_kernel) against the pair loops on the vector
units, HIP events around each.

    python tools/antspace_big_bench.py [nsrc] [nft]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hydra_amd import _dev                      # noqa: E402
from hydra_amd.ptsrc_sampler import FactoredProj  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nft = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nant = 350
dev = _dev.device()
g = torch.Generator(device=dev).manual_seed(1)
V = torch.randn((nft, 1, nant, ns), dtype=torch.complex128, device=dev, generator=g)
ii, jj = np.triu_indices(nant, 1)
fac = FactoredProj(V, ii.astype(np.int32), jj.astype(np.int32))
nbl = ii.size
x = torch.randn(ns, dtype=torch.float64, device=dev, generator=g)
z = torch.randn((nbl, nft, 1), dtype=torch.complex128, device=dev, generator=g)


def timed(fn, reps=3):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


flops = 8.0 * nbl * ns * nft
res = {}
for name, enable in (("matrix cores", 1), ("pair loop", 0)):
    prev = _dev._lib.query("hydra_debug_adjoint_mfma", enable)
    fac.antspace_adjoint = bool(enable)
    ms_a, ya = timed(lambda: fac.apply(x))
    ms_t, yt = timed(lambda: fac.adjoint(z))
    _dev._lib.query("hydra_debug_adjoint_mfma", prev)
    res[name] = (ya, yt)
    print(f"{name:13s} R x {ms_a:8.2f} ms = {flops / ms_a / 1e9:6.1f} TFLOP/s   "
          f"R^H z {ms_t:8.2f} ms = {flops / ms_t / 1e9:6.1f} TFLOP/s   (8 Nbl Ns flops per (f,t))")
a, b = res["matrix cores"], res["pair loop"]
print("R x   relative difference", float((a[0] - b[0]).abs().max() / b[0].abs().max()))
print("R^H z relative difference", float((a[1] - b[1]).abs().max() / b[1].abs().max()))
