#!/usr/bin/env python
"""The batched gain operator at config 4 (331 antennas, 64 x 64 samples, 16 modes): 8 vectors per
application through the matrix-core pair kernel against the tiled kernel, HIP events around the
whole application and around the pair kernel alone.

    python tools/gain_batch_bench.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import hydra_amd as hy                          # noqa: E402
from hydra_amd import _dev, gain_sampler as gs  # noqa: E402

nf = nt = 64
ant_pos = hy.utils.build_hex_array(hex_spec=(11, 21), d=14.6)
nant = len(ant_pos)
ants = np.arange(nant)
antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
nbl = len(antpairs)
dev = _dev.device()
gen = torch.Generator(device=dev).manual_seed(1)
freqs, times = np.linspace(100.0, 120.0, nf), np.linspace(0.2, 0.5, nt)
Fb, _, _ = hy.utils.partial_fourier_basis_2d_from_nmax(freqs, times, 2, 2, 20.0, 0.3)
nmodes = Fb.shape[0]
ggv = torch.randn((nbl, nf, nt), dtype=torch.complex128, device=dev, generator=gen)
w = torch.full((nbl, nf, nt), 400.0, dtype=torch.float64, device=dev)
ps = torch.full((nmodes,), 0.05, dtype=torch.float64, device=dev)
A_real, A_imag = gs.proj_operator(ants, antpairs)
op = gs.GainOperator(w, ps, A_real, A_imag, ggv, torch.as_tensor(Fb, device=dev))
X = torch.randn((8, op.n), dtype=torch.float64, device=dev, generator=gen)


def timed(fn, reps=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


op.Wb
op.batch_mfma = True
ms_m, Ym = timed(lambda: op.local(X))
op.batch_mfma = False
ms_t, Yt = timed(lambda: op.local(X))
print(f"8-vector application: matrix cores {ms_m:.3f} ms, tiled kernel {ms_t:.3f} ms, "
      f"relative difference {float((Ym - Yt).abs().max() / Yt.abs().max()):.2e}")
g = torch.randn((8, nant, nf * nt), dtype=torch.complex128, device=dev, generator=gen)
y = torch.empty_like(g)
ms_p, _ = timed(lambda: _dev.call("hydra_gain_pairs_mfma", _dev.ptr(g), 8, nant, nf * nt,
                                  _dev.ptr(op.Wb), _dev.ptr(op.sw), _dev.ptr(y), _dev.stream_ptr()), 20)
byts = 8.0 * nbl * nf * nt
print(f"pair kernel alone: {ms_p:.3f} ms = {byts / ms_p / 1e9:.2f} TB/s on 8 M bytes "
      f"({byts / ms_p / 1e9 / 8:.2f} of 8 TB/s); stored {op.Wb.numel() * 8 / 1e9:.2f} GB")
nap, nftp = (nant + 15) // 16 * 16, nf * nt
Gc = torch.randn((nftp, nap, 16), dtype=torch.float64, device=dev, generator=gen)
Yc = torch.empty_like(Gc)
# (the blocked form of the pair kernel alone: through the one-call entry point with the basis
# kernels' share taken from rocprofv3 --kernel-trace)
x1 = X[0].contiguous()
ms_1, _ = timed(lambda: op.local(x1))
print(f"one vector (streamed kernel): {ms_1:.3f} ms")
