#!/usr/bin/env python
"""Diagnostic: phase shares of the fused Gram kernel from the stamped build
(make -C hydra_amd/csrc stamps).  Run on the GPU box:
    HYDRA_HIP_LIBRARY=hydra_amd/libhydra_hip_stamps.so python tools/gram_stamps.py
Reports SHARES only -- the stamped build is slower than the product kernel."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hydra_amd as hy
from hydra_amd import _lib, _dev, ptsrc_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "mid"
hex_spec, ns, nf, nt = bench.CONFIGS[cfg]
inp = bench.synth_inputs(hex_spec, ns, nf, nt)
beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in inp["ant_pos"]]
fac = ptsrc_sampler.calc_proj_operator(inp["ra"], inp["dec"], inp["fluxes"], inp["ant_pos"],
                                       inp["antpairs"], inp["freqs"], inp["times"], beams,
                                       materialize=False)
nant, nbl = len(inp["ant_pos"]), len(inp["antpairs"])
dev = _dev.device()
gains = torch.as_tensor(inp["g0"], device=dev)[:, None, None].expand(nant, nf, nt).contiguous()
w = torch.full((nbl, nf, nt), 400.0, dtype=torch.float64, device=dev)
lib = _lib.load()
buf = (ctypes.c_ulonglong * 8)()
fac.allow_uniform = False
fac.gram(gains, w); torch.cuda.synchronize()
lib.hydra_debug_stamps(buf, 1)
fac.gram(gains, w); torch.cuda.synchronize()
lib.hydra_debug_stamps(buf, 1)
fill, issue, mfma, valu, steps, cyc, ticks = [buf[i] for i in range(7)]
print(f"config {cfg}: workgroup 0 wave 0: total cycles {cyc}, clock {cyc / ticks * 0.1:.2f} GHz")
print(f"  segment change (barrier wait + LDS fill) {fill / cyc * 100:5.1f} %")
if steps:      # built with -DHYDRA_GRAM_STAMPS_INNER (these stamps perturb the loop they time)
    for name, v in [("load issue", issue), ("MFMA burst (32)", mfma), ("VALU batch", valu)]:
        print(f"  {name:16s} {v / cyc * 100:5.1f} %   {v / steps:8.1f} cycles per k-step")
tr = (ctypes.c_ulonglong * 64)()
if hasattr(lib, "hydra_debug_trace") and lib.hydra_debug_trace(tr) == 0:
    t = np.array(list(tr), dtype=np.int64).reshape(8, 8)[:, :5]
    t0 = t[t[:, 0] > 0, 0].min() if (t[:, 0] > 0).any() else 0
    print("per-wave timeline of workgroup 0, third (f,t) (cycles from the first barrier arrival):")
    print("  wave  at-barrier-1  past-barrier-1  DMA-landed  past-barrier-2  loop-end   loop")
    for w in range(8):
        if t[w, 0] > 0:
            r = t[w] - t0
            print(f"  {w:4d}  {r[0]:12d}  {r[1]:14d}  {r[2]:10d}  {r[3]:14d}  {r[4]:8d}  {r[4]-r[3]:6d}")
