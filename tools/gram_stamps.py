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
print(f"config {cfg}: workgroup 0 wave 0: k-steps {steps}, total cycles {cyc}, clock {cyc / ticks * 0.1:.2f} GHz")
for name, v in [("LDS fill", fill), ("load issue", issue), ("MFMA burst (32)", mfma), ("VALU batch", valu)]:
    print(f"  {name:16s} {v / cyc * 100:5.1f} %   {v / max(steps, 1):8.1f} cycles per k-step")
print(f"  accounted        {(fill + issue + mfma + valu) / cyc * 100:5.1f} %")
