#!/usr/bin/env python
"""Diagnostic: where a wave of the closed-form Gram kernel spends its cycles, from the stamped
build (make -C hydra_amd/csrc stamps).  Run on the GPU box:
    HYDRA_HIP_LIBRARY=hydra_amd/libhydra_hip_stamps.so python tools/gram_uniform_stamps.py [c2]
Reports SHARES of wave 0 of workgroup 0 -- the stamped build is slower than the product kernel."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hydra_amd as hy
from hydra_amd import _lib, _dev, ptsrc_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
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
fac.gram(gains, w); torch.cuda.synchronize()
assert fac.last_gram_kernel == "gram_uniform_kernel"
lib.hydra_debug_stamps(buf, 1)
fac.gram(gains, w); torch.cuda.synchronize()
lib.hydra_debug_stamps(buf, 1)
bar, dma, ks, ep, nphase, cyc = [buf[i] for i in range(6)]
print(f"config {cfg}: workgroup 0 wave 0: {nphase} phases, {cyc} cycles")
for name, v in [("barrier wait", bar), ("DMA issue", dma), ("k-steps", ks), ("epilogue", ep)]:
    print(f"  {name:14s} {v / cyc * 100:5.1f} %   {v / max(nphase, 1):8.1f} cycles per phase")
print(f"  accounted      {(bar + dma + ks + ep) / cyc * 100:5.1f} %")
