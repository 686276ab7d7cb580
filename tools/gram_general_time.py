#!/usr/bin/env python
"""Time the general (per-baseline-weight) Gram of a bench configuration, whole entry point
(row factors + gram_fused_kernel + slab sum).  For comparing build variants on the GPU box:
    HYDRA_HIP_LIBRARY=path/to/variant.so python tools/gram_general_time.py c2"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hydra_amd as hy
from hydra_amd import _dev, ptsrc_sampler

cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
fac.allow_uniform = os.environ.get("GRAM_CLOSED", "0") == "1"   # 1: time the closed-form path
G = fac.gram(gains, w)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    G = fac.gram(gains, w)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 2.0 * nbl * nf * nt * ns * (ns + 1)
print(f"{os.environ.get('HYDRA_HIP_LIBRARY', 'product')}: config {cfg} general Gram {ms:.1f} ms "
      f"= {flops / ms / 1e9:.2f} TFLOP/s = {flops / ms / 1e9 / 78.6:.3f} of peak; checksum {float(G.sum()):.12e}")
