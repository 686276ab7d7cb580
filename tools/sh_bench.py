#!/usr/bin/env python
"""Time the SH pixel->mode projection (hydra_sh_project) at BASELINE config-3 shapes on a few
(f,t) slices:  python tools/sh_bench.py [nf] [nt]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hydra_amd as hy
from hydra_amd import vis_simulator as vs

nf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lmax, nside = 40, 32
ant_pos = hy.utils.build_hex_array(hex_spec=(7, 13), d=14.6)
nant = len(ant_pos)
names = list(ant_pos.keys())
pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
bi = np.array([p[0] for p in pairs], dtype=np.int32)
bj = np.array([p[1] for p in pairs], dtype=np.int32)
freqs = np.linspace(100.0, 120.0, max(nf, 2))[:nf]
lsts = np.linspace(0.2, 0.5, max(nt, 2))[:nt]
beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
V = vs.sh_voltages(nside, ant_pos, freqs * 1e6, lsts, beams)
ymap = vs.sh_mode_maps(lmax, nside, 1.0, packed=True)
npix, nm = ymap.shape
up = (V[0, :, 0, :].abs() > 0).sum(dim=1).cpu().numpy()
print(f"nant {nant} nbl {len(pairs)} npix {npix} nm {nm} nf {nf} nt {nt}; pixels above horizon per t: {up}")
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = vs.sh_project(V, bi, bj, ymap)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fl = 4.0 * len(pairs) * nf * nt * npix * nm
    print(f"sh_project: {dt*1e3:.1f} ms  {fl/dt/1e12:.2f} TFLOP/s (algorithmic 4 M Npix Nm)")
