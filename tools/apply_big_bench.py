"""Time R a of the SH head without the response matrix (StreamedSHOperator.apply) at config 3's
array (127 antennas, nside 32, 40 LSTs) for 16 of the 128 channels: pair loop on the vector units
against the antenna-space kernel (apply_V_big_kernel).  python tools/apply_big_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hydra_amd as hy
from hydra_amd import sh_sampler, _dev
ant_pos = hy.utils.build_hex_array(hex_spec=(7, 13), d=14.6)
freqs = np.linspace(100.0, 120.0, 16); times = np.linspace(0.2, 0.5, 40)
beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
op = sh_sampler.StreamedSHOperator(freqs, times, beams, ant_pos, 40, 32, time_block=1)
a = torch.randn(op.shape[1], dtype=torch.float64, device="cuda")
for enable in (0, 1):
    prev = _dev._lib.query("hydra_debug_adjoint_mfma", enable)
    y = op.apply(a); torch.cuda.synchronize()
    t0 = time.perf_counter(); y = op.apply(a); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("antenna space" if enable else "pair loop", f"{dt*1e3:.1f} ms for 16 channels -> x8 = {dt*8e3:.0f} ms at 128", flush=True)
    if enable: print("relerr", float((y - yref).abs().max() / yref.abs().max()))
    yref = y
    _dev._lib.query("hydra_debug_adjoint_mfma", prev)
