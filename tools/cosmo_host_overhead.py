"""Host time against GPU time of one CG iteration of the cosmological-field head's operator
(FFT prior + batched symmetric product + fused CG step): is the loop launch bound?"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import hydra_amd  # noqa: F401
from hydra_amd import _dev, cosmo_sampler as cs

nf, nx = 64, 32
npix = nx * nx
torch.manual_seed(0)
G = torch.randn(nf, npix, npix, dtype=torch.float64, device="cuda")
G = G + G.transpose(1, 2).clone()
p3d = torch.rand(nf, nx, nx, dtype=torch.float64, device="cuda") + 0.5
x = torch.randn(nf, nx, nx, dtype=torch.float64, device="cuda")
n = x.numel()
r, p = torch.randn(n, dtype=torch.float64, device="cuda"), torch.randn(n, dtype=torch.float64, device="cuda")
xs = torch.zeros(n, dtype=torch.float64, device="cuda")
st = _dev.zeros((_dev._lib.query("hydra_cg_state_len"),), torch.float64)


def iteration():
    ap = cs.apply_lhs_operator(p.reshape(nf, nx, nx), G, p3d).flatten()
    _dev.call("hydra_cg_step", n, _dev.ptr(ap), _dev.ptr(xs), _dev.ptr(r), _dev.ptr(p), _dev.ptr(st),
              1e-30, 0, _dev.stream_ptr())


for _ in range(20):
    iteration()
torch.cuda.synchronize()
reps = 300
t0 = time.perf_counter()
for _ in range(reps):
    iteration()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"per iteration: host (enqueue only) {t_host / reps * 1e6:.1f} us, host + GPU drain {t_all / reps * 1e6:.1f} us")


def host_cost(name, fn, reps=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"  {name:28s} {t / reps * 1e6:6.1f} us enqueue")


half = cs._half_spectrum(p3d, -1.0)
X = torch.fft.rfftn(x)
ws = _dev.workspace(_dev._lib.query("hydra_symv_workspace_bytes", npix, nf))
y = torch.empty(nf, npix, dtype=torch.float64, device="cuda")
ap = torch.randn(n, dtype=torch.float64, device="cuda")
host_cost("rfftn", lambda: torch.fft.rfftn(x))
host_cost("filter multiply", lambda: half * X)
host_cost("irfftn", lambda: torch.fft.irfftn(X, s=tuple(x.shape)))
host_cost("hydra_symv_batched", lambda: _dev.call("hydra_symv_batched", _dev.ptr(G), npix, nf, _dev.ptr(x), _dev.ptr(y), 1,
                                                  _dev.ptr(ws), ws.numel(), _dev.stream_ptr()))
host_cost("hydra_cg_step", lambda: _dev.call("hydra_cg_step", n, _dev.ptr(ap), _dev.ptr(xs), _dev.ptr(r), _dev.ptr(p),
                                             _dev.ptr(st), 1e-30, 0, _dev.stream_ptr()))
host_cost("apply_lhs_operator", lambda: cs.apply_lhs_operator(x, G, p3d))
