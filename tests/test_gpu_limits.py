"""
Sizes beyond one launch dimension (grid.y is capped at 65 535 workgroups): more than 65 535
baselines in apply_gains / the row factors of precompute_mpi / gain_sampler.apply_proj (a
complete array of 363+ antennas has that many), and more than 65 535 (f,t) samples in simulate_vis
and the subarr_ant cube.  The reference has no such limits (utils.py:63-110,
ptsrc_sampler.py:53-59, gain_sampler.py:81-106, vis_simulator.py:290-297,565-568).
"""
import numpy as np
import pytest

from conftest import elem_err, relerr

pytestmark = pytest.mark.gpu


def test_more_than_65535_baselines():
    import hydra_amd as hy
    from hydra_amd import gain_sampler as gs, _dev
    import torch
    nant, nf, nt = 364, 1, 2
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    assert nbl == 66066 > 65535
    rng = np.random.RandomState(0)
    v = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    g = (1.0 + 1.0j) + 0.1 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    x = 0.1 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    i1 = np.array([p[0] for p in antpairs])
    i2 = np.array([p[1] for p in antpairs])
    want = v * g[i1] * g[i2].conj() * (1.0 + x[i1] + x[i2].conj())      # utils.py:102-109
    got = hy.utils.apply_gains(v, g, ants, antpairs, perturbation=x)
    assert elem_err(got, want, rtol=1e-13, floor=1e-15) <= 1.0
    # row factors g_i conj(g_j) sqrt(w) (ptsrc_sampler.py:53-71)
    w = 300.0 + 100.0 * rng.rand(nbl, nf, nt)
    gg = _dev.empty((nbl, nf, nt), torch.complex128)
    rows = _dev.empty((nbl, nf, nt), torch.complex128)
    gd, wd, i1d, i2d = _dev.c128(g), _dev.f64(w), _dev.i32(i1), _dev.i32(i2)   # alive over the call
    _dev.call("hydra_row_factors", _dev.ptr(gd), _dev.ptr(wd), _dev.ptr(i1d), _dev.ptr(i2d), nbl,
              nf, nt, _dev.ptr(gg), _dev.ptr(rows), _dev.stream_ptr())
    assert elem_err(gg.cpu().numpy(), g[i1] * g[i2].conj(), rtol=1e-13, floor=1e-15) <= 1.0
    assert elem_err(rows.cpu().numpy(), g[i1] * g[i2].conj() * np.sqrt(w), rtol=1e-13,
                    floor=1e-15) <= 1.0
    # gain projector (x_i + conj x_j) V (gain_sampler.py:81-106)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    got = gs.apply_proj(x, A_real, A_imag, v)
    assert elem_err(got, (x[i1] + x[i2].conj()) * v, rtol=1e-13, floor=1e-15) <= 1.0


def test_more_than_65535_freq_time_samples():
    import hydra_amd as hy
    from oracle import beams as obeams, helpers, response
    ants = helpers.build_hex_array((2, 3))          # 7 antennas, the first 3 used
    ants = {k: ants[k] for k in list(ants)[:3]}
    nf, nt, ns = 256, 257, 3
    assert nf * nt > 65535
    freqs = np.linspace(100e6, 120e6, nf)
    lsts = np.linspace(0.2, 0.5, nt)
    ra, dec = np.array([0.3, 0.35, 0.4]), np.array([-0.5, -0.55, -0.45])
    fluxes = np.ones((ns, nf)) * np.array([1.0, 2.0, 0.5])[:, None]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ants]
    vis = hy.vis_simulator.simulate_vis(ants, fluxes, ra, dec, freqs, lsts, bh)
    sub = hy.vis_simulator.simulate_vis_per_source(ants, fluxes, ra, dec, freqs, lsts, bh,
                                                   subarr_ant=1)
    assert vis.shape == (nf, nt, 3, 3) and sub.shape == (nf, nt, 3, ns)
    # the oracle on the first, a middle and the LAST (f,t) samples (the ones past 65 535)
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ants]
    for fsel, tsel in (([0, 255], [0, 256]), ([128, 254], [100, 255])):
        ref = response.simulate_vis_per_source(ants, fluxes[:, fsel], ra, dec, freqs[fsel],
                                               lsts[tsel], bo)
        got = vis[np.ix_(fsel, tsel)]
        assert relerr(got, ref.sum(axis=-1)) < 1e-10
        ref_sub = response.simulate_vis_per_source(ants, fluxes[:, fsel], ra, dec, freqs[fsel],
                                                   lsts[tsel], bo, subarr_ant=1)
        assert relerr(sub[np.ix_(fsel, tsel)], ref_sub) < 1e-10
