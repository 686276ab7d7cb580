#!/usr/bin/env python
"""
Golden vectors for the cosmological-field and diffuse-region heads, from the REFERENCE's own
functions.  Build container only (needs /root/reference); the .npz files are committed.

  cosmo.npz            hydra/cosmo_sampler.py (numpy/scipy only): calculate_cosmo_fns,
                       make_cosmo_field_grid, comoving_fourier_modes, calculate_pspec_on_grid,
                       apply_S, apply_lhs_operator, precompute_mpi (one-rank communicator)
  region_segments.npz  hydra/region_sampler.segmented_diffuse_sky_model_pixels, no smoothing
                       (module imported with the placeholder technique of make_golden.py; its
                       simulator-dependent calc_proj_operator is NOT run)

Usage:  python tests/golden/make_golden_cosmo.py
"""
import argparse
import importlib
import os

import numpy as np

from make_golden import OUT, load_reference


class _OneRank:
    def Get_rank(self):
        return 0

    def Reduce(self, src, dst, op=None, root=0):
        dst[...] = src


def main():
    load_reference()
    cs = importlib.import_module("hydra.cosmo_sampler")
    rs = importlib.import_module("hydra.region_sampler")
    cs.MPI_SUM = None

    # ------------------------------------------------------------------ grid, modes, prior
    args = argparse.Namespace(cosmo_field_ra_bounds=(10.0, 16.0), cosmo_field_dec_bounds=(-33.0, -28.0),
                              cosmo_field_ra_ngrid=5, cosmo_field_dec_ngrid=4)
    g_ra, g_dec = cs.make_cosmo_field_grid(args)
    freqs = np.linspace(100.0, 104.0, 6)
    cosmo = {"h": 0.69, "omega_m": 0.31}
    hz_fn, r_fn = cs.calculate_cosmo_fns(**cosmo)
    zs = np.array([0.0, 0.3, 6.5, 12.9])
    x = np.unique(g_ra)
    y = np.unique(g_dec)
    kx, ky, knu = cs.comoving_fourier_modes(x=x, y=y, freqs=freqs, **cosmo)
    kbins = np.linspace(0.0, 10.0, 25)
    pk = 0.1 * kbins + np.ones(kbins.size)
    pspec3d = cs.calculate_pspec_on_grid(kbins=kbins, pspec=pk, x=x, y=y, freqs=freqs, **cosmo)

    rng = np.random.RandomState(41)
    cube = rng.randn(*pspec3d.shape)
    s_plus = cs.apply_S(cube, pspec3d, exponent=1)
    s_inv = cs.apply_S(cube, pspec3d, exponent=-1.0)
    s_mhalf = cs.apply_S(cube, pspec3d, exponent=-0.5)

    # ------------------------------------------------------------------ linear system
    nant, nt = 5, 4
    nf = freqs.size
    npix = g_ra.size
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    proj = rng.randn(nbl, nf, nt, npix) + 1j * rng.randn(nbl, nf, nt, npix)
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    inv_noise = rng.uniform(0.5, 3.0, size=(nbl, nf, nt))
    inv_noise[2, 1, :] = 0.0
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    times = np.arange(nt) * 0.1
    seed = 4242
    np.random.seed(seed)
    op, rhs = cs.precompute_mpi(_OneRank(), ants, antpairs, freqs, freqs, times, proj, data,
                                inv_noise, gains, pspec3d, realisation=True)
    np.random.seed(seed)            # replay: omega_n Re, Im (cosmo_sampler.py:308-312), omega_s (:346)
    om_re = np.random.randn(nbl, nf, nt)
    om_im = np.random.randn(nbl, nf, nt)
    omega_n = (1.0 * om_re + 1.0j * om_im) / np.sqrt(2.0)
    omega_s = np.random.randn(*pspec3d.shape)
    op0, rhs0 = cs.precompute_mpi(_OneRank(), ants, antpairs, freqs, freqs, times, proj, data,
                                  inv_noise, gains, pspec3d, realisation=False)
    # a rank holding only channels 2..4 of the 6 (rows of the other channels stay zero)
    sl = slice(2, 5)
    np.random.seed(seed)
    op_part, rhs_part = cs.precompute_mpi(
        _OneRank(), ants, antpairs, freqs, freqs[sl], times, proj[:, sl], data[:, sl],
        inv_noise[:, sl], gains[:, sl], pspec3d, realisation=False)
    xvec = rng.randn(*pspec3d.shape)
    lhs = cs.apply_lhs_operator(xvec, op, pspec3d)

    np.savez(os.path.join(OUT, "cosmo.npz"), g_ra=g_ra, g_dec=g_dec, freqs=freqs, zs=zs,
             hz=hz_fn(zs), r=r_fn(zs), kx=kx, ky=ky, knu=knu, kbins=kbins, pk=pk, pspec3d=pspec3d,
             cube=cube, s_plus=s_plus, s_inv=s_inv, s_mhalf=s_mhalf, ants=ants,
             antpairs=np.array(antpairs), proj=proj, gains=gains, inv_noise=inv_noise, data=data,
             times=times, seed=seed, omega_n=omega_n, omega_s=omega_s, op=op, rhs=rhs, op0=op0,
             rhs0=rhs0, op_part=op_part, rhs_part=rhs_part, xvec=xvec, lhs=lhs)

    # ------------------------------------------------------------------ region segmentation
    rng = np.random.RandomState(43)
    npx = 192
    fr = np.array([100.0, 110.0, 120.0])
    beta = rng.uniform(-3.0, -2.2, size=npx)
    amp = rng.uniform(10.0, 500.0, size=npx)
    sky = amp[:, None] * (fr[None, :] / 100.0) ** beta[:, None]
    ra = rng.uniform(0.0, 2.0 * np.pi, size=npx)
    dec = rng.uniform(-1.0, 0.2, size=npx)
    segs = {}
    for nreg in (3, 8):
        idx = rs.segmented_diffuse_sky_model_pixels(ra, dec, sky, fr, nreg, smoothing_fwhm=None)
        segs[f"n{nreg}_count"] = len(idx)
        for q, ix in enumerate(idx):
            segs[f"n{nreg}_{q}"] = ix
    np.savez(os.path.join(OUT, "region_segments.npz"), ra=ra, dec=dec, sky=sky, freqs=fr, **segs)
    print("cosmo / region golden vectors written to", OUT)


if __name__ == "__main__":
    main()
