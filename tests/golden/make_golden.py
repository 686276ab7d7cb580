#!/usr/bin/env python
"""
Generate the golden vectors under tests/golden/ from the REFERENCE's own
pure-numpy functions.  Run in the build container only (needs /root/reference);
the resulting .npz files are committed, the reference never travels.

Technique: the reference package cannot be imported as a
whole (astropy / pyuvdata / matvis / healpy / ... are not installed), so empty
placeholder modules are registered for those names -- they are never *called* by
the functions used here -- and the needed reference modules are imported one by
one, skipping hydra/__init__.py.  Only functions whose arithmetic is pure numpy /
scipy are executed: ptsrc_sampler.precompute_mpi, sh_sampler.{get_em_ell_idx,
alms2healpy, construct_rhs_no_rot[_mpi], apply_lhs_no_rot[_mpi]},
linear_solver.cg, gain_sampler.{proj_operator, apply_proj, apply_proj_conj},
utils.{apply_gains, build_hex_array, get_flux_from_ptsrc_amp,
partial_fourier_basis_2d_from_nmax, extract_vis_from_sim, flatten_vector,
reconstruct_vector}, example.generate_random_ptsrc_catalogue.

The visibility simulator (vis_simulator.py) is NOT run: its arithmetic depends on
the missing third-party packages, and no stand-ins are written for them.

Usage:  python tests/golden/make_golden.py
"""
import importlib
import os
import sys
import types
from unittest import mock

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Placeholder(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return mock.MagicMock(name=f"{self.__name__}.{name}")


def load_reference():
    names = [
        "matvis", "matvis.core", "matvis.core.beams", "pyuvsim", "pyuvdata",
        "pyuvdata.analytic_beam", "astropy", "astropy.constants", "astropy.units",
        "astropy.coordinates", "astropy.coordinates.builtin_frames", "astropy.time",
        "healpy", "spherical", "quaternionic", "pygdsm", "hera_sim", "hera_sim.beams",
        "pylab",
    ]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = _Placeholder(n)
    # isinstance() targets used by utils.get_beam_interp_shape need real classes
    for modname, attr in [("pyuvdata", "UVBeam"), ("pyuvdata", "BeamInterface"),
                          ("pyuvdata.analytic_beam", "AnalyticBeam")]:
        setattr(sys.modules[modname], attr, type(attr, (), {}))
    pkg = types.ModuleType("hydra")
    pkg.__path__ = [os.path.join(REF, "hydra")]
    sys.modules["hydra"] = pkg
    mods = {}
    for m in ["utils", "linear_solver", "ptsrc_sampler", "sh_sampler", "gain_sampler",
              "example"]:
        mods[m] = importlib.import_module("hydra." + m)
    return mods


def main():
    ref = load_reference()
    utils, ls, ps, sh, gs, ex = (ref[k] for k in ["utils", "linear_solver",
                                                   "ptsrc_sampler", "sh_sampler",
                                                   "gain_sampler", "example"])

    # ------------------------------------------------------------------ helpers
    np.random.seed(0)
    ra, dec, amps = ex.generate_random_ptsrc_catalogue(
        25, ra_bounds=(0.0, 1.0), dec_bounds=(-0.6, 0.4), logflux_bounds=(-1.0, 2.0))
    freqs = np.linspace(100.0, 120.0, 6)
    fluxes = utils.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    fluxes_curv = utils.get_flux_from_ptsrc_amp(amps, freqs, -2.7, curv_ptsrc=0.3)
    hex34 = utils.build_hex_array(hex_spec=(3, 4), d=14.6)
    hex47 = utils.build_hex_array(hex_spec=(4, 7), d=14.6)
    hexrows = utils.build_hex_array(ants_per_row=[2, 3, 2], d=10.0)
    times = np.linspace(0.2, 0.5, 5)
    Fb, kf, kt = utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 2, Lfreq=freqs[-1] - freqs[0], Ltime=times[-1] - times[0],
        freq0=freqs[0], time0=times[0], shape0=(freqs.size, times.size),
        positive_only=True)
    Fb2, kf2, kt2 = utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 1, Lfreq=25.0, Ltime=0.4, positive_only=False)
    np.savez(
        os.path.join(OUT, "helpers.npz"),
        ra=ra, dec=dec, amps=amps, freqs=freqs, fluxes=fluxes, fluxes_curv=fluxes_curv,
        hex34=np.array([hex34[k] for k in hex34]), hex47=np.array([hex47[k] for k in hex47]),
        hexrows=np.array([hexrows[k] for k in hexrows]), times=times,
        Fb=Fb, kf=kf, kt=kt, Fb2=Fb2, kf2=kf2, kt2=kt2)

    # ------------------------------------------------------------- apply_gains
    rng = np.random.RandomState(3)
    nant, nf, nt = 5, 4, 3
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    v = rng.randn(len(antpairs), nf, nt) + 1j * rng.randn(len(antpairs), nf, nt)
    g = 1.0 + 0.1 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    pert = 0.05 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    out_plain = utils.apply_gains(v, g, ants, antpairs)
    out_pert = utils.apply_gains(v, g, ants, antpairs, perturbation=pert)
    simvis = rng.randn(nf, nt, nant, nant) + 1j * rng.randn(nf, nt, nant, nant)
    extracted = utils.extract_vis_from_sim(ants, antpairs, simvis)
    flat = utils.flatten_vector(v)
    recon = utils.reconstruct_vector(flat, v.shape)
    np.savez(os.path.join(OUT, "gains.npz"), ants=ants, antpairs=np.array(antpairs), v=v,
             g=g, pert=pert, out_plain=out_plain, out_pert=out_pert, simvis=simvis,
             extracted=extracted, flat=flat, recon=recon)

    # --------------------------------------------- ptsrc precompute_mpi (two cases)
    for tag, (nant, nf, nt, ns, uniform) in {
        "small": (4, 3, 5, 7, False),
        "c1like": (10, 4, 6, 40, True),
    }.items():
        rng = np.random.RandomState(11 if tag == "small" else 12)
        ants = np.arange(nant)
        antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
        nbl = len(antpairs)
        proj = rng.randn(nbl, nf, nt, ns) + 1j * rng.randn(nbl, nf, nt, ns)
        gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
        if uniform:
            inv_noise = np.full((nbl, nf, nt), 400.0)
        else:
            inv_noise = rng.uniform(0.5, 3.0, size=(nbl, nf, nt))
            inv_noise[1, 0, :] = 0.0           # flagged samples
        prior = rng.uniform(0.05, 0.2, size=ns)
        truth = 1.0 + 0.1 * rng.randn(ns)
        data = (proj.reshape(-1, ns) @ truth).reshape(nbl, nf, nt)
        data = utils.apply_gains(data, gains, ants, antpairs)
        data = data + 0.05 * (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt))
        seed = 12345
        np.random.seed(seed)
        A, b = ps.precompute_mpi(None, ants, antpairs, freqs[:nf], np.arange(nt) * 0.1,
                                 proj, data, inv_noise, gains, prior, realisation=True)
        # replay the draws in the reference's order (ptsrc_sampler.py:109-116,138)
        np.random.seed(seed)
        om_re = np.random.randn(nbl, nf, nt)
        om_im = np.random.randn(nbl, nf, nt)
        omega_n = (1.0 * om_re + 1.0j * om_im) / np.sqrt(2.0)
        omega_a = np.random.randn(ns)
        A0, b0 = ps.precompute_mpi(None, ants, antpairs, freqs[:nf], np.arange(nt) * 0.1,
                                   proj, data, inv_noise, gains, prior, realisation=False)
        np.savez(os.path.join(OUT, f"ptsrc_{tag}.npz"), ants=ants,
                 antpairs=np.array(antpairs), proj=proj, gains=gains, inv_noise=inv_noise,
                 prior=prior, data=data, seed=seed, omega_n=omega_n, omega_a=omega_a,
                 A=A, b=b, A0=A0, b0=b0)

    # ------------------------------------------------------------------- SH algebra
    rng = np.random.RandomState(21)
    lmax = 4
    ems, ells, idx = sh.get_em_ell_idx(lmax)
    nm = len(ems)
    nbl, nf, nt = 6, 3, 4
    M = nbl * nf * nt
    R = rng.randn(M, nm) + 1j * rng.randn(M, nm)
    inv_noise = rng.uniform(0.5, 2.0, size=(nbl, nf, nt))
    inv_prior = rng.uniform(1.0, 100.0, size=nm)
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    omega_a = rng.randn(nm)
    omega_n = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    a0 = rng.randn(nm)
    acr = rng.randn(nm)
    rhs = sh.construct_rhs_no_rot(data.flatten(), inv_noise.flatten(), inv_prior, omega_a,
                                  omega_n.flatten(), a0, R)
    rhs_mpi = sh.construct_rhs_no_rot_mpi(None, data, inv_noise, inv_prior, omega_a.copy(),
                                          omega_n, a0, R)
    lhs = sh.apply_lhs_no_rot(acr, inv_noise.flatten(), inv_prior, R)
    lhs_mpi = sh.apply_lhs_no_rot_mpi(None, acr.copy(), inv_noise, inv_prior, R)
    packed = rng.randn(nm)
    hp_modes = sh.alms2healpy(packed, lmax)
    x_cg = ls.cg(None, rhs_mpi, maxiters=1000, abs_tol=1e-8, use_norm_tol=False, x0=a0.copy(),
                 linear_op=lambda x: sh.apply_lhs_no_rot_mpi(None, x, inv_noise, inv_prior, R))
    np.savez(os.path.join(OUT, "sh_algebra.npz"), lmax=lmax, ems=np.array(ems),
             ells=np.array(ells), idx=np.array(idx), R=R, inv_noise=inv_noise,
             inv_prior=inv_prior, data=data, omega_a=omega_a, omega_n=omega_n, a0=a0, acr=acr,
             rhs=rhs, rhs_mpi=rhs_mpi, lhs=lhs, lhs_mpi=lhs_mpi, packed=packed,
             hp_modes=hp_modes, x_cg=x_cg)
    ems10, ells10, idx10 = sh.get_em_ell_idx(10)
    np.savez(os.path.join(OUT, "sh_layout_lmax10.npz"), ems=np.array(ems10),
             ells=np.array(ells10), idx=np.array(idx10))

    # ----------------------------------------------------------------------- CG
    np.random.seed(1)                       # hydra/tests/test_linear_solver.py:12-41
    n = 401
    xtrue = np.random.randn(n)
    A2 = np.eye(n) + 0.1 * np.random.randn(n, n)
    A2 = 0.5 * (A2 + A2.T)
    b2 = A2 @ xtrue
    x2 = ls.cg(A2, b2, maxiters=1000, abs_tol=1e-8, use_norm_tol=False)
    x2n = ls.cg(A2, b2, maxiters=1000, abs_tol=1e-6, use_norm_tol=True)
    x2_5 = ls.cg(A2, b2, maxiters=5, abs_tol=1e-8)
    rng = np.random.RandomState(5)
    B = rng.randn(60, 37)
    A3 = np.eye(37) + 0.05 * (B.T @ B)
    b3 = rng.randn(37)
    x03 = rng.randn(37)
    x3 = ls.cg(A3, b3, maxiters=1000, abs_tol=1e-10, x0=x03.copy())
    np.savez(os.path.join(OUT, "cg.npz"), xtrue=xtrue, A2=A2, b2=b2, x2=x2, x2n=x2n, x2_5=x2_5,
             A3=A3, b3=b3, x03=x03, x3=x3)

    # --------------------------------------------------------------- gain head
    rng = np.random.RandomState(31)
    nant, nf, nt = 6, 5, 4
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    x = rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt)
    mv = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    vproj = gs.apply_proj(x, A_real, A_imag, mv)
    vv = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    gconj = gs.apply_proj_conj(vv, A_real, A_imag, mv, (nant, nf, nt))
    fr = np.linspace(100.0, 120.0, nf)
    tm = np.linspace(0.2, 0.5, nt)
    Fb, _, _ = utils.partial_fourier_basis_2d_from_nmax(
        fr, tm, 2, 2, Lfreq=fr[-1] - fr[0], Ltime=tm[-1] - tm[0], freq0=fr[0], time0=tm[0],
        shape0=(nf, nt), positive_only=True)
    nmodes = Fb.shape[0]
    ps_sqrt = rng.uniform(0.05, 0.2, size=nmodes)
    inv_noise = rng.uniform(0.5, 2.0, size=(nbl, nf, nt))
    xvec = rng.randn(2 * nant * nmodes)
    resid = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)

    class _Comm:                      # one-rank communicator for the *_mpi entry points
        def Get_rank(self):
            return 0

        def Bcast(self, buf, root=0):
            return None

        def Allreduce(self, src, dst, op=None):
            dst[...] = np.asarray(src).reshape(dst.shape)

    gs.MPI_SUM = None
    opx = gs.apply_operator_mpi(_Comm(), xvec, inv_noise, ps_sqrt, A_real, A_imag, mv, Fb)
    seed = 77
    bvec = gs.construct_rhs_mpi(_Comm(), resid, inv_noise, ps_sqrt, A_real, A_imag, mv, Fb,
                                realisation=True, seed=seed)
    np.savez(os.path.join(OUT, "gain.npz"), ants=ants, antpairs=np.array(antpairs),
             A_real=A_real.toarray(), A_imag=A_imag.toarray(), x=x, mv=mv, vproj=vproj, vv=vv,
             gconj=gconj, Fb=Fb, ps_sqrt=ps_sqrt, inv_noise=inv_noise, xvec=xvec, opx=opx,
             resid=resid, seed=seed, bvec=bvec)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
