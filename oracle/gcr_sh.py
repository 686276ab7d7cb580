"""
Oracle (test infrastructure): spherical-harmonic GCR pieces, restating
``hydra/sh_sampler.py`` (algebra PINNED by tests/golden/sh_*.npz; the response
operator itself inherits the unpinned third-party boundary of ``response``).
"""
import numpy as np

from . import response, sphere


def get_em_ell_idx(lmax):
    """Parameter-vector layout [Re a_lm for all (m, l); Im a_lm for m >= 1],
    m-major (sh_sampler.py:28-62).  Returns (ems, ells, idx) lists."""
    ems, ells = [], []
    for part_first_m in (0, 1):
        for em in range(part_first_m, lmax + 1):
            for ell in range(em, lmax + 1):
                ems.append(em)
                ells.append(ell)
    return ems, ells, list(range(len(ems)))


def vis_proj_operator_no_rot(freqs, lsts, beams, ant_pos, lmax, nside,
                             latitude=-0.5361913261514378, include_autos=False,
                             autos_only=False, ref_freq=100.0, spectral_idx=0.0):
    """sh_sampler.py:65-196.  ``freqs`` in MHz."""
    ell, m, vis_alm = response.simulate_vis_per_alm(
        lmax=lmax, nside=nside, ants=ant_pos, freqs=freqs * 1e6, lsts=lsts,
        beams=beams, latitude=latitude)
    nl = len(ell)
    # drop the (lmax+1) imaginary m=0 columns            (:127-131)
    keep = np.concatenate((np.arange(nl), np.arange(nl + lmax + 1, 2 * nl)))
    vis_alm = vis_alm[..., keep]
    names = list(ant_pos.keys())
    cross, autos = [], []
    for i in names:                                     # (:137-150) keys used as indices
        for j in names:
            if include_autos:
                if j >= i:
                    cross.append((names[i], names[j]))
            elif autos_only:
                if j == i:
                    cross.append((names[i], names[j]))
            else:
                if j == i:
                    autos.append((names[i], names[j]))
                if j > i:
                    cross.append((names[i], names[j]))
    nm = 2 * nl - (lmax + 1)
    spec = ((freqs / ref_freq) ** spectral_idx)[np.newaxis, :, np.newaxis, np.newaxis]

    def gather(pairs):
        out = np.zeros((len(pairs), len(freqs), len(lsts), nm), dtype=np.complex128)
        for k, (a1, a2) in enumerate(pairs):
            out[k] = vis_alm[:, :, names.index(a1), names.index(a2), :]
        return out

    resp = gather(cross)
    resp *= spec
    resp2d = resp.reshape(len(cross) * len(freqs) * len(lsts), nm)
    if not autos_only and not include_autos:
        au = gather(autos)           # note: no spectral factor on autos (:176-192)
        return resp2d, au.reshape(len(autos) * len(freqs) * len(lsts), nm), ell, m
    return resp2d, ell, m


def alms2healpy(alms, lmax):
    """[Re; Im(m>=1)] -> healpy complex a_lm  (sh_sampler.py:199-232)."""
    alms = np.asarray(alms)
    if alms.ndim == 1:
        split = int((np.size(alms) + (lmax + 1)) / 2)
        imag = np.concatenate((np.zeros(lmax + 1), alms[split:]))
        return alms[:split] + 1.0j * imag
    if alms.ndim == 2:
        return np.array([alms2healpy(a, lmax) for a in alms])
    raise ValueError("alms array must either have shape (Nmodes,) or (Nmaps, Nmodes)")


def healpy2alms(healpy_modes):
    """healpy complex a_lm -> [Re; Im(m>=1)]  (sh_sampler.py:235-266)."""
    healpy_modes = np.asarray(healpy_modes)
    if healpy_modes.ndim == 1:
        lmax = sphere.getlmax(healpy_modes.size)
        return np.concatenate((healpy_modes.real, healpy_modes.imag[(lmax + 1):]))
    if healpy_modes.ndim == 2:
        return np.array([healpy2alms(a) for a in healpy_modes])
    raise ValueError("Input array must have shape (Nmodes,) or (Nmaps, Nmodes).")


def construct_rhs_no_rot(data, inv_noise_var, inv_prior_var, omega_0, omega_1, a_0,
                         vis_response):
    """sh_sampler.py:370-386 (and the comm=None branch of :405-448, which
    flattens its inputs first)."""
    data = np.asarray(data).flatten()
    inv_noise_var = np.asarray(inv_noise_var).flatten()
    omega_1 = np.asarray(omega_1).flatten()
    re = vis_response.real.T @ (inv_noise_var * data.real
                                + np.sqrt(inv_noise_var) * omega_1.real)
    im = vis_response.imag.T @ (inv_noise_var * data.imag
                                + np.sqrt(inv_noise_var) * omega_1.imag)
    prior = inv_prior_var * a_0 + np.sqrt(inv_prior_var) * omega_0
    return re + im + prior


def apply_lhs_no_rot(a_cr, inv_noise_var, inv_prior_var, vis_response):
    """sh_sampler.py:389-402 -- left-associative, i.e. the (Nm, Nm) Gram is
    formed on every call, exactly as the reference does."""
    w = np.asarray(inv_noise_var).flatten()[:, np.newaxis]
    re = vis_response.real.T @ (w * vis_response.real) @ a_cr
    im = vis_response.imag.T @ (w * vis_response.imag) @ a_cr
    return re + im + inv_prior_var * a_cr


def apply_lhs_no_rot_matfree(a_cr, inv_noise_var, inv_prior_var, vis_response):
    """Same operator, right-associated: R^T (N^-1 (R a)).  Differs from
    ``apply_lhs_no_rot`` at rounding level only."""
    w = np.asarray(inv_noise_var).flatten()
    re = vis_response.real.T @ (w * (vis_response.real @ a_cr))
    im = vis_response.imag.T @ (w * (vis_response.imag @ a_cr))
    return re + im + inv_prior_var * a_cr
