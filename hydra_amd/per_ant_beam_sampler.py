"""
Consumers of the visibility simulator in the per-antenna beam head on MI355X -- drop-in for
the corresponding functions of ``hydra.per_ant_beam_sampler``.

    get_bess_sky_contraction(...)          per_ant_beam_sampler.py:321-422
    get_bess_to_vis_from_contraction(...)  per_ant_beam_sampler.py:424-465
    get_bess_to_vis(...)                   per_ant_beam_sampler.py:468-583
    get_bess_matr, get_bess_outer, get_ant_inds, select_subarr   :115-161, 254-318

The reference simulates the per-source visibilities of a uniform-beam array one time sample at a
time, adds the cube to its conjugate transpose and contracts it over sources with the
Fourier-Bessel matrices (``tensordot``).  Here the antenna voltages of all (f,t) are generated
once and the pair products are formed inside the contraction kernel
(``hydra_bess_sky_contraction``, csrc/beamhead.hip); the per-source cube never exists.

The rest of the beam head (priors, covariance operators, plotting) is outside the hot path and is
not provided.  ``polarized=True`` raises ``NotImplementedError``; ``precision`` is accepted for
signature compatibility, the arithmetic is always double precision (the reference's output array
is complex128 in either case, :402).
"""
import numpy as np
import torch
from scipy.special import jn, jn_zeros

from . import _dev
from .beams import UniformBeam
from .vis_simulator import DEFAULT_LATITUDE, voltages_for


def get_ant_inds(ant_samp_ind, nants):
    """Boolean mask, true everywhere but at the antenna being sampled (:254-275)."""
    return np.arange(nants) != ant_samp_ind


def select_subarr(arr, ant_samp_ind, Nants):
    """``arr[:, :, :, :, other antennas, ant_samp_ind]`` of anything visibility-shaped (:277-298)."""
    return arr[:, :, :, :, get_ant_inds(ant_samp_ind, Nants), ant_samp_ind]


def get_bess_matr(nmodes, mmodes, rho, phi):
    """Sparse Fourier-Bessel basis evaluated at the polar positions ``rho``, ``phi`` (Ntimes, Nsrc)
    of the sources: (Ntimes, Nsrc, len(nmodes)) complex (:115-161).  Host set-up (scipy Bessel
    functions), as in the reference."""
    nmodes, mmodes = np.asarray(nmodes), np.asarray(mmodes)
    unique_n, ninv = np.unique(nmodes, return_inverse=True)
    bzeros = jn_zeros(0, np.amax(unique_n))
    bess_vals = jn(0, bzeros[np.newaxis, np.newaxis, :] * rho[:, :, np.newaxis])[:, :, ninv] \
        / jn(1, bzeros)[ninv]
    unique_m, minv = np.unique(mmodes, return_inverse=True)
    az_vals = np.exp(1.0j * unique_m[np.newaxis, np.newaxis, :] * phi[:, :, np.newaxis])[:, :, minv] \
        / np.sqrt(np.pi)
    return bess_vals * az_vals


def get_bess_outer(bess_matr):
    """Outer product of the design matrix with itself, conjugate on the first basis index:
    ``out[t,s,i,j] = conj(bess[t,s,i]) bess[t,s,j]`` (:301-318)."""
    if _dev.is_torch(bess_matr):
        return bess_matr[:, :, None] * bess_matr.conj()[:, :, :, None]
    return bess_matr[:, :, np.newaxis] * bess_matr.conj()[:, :, :, np.newaxis]


def _uniform_voltages(ants, fluxes, ra, dec, freqs, lsts, latitude):
    beams = [UniformBeam() for _ in range(len(ants))]
    return voltages_for(ants, fluxes, ra, dec, freqs, lsts, beams, latitude)


def get_bess_sky_contraction(bess_outer, ants, fluxes, ra, dec, freqs, lsts, polarized=False,
                             precision=1, latitude=DEFAULT_LATITUDE, use_feed="x", outer=True,
                             ref_beam_response=None):
    """Contract the (outer product of the) Fourier-Bessel design matrix with the sky model times
    the fringe pattern: a visibility simulation of the array for each pair of beam basis
    functions (:321-422).  ``freqs`` in Hz.  ``bess_outer`` (Ntimes, Nsrc, Nc, Nc), or
    (Ntimes, Nsrc, Nc) with ``outer=False``; ``ref_beam_response`` (Nfreqs, Ntimes, Nsrc).

    Returns (1, 1, Nfreqs, Ntimes, Nants, Nants, Nc[, Nc]) complex128 (numpy, or a GPU tensor if
    ``bess_outer`` is one)."""
    if polarized:
        raise NotImplementedError("polarized beams are outside the GCR hot path")
    B = _dev.c128(bess_outer)
    nt, ns, nc = int(B.shape[0]), int(B.shape[1]), int(B.shape[-1])
    assert B.dim() == (4 if outer else 3), "bess_outer does not match the `outer` setting"
    nq = nc * nc if outer else nc
    V = _uniform_voltages(ants, fluxes, ra, dec, freqs, lsts, latitude)
    nf, nt_v, nant, ns_v = V.shape
    assert (nt_v, ns_v) == (nt, ns), "bess_outer must have shape (Ntimes, Nsrc, ...)"
    ref = None
    if ref_beam_response is not None:
        ref = _dev.c128(ref_beam_response)
        assert tuple(ref.shape) == (nf, nt, ns)
    out = _dev.empty((nf, nt, nant, nant, nq), torch.complex128)
    _dev.call("hydra_bess_sky_contraction", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(B), nq,
              _dev.ptr(ref), -1, _dev.ptr(out), _dev.stream_ptr())
    shape = (1, 1, nf, nt, nant, nant, nc) + ((nc,) if outer else ())
    return _dev.like(bess_outer, out.reshape(shape))


def _coeffs(beam_coeffs, nf, nant):
    c = _dev.c128(beam_coeffs)
    assert c.dim() == 5 and tuple(c.shape[1:]) == (nf, nant, 1, 1), \
        "beam_coeffs must have shape (ncoeff, NFREQS, NANTS, 1, 1)"
    return c.reshape(c.shape[0], nf, nant).contiguous()


def get_bess_to_vis_from_contraction(bess_sky_contraction, beam_coeffs, Nants, ant_samp_ind,
                                     ref_contraction=False):
    """Linear operator from the sampled antenna's beam coefficients to visibilities (:424-465):
    the conjugated coefficients of the other antennas (``beam_coeffs`` (Nbasis, Nfreqs, Nants, 1,
    1)) contracted with column ``ant_samp_ind`` of the sky contraction.  Returns
    (1, 1, Nfreqs, Ntimes, Nants-1, Nbasis), or (1, 1, Nfreqs, Ntimes, Nants-1) with
    ``ref_contraction``."""
    C = _dev.c128(bess_sky_contraction)
    assert tuple(C.shape[:2]) == (1, 1), "polarized contractions are outside the GCR hot path"
    nf, nt, nant = int(C.shape[2]), int(C.shape[3]), int(C.shape[4])
    assert nant == Nants and C.shape[5] == nant
    nc = int(C.shape[6])
    nb = 1 if ref_contraction else nc
    assert C.dim() == (7 if ref_contraction else 8)
    coef = _coeffs(beam_coeffs, nf, nant)
    assert coef.shape[0] == nc
    out = _dev.empty((nf, nt, nant - 1, nb), torch.complex128)
    _dev.call("hydra_bess_to_vis", _dev.ptr(C), _dev.ptr(coef), nf, nt, nant, nc, nb,
              int(ant_samp_ind), 0, _dev.ptr(out), _dev.stream_ptr())
    shape = (1, 1, nf, nt, nant - 1) + (() if ref_contraction else (nc,))
    return _dev.like(bess_sky_contraction, out.reshape(shape))


def get_bess_to_vis(bess_matr, ants, fluxes, ra, dec, freqs, lsts, beam_coeffs, ant_samp_ind,
                    polarized=False, precision=1, latitude=DEFAULT_LATITUDE, use_feed="x"):
    """The same operator straight from the design matrix ``bess_matr`` (Ntimes, Nsrc, Nbasis)
    (:468-583; the reference calls it the slower alternative of the route through
    ``get_bess_sky_contraction``).  Only the sampled antenna's column of the pair matrix is
    contracted.  Returns (1, 1, Nfreqs, Ntimes, Nants-1, Nbasis)."""
    if polarized:
        raise NotImplementedError("polarized beams are outside the GCR hot path")
    bm = _dev.c128(bess_matr)
    nt, ns, nc = (int(x) for x in bm.shape)
    B = (bm[:, :, None] * bm.conj()[:, :, :, None]).contiguous()      # get_bess_outer
    V = _uniform_voltages(ants, fluxes, ra, dec, freqs, lsts, latitude)
    nf, _, nant, _ = V.shape
    col = _dev.empty((nf, nt, nant, nc * nc), torch.complex128)
    _dev.call("hydra_bess_sky_contraction", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(B), nc * nc,
              _dev.ptr(None), int(ant_samp_ind), _dev.ptr(col), _dev.stream_ptr())
    coef = _coeffs(beam_coeffs, nf, nant)
    out = _dev.empty((nf, nt, nant - 1, nc), torch.complex128)
    _dev.call("hydra_bess_to_vis", _dev.ptr(col), _dev.ptr(coef), nf, nt, nant, nc, nc,
              int(ant_samp_ind), 1, _dev.ptr(out), _dev.stream_ptr())
    return _dev.like(bess_matr, out.reshape(1, 1, nf, nt, nant - 1, nc))
