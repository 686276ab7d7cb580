"""
Oracle (test infrastructure): the consumers of the visibility simulator in the per-antenna
beam head, restating ``hydra/per_ant_beam_sampler.py``.

  get_ant_inds, get_bess_matr, get_bess_outer, get_bess_to_vis_from_contraction
        per_ant_beam_sampler.py:254-275, 115-161, 301-318, 424-465 -- PINNED by
        tests/golden/beam.npz, generated from the reference's own functions
  get_bess_sky_contraction, get_bess_to_vis
        per_ant_beam_sampler.py:321-422, 468-583 -- the tensor algebra follows the reference line
        by line, the per-source visibilities come from the oracle's simulator with a uniform
        beam, so these inherit its status ("parity unpinned" at the geometry boundary, see
        oracle/response.py)
"""
import numpy as np
from scipy.special import jn, jn_zeros

from . import response
from .beams import UniformBeam


def get_ant_inds(ant_samp_ind, nants):
    """Mask of the antennas that are not being sampled (:254-275)."""
    return np.arange(nants) != ant_samp_ind


def get_bess_matr(nmodes, mmodes, rho, phi):
    """Sparse Fourier-Bessel basis at polar positions rho, phi of shape (Nt, Ns) (:115-161).
    Radial part J0(z_k rho) / J1(z_k) with z_k the zeros of J0, azimuthal exp(i m phi)/sqrt(pi);
    the radial column of mode n is picked by its rank among the unique n, as in the reference."""
    uniq_n, n_rank = np.unique(nmodes, return_inverse=True)
    zeros = jn_zeros(0, np.amax(uniq_n))
    radial = jn(0, zeros[np.newaxis, np.newaxis, :] * rho[:, :, np.newaxis])[:, :, n_rank] \
        / jn(1, zeros)[n_rank]
    uniq_m, m_rank = np.unique(mmodes, return_inverse=True)
    azim = np.exp(1.0j * uniq_m[np.newaxis, np.newaxis, :] * phi[:, :, np.newaxis])[:, :, m_rank] \
        / np.sqrt(np.pi)
    return radial * azim


def get_bess_outer(bess_matr):
    """out[t,s,i,j] = conj(bess[t,s,i]) bess[t,s,j]  (:301-318)."""
    return bess_matr[:, :, np.newaxis] * bess_matr.conj()[:, :, :, np.newaxis]


def get_bess_sky_contraction(bess_outer, ants, fluxes, ra, dec, freqs, lsts,
                             latitude=-30.7215 * np.pi / 180.0, outer=True,
                             ref_beam_response=None):
    """(1, 1, Nf, Nt, Nant, Nant, Nc[, Nc]): per time, the uniform-beam per-source visibilities
    made Hermitian in the antenna pair (which doubles the autos) contracted over sources with
    ``bess_outer[t]``  (:385-422).  ``freqs`` in Hz."""
    nf, nt, nant = len(freqs), len(lsts), len(ants)
    nc = bess_outer.shape[-1]
    shape = [1, 1, nf, nt, nant, nant, nc] + ([nc] if outer else [])
    out = np.zeros(shape, dtype=complex)
    beams = [UniformBeam() for _ in range(nant)]
    for t in range(nt):
        sky = response.simulate_vis_per_source(ants, fluxes, ra, dec, freqs, lsts[t:t + 1],
                                               beams=beams, latitude=latitude)
        sky = sky[np.newaxis, np.newaxis, :]
        if ref_beam_response is not None:
            sky = sky * ref_beam_response[:, None, None, None, t]
        sky = sky + sky.swapaxes(4, 5).conj()
        out[:, :, :, t] = np.tensordot(sky[:, :, :, 0], bess_outer[t], axes=((-1,), (0,)))
    return out


def get_bess_to_vis_from_contraction(bess_sky_contraction, beam_coeffs, nants, ant_samp_ind,
                                     ref_contraction=False):
    """Operator from the sampled antenna's beam coefficients to visibilities (:424-465):
    conj of the other antennas' coefficients (Nc, Nf, Nant, Npol, Npol) contracted with the
    column ``ant_samp_ind`` of the sky contraction."""
    keep = get_ant_inds(ant_samp_ind, nants)
    res = beam_coeffs.transpose((2, 3, 1, 0, 4))[keep]          # bfApQ -> ApfbQ
    spec = "ApfbQ,qQftAb->pqftA" if ref_contraction else "ApfbQ,qQftAbB->pqftAB"
    return np.einsum(spec, res.conj(), bess_sky_contraction[:, :, :, :, keep, ant_samp_ind],
                     optimize=True)


def get_bess_to_vis(bess_matr, ants, fluxes, ra, dec, freqs, lsts, beam_coeffs, ant_samp_ind,
                    latitude=-30.7215 * np.pi / 180.0):
    """The same operator without the intermediate contraction (:468-583): the sampled antenna's
    row of the uniform-beam per-source visibilities, weighted by the other antennas' beams on
    the sky, projected on the basis.  Returns (1, 1, Nf, Nt, Nant-1, Nc)."""
    nants = len(ants)
    keep = get_ant_inds(ant_samp_ind, nants)
    beams = [UniformBeam() for _ in range(nants)]
    sky = response.simulate_vis_per_source(ants, fluxes, ra, dec, freqs, lsts, beams=beams,
                                           latitude=latitude, subarr_ant=ant_samp_ind)
    sky = sky[:, :, keep][np.newaxis, np.newaxis, :]
    res = beam_coeffs.transpose((2, 3, 1, 0, 4))[keep]          # BfaqQ -> aqfBQ
    on_sky = np.tensordot(res.conj(), bess_matr.conj(), axes=((3,), (2,))) \
        .transpose((3, 1, 2, 4, 0, 5))                          # Qqftas
    sky = (on_sky[:, :, np.newaxis] * sky[:, np.newaxis]).sum(axis=0)
    out = np.zeros((1, 1, len(freqs), len(lsts), nants - 1, res.shape[-2]), dtype=sky.dtype)
    for t in range(len(lsts)):
        out[:, :, :, t] = np.tensordot(sky[:, :, :, t], bess_matr[t], axes=((-1,), (0,)))
    return out
