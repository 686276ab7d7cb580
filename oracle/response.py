"""
Oracle (test infrastructure): the visibility response simulator,
restating ``hydra/vis_simulator.py`` (unpolarised, precision=2 -- the only mode
the GCR hot path uses: ``ptsrc_sampler.py:202-205``).

The operation order of the reference is kept so rounding matches:
  Isqrt = sqrt(0.5 I)                      vis_simulator.py:214
  crd_top = eq2top @ crd_eq                :233
  horizon mask tz > 0                      :237
  (az, za) from (tx, ty)                   :247
  sqrt(power beam) per beam                :249-253, 347-359
  tau = antpos @ crd_top / c               :261-262
  v = exp(1j * (2 pi nu * tau)) * Isqrt    :267-271
  v[:, below horizon] *= 0                 :272
  v = A_s[beam_idx] * v                    :276
  vis[t, i, j>=i, s] = conj(v_i) v_j       :281-289   (j < i left at zero)

Third-party boundary (geometry, beams): PARITY UNPINNED, see oracle/__init__.py.
"""
import warnings
import numpy as np

from . import geometry, sphere


def _default_beam_idx(nant, nbeam, beam_idx):
    # vis_simulator.py:62-76 (run_checks)
    if beam_idx is None:
        if nbeam == 1:
            return np.zeros(nant, dtype=int)
        if nbeam == nant:
            return np.arange(nant, dtype=int)
        raise ValueError(
            "If number of beams provided is not 1 or nant, beam_idx must be provided."
        )
    beam_idx = np.asarray(beam_idx)
    assert beam_idx.shape == (nant,), "beam_idx must be length nant"
    assert all(0 <= i < nbeam for i in beam_idx), (
        "beam_idx contains indices greater than the number of beams")
    return beam_idx


def beam_sqrt_power(bm, freq, az, za, force_no_beam_sqrt=False):
    """``get_interp_beam_for_sim`` for the unpolarised case
    (vis_simulator.py:347-359)."""
    resp = bm.compute_response(az_array=az, za_array=za,
                               freq_array=np.atleast_1d(freq))[0]
    if force_no_beam_sqrt:
        return resp[0, 0, :]
    return np.sqrt(resp[0, 0, :])


def antenna_voltages(antpos, freq, eq2top, crd_eq, Isqrt, beam_list, beam_idx,
                     force_no_beam_sqrt=False):
    """Complex voltage per (antenna, source) at one frequency and time:
    vis_simulator.py:233-276.  Returns (Nant, Nsrc) complex128."""
    tx, ty, tz = crd_top = np.dot(eq2top, crd_eq)
    above_horizon = tz > 0
    az, za = geometry.enu_to_az_za(enu_e=tx, enu_n=ty, orientation="uvbeam")
    nbeam = len(beam_list)
    A_s = np.zeros((nbeam, tx.size), dtype=np.complex128)
    for k, bm in enumerate(beam_list):
        A_s[k] = beam_sqrt_power(bm, freq, az, za, force_no_beam_sqrt)
    if np.any(np.isinf(A_s)) or np.any(np.isnan(A_s)):
        raise ValueError("Beam interpolation resulted in an invalid value")
    tau = np.dot(antpos, crd_top)
    tau /= geometry.C_LIGHT
    ang_freq = 2.0 * np.pi * freq
    v = np.exp(1.0j * (ang_freq * tau))
    v *= Isqrt
    v[:, ~above_horizon] *= 0.0
    return A_s[beam_idx] * v


def vis_sim_per_source(antpos, freq, eq2tops, crd_eq, I_sky, beam_list,
                       beam_idx=None, subarr_ant=None, force_no_beam_sqrt=False):
    """vis_simulator.py:108-300 (polarized=False, precision=2).
    Returns (Nt, Nant, Nant, Nsrc), or (Nt, Nant, Nsrc) with ``subarr_ant``."""
    antpos = np.asarray(antpos, dtype=np.float64)
    nant, ncrd = antpos.shape
    assert ncrd == 3, "antpos must have shape (NANTS, 3)."
    ntimes, n1, n2 = eq2tops.shape
    assert n1 == 3 and n2 == 3, "eq2tops must have shape (NTIMES, 3, 3)."
    ncrd, nsrcs = crd_eq.shape
    assert ncrd == 3, "crd_eq must have shape (3, NSRCS)."
    assert I_sky.ndim == 1 and I_sky.shape[0] == nsrcs, "I_sky must have shape (NSRCS,)."
    beam_idx = _default_beam_idx(nant, len(beam_list), beam_idx)

    Isqrt = np.sqrt(0.5 * I_sky).astype(np.float64)
    crd_eq = crd_eq.astype(np.float64)
    if subarr_ant is None:
        vis = np.zeros((ntimes, nant, nant, nsrcs), dtype=np.complex128)
    else:
        vis = np.zeros((ntimes, nant, nsrcs), dtype=np.complex128)
    for t in range(ntimes):
        v = antenna_voltages(antpos, freq, eq2tops[t].astype(np.float64), crd_eq,
                             Isqrt, beam_list, beam_idx, force_no_beam_sqrt)
        if subarr_ant is None:
            for i in range(nant):
                vis[t, i, i:, :] = v[i].conj()[np.newaxis, :] * v[i:, :]
        else:
            vis[t] = v.conj() * v[subarr_ant][np.newaxis, :]
    return vis


def simulate_vis_per_source(ants, fluxes, ra, dec, freqs, lsts, beams,
                            latitude=-30.7215 * np.pi / 180.0, subarr_ant=None,
                            force_no_beam_sqrt=False):
    """vis_simulator.py:363-515.  ``freqs`` in Hz.
    Returns (Nf, Nt, Nant, Nant, Nsrc) complex128."""
    ra = np.asarray(ra)
    dec = np.asarray(dec)
    freqs = np.asarray(freqs)
    lsts = np.asarray(lsts)
    assert len(ants) == len(beams), (
        "The `beams` list must have as many entries as the ``ants`` dict.")
    assert fluxes.shape == (ra.size, freqs.size), (
        "The `fluxes` array must have shape (NSRCS, NFREQS).")
    if np.any(ra < 0.0) or np.any(ra > 2.0 * np.pi):
        warnings.warn("One or more ra values is outside the allowed range (0, 2 pi)",
                      RuntimeWarning)
    if np.any(dec < -np.pi) or np.any(dec > np.pi):
        warnings.warn("One or more dec values is outside the allowed range (-pi, +pi)",
                      RuntimeWarning)
    antpos = np.array([ants[k] for k in ants.keys()])
    crd_eq = geometry.point_source_crd_eq(ra, dec)
    eq2tops = np.array([geometry.eci_to_enu_matrix(lst, latitude) for lst in lsts])
    out = []
    for i in range(freqs.size):
        out.append(vis_sim_per_source(antpos, freqs[i], eq2tops, crd_eq, fluxes[:, i],
                                      beam_list=list(beams), subarr_ant=subarr_ant,
                                      force_no_beam_sqrt=force_no_beam_sqrt))
    return np.asarray(out)


def simulate_vis(*args, **kwargs):
    """vis_simulator.py:518-568: sum of the per-source cube over sources."""
    return np.sum(simulate_vis_per_source(*args, **kwargs), axis=-1)


def simulate_vis_per_alm(lmax, nside, ants, freqs, lsts, beams,
                         latitude=-30.7215 * np.pi / 180.0, amplitude=1.0):
    """vis_simulator.py:571-736.  ``freqs`` in Hz.  Returns (ell, m, vis) with
    vis of shape (Nf, Nt, Nant, Nant, 2*nalm)."""
    freqs = np.atleast_1d(freqs)
    lsts = np.atleast_1d(lsts)
    ell, m = sphere.getlm(lmax)
    npix = sphere.nside2npix(nside)
    pix_area = 4.0 * np.pi / npix
    # reference: dec, ra = hp.pix2ang(...); dec = dec - pi/2   (:659-661)
    dec, ra = sphere.pix2ang_ring(nside, np.arange(npix))
    dec = dec - 0.5 * np.pi
    fluxes = np.ones((npix, freqs.size))
    vis_pix = simulate_vis_per_source(ants=ants, fluxes=fluxes, ra=ra, dec=dec,
                                      freqs=freqs, lsts=lsts, beams=beams,
                                      latitude=latitude)
    shape = list(vis_pix.shape)
    shape[-1] = 2 * ell.size
    vis = np.zeros(shape, dtype=np.complex128)
    maps = sphere.unit_mode_maps(lmax, nside)
    for n in range(2 * ell.size):
        skymap = maps[n] * pix_area * amplitude
        vis[..., n] = np.sum(vis_pix * skymap, axis=-1)
    return ell, m, vis
