"""
Oracle (test infrastructure): small numpy helpers the hot path and its driver
call, restating ``hydra/utils.py`` and ``hydra/example.py`` (PINNED by
tests/golden/helpers_*.npz generated from the reference functions).
"""
import numpy as np


def apply_gains(v, gains, ants, antpairs, perturbation=None, inline=False):
    """v[k] *= g_i conj(g_j) (1 + x_i + conj(x_j))   (utils.py:63-110)."""
    out = v if inline else v.copy()
    assert v.shape[0] == len(antpairs), (
        "Input array `v` has shape that is incompatible with `antpairs`")
    ants = np.asarray(ants)
    for k, (a1, a2) in enumerate(antpairs):
        i1 = np.where(ants == a1)[0][0]
        i2 = np.where(ants == a2)[0][0]
        fac = 1.0
        if perturbation is not None:
            fac = 1.0 + perturbation[i1] + perturbation[i2].conj()
        out[k, :, :] *= gains[i1] * gains[i2].conj() * fac
    return out


def build_hex_array(hex_spec=(3, 4), ants_per_row=None, d=14.6):
    """Hexagonally packed antenna dict {id: (x, y, 0)}   (utils.py:374-419)."""
    if ants_per_row is None:
        up = np.arange(hex_spec[0], hex_spec[1] + 1).tolist()
        ants_per_row = up[:-1] + up[::-1]
    dy = d * np.sqrt(3) / 2.0
    ants, k = {}, -1
    for j, n in enumerate(ants_per_row):
        y = -0.5 * dy * (len(ants_per_row) - 1) + dy * j
        x = np.linspace(-d * (n - 1) / 2.0, d * (n - 1) / 2.0, n)
        for i in range(n):
            k += 1
            ants[k] = (x[i], y, 0.0)
    return ants


def get_flux_from_ptsrc_amp(ptsrc_amps, freqs, beta_ptsrc, curv_ptsrc=None,
                            ref_freq=100.0):
    """S (nu/nu_ref)^(beta + c ln(nu/nu_ref)), shape (Nsrc, Nf)  (utils.py:461-501)."""
    if isinstance(beta_ptsrc, float):
        beta_ptsrc = beta_ptsrc * np.ones_like(ptsrc_amps)
    assert ptsrc_amps.size == beta_ptsrc.size
    if curv_ptsrc is not None and isinstance(curv_ptsrc, float):
        curv_ptsrc = curv_ptsrc * np.ones_like(ptsrc_amps)
    if curv_ptsrc is None:
        curv_ptsrc = np.zeros_like(ptsrc_amps)
    assert ptsrc_amps.size == curv_ptsrc.size
    x = freqs / ref_freq
    idx = beta_ptsrc[:, np.newaxis] + curv_ptsrc[:, np.newaxis] * np.log(x)[np.newaxis, :]
    return ptsrc_amps[:, np.newaxis] * x[np.newaxis, :] ** idx


def generate_random_ptsrc_catalogue(Nptsrc, ra_bounds, dec_bounds,
                                    logflux_bounds=(-1.0, 2.0)):
    """Uniform-on-sphere random catalogue from the global numpy RNG, draws in
    the reference's order: ra, U (for dec), log-flux  (hydra/example.py:21-72)."""
    ra_lo, ra_hi = min(ra_bounds), max(ra_bounds)
    dec_lo, dec_hi = min(dec_bounds), max(dec_bounds)
    lf_lo, lf_hi = min(logflux_bounds), max(logflux_bounds)
    ra = np.random.uniform(low=ra_lo, high=ra_hi, size=Nptsrc)
    U = np.random.uniform(low=0, high=1, size=Nptsrc)
    dsin = np.sin(dec_hi) - np.sin(dec_lo)
    dec = np.arcsin(U * dsin + np.sin(dec_lo))
    amps = 10.0 ** np.random.uniform(low=lf_lo, high=lf_hi, size=Nptsrc)
    return ra, dec, amps


def extract_vis_from_sim(ants, antpairs, sim_vis):
    """(Nf, Nt, Nant, Nant) -> (Nbl, Nf, Nt), un-conjugated order  (utils.py:159-198)."""
    nf, nt, _, _ = sim_vis.shape
    ants = np.asarray(ants)
    out = np.zeros((len(antpairs), nf, nt), dtype=sim_vis.dtype)
    for k, (a1, a2) in enumerate(antpairs):
        if a1 > a2:
            a1, a2 = a2, a1
        i1 = np.where(ants == a1)[0][0]
        i2 = np.where(ants == a2)[0][0]
        out[k] = sim_vis[:, :, i1, i2]
    return out


def chunk_grid(nworkers, nfreqs, ntimes, fchunks, tchunks=1):
    """Worker -> (freq idxs, time idxs) map of ``freqs_times_for_worker``
    (utils.py:313-371), without the communicator."""
    assert nworkers <= fchunks * tchunks, (
        "There are more workers than time and frequency chunks")
    grid = np.arange(fchunks * tchunks).reshape((fchunks, tchunks))
    fsplit = np.array_split(np.arange(nfreqs), fchunks)
    tsplit = np.array_split(np.arange(ntimes), tchunks)
    out = {}
    for w in range(nworkers):
        fi, ti = np.where(grid == w)
        fi, ti = int(fi[0]), int(ti[0])
        out[w] = (fi, ti, fsplit[fi], tsplit[ti])
    return out


def partial_fourier_basis_2d(freqs, times, nfreq, ntime, Lfreq, Ltime, freq0=None,
                             time0=None, shape0=None):
    """exp(i (k_f (f - f0) + k_t (t - t0))) / sqrt(Nf Nt), shape (Nmodes, Nf, Nt)
    (utils.py:625-669)."""
    time0 = times[0] if time0 is None else time0
    freq0 = freqs[0] if freq0 is None else freq0
    nf, nt = (freqs.size, times.size) if shape0 is None else shape0
    nfreq = np.atleast_1d(nfreq)
    ntime = np.atleast_1d(ntime)
    t2d, f2d = np.meshgrid(times - time0, freqs - freq0)
    kf = 2.0 * np.pi * nfreq / Lfreq
    kt = 2.0 * np.pi * ntime / Ltime
    basis = np.exp(1.0j * ((kf[:, None, None] * f2d[None, :, :])
                           + (kt[:, None, None] * t2d[None, :, :]))) / np.sqrt(nf * nt)
    return basis, kf, kt


def partial_fourier_basis_2d_from_nmax(freqs, times, nmaxfreq, nmaxtime, Lfreq, Ltime,
                                       freq0=None, time0=None, shape0=None,
                                       positive_only=False):
    """utils.py:672-710."""
    if positive_only:
        nf1, nt1 = np.arange(0, nmaxfreq), np.arange(0, nmaxtime)
    else:
        nf1, nt1 = np.arange(-nmaxfreq, nmaxfreq), np.arange(-nmaxtime, nmaxtime)
    nfreq, ntime = np.meshgrid(nf1, nt1)
    return partial_fourier_basis_2d(freqs, times, nfreq.flatten(), ntime.flatten(),
                                    Lfreq, Ltime, freq0, time0, shape0)


def flatten_vector(v):
    """utils.py:21-36 (no reduced_idxs)."""
    return np.concatenate((v.real.flatten(), v.imag.flatten()))


def reconstruct_vector(v, shape):
    """utils.py:39-60 (no reduced_idxs)."""
    return (v[: v.size // 2] + 1.0j * v[v.size // 2:]).reshape(shape)
