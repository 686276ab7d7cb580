"""
Utilities used by the GCR hot path and its driver -- drop-in for the corresponding
functions of ``hydra.utils`` (reference: hydra/utils.py).  ``apply_gains`` runs on the GPU;
the rest is small host-side set-up (array layout, SEDs, chunk maps, Fourier basis) that the
reference also does once per run in numpy.
"""
import numpy as np
import torch

from . import _dev, _tables


def flatten_vector(v, reduced_idxs=None):
    """Complex array -> [real block, imaginary block] (utils.py:21-36)."""
    out = np.concatenate((v.real.flatten(), v.imag.flatten()))
    return out if reduced_idxs is None else out[reduced_idxs]


def reconstruct_vector(v, shape, reduced_idxs=None):
    """Inverse of ``flatten_vector`` (utils.py:39-60)."""
    if reduced_idxs is not None:
        vv = np.zeros(2 * int(np.prod(shape)), dtype=v.dtype)
        vv[reduced_idxs] = v[:]
    else:
        vv = v
    half = vv.size // 2
    return (vv[:half] + 1.0j * vv[half:]).reshape(shape)


def apply_gains(v, gains, ants, antpairs, perturbation=None, inline=False):
    """v[k] *= g_i conj(g_j) (1 + x_i + conj(x_j)) for every baseline k = (i, j)
    (utils.py:63-110).  ``v`` (Nbl,Nf,Nt), ``gains``/``perturbation`` (Nants,Nf,Nt)."""
    assert v.shape[0] == len(antpairs), (
        "Input array `v` has shape that is incompatible with `antpairs`")
    bl_i, bl_j = _tables.baseline_indices(ants, antpairs)
    nbl, nf, nt = v.shape
    vd = _dev.c128(v)
    if _dev.is_torch(v) and not inline and vd.data_ptr() == v.data_ptr():
        vd = vd.clone()
    g = _dev.c128(gains)
    g = g.expand(g.shape[0], nf, nt).contiguous() if tuple(g.shape[1:]) != (nf, nt) else g
    pert = None
    if perturbation is not None:
        pert = _dev.c128(perturbation)
        pert = pert.expand(pert.shape[0], nf, nt).contiguous()
    bi_d, bj_d = _dev.i32(bl_i), _dev.i32(bl_j)       # keep alive across the launch
    _dev.call("hydra_apply_gains", _dev.ptr(vd), _dev.ptr(g), _dev.ptr(pert), _dev.ptr(bi_d),
              _dev.ptr(bj_d), nbl, nf, nt, _dev.stream_ptr())
    if _dev.is_torch(v):
        if inline and vd.data_ptr() != v.data_ptr():
            v.copy_(vd)
            return v
        return vd
    res = vd.cpu().numpy()
    if inline:
        v[...] = res
        return v
    return res


def extract_vis_from_sim(ants, antpairs, sim_vis):
    """(Nf,Nt,Nants,Nants) -> (Nbl,Nf,Nt) for the requested pairs, un-conjugated order
    (utils.py:159-198)."""
    nf, nt, _, _ = sim_vis.shape
    ants = np.asarray(ants)
    xp_torch = _dev.is_torch(sim_vis)
    out = (torch.zeros((len(antpairs), nf, nt), dtype=sim_vis.dtype, device=sim_vis.device)
           if xp_torch else np.zeros((len(antpairs), nf, nt), dtype=sim_vis.dtype))
    for k, (a1, a2) in enumerate(antpairs):
        if a1 > a2:
            a1, a2 = a2, a1
        i1 = int(np.where(ants == a1)[0][0])
        i2 = int(np.where(ants == a2)[0][0])
        out[k] = sim_vis[:, :, i1, i2]
    return out


def timing_info(fname, iter, task, duration, verbose=True):
    """Append a timing line to ``fname`` (utils.py:291-310)."""
    with open(fname, "a") as f:
        f.write("%05d %7.5f %s\n" % (iter, duration, task))
        if verbose:
            print("%s took %3.2f sec" % (task, duration))


def freqs_times_for_worker(comm, freqs, times, fchunks, tchunks=1):
    """(freq idxs, time idxs, worker map) of this rank on the fchunks x tchunks grid
    (utils.py:313-371)."""
    myid, nworkers = comm.Get_rank(), comm.Get_size()
    assert nworkers <= fchunks * tchunks, "There are more workers than time and frequency chunks"
    grid = np.arange(fchunks * tchunks).reshape((fchunks, tchunks))
    fsplit = np.array_split(np.arange(np.size(freqs)), fchunks)
    tsplit = np.array_split(np.arange(np.size(times)), tchunks)
    worker_map = {}
    for w in range(nworkers):
        fi, ti = np.where(grid == w)
        fi, ti = int(fi[0]), int(ti[0])
        worker_map[w] = (fi, ti, fsplit[fi], tsplit[ti])
    fi, ti = worker_map[myid][:2]
    return fsplit[fi], tsplit[ti], worker_map


def convert_to_tops(ra, dec, lsts, latitude, precision=1):
    """Topocentric direction cosines (l, m, n) = (east, north, up) of the directions (ra, dec)
    at the given LSTs from ``latitude`` (utils.py:422-458): three arrays (Nt, Nsrc).  The
    equatorial unit vectors and the ECI->ENU matrices come from the same device kernels the
    simulator uses (hydra_crd_eq / hydra_eq2tops = matvis.coordinates.point_source_crd_eq /
    eci_to_enu_matrix); as in the reference, ``precision=1`` rounds the matrices (not the unit
    vectors) to float32 before the product."""
    ra_d, dec_d = _dev.f64(np.atleast_1d(ra)), _dev.f64(np.atleast_1d(dec))
    lsts_d = _dev.f64(np.atleast_1d(lsts))
    ns, nt = ra_d.numel(), lsts_d.numel()
    crd_eq = _dev.empty((3, ns), torch.float64)
    eq2tops = _dev.empty((nt, 3, 3), torch.float64)
    _dev.call("hydra_crd_eq", _dev.ptr(ra_d), _dev.ptr(dec_d), ns, _dev.ptr(crd_eq),
              _dev.stream_ptr())
    _dev.call("hydra_eq2tops", _dev.ptr(lsts_d), nt, float(latitude), _dev.ptr(eq2tops),
              _dev.stream_ptr())
    real_dtype = np.float32 if precision == 1 else np.float64
    crd = crd_eq.cpu().numpy()
    mats = eq2tops.cpu().numpy().astype(real_dtype)
    tops = np.array([np.dot(m, crd) for m in mats])          # (Nt, 3, Nsrc)
    return tops[:, 0], tops[:, 1], tops[:, 2]


def get_beam_interp_shape(beam):
    """Whether a beam object's ``interp`` return carries the (old pyuvdata) spectral-window axis
    (utils.py:542-562): False for analytic beams and beam interfaces, ``not
    beam.future_array_shapes`` for UVBeam-like objects, False for anything else."""
    if hasattr(beam, "compute_response") and not hasattr(beam, "future_array_shapes"):
        return False
    if hasattr(beam, "future_array_shapes"):
        return not beam.future_array_shapes
    return False


def build_hex_array(hex_spec=(3, 4), ants_per_row=None, d=14.6):
    """Antenna dict {id: (x, y, 0.)} of a hexagonally packed array (utils.py:374-419)."""
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


def get_flux_from_ptsrc_amp(ptsrc_amps, freqs, beta_ptsrc, curv_ptsrc=None, ref_freq=100.0):
    """Power-law SEDs with optional curvature, shape (Nsrc, Nfreqs) (utils.py:461-501)."""
    if isinstance(beta_ptsrc, float):
        beta_ptsrc = beta_ptsrc * np.ones_like(ptsrc_amps)
    assert ptsrc_amps.size == beta_ptsrc.size
    if curv_ptsrc is not None and isinstance(curv_ptsrc, float):
        curv_ptsrc = curv_ptsrc * np.ones_like(ptsrc_amps)
    if curv_ptsrc is None:
        curv_ptsrc = np.zeros_like(ptsrc_amps)
    assert ptsrc_amps.size == curv_ptsrc.size
    x = freqs / ref_freq
    spec_idx = beta_ptsrc[:, np.newaxis] + curv_ptsrc[:, np.newaxis] * np.log(x)[np.newaxis, :]
    return ptsrc_amps[:, np.newaxis] * x[np.newaxis, :] ** spec_idx


def partial_fourier_basis_2d(freqs, times, nfreq, ntime, Lfreq, Ltime, freq0=None, time0=None,
                             shape0=None):
    """Incomplete 2-D Fourier basis (Nmodes,Nf,Nt) and its wavenumbers (utils.py:625-669)."""
    time0 = times[0] if time0 is None else time0
    freq0 = freqs[0] if freq0 is None else freq0
    nf_norm, nt_norm = (freqs.size, times.size) if shape0 is None else shape0
    nfreq, ntime = np.atleast_1d(nfreq), np.atleast_1d(ntime)
    assert nfreq.ndim == 1 and ntime.ndim == 1 and freqs.ndim == 1 and times.ndim == 1
    t2d, f2d = np.meshgrid(times - time0, freqs - freq0)
    kfreq = 2.0 * np.pi * nfreq / Lfreq
    ktime = 2.0 * np.pi * ntime / Ltime
    phase = (kfreq[:, None, None] * f2d[None, :, :]) + (ktime[:, None, None] * t2d[None, :, :])
    return np.exp(1.0j * phase) / np.sqrt(nf_norm * nt_norm), kfreq, ktime


def partial_fourier_basis_2d_from_nmax(freqs, times, nmaxfreq, nmaxtime, Lfreq, Ltime,
                                       freq0=None, time0=None, shape0=None,
                                       positive_only=False):
    """Fourier basis for wavenumbers -nmax <= n < nmax (or 0 <= n < nmax) (utils.py:672-710)."""
    if positive_only:
        nf1, nt1 = np.arange(0, nmaxfreq), np.arange(0, nmaxtime)
    else:
        nf1, nt1 = np.arange(-nmaxfreq, nmaxfreq), np.arange(-nmaxtime, nmaxtime)
    nfreq, ntime = np.meshgrid(nf1, nt1)
    return partial_fourier_basis_2d(freqs, times, nfreq.flatten(), ntime.flatten(), Lfreq, Ltime,
                                    freq0, time0, shape0)


def status(myid, message, colour=None):
    """Coloured status line (utils.py:713-739)."""
    colours = {"r": "\033[91m", "g": "\033[92m", "y": "\033[93m", "b": "\033[94m",
               "m": "\033[95m", "c": "\033[96m", "bold": "\033[1m", "ul": "\033[4m"}
    tag = "" if myid is None else "[%d]" % myid
    if colour is not None:
        print("%s%s %s%s" % (colours[colour], tag, message, "\033[0m"))
    else:
        print("%s %s" % (tag, message))


def load_gain_model(gain_model_file, lst_pad=[0, 0], freq_pad=[0, 0], pad_value=1.0):
    """Complex gain model (Nants, Nfreqs, Ntimes) from a ``.npy`` file, surrounded by
    ``freq_pad`` / ``lst_pad`` = [before, after] samples of ``pad_value`` (utils.py:113-157)."""
    model = np.load(gain_model_file)
    nants, nf, nt = model.shape
    out = np.full((nants, nf + freq_pad[0] + freq_pad[1], nt + lst_pad[0] + lst_pad[1]),
                  pad_value, dtype=model.dtype)
    out[:, freq_pad[0]:freq_pad[0] + nf, lst_pad[0]:lst_pad[0] + nt] = model
    return out


def extend_coords_with_padding(arr, pad=[0, 0]):
    """An equally spaced coordinate array continued by ``pad`` = [before, after] extrapolated
    samples (utils.py:261-288)."""
    arr = np.asarray(arr)
    step = arr[1] - arr[0]
    head = arr[0] - step * (np.arange(pad[0]) + 1.0)[::-1]
    tail = arr[-1] + step * (1.0 + np.arange(pad[1]))
    return np.concatenate([head, arr, tail]).astype(arr.dtype)


def gain_prior_pspec_sqrt(lsts, freqs, gain_prior_amp, gain_prior_sigma_frate=None,
                          gain_prior_sigma_delay=None, gain_prior_zeropoint_std=None, frate0=0.0,
                          delay0=0.0):
    """Square root of the gain prior power spectrum on the (delay, fringe-rate) FFT grid,
    (Nfreqs, Ntimes): flat at ``gain_prior_amp``, optionally Gaussian-tapered in fringe rate
    (mHz, centre ``frate0``) and / or delay (ns, centre ``delay0``), the (0,0) mode optionally
    pinned to ``gain_prior_zeropoint_std`` (utils.py:565-622).  ``lsts`` in radians, ``freqs`` in MHz."""
    lsts, freqs = np.asarray(lsts), np.asarray(freqs)
    secs = 24.0 * 60.0 * 60.0 * lsts / (2.0 * np.pi)
    frate = 1e3 * np.fft.fftfreq(secs.size, d=secs[1] - secs[0])       # mHz
    delay = 1e3 * np.fft.fftfreq(freqs.size, d=freqs[1] - freqs[0])    # ns
    ps = gain_prior_amp * np.ones((freqs.size, secs.size))
    if gain_prior_sigma_frate is not None:
        ps *= np.exp(-0.5 * ((frate[np.newaxis, :] - frate0) / gain_prior_sigma_frate) ** 2.0)
    if gain_prior_sigma_delay is not None:
        ps *= np.exp(-0.5 * ((delay[:, np.newaxis] - delay0) / gain_prior_sigma_delay) ** 2.0)
    if gain_prior_zeropoint_std is not None:
        ps[delay == 0.0, frate == 0.0] = gain_prior_zeropoint_std
    return ps
