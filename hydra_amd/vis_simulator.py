"""
Visibility response simulator on MI355X -- drop-in for ``hydra.vis_simulator``
(reference: hydra/vis_simulator.py).  Same function names, argument meaning and
return shapes; the arithmetic runs in libhydra_hip.so (csrc/response.hip,
csrc/sh.hip).

The reference builds the per-source cube from complex antenna voltages
``v[a, s]`` (vis_simulator.py:261-276) and their outer product (:281-289).  The
voltages ``V (Nf, Nt, Nant, Ns)`` are the natural device-resident form: they are
(Nant-1)/2 times smaller than the baseline cube and every downstream product
(baseline gather, Gram, R x, R^H z, SH projection) is generated from them.
``antenna_voltages`` exposes that form; the reference-shaped functions are thin
wrappers over it.
"""
import warnings

import numpy as np
import torch

from . import _dev, beams as _beams

DEFAULT_LATITUDE = -30.7215 * np.pi / 180.0


def run_checks(antpos, freq, eq2tops, crd_eq, I_sky, beam_list, precision=2, polarized=False,
               beam_idx=None, freq_interp_kind="cubic"):
    """Input validation of the reference (vis_simulator.py:19-105); returns ``beam_idx``."""
    assert precision in {1, 2}
    nant, ncrd = antpos.shape
    assert ncrd == 3, "antpos must have shape (NANTS, 3)."
    ntimes, ncrd1, ncrd2 = eq2tops.shape
    assert ncrd1 == 3 and ncrd2 == 3, "eq2tops must have shape (NTIMES, 3, 3)."
    ncrd, nsrcs = crd_eq.shape
    assert ncrd == 3, "crd_eq must have shape (3, NSRCS)."
    assert I_sky.ndim == 1 and I_sky.shape[0] == nsrcs, "I_sky must have shape (NSRCS,)."
    nbeam = len(beam_list)
    if beam_idx is None:
        if nbeam == 1:
            beam_idx = np.zeros(nant, dtype=int)
        elif nbeam == nant:
            beam_idx = np.arange(nant, dtype=int)
        else:
            raise ValueError(
                "If number of beams provided is not 1 or nant, beam_idx must be provided.")
    else:
        beam_idx = np.asarray(beam_idx)
        assert beam_idx.shape == (nant,), "beam_idx must be length nant"
        assert all(0 <= i < nbeam for i in beam_idx), (
            "beam_idx contains indices greater than the number of beams")
    _beams.check_beams(beam_list, polarized)
    return beam_idx


def _np(x):
    return x.detach().cpu().numpy() if _dev.is_torch(x) else np.asarray(x)


def _require_precision(precision, polarized):
    if polarized:
        raise NotImplementedError(
            "polarized=True is outside the GCR hot path (the heads call with polarized=False, "
            "ptsrc_sampler.py:202)")
    if precision != 2:
        raise NotImplementedError(
            "only precision=2 (float64/complex128) is implemented: it is the only mode the GCR "
            "path uses (ptsrc_sampler.py:203)")


def _host_beam_values(uniq, freqs, eq2tops_d, crd_eq_d, nt, ns):
    """Evaluate arbitrary beam objects on the host at the az/za computed on the device
    (get_interp_beam_for_sim, vis_simulator.py:347-349): power values (Nbeam,Nf,Nt,Ns)."""
    az = _dev.empty((nt, ns), torch.float64)
    za = _dev.empty((nt, ns), torch.float64)
    up = _dev.empty((nt, ns), torch.uint8)
    _dev.call("hydra_az_za", _dev.ptr(eq2tops_d), nt, _dev.ptr(crd_eq_d), ns, _dev.ptr(az),
              _dev.ptr(za), _dev.ptr(up), _dev.stream_ptr())
    vals = _dev.empty((len(uniq), freqs.size, nt, ns), torch.float64)
    az_h = za_h = None
    for k, bm in enumerate(uniq):
        if hasattr(bm, "device_response"):      # gridded beams: spline evaluated on the GPU
            vals[k] = bm.device_response(az, za, freqs)
            continue
        if az_h is None:
            az_h, za_h = az.cpu().numpy(), za.cpu().numpy()
        host = np.empty((freqs.size, nt, ns), dtype=np.float64)
        for t in range(nt):
            resp = bm.compute_response(az_array=az_h[t], za_array=za_h[t],
                                       freq_array=np.atleast_1d(freqs))
            host[:, t, :] = np.asarray(resp)[0][0]
        vals[k] = _dev.f64(host)
    return vals


def antenna_voltages(antpos, freqs, eq2tops, crd_eq, fluxes, beam_list, beam_idx=None,
                     force_no_beam_sqrt=False):
    """Complex voltages V (Nf, Nt, Nant, Ns) as a GPU tensor (vis_simulator.py:214,233-276).

    ``freqs`` in Hz, ``fluxes`` (Ns, Nf), ``eq2tops`` (Nt,3,3), ``crd_eq`` (3,Ns).
    Raises ValueError if a beam evaluates to NaN/Inf (vis_simulator.py:256-257).
    """
    antpos_d = _dev.f64(antpos)
    freqs_h = np.atleast_1d(np.asarray(freqs.cpu() if _dev.is_torch(freqs) else freqs,
                                       dtype=np.float64))
    freqs_d = _dev.f64(freqs_h)
    eq2tops_d = _dev.f64(eq2tops)
    crd_eq_d = _dev.f64(crd_eq)
    fluxes_d = _dev.f64(fluxes)
    nant, nf, nt, ns = antpos_d.shape[0], freqs_d.numel(), eq2tops_d.shape[0], crd_eq_d.shape[1]
    assert tuple(fluxes_d.shape) == (ns, nf), "The `fluxes` array must have shape (NSRCS, NFREQS)."
    nb = len(beam_list)
    if beam_idx is None:
        if nb == 1:
            beam_idx = np.zeros(nant, dtype=int)
        elif nb == nant:
            beam_idx = np.arange(nant, dtype=int)
        else:
            raise ValueError(
                "If number of beams provided is not 1 or nant, beam_idx must be provided.")
    uniq, remap = _beams.group_beams(beam_list)
    idx = remap[np.asarray(beam_idx, dtype=np.int64)].astype(np.int32)
    specs = [b.hip_spec() if hasattr(b, "hip_spec") else None for b in uniq]
    kinds = {s[0] for s in specs if s is not None}
    vals_d = None
    if all(s is not None for s in specs) and len(kinds) == 1:
        kind = kinds.pop()
        params = np.asarray([s[1] for s in specs], dtype=np.float64)
    else:
        kind = _beams.HYDRA_BEAM_VALUES
        params = np.zeros((len(uniq), _beams.HYDRA_BEAM_NPARAMS))
        vals_d = _dev.f64(_host_beam_values(uniq, freqs_h, eq2tops_d, crd_eq_d, nt, ns))
    params_d = _dev.f64(params)
    idx_d = _dev.i32(idx)
    V = _dev.empty((nf, nt, nant, ns), torch.complex128)
    bad = _dev.zeros((1,), torch.int32)
    _dev.call("hydra_voltages", _dev.ptr(antpos_d), nant, _dev.ptr(freqs_d), nf,
              _dev.ptr(eq2tops_d), nt, _dev.ptr(crd_eq_d), ns, _dev.ptr(fluxes_d), int(kind),
              _dev.ptr(params_d), len(uniq), _dev.ptr(idx_d), _dev.ptr(vals_d),
              int(bool(force_no_beam_sqrt)), _dev.ptr(V), _dev.ptr(bad), _dev.stream_ptr())
    if int(bad.item()) != 0:
        raise ValueError("Beam interpolation resulted in an invalid value")
    # one beam model for every antenna: |V[f,t,a,s]| does not depend on a (used by the
    # closed-form Gram, csrc/gram.hip)
    V.hydra_same_beam = bool(np.all(idx == idx[0]))
    return V


def _geometry(ra, dec, lsts, latitude):
    ra_d, dec_d, lsts_d = _dev.f64(ra), _dev.f64(dec), _dev.f64(np.atleast_1d(lsts))
    ns, nt = ra_d.numel(), lsts_d.numel()
    crd_eq = _dev.empty((3, ns), torch.float64)
    eq2tops = _dev.empty((nt, 3, 3), torch.float64)
    _dev.call("hydra_crd_eq", _dev.ptr(ra_d), _dev.ptr(dec_d), ns, _dev.ptr(crd_eq),
              _dev.stream_ptr())
    _dev.call("hydra_eq2tops", _dev.ptr(lsts_d), nt, float(latitude), _dev.ptr(eq2tops),
              _dev.stream_ptr())
    return crd_eq, eq2tops


def _cube_from_voltages(V, subarr_ant=None):
    nf, nt, nant, ns = V.shape
    if subarr_ant is None:
        out = _dev.empty((nf, nt, nant, nant, ns), torch.complex128)
        _dev.call("hydra_vis_cube", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(out),
                  _dev.stream_ptr())
    else:
        out = _dev.empty((nf, nt, nant, ns), torch.complex128)
        _dev.call("hydra_vis_subarr", _dev.ptr(V), nf, nt, nant, ns, int(subarr_ant),
                  _dev.ptr(out), _dev.stream_ptr())
    return out


def vis_sim_per_source(antpos, freq, eq2tops, crd_eq, I_sky, beam_list, precision=2,
                       polarized=False, beam_idx=None, subarr_ant=None,
                       force_no_beam_sqrt=False):
    """One frequency channel (vis_simulator.py:108-300): returns (NTIMES, NANTS, NANTS, NSRCS)
    complex128, upper triangle incl. autos populated, lower triangle zero; or
    (NTIMES, NANTS, NSRCS) with ``subarr_ant``."""
    _require_precision(precision, polarized)
    beam_idx = run_checks(_np(antpos), freq, _np(eq2tops), _np(crd_eq), _np(I_sky), beam_list,
                          precision, polarized, beam_idx)
    fl = _dev.f64(I_sky).reshape(-1, 1)
    V = antenna_voltages(antpos, np.array([float(freq)]), eq2tops, crd_eq, fl, beam_list,
                         beam_idx, force_no_beam_sqrt)
    return _dev.like(I_sky, _cube_from_voltages(V, subarr_ant)[0])


def get_interp_beam_for_sim(bm, freq, az, za, polarized=True, force_no_beam_sqrt=False,
                            freq_interp_kind="cubic"):
    """Host-side evaluation of one beam object (vis_simulator.py:302-360), kept for callers
    that use it directly; the simulator itself evaluates analytic beams in-kernel."""
    if polarized:
        raise NotImplementedError("polarized beams are outside the GCR hot path")
    resp = np.asarray(bm.compute_response(az_array=az, za_array=za,
                                          freq_array=np.atleast_1d(freq)))[0]
    if force_no_beam_sqrt:
        return resp[0, 0, :]
    return np.sqrt(resp[0, 0, :])


def _size(x):
    return int(x.numel()) if _dev.is_torch(x) else int(np.size(x))


def _wrapper_checks(ants, fluxes, ra, dec, freqs, beams):
    # vis_simulator.py:426-445
    assert len(ants) == len(beams), (
        "The `beams` list must have as many entries as the ``ants`` dict.")
    assert tuple(fluxes.shape) == (_size(ra), _size(freqs)), (
        "The `fluxes` array must have shape (NSRCS, NFREQS).")
    ra_h = np.asarray(ra.cpu() if _dev.is_torch(ra) else ra)
    dec_h = np.asarray(dec.cpu() if _dev.is_torch(dec) else dec)
    if np.any(ra_h < 0.0) or np.any(ra_h > 2.0 * np.pi):
        warnings.warn("One or more ra values is outside the allowed range (0, 2 pi)",
                      RuntimeWarning)
    if np.any(dec_h < -np.pi) or np.any(dec_h > np.pi):
        warnings.warn("One or more dec values is outside the allowed range (-pi, +pi)",
                      RuntimeWarning)


def voltages_for(ants, fluxes, ra, dec, freqs, lsts, beams, latitude=DEFAULT_LATITUDE,
                 force_no_beam_sqrt=False):
    """Voltages V (Nf,Nt,Nant,Ns) for the wrapper-level arguments of
    ``simulate_vis_per_source`` (``freqs`` in Hz).  GPU tensor."""
    _wrapper_checks(ants, fluxes, ra, dec, freqs, beams)
    _beams.check_beams(beams)
    antpos = np.array([ants[k] for k in ants.keys()], dtype=np.float64)
    crd_eq, eq2tops = _geometry(ra, dec, lsts, latitude)
    return antenna_voltages(antpos, freqs, eq2tops, crd_eq, fluxes, list(beams), None,
                            force_no_beam_sqrt)


def simulate_vis_per_source(ants, fluxes, ra, dec, freqs, lsts, beams, polarized=False,
                            precision=2, latitude=DEFAULT_LATITUDE, use_feed="x",
                            subarr_ant=None, force_no_beam_sqrt=False):
    """Per-source visibilities (vis_simulator.py:363-515): (NFREQS, NTIMES, NANTS, NANTS,
    NSRCS) complex128 (or (NFREQS, NTIMES, NANTS, NSRCS) with ``subarr_ant``).  ``freqs`` in
    Hz.  Returns numpy unless ``fluxes`` is a torch tensor."""
    _require_precision(precision, polarized)
    V = voltages_for(ants, fluxes, ra, dec, freqs, lsts, beams, latitude, force_no_beam_sqrt)
    return _dev.like(fluxes, _cube_from_voltages(V, subarr_ant))


def simulate_vis(*args, **kwargs):
    """Sum over sources of ``simulate_vis_per_source`` (vis_simulator.py:518-568), computed
    by a wave-level reduction over sources without materialising the per-source cube:
    (NFREQS, NTIMES, NANTS, NANTS) complex128."""
    names = ["ants", "fluxes", "ra", "dec", "freqs", "lsts", "beams", "polarized", "precision",
             "latitude", "use_feed", "subarr_ant", "force_no_beam_sqrt"]
    kw = dict(zip(names, args))
    kw.update(kwargs)
    kw.pop("multiprocess", None)
    _require_precision(kw.get("precision", 2), kw.get("polarized", False))
    if kw.get("subarr_ant") is not None:
        per = simulate_vis_per_source(**kw)
        return per.sum(axis=-1) if not _dev.is_torch(per) else per.sum(dim=-1)
    V = voltages_for(kw["ants"], kw["fluxes"], kw["ra"], kw["dec"], kw["freqs"], kw["lsts"],
                     kw["beams"], kw.get("latitude", DEFAULT_LATITUDE),
                     kw.get("force_no_beam_sqrt", False))
    nf, nt, nant, ns = V.shape
    out = _dev.empty((nf, nt, nant, nant), torch.complex128)
    _dev.call("hydra_vis_sum", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(out), _dev.stream_ptr())
    return _dev.like(kw["fluxes"], out)


def healpix_pixel_coords(nside):
    """(theta, phi) of the RING pixel centres as GPU tensors (hp.pix2ang, :659)."""
    npix = 12 * int(nside) ** 2
    theta = _dev.empty((npix,), torch.float64)
    phi = _dev.empty((npix,), torch.float64)
    _dev.call("hydra_healpix_pix2ang", int(nside), _dev.ptr(theta), _dev.ptr(phi),
              _dev.stream_ptr())
    return theta, phi


def getlm(lmax):
    """(ell, m) in healpy order -- ``hp.Alm().getlm(lmax)`` (vis_simulator.py:654)."""
    m = np.concatenate([np.full(lmax + 1 - k, k) for k in range(lmax + 1)])
    ell = np.concatenate([np.arange(k, lmax + 1) for k in range(lmax + 1)])
    return ell.astype(np.int64), m.astype(np.int64)


def sh_mode_maps(lmax, nside, amplitude=1.0, packed=False):
    """Ymap (Npix, Ncols) GPU tensor: alm2map of each unit mode times pixel area
    (vis_simulator.py:702-718)."""
    npix = 12 * int(nside) ** 2
    nalm = (lmax + 1) * (lmax + 2) // 2
    ncols = 2 * nalm - (lmax + 1) if packed else 2 * nalm
    ymap = _dev.empty((npix, ncols), torch.float64)
    _dev.call("hydra_sh_mode_maps", int(lmax), int(nside), float(amplitude), int(bool(packed)),
              _dev.ptr(ymap), _dev.stream_ptr())
    return ymap


def sh_voltages(nside, ants, freqs, lsts, beams, latitude=DEFAULT_LATITUDE, flux_f=None):
    """Voltages of the HEALPix pixels treated as unit-flux sources, with the reference's
    coordinate convention dec = theta - pi/2, ra = phi (vis_simulator.py:657-664).  The tensor
    carries ``hydra_pixidx`` / ``hydra_npix_t``: per time the indices of the pixels above the
    horizon (the others have exactly zero voltage, vis_simulator.py:237,272)."""
    theta, phi = healpix_pixel_coords(nside)
    dec = theta - 0.5 * np.pi
    freqs = np.atleast_1d(np.asarray(freqs, dtype=np.float64))
    fluxes = torch.ones((theta.numel(), freqs.size), dtype=torch.float64, device=theta.device)
    if flux_f is not None:      # a per-channel flux of every pixel: scales conj(V_i) V_j by flux_f / 1
        fluxes = fluxes * _dev.f64(flux_f)[None, :]
    _beams.check_beams(beams)
    assert len(ants) == len(beams), (
        "The `beams` list must have as many entries as the ``ants`` dict.")
    antpos = np.array([ants[k] for k in ants.keys()], dtype=np.float64)
    crd_eq, eq2tops = _geometry(phi, dec, lsts, latitude)
    V = antenna_voltages(antpos, freqs, eq2tops, crd_eq, fluxes, list(beams))
    V.hydra_pixidx, V.hydra_npix_t = _horizon_lists(eq2tops, crd_eq)
    return V


def _horizon_lists(eq2tops, crd_eq):
    """(pixidx (Nt, npix_max) int32, npix_t (Nt) int32): sources with tz > 0 at each time, in
    ascending order; rows are padded with their last valid entry."""
    nt, ns = eq2tops.shape[0], crd_eq.shape[1]
    az = _dev.empty((nt, ns), torch.float64)
    za = _dev.empty((nt, ns), torch.float64)
    up = _dev.empty((nt, ns), torch.uint8)
    _dev.call("hydra_az_za", _dev.ptr(eq2tops), nt, _dev.ptr(crd_eq), ns, _dev.ptr(az),
              _dev.ptr(za), _dev.ptr(up), _dev.stream_ptr())
    counts = up.sum(dim=1).to(torch.int32)
    npix_max = max(int(counts.max().item()), 1)
    # stable sort puts the flagged sources first, in ascending index order
    order = torch.argsort(up.to(torch.int16), dim=1, descending=True, stable=True)[:, :npix_max]
    pos = torch.arange(npix_max, device=up.device)[None, :]
    last = torch.clamp(counts.long() - 1, min=0)[:, None]
    pixidx = torch.gather(order, 1, torch.minimum(pos.expand(nt, -1), last))
    return pixidx.to(torch.int32).contiguous(), counts.contiguous()


_SH_TILE_I, _SH_TILE_J = 8, 16


def _sh_tiles(bl_i, bl_j, nant):
    """Antenna-block tiles for ``hydra_sh_project_tiled``: (tiles (ntile,2), bmap (ntile,128))
    or None if a pair occurs twice (then the untiled kernel is used)."""
    bl_i = np.asarray(bl_i, dtype=np.int64)
    bl_j = np.asarray(bl_j, dtype=np.int64)
    key = bl_i * nant + bl_j
    if np.unique(key).size != key.size:
        return None
    ti, tj = bl_i // _SH_TILE_I, bl_j // _SH_TILE_J
    nti = (nant + _SH_TILE_I - 1) // _SH_TILE_I
    tkey = ti * ((nant + _SH_TILE_J - 1) // _SH_TILE_J) + tj
    uniq, inv = np.unique(tkey, return_inverse=True)
    ntj = (nant + _SH_TILE_J - 1) // _SH_TILE_J
    tiles = np.stack([(uniq // ntj) * _SH_TILE_I, (uniq % ntj) * _SH_TILE_J], axis=1).astype(np.int32)
    bmap = np.full((uniq.size, _SH_TILE_I * _SH_TILE_J), -1, dtype=np.int32)
    slot = (bl_i - tiles[inv, 0]) * _SH_TILE_J + (bl_j - tiles[inv, 1])
    bmap[inv, slot] = np.arange(bl_i.size, dtype=np.int32)
    assert nti > 0
    return np.ascontiguousarray(tiles), np.ascontiguousarray(bmap)


def sh_project(V, bl_i, bl_j, ymap, scale_f=None, out=None, f0=0, nf_total=None):
    """out (Nbl, Nf, Nt, Ncols) = sum_p conj(V_i,p) V_j,p Ymap[p, :] on the matrix cores.

    With the horizon lists attached by ``sh_voltages`` the tiled kernel is used (antenna-block
    tiles, only pixels above the horizon); ``out`` / ``f0`` / ``nf_total`` let a caller fill a
    long frequency axis block by block."""
    nf, nt, nant, npix = V.shape
    bi, bj = _dev.i32(bl_i), _dev.i32(bl_j)
    nbl, nm = bi.numel(), ymap.shape[1]
    nf_total = nf if nf_total is None else int(nf_total)
    if out is None:
        out = _dev.empty((nbl, nf_total, nt, nm), torch.complex128)
    assert tuple(out.shape) == (nbl, nf_total, nt, nm) and out.is_contiguous()
    sc = None if scale_f is None else _dev.f64(scale_f)
    tiles = _sh_tiles(np.asarray(bl_i), np.asarray(bl_j), nant) \
        if getattr(V, "hydra_pixidx", None) is not None else None
    if tiles is not None:
        nmp = (nm + 127) // 128 * 128
        ymap_p = getattr(ymap, "hydra_padded", None)
        if ymap_p is None or ymap_p.shape[1] != nmp:
            ymap_p = torch.zeros((npix, nmp), dtype=torch.float64, device=ymap.device)
            ymap_p[:, :nm] = ymap
            ymap.hydra_padded = ymap_p
        tiles_d, bmap_d = _dev.i32(tiles[0]), _dev.i32(tiles[1])
        pixidx, npix_t = V.hydra_pixidx, V.hydra_npix_t
        step = max(1, 65535 // nt)
        for a in range(0, nf, step):                  # grid.z limit: whole channels per launch
            b = min(nf, a + step)
            Vb = V[a:b]
            _dev.call("hydra_sh_project_tiled", _dev.ptr(Vb), b - a, nt, nant, npix,
                      _dev.ptr(pixidx), _dev.ptr(npix_t), int(pixidx.shape[1]), _dev.ptr(tiles_d),
                      _dev.ptr(bmap_d), int(tiles[0].shape[0]), _dev.ptr(ymap_p), nmp, nm,
                      _dev.ptr(None if sc is None else sc[a:b].contiguous()), f0 + a, nf_total,
                      _dev.ptr(out), _dev.stream_ptr())
        return out
    if nf * nt > 65535 or f0 != 0 or nf_total != nf:
        raise NotImplementedError("the untiled projection handles one block of <= 65535 (f,t) samples")
    _dev.call("hydra_sh_project", _dev.ptr(V), nf, nt, nant, npix, _dev.ptr(bi), _dev.ptr(bj), nbl,
              _dev.ptr(ymap), nm, _dev.ptr(sc), _dev.ptr(out), _dev.stream_ptr())
    return out


def simulate_vis_per_alm(lmax, nside, ants, freqs, lsts, beams, polarized=False, precision=2,
                         latitude=DEFAULT_LATITUDE, use_feed="x", multiprocess=True,
                         amplitude=1.0, logfile=None):
    """Visibility response per spherical-harmonic mode (vis_simulator.py:571-736):
    returns (ell, m, vis) with vis (NFREQS, NTIMES, NANTS, NANTS, 2*nalm) complex128 numpy,
    first nalm columns the real-part modes, last nalm the imaginary-part modes."""
    _require_precision(precision, polarized)
    freqs = np.atleast_1d(freqs)
    lsts = np.atleast_1d(lsts)
    ell, m = getlm(lmax)
    V = sh_voltages(nside, ants, freqs, lsts, beams, latitude)
    nant = V.shape[2]
    ii, jj = np.meshgrid(np.arange(nant), np.arange(nant), indexing="ij")
    keep = jj >= ii
    bl_i, bl_j = ii[keep].astype(np.int32), jj[keep].astype(np.int32)
    ymap = sh_mode_maps(lmax, nside, amplitude, packed=False)
    tri = sh_project(V, bl_i, bl_j, ymap)               # (Npair, Nf, Nt, 2 nalm)
    nf, nt = V.shape[0], V.shape[1]
    vis = torch.zeros((nf, nt, nant, nant, 2 * ell.size), dtype=torch.complex128,
                      device=V.device)
    vis[:, :, torch.as_tensor(bl_i, device=V.device).long(),
        torch.as_tensor(bl_j, device=V.device).long(), :] = tri.permute(1, 2, 0, 3)
    return ell, m, vis.cpu().numpy()


def simulate_vis_per_region(lmax, nside, ants, freqs, lsts, beams, polarized=False, precision=2,
                            latitude=DEFAULT_LATITUDE, use_feed="x", multiprocess=True,
                            amplitude=1.0, logfile=None):
    """The reference keeps a second copy of ``simulate_vis_per_alm`` under this name
    (vis_simulator.py:739-906; it only adds ``multiprocess`` to the inner call): same result."""
    return simulate_vis_per_alm(lmax, nside, ants, freqs, lsts, beams, polarized=polarized,
                                precision=precision, latitude=latitude, use_feed=use_feed,
                                multiprocess=multiprocess, amplitude=amplitude, logfile=logfile)
