"""
Diffuse-region head on MI355X -- drop-in for the parts of ``hydra.region_sampler`` on the GCR
path (reference: hydra/region_sampler.py).

    calc_proj_operator(...)                 -> (Nbl, Nf, Nt, Nregions)    region_sampler.py:148-223
    segmented_diffuse_sky_model_pixels(...) -> list of pixel index arrays  region_sampler.py:85-145

The region amplitudes are then sampled by ``ptsrc_sampler.precompute_mpi`` on the (optionally
concatenated) operator exactly as in the driver (example.py:526-533,878-889).

The reference runs one ``simulate_vis`` per region; here the antenna voltages of ALL pixels are
generated once, ordered by region, and one kernel launch forms the per-region sums for every
(baseline, f, t) (``hydra_vis_regions``, csrc/gemv.hip).

``get_diffuse_sky_model_pixels`` (region_sampler.py:10-82) loads a pygdsm sky model and converts
coordinates with astropy/healpy; it is a data source outside the hot path and is not provided.
"""
import numpy as np
import torch

from . import _dev
from .vis_simulator import voltages_for


def get_diffuse_sky_model_pixels(freqs, nside=32, sky_model="gsm2016"):
    raise NotImplementedError(
        "get_diffuse_sky_model_pixels needs the pygdsm sky-model data (region_sampler.py:10-82), "
        "which hydra_amd does not ship: pass pixel ra, dec and (Npix, Nfreq) spectra in Jy "
        "to calc_proj_operator directly")


def segmented_diffuse_sky_model_pixels(ra, dec, sky_maps, freqs, nregions, smoothing_fwhm=None):
    """Lists of pixel indices, one per region, of roughly equal size in a crude spectral index
    taken from the first two channels (region_sampler.py:121-145).  Regions can be disconnected.
    ``smoothing_fwhm`` (a healpy map smoothing of the region labels) is not supported."""
    if smoothing_fwhm is not None:
        raise NotImplementedError("smoothing_fwhm needs healpy.smoothing (region_sampler.py:136-138)")
    sky_maps = np.asarray(sky_maps)
    freqs = np.asarray(freqs)
    beta = np.log(sky_maps[:, 0] / sky_maps[:, 1]) / np.log(freqs[0] / freqs[1])
    bounds = np.sort(beta)[:: beta.size // nregions]
    regions = np.zeros(beta.size, dtype=int)
    # closed intervals on both sides, later intervals win on ties, as in the reference
    for i in range(bounds.size - 1):
        regions[(beta >= bounds[i]) & (beta <= bounds[i + 1])] = i
    return [np.nonzero(regions == i)[0] for i in np.unique(regions)]


def calc_proj_operator(region_pixel_ra, region_pixel_dec, region_fluxes, region_idxs, ant_pos,
                       antpairs, freqs, times, beams, latitude=-0.5361913261514378):
    """Visibility of each sky region at unit amplitude, per (baseline, frequency, time): the
    projection operator from region amplitudes to visibilities, gains not included
    (region_sampler.py:148-223).  ``freqs`` in MHz, ``times`` = LSTs in rad; ``region_fluxes``
    (Npix, Nf) in Jy; ``region_idxs`` list of pixel-index arrays (a pixel may appear in several
    regions or in none).

    Returns (Nbl, Nf, Nt, Nregions) complex128 (numpy, or a GPU tensor if ``region_fluxes`` is
    one)."""
    nreg = len(region_idxs)
    assert nreg > 0, "region_idxs must hold at least one region"
    idx = [np.asarray(ix, dtype=np.int64).ravel() for ix in region_idxs]
    seg = np.zeros(nreg + 1, dtype=np.int32)
    seg[1:] = np.cumsum([ix.size for ix in idx])
    order = np.concatenate(idx)
    assert order.size > 0, "regions hold no pixels"
    freqs_h = np.asarray(freqs.cpu() if _dev.is_torch(freqs) else freqs, dtype=np.float64)
    if _dev.is_torch(region_fluxes):
        od = torch.as_tensor(order, device=region_fluxes.device)
        flux = region_fluxes[od]
    else:
        flux = np.asarray(region_fluxes)[order]
    ra = np.asarray(region_pixel_ra.cpu() if _dev.is_torch(region_pixel_ra) else region_pixel_ra)
    dec = np.asarray(region_pixel_dec.cpu() if _dev.is_torch(region_pixel_dec) else region_pixel_dec)
    V = voltages_for(ant_pos, flux, ra[order], dec[order], freqs_h * 1e6, times, beams, latitude)
    nf, nt, nant, ns = V.shape
    names = list(ant_pos.keys())
    bl_i = np.array([names.index(bl[0]) for bl in antpairs], dtype=np.int32)
    bl_j = np.array([names.index(bl[1]) for bl in antpairs], dtype=np.int32)
    nbl = bl_i.size
    # only vis[..., i, j >= i] is filled by the simulator (vis_simulator.py:281-289): a pair
    # listed the other way round picks up zeros, as in ptsrc_sampler.calc_proj_operator
    flipped = bl_i > bl_j
    out = _dev.empty((nbl, nf, nt, nreg), torch.complex128)
    bl_i_d, bl_j_d, seg_d = _dev.i32(bl_i), _dev.i32(bl_j), _dev.i32(seg)   # alive over the call
    _dev.call("hydra_vis_regions", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(bl_i_d),
              _dev.ptr(bl_j_d), nbl, _dev.host_ptr(seg), _dev.ptr(seg_d), nreg,
              _dev.ptr(out), _dev.stream_ptr())
    if flipped.any():
        out[torch.as_tensor(np.nonzero(flipped)[0], device=out.device)] = 0
    return _dev.like(region_fluxes, out)
