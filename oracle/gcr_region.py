"""
Oracle (test infrastructure): diffuse-region head, restating the numpy parts of
``hydra/region_sampler.py``.

  segment_pixels      region_sampler.py:85-145 without the optional healpy smoothing
                      (PINNED by tests/golden/region_segments.npz from the reference function)
  calc_proj_operator  region_sampler.py:148-223: one ``simulate_vis`` per region (sum over the
                      region's pixels), gathered per baseline.  Runs on the oracle's own
                      simulator, so it inherits that simulator's status ("parity unpinned" at
                      the geometry/beam boundary, see oracle/response.py).
The sky-model loader (:10-82) needs pygdsm/astropy/healpy data and is not restated.
"""
import numpy as np

from . import response


def segment_pixels(ra, dec, sky_maps, freqs, nregions):
    """Pixel index lists of regions of roughly equal size in a crude two-channel spectral index
    (region_sampler.py:121-145, smoothing_fwhm=None)."""
    beta = np.log(sky_maps[:, 0] / sky_maps[:, 1]) / np.log(freqs[0] / freqs[1])
    edges = np.sort(beta)[:: beta.size // nregions]
    label = np.zeros(beta.size, dtype=int)
    for i in range(edges.size - 1):
        label[np.where(np.logical_and(beta >= edges[i], beta <= edges[i + 1]))] = i
    return [np.where(label == i)[0] for i in np.sort(np.unique(label))]


def calc_proj_operator(region_pixel_ra, region_pixel_dec, region_fluxes, region_idxs, ant_pos,
                       antpairs, freqs, times, beams, latitude=-0.5361913261514378):
    """(Nbl, Nf, Nt, Nregions) complex128: visibility of each region at unit amplitude
    (region_sampler.py:192-223).  ``freqs`` in MHz."""
    names = list(ant_pos.keys())
    out = np.zeros((len(antpairs), freqs.size, times.size, len(region_idxs)), dtype=np.complex128)
    for j, idx in enumerate(region_idxs):
        vis = response.simulate_vis(
            ants=ant_pos, fluxes=region_fluxes[idx, :], ra=region_pixel_ra[idx],
            dec=region_pixel_dec[idx], freqs=freqs * 1e6, lsts=times, beams=beams,
            latitude=latitude)
        for i, (a1, a2) in enumerate(antpairs):
            out[i, :, :, j] = vis[:, :, names.index(a1), names.index(a2)]
    return out
