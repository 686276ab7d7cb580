"""
Random catalogue and example simulation used by the driver and the tests -- drop-in for
``hydra.example`` (reference: hydra/example.py).
"""
import numpy as np

from .beams import GaussianBeam, PolyBeam
from .utils import build_hex_array, extract_vis_from_sim, get_flux_from_ptsrc_amp
from .vis_simulator import simulate_vis


def generate_random_ptsrc_catalogue(Nptsrc, ra_bounds, dec_bounds, logflux_bounds=(-1.0, 2.0)):
    """Random point sources, uniform on the sphere within the bounds, log-uniform in flux.
    Draws from the global numpy RNG in the reference's order: RA, U (for Dec), log-flux
    (hydra/example.py:21-72).  Returns (ra, dec, ptsrc_amps)."""
    ra_low, ra_high = min(ra_bounds), max(ra_bounds)
    dec_low, dec_high = min(dec_bounds), max(dec_bounds)
    lf_low, lf_high = min(logflux_bounds), max(logflux_bounds)
    ra = np.random.uniform(low=ra_low, high=ra_high, size=Nptsrc)
    U = np.random.uniform(low=0, high=1, size=Nptsrc)
    dsin = np.sin(dec_high) - np.sin(dec_low)
    dec = np.arcsin(U * dsin + np.sin(dec_low))
    ptsrc_amps = 10.0 ** np.random.uniform(low=lf_low, high=lf_high, size=Nptsrc)
    return ra, dec, ptsrc_amps


#: Chebyshev coefficients of the PolyBeam fitted to the HERA Fagnoni beam (hydra/example.py:145-164)
HERA_POLYBEAM_COEFFS = [
    0.29778665, -0.44821433, 0.27338272, -0.10030698, -0.01195859, 0.06063853, -0.04593295,
    0.0107879, 0.01390283, -0.01881641, -0.00177106, 0.01265177, -0.00568299, -0.00333975,
    0.00452368, 0.00151808, -0.00593812, 0.00351559,
]


def example_beams(beam_type, nants):
    """One beam object per antenna as the reference's example builds them
    (hydra/example.py:142-172): the HERA PolyBeam for a ``beam_type`` containing 'polybeam',
    else a 14 m Gaussian."""
    if "polybeam" in beam_type.lower():
        return [PolyBeam(HERA_POLYBEAM_COEFFS, spectral_index=-0.6975, ref_freq=1.0e8)
                for _ in range(nants)]
    return [GaussianBeam(diameter=14.0) for _ in range(nants)]


def run_example_simulation(times, freqs, output_dir, ra, dec, ptsrc_amps, array_latitude,
                           beam_type="gaussian", hex_array=(3, 4)):
    """Example visibility simulation on a hexagonal array (hydra/example.py:75-198).
    ``freqs`` in MHz, ``times`` LSTs in rad.  Returns (model0 (Nbl,Nf,Nt), fluxes, beams,
    (ants, ant_pos, antpairs, ants1, ants2))."""
    hex_array = tuple(hex_array)
    assert len(hex_array) == 2, "hex-array argument must have length 2."
    ant_pos = build_hex_array(hex_spec=hex_array, d=14.6)
    ants = np.array(list(ant_pos.keys()))
    antpairs = [(i, j) for i in range(len(ants)) for j in range(i, len(ants)) if i != j]
    ants1, ants2 = list(zip(*antpairs))
    fluxes = get_flux_from_ptsrc_amp(ptsrc_amps, freqs, -2.7)
    beams = example_beams(beam_type, len(ants))
    sim = simulate_vis(ants=ant_pos, fluxes=fluxes, ra=ra, dec=dec, freqs=freqs * 1e6, lsts=times,
                       beams=beams, polarized=False, precision=2, latitude=array_latitude,
                       use_feed="x")
    model0 = extract_vis_from_sim(ants, antpairs, sim)
    return model0, fluxes, beams, (ants, ant_pos, antpairs, ants1, ants2)
