"""
Command-line flags of the Gibbs driver -- drop-in for ``hydra.config.get_config``
(reference: hydra/config.py:5-589): the same flag names, destinations, types, ``nargs`` and
defaults, so that a command line written for the reference's ``example.py`` parses to the same
``argparse.Namespace`` here (checked flag by flag against the reference's parser in
tests/test_product_helpers.py).  The flags are kept as a table; the help texts are this
package's own.
"""
import argparse

import numpy as np

# (flag, dest, kind, default, nargs, help); kind "flag" = store_true
_FLAGS = [
    # what to sample
    ("--seed", "seed", int, 0, None, "random seed (each worker adds its rank)"),
    ("--gains", "sample_gains", "flag", False, None, "sample the linearised antenna gains"),
    ("--cosmo", "sample_cosmo_field", "flag", False, None, "sample the cosmological field"),
    ("--ptsrc", "sample_ptsrc", "flag", False, None, "sample point-source amplitudes"),
    ("--regions", "sample_regions", "flag", False, None, "sample diffuse-region amplitudes"),
    ("--beam", "sample_beam", "flag", False, None, "sample the primary beam"),
    ("--sh", "sample_sh", "flag", False, None, "sample spherical-harmonic modes"),
    ("--cl", "sample_sh_pspec", "flag", False, None, "sample the spherical-harmonic power spectrum"),
    ("--pspec", "sample_pspec", "flag", False, None, "sample the 21-cm power spectrum"),
    # output control
    ("--debug", "debug", "flag", False, None, "verbose progress output"),
    ("--stats", "calculate_stats", "flag", False, None, "statistics of the samples"),
    ("--diagnostics", "output_diagnostics", "flag", False, None, "write diagnostic arrays"),
    ("--timing", "save_timing_info", "flag", False, None, "append step timings to timing.dat"),
    ("--plotting", "plotting", "flag", False, None, "make plots"),
    # array and data
    ("--latitude", "latitude", float, -30.7215, None, "array latitude, degrees"),
    ("--hex-array", "hex_array", int, (3, 4), "+", "antennas in the shortest and longest hex row"),
    ("--Nptsrc", "Nptsrc", int, 100, None, "number of point sources"),
    ("--Ntimes", "Ntimes", int, 30, None, "number of LSTs"),
    ("--Nfreqs", "Nfreqs", int, 60, None, "number of frequency channels"),
    ("--Niters", "Niters", int, 100, None, "Gibbs iterations"),
    ("--sigma-noise", "sigma_noise", float, 0.05, None, "noise rms per visibility"),
    ("--solver", "solver_name", str, "cg", None, "cg | gmres | bicgstab | mpicg"),
    ("--output-dir", "output_dir", str, "./output", None, "directory for the output files"),
    ("--ra-bounds", "ra_bounds", float, (0, 1), 2, "RA range of the catalogue, rad"),
    ("--dec-bounds", "dec_bounds", float, (-0.6, 0.4), 2, "Dec range of the catalogue, rad"),
    ("--lst-bounds", "lst_bounds", float, (0.2, 0.5), 2, "LST range, rad"),
    ("--freq-bounds", "freq_bounds", float, (100.0, 120.0), 2, "frequency range, MHz"),
    # point sources
    ("--ptsrc-amp-prior-level", "ptsrc_amp_prior_level", float, 0.1, None,
     "fractional prior width of the source amplitudes"),
    ("--calsrc-std", "calsrc_std", float, -1.0, None,
     "prior width of the calibration source (negative: none)"),
    ("--calsrc-radius", "calsrc_radius", float, 10.0, None,
     "search radius for the calibration source around the zenith, degrees"),
    # simulated diffuse sky
    ("--sim-diffuse-sky-model", "sim_diffuse_sky_model", str, "none", None,
     "diffuse model added to the simulated data"),
    ("--sim-diffuse-nside", "sim_diffuse_nside", int, 16, None, "HEALPix nside of that model"),
    # cosmological field
    ("--cosmo-ra-bounds", "cosmo_field_ra_bounds", float, (0, 60), 2, "field RA range, degrees"),
    ("--cosmo-dec-bounds", "cosmo_field_dec_bounds", float, (-40.0, -20.0), 2,
     "field Dec range, degrees"),
    ("--cosmo-ra-ngrid", "cosmo_field_ra_ngrid", int, 10, None, "field grid points in RA"),
    ("--cosmo-dec-ngrid", "cosmo_field_dec_ngrid", int, 10, None, "field grid points in Dec"),
    # gains
    ("--gain-prior-amp", "gain_prior_amp", float, 0.1, None, "prior amplitude of the gain modes"),
    ("--gain-nmax-freq", "gain_nmaxfreq", int, 2, None, "highest gain Fourier mode in frequency"),
    ("--gain-nmax-time", "gain_nmaxtime", int, 2, None, "highest gain Fourier mode in time"),
    ("--gain-prior-zero-mode-std", "gain_prior_zero_mode_std", float, None, None,
     "prior width of the gain zero mode"),
    ("--gain-only-positive-modes", "gain_only_positive_modes", bool, True, None,
     "use only non-negative wavenumbers"),
    ("--sim-gain-amp-std", "sim_gain_amp_std", float, 0.05, None,
     "rms of the simulated gain perturbations"),
    # beam
    ("--beam-sim-type", "beam_sim_type", str, "gaussian", None, "gaussian | polybeam"),
    ("--beam-prior-std", "beam_prior_std", float, 1, None, "prior width of the beam coefficients"),
    ("--beam-nmax", "beam_nmax", int, 16, None, "radial Fourier-Bessel modes"),
    ("--beam-mmax", "beam_mmax", int, 0, None, "azimuthal Fourier-Bessel modes"),
    ("--rho-const", "rho_const", float, np.sqrt(1 - np.cos(np.pi * 23 / 45)), None,
     "radial scale of the Fourier-Bessel basis"),
    # spherical harmonics
    ("--sim-sh-lmax", "sim_sh_lmax", int, 8, None, "lmax of the simulated diffuse sky"),
    ("--sim-sh-nside", "sim_sh_nside", int, 16, None, "nside of the simulated diffuse sky"),
    ("--sh-lmax", "sh_lmax", int, 8, None, "lmax of the sampled modes"),
    ("--sh-nside", "sh_nside", int, 16, None, "nside of the response pixelisation"),
    ("--sh-prior-std", "sh_prior_std", float, 0.1, None, "prior width of the modes"),
    ("--sh-ref-freq", "sh_ref_freq", float, 100.0, None, "reference frequency, MHz"),
    ("--sh-spectral-idx", "sh_spectral_idx", float, 0.0, None, "spectral index of the modes"),
    # regions
    ("--region-nregions", "region_nregions", int, 10, None, "number of sky regions"),
    ("--region-smoothing-fwhm", "region_smoothing_fwhm", float, None, None,
     "smoothing of the segmented map, degrees"),
    ("--region-sky-model", "region_sky_model", str, "gsm2016", None, "sky model to segment"),
    ("--region-nside", "region_nside", int, 16, None, "nside of the segmented map"),
    ("--region-amp-prior-level", "region_amp_prior_level", float, 0.1, None,
     "fractional prior width of the region amplitudes"),
]


def build_parser():
    parser = argparse.ArgumentParser(
        description="Gibbs sampler for 21-cm arrays: the GCR heads on MI355X (flags of "
                    "hydra/config.py)")
    for flag, dest, kind, default, nargs, text in _FLAGS:
        if kind == "flag":
            parser.add_argument(flag, action="store_true", required=False, dest=dest, help=text)
        else:
            kw = dict(type=kind, action="store", default=default, required=False, dest=dest,
                      help=text)
            if nargs is not None:
                kw["nargs"] = nargs
            parser.add_argument(flag, **kw)
    return parser


def get_config(argv=None):
    """Parse the command line (``argv=None``: ``sys.argv``) into an ``argparse.Namespace``."""
    return build_parser().parse_args(argv)
