"""
Oracle (test infrastructure): analytic primary beams.

The reference evaluates beams through ``pyuvdata.BeamInterface.compute_response``
(``hydra/vis_simulator.py:347-359``) on objects such as
``pyuvdata.GaussianBeam(diameter=14.)`` (``hydra/example.py:170-172``).  pyuvdata
(>=3.2.3, ``requirements.txt:9``) is not installed: PARITY UNPINNED, the
Gaussian formula below restates pyuvdata's published definition.

A beam here is any object with

    compute_response(az_array=, za_array=, freq_array=, **kw)
        -> ndarray (1, 1, Nfreq, Npts) of *power*-beam values

so that ``resp[0][0, 0, :]`` is what ``get_interp_beam_for_sim`` indexes
(``vis_simulator.py:347-359``).
"""
import numpy as np
from .geometry import C_LIGHT


class GaussianBeam:
    """Achromatic-shape Gaussian beam of an aperture of ``diameter`` metres.

    E-field: exp(-za^2 / (2 sigma^2)); power: exp(-za^2 / sigma^2), with
    sigma(nu) = arcsin(2.2150894 * lambda / (pi * D)) * 2 / 2.355
    (Gaussian fitted to an Airy disc).  Alternatively a fixed ``sigma`` (rad) at
    ``reference_frequency`` scaled as (nu/nu_ref)^spectral_index.
    """

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, diameter=None, sigma=None, spectral_index=0.0,
                 reference_frequency=None):
        if (diameter is None) == (sigma is None):
            raise ValueError("give exactly one of diameter, sigma")
        self.diameter = diameter
        self.sigma = sigma
        self.spectral_index = spectral_index
        self.reference_frequency = reference_frequency

    def get_sigmas(self, freq_array):
        freq_array = np.asarray(freq_array, dtype=np.float64)
        if self.diameter is not None:
            wavelengths = C_LIGHT / freq_array
            scalar = 2.2150894
            return np.arcsin(scalar * wavelengths / (np.pi * self.diameter)) * 2 / 2.355
        ref = self.reference_frequency
        if ref is None:
            return self.sigma * np.ones_like(freq_array)
        return self.sigma * (freq_array / ref) ** self.spectral_index

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        za = np.asarray(za_array, dtype=np.float64)
        sig = self.get_sigmas(np.atleast_1d(freq_array))
        power = np.exp(-(za[np.newaxis, :] ** 2) / (sig[:, np.newaxis] ** 2))
        return power[np.newaxis, np.newaxis, :, :]


class PolyBeam:
    """``hera_sim.beams.PolyBeam`` (hydra/example.py:143-168; hera_sim is unpinned in
    requirements.txt and not installed: PARITY UNPINNED, restated from its published definition).
    E-field: chebval(x, coeffs) / chebval(-1, coeffs), x = 2 sin(za / fscale) - 1,
    fscale = (nu / ref_freq)**spectral_index, i.e. normalised to 1 at zenith; power = E**2."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, beam_coeffs, spectral_index=0.0, ref_freq=1.0e8):
        self.beam_coeffs = np.asarray(beam_coeffs, dtype=np.float64)
        self.spectral_index = spectral_index
        self.ref_freq = ref_freq

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        from numpy.polynomial.chebyshev import chebval
        za = np.asarray(za_array, dtype=np.float64)
        fscale = (np.atleast_1d(freq_array) / self.ref_freq) ** self.spectral_index
        x = 2.0 * np.sin(za[np.newaxis, :] / fscale[:, np.newaxis]) - 1.0
        efield = chebval(x, self.beam_coeffs) / chebval(-1.0, self.beam_coeffs)
        return (efield ** 2)[np.newaxis, np.newaxis, :, :]


class UniformBeam:
    """Unit response everywhere (pyuvdata ``UniformBeam``)."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        nf = np.atleast_1d(freq_array).size
        return np.ones((1, 1, nf, np.asarray(za_array).size))


class TabulatedBeam:
    """Power beam given on a regular (za, az) grid per frequency, bilinear in
    (za, az) and linear in frequency.  Stand-in for a gridded ``UVBeam`` in
    az_za coordinates; used to exercise the table-lookup path of the HIP
    generator.  Not a restatement of pyuvdata's spline.
    """

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, freqs_hz, za_grid, az_grid, power):
        self.freqs = np.asarray(freqs_hz, dtype=np.float64)
        self.za_grid = np.asarray(za_grid, dtype=np.float64)
        self.az_grid = np.asarray(az_grid, dtype=np.float64)
        self.power = np.asarray(power, dtype=np.float64)
        assert self.power.shape == (self.freqs.size, self.za_grid.size, self.az_grid.size)

    def _interp_one(self, tab, az, za):
        dza = self.za_grid[1] - self.za_grid[0]
        daz = self.az_grid[1] - self.az_grid[0]
        nza, naz = self.za_grid.size, self.az_grid.size
        x = (za - self.za_grid[0]) / dza
        y = (az - self.az_grid[0]) / daz
        i0 = np.clip(np.floor(x).astype(np.int64), 0, nza - 2)
        j0 = np.floor(y).astype(np.int64)
        fx = x - i0
        fy = y - j0
        j0 = np.mod(j0, naz)
        j1 = np.mod(j0 + 1, naz)
        i1 = i0 + 1
        return ((1 - fx) * ((1 - fy) * tab[i0, j0] + fy * tab[i0, j1])
                + fx * ((1 - fy) * tab[i1, j0] + fy * tab[i1, j1]))

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        az = np.asarray(az_array, dtype=np.float64)
        za = np.asarray(za_array, dtype=np.float64)
        out = []
        for f in np.atleast_1d(freq_array):
            k = np.clip(np.searchsorted(self.freqs, f) - 1, 0, self.freqs.size - 2)
            w = (f - self.freqs[k]) / (self.freqs[k + 1] - self.freqs[k])
            out.append((1 - w) * self._interp_one(self.power[k], az, za)
                       + w * self._interp_one(self.power[k + 1], az, za))
        return np.asarray(out)[np.newaxis, np.newaxis, :, :]


class SplineBeam:
    """Power beam on a regular (za, az) grid per frequency, interpolated the way pyuvdata's
    ``UVBeam`` does for ``interpolation_function="az_za_simple"`` (the reference calls it with
    ``reuse_spline=True, freq_interp_kind="cubic"``, vis_simulator.py:340-349): the grid is
    first interpolated to the requested frequency with ``scipy.interpolate.interp1d(kind=...)``
    and then evaluated with a bicubic ``scipy.interpolate.RectBivariateSpline(za, az, grid)``.
    A grid that covers the full circle in azimuth is extended by one wrapped column at
    az0 + 2 pi.  pyuvdata itself is not installed: the scipy calls are the pinned part, the
    wrap / clamping conventions are restated from memory (PARITY UNPINNED for those)."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, freqs_hz, za_grid, az_grid, power, freq_interp_kind="cubic"):
        self.freqs = np.asarray(freqs_hz, dtype=np.float64)
        self.za_grid = np.asarray(za_grid, dtype=np.float64)
        az = np.asarray(az_grid, dtype=np.float64)
        power = np.asarray(power, dtype=np.float64)
        assert power.shape == (self.freqs.size, self.za_grid.size, az.size)
        daz = az[1] - az[0]
        if np.isclose(az[-1] - az[0] + daz, 2.0 * np.pi):
            az = np.concatenate([az, [az[0] + 2.0 * np.pi]])
            power = np.concatenate([power, power[:, :, :1]], axis=2)
        self.az_grid, self.power, self.kind = az, power, freq_interp_kind

    def grid_at(self, freq):
        from scipy.interpolate import interp1d
        if self.freqs.size == 1:
            return self.power[0]
        return interp1d(self.freqs, self.power, kind=self.kind, axis=0)(freq)

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        from scipy.interpolate import RectBivariateSpline
        az = np.asarray(az_array, dtype=np.float64)
        za = np.asarray(za_array, dtype=np.float64)
        out = [RectBivariateSpline(self.za_grid, self.az_grid, self.grid_at(f), kx=3, ky=3)
               .ev(za, az) for f in np.atleast_1d(freq_array)]
        return np.asarray(out)[np.newaxis, np.newaxis, :, :]


class HealpixBeam:
    """Power beam tabulated on a HEALPix map (RING) per frequency -- a pyuvdata ``UVBeam`` in
    healpix coordinates with ``interpolation_function = "healpix_simple"`` and
    ``freq_interp_kind = "linear"``, as the reference's SH response script sets it up
    (scripts/precompute_response_sh_singlefreq_general.py:106-110): bilinear interpolation over
    the four nearest pixels (theta = za, phi = az) and linear interpolation between channels.
    Parity unpinned: the arithmetic is pyuvdata's / astropy-healpix's, restated from the HEALPix
    bilinear scheme (oracle.sphere.interp_weights_ring)."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, freqs_hz, power):
        self.freqs = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
        self.power = np.atleast_2d(np.asarray(power, dtype=np.float64))
        assert self.power.shape[0] == self.freqs.size
        self.nside = int(round(np.sqrt(self.power.shape[1] / 12.0)))
        assert 12 * self.nside ** 2 == self.power.shape[1]

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        from . import sphere
        from scipy.interpolate import interp1d
        az = np.asarray(az_array, dtype=np.float64)
        za = np.asarray(za_array, dtype=np.float64)
        pix, wgt = sphere.interp_weights_ring(self.nside, za.ravel(), az.ravel())
        freq = np.atleast_1d(np.asarray(freq_array, dtype=np.float64))
        if self.freqs.size == 1:
            maps = np.repeat(self.power, freq.size, axis=0)
        else:
            maps = interp1d(self.freqs, self.power, kind="linear", axis=0)(freq)
        vals = np.sum(maps[:, pix] * wgt[None], axis=1)          # (Nfreq, Npts)
        return vals.reshape((1, 1, freq.size) + az.shape)
