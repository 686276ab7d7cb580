"""
Primary-beam objects accepted by ``hydra_amd.vis_simulator``.

The reference hands pyuvdata beam objects to ``BeamInterface.compute_response``
(hydra/vis_simulator.py:347-359, hydra/example.py:170-172).  Here a beam is any
object with

    compute_response(az_array=, za_array=, freq_array=, **kw) -> (1, 1, Nfreq, Npts) power

Analytic beams defined in this module additionally carry ``hip_spec()``, which
lets the HIP generator evaluate them in-kernel (no host round trip).  Any other
object (e.g. a real pyuvdata beam, when that package is installed) is evaluated
through its own ``compute_response`` on the host and uploaded as a value table
(HYDRA_BEAM_VALUES).
"""
import numpy as np

C_LIGHT = 299792458.0

HYDRA_BEAM_GAUSS_DIAMETER = 0
HYDRA_BEAM_GAUSS_SIGMA = 1
HYDRA_BEAM_UNIFORM = 2
HYDRA_BEAM_VALUES = 3
HYDRA_BEAM_POLY = 4
HYDRA_BEAM_NPARAMS = 32
HYDRA_BEAM_POLY_MAXCOEF = HYDRA_BEAM_NPARAMS - 3


def _pad(params):
    return list(params) + [0.0] * (HYDRA_BEAM_NPARAMS - len(params))


class GaussianBeam:
    """Gaussian power beam exp(-za^2/sigma^2); ``diameter`` (m) sets
    sigma(nu) = arcsin(2.2150894 lambda / (pi D)) * 2 / 2.355 (pyuvdata's
    ``GaussianBeam(diameter=...)``), or a fixed ``sigma`` (rad) with optional
    power-law scaling from ``reference_frequency``."""

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

    def hip_spec(self):
        if self.diameter is not None:
            return HYDRA_BEAM_GAUSS_DIAMETER, _pad([float(self.diameter)])
        ref = 0.0 if self.reference_frequency is None else float(self.reference_frequency)
        return HYDRA_BEAM_GAUSS_SIGMA, _pad([float(self.sigma), ref, float(self.spectral_index)])

    def get_sigmas(self, freq_array):
        freq_array = np.asarray(freq_array, dtype=np.float64)
        if self.diameter is not None:
            return np.arcsin(2.2150894 * (C_LIGHT / freq_array) / (np.pi * self.diameter)) * 2 / 2.355
        if self.reference_frequency is None:
            return self.sigma * np.ones_like(freq_array)
        return self.sigma * (freq_array / self.reference_frequency) ** self.spectral_index

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        za = np.asarray(za_array, dtype=np.float64)
        sig = self.get_sigmas(np.atleast_1d(freq_array))
        return np.exp(-(za[None, :] ** 2) / (sig[:, None] ** 2))[None, None]


class UniformBeam:
    """Unit response everywhere."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def hip_spec(self):
        return HYDRA_BEAM_UNIFORM, _pad([])

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        return np.ones((1, 1, np.atleast_1d(freq_array).size, np.asarray(za_array).size))


class PolyBeam:
    """Chebyshev-polynomial beam of ``hera_sim.beams.PolyBeam`` (hydra/example.py:143-168):
    E(za, nu) = T(x) / T(-1) with x = 2 sin(za / (nu / ref_freq)**spectral_index) - 1 and T the
    Chebyshev series with coefficients ``beam_coeffs``; the power beam is E**2.  Evaluated inside
    the generator kernel.  hera_sim is not installed here, so the formula restates its published
    definition: PARITY UNPINNED (as for the other third-party beams)."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, beam_coeffs, spectral_index=0.0, ref_freq=1.0e8):
        self.beam_coeffs = np.atleast_1d(np.asarray(beam_coeffs, dtype=np.float64))
        if not 1 <= self.beam_coeffs.size <= HYDRA_BEAM_POLY_MAXCOEF:
            raise ValueError(f"PolyBeam takes 1..{HYDRA_BEAM_POLY_MAXCOEF} coefficients")
        self.spectral_index = float(spectral_index)
        self.ref_freq = float(ref_freq)

    def hip_spec(self):
        return HYDRA_BEAM_POLY, _pad([self.ref_freq, self.spectral_index, float(self.beam_coeffs.size)]
                                     + [float(c) for c in self.beam_coeffs])

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        from numpy.polynomial.chebyshev import chebval
        za = np.asarray(za_array, dtype=np.float64)
        fscale = (np.atleast_1d(freq_array) / self.ref_freq) ** self.spectral_index
        x = 2.0 * np.sin(za[None, :] / fscale[:, None]) - 1.0
        e = chebval(x, self.beam_coeffs) / chebval(-1.0, self.beam_coeffs)
        return (e ** 2)[None, None]


def check_beams(beams, polarized=False):
    """The validation of ``run_checks`` (vis_simulator.py:92-104)."""
    if polarized:
        raise NotImplementedError("polarized=True is not on the GCR hot path")
    for b in beams:
        bad_type = getattr(b, "beam_type", "power") != "power"
        bad_npol = getattr(b, "Npols", 1) > 1
        pol = getattr(b, "polarization_array", [-5])
        if bad_type or bad_npol or pol[0] not in (-5, -6):
            raise ValueError(
                "beam type must be power and have only one pol (either xx or yy) if "
                "polarized=False")


def group_beams(beams):
    """Deduplicate beam objects: returns (unique_beams, beam_idx per input beam).
    Two analytic beams with equal ``hip_spec()`` are the same beam; other objects are
    compared by identity.  (The reference evaluates one beam per antenna even when they
    are identical, vis_simulator.py:248-253.)"""
    uniq, keys, idx = [], [], []
    for b in beams:
        key = ("spec",) + tuple(map(repr, b.hip_spec())) if hasattr(b, "hip_spec") else ("id", id(b))
        if key in keys:
            idx.append(keys.index(key))
        else:
            keys.append(key)
            uniq.append(b)
            idx.append(len(uniq) - 1)
    return uniq, np.asarray(idx, dtype=np.int32)


class GriddedBeam:
    """Power beam tabulated on a regular (za, az) grid per frequency -- the role of a pyuvdata
    ``UVBeam`` in az_za coordinates (BASELINE config 5 "UVBeam interp").

    Interpolation follows pyuvdata's ``az_za_simple`` recipe as the reference drives it
    (``reuse_spline=True, freq_interp_kind="cubic"``, vis_simulator.py:340-349): cubic
    ``interp1d`` in frequency on the grid, then a bicubic ``RectBivariateSpline`` in (za, az).
    The spline FIT (small: one grid per channel) is done once per channel on the host with
    scipy and cached; the EVALUATION at Nsrc x Ntimes points per channel -- the part the
    reference repeats per antenna, time and frequency -- runs on the GPU
    (``hydra_beam_bspline``).  A full-circle azimuth grid is extended by a wrapped column."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, freqs_hz, za_grid, az_grid, power, freq_interp_kind="cubic"):
        self.freqs = np.asarray(freqs_hz, dtype=np.float64)
        self.za_grid = np.asarray(za_grid, dtype=np.float64)
        az = np.asarray(az_grid, dtype=np.float64)
        power = np.asarray(power, dtype=np.float64)
        if power.shape != (self.freqs.size, self.za_grid.size, az.size):
            raise ValueError("power must have shape (Nfreq, Nza, Naz)")
        daz = az[1] - az[0]
        if np.isclose(az[-1] - az[0] + daz, 2.0 * np.pi):
            az = np.concatenate([az, [az[0] + 2.0 * np.pi]])
            power = np.concatenate([power, power[:, :, :1]], axis=2)
        if self.za_grid.size < 4 or az.size < 4:
            raise ValueError("a bicubic spline needs at least 4 grid points per axis")
        self.az_grid, self.power, self.kind = az, power, freq_interp_kind
        self._fits = {}

    def _grid_at(self, freq):
        from scipy.interpolate import interp1d
        if self.freqs.size == 1:
            return self.power[0]
        return interp1d(self.freqs, self.power, kind=self.kind, axis=0)(freq)

    def _fit(self, freq):
        """(tx, ty, coefficients) of the interpolating bicubic spline at ``freq`` (cached)."""
        key = float(freq)
        if key not in self._fits:
            from scipy.interpolate import RectBivariateSpline
            spl = RectBivariateSpline(self.za_grid, self.az_grid, self._grid_at(freq), kx=3, ky=3)
            tx, ty = spl.get_knots()
            self._fits[key] = (np.ascontiguousarray(tx), np.ascontiguousarray(ty),
                               np.ascontiguousarray(spl.get_coeffs()))
        return self._fits[key]

    def device_response(self, az, za, freqs_hz):
        """Power values (Nfreq,) + az.shape as a GPU tensor; ``az``, ``za`` GPU tensors."""
        import torch
        from . import _dev
        freqs_hz = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
        fits = [self._fit(f) for f in freqs_hz]
        tx, ty = fits[0][0], fits[0][1]
        coef = _dev.f64(np.stack([f[2] for f in fits]))
        tx_d, ty_d = _dev.f64(tx), _dev.f64(ty)
        az_d, za_d = _dev.f64(az), _dev.f64(za)
        npts = az_d.numel()
        out = _dev.empty((freqs_hz.size,) + tuple(az_d.shape), torch.float64)
        _dev.call("hydra_beam_bspline", _dev.ptr(tx_d), tx.size, _dev.ptr(ty_d), ty.size,
                  _dev.ptr(coef), freqs_hz.size, _dev.ptr(za_d), _dev.ptr(az_d), npts,
                  _dev.ptr(out), _dev.stream_ptr())
        return out

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        """Host-array interface of the beam protocol (evaluated on the GPU as well)."""
        out = self.device_response(np.asarray(az_array, dtype=np.float64),
                                   np.asarray(za_array, dtype=np.float64), freq_array)
        return out.cpu().numpy()[np.newaxis, np.newaxis, :, :]


class HealpixBeam:
    """Power beam tabulated on a HEALPix map (RING ordering) per frequency -- the role of a
    pyuvdata ``UVBeam`` in healpix coordinates with ``interpolation_function =
    "healpix_simple"`` and ``freq_interp_kind = "linear"``
    (scripts/precompute_response_sh_singlefreq_general.py:106-110): bilinear interpolation over
    the four nearest pixels and linear interpolation between channels, evaluated on the GPU
    (``hydra_beam_healpix``)."""

    beam_type = "power"
    Npols = 1
    polarization_array = np.array([-5])

    def __init__(self, freqs_hz, power):
        self.freqs = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
        power = np.atleast_2d(np.asarray(power, dtype=np.float64))
        if power.shape[0] != self.freqs.size:
            raise ValueError("power must have shape (Nfreq, Npix)")
        self.nside = int(round(np.sqrt(power.shape[1] / 12.0)))
        if 12 * self.nside ** 2 != power.shape[1]:
            raise ValueError("Npix is not 12 nside^2")
        if self.freqs.size > 1 and np.any(np.diff(self.freqs) <= 0):
            raise ValueError("tabulated frequencies must increase")
        self.power = power
        self._maps_d = None

    def _channel_weights(self, freqs_hz):
        """Neighbouring tabulated channels and the weight of the upper one (scipy interp1d
        'linear': a frequency outside the table is an error)."""
        f = np.atleast_1d(np.asarray(freqs_hz, dtype=np.float64))
        if self.freqs.size == 1:
            z = np.zeros(f.size, dtype=np.int32)
            return z, z, np.zeros(f.size)
        if f.min() < self.freqs[0] or f.max() > self.freqs[-1]:
            raise ValueError("A value in freq_array is outside the tabulated frequency range.")
        hi = np.clip(np.searchsorted(self.freqs, f, side="left"), 1, self.freqs.size - 1)
        lo = hi - 1
        w = (f - self.freqs[lo]) / (self.freqs[hi] - self.freqs[lo])
        return lo.astype(np.int32), hi.astype(np.int32), w

    def device_response(self, az, za, freqs_hz):
        """Power values (Nfreq,) + az.shape as a GPU tensor; ``az``, ``za`` GPU tensors."""
        import torch
        from . import _dev
        if self._maps_d is None:
            self._maps_d = _dev.f64(self.power)
        lo, hi, w = self._channel_weights(freqs_hz)
        az_d, za_d = _dev.f64(az), _dev.f64(za)
        out = _dev.empty((lo.size,) + tuple(az_d.shape), torch.float64)
        lo_d, hi_d, w_d = _dev.i32(lo), _dev.i32(hi), _dev.f64(w)
        _dev.call("hydra_beam_healpix", _dev.ptr(self._maps_d), self.freqs.size, self.nside,
                  _dev.ptr(lo_d), _dev.ptr(hi_d), _dev.ptr(w_d), lo.size, _dev.ptr(za_d),
                  _dev.ptr(az_d), az_d.numel(), _dev.ptr(out), _dev.stream_ptr())
        return out

    def compute_response(self, *, az_array, za_array, freq_array, **_unused):
        """Host-array interface of the beam protocol (evaluated on the GPU as well)."""
        out = self.device_response(np.asarray(az_array, dtype=np.float64),
                                   np.asarray(za_array, dtype=np.float64), freq_array)
        return out.cpu().numpy()[np.newaxis, np.newaxis, :, :]
