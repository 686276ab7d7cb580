"""
Oracle (test infrastructure): cosmological-field GCR head, restating
``hydra/cosmo_sampler.py`` (PINNED by tests/golden/cosmo.npz, generated from the
reference's own functions -- the module needs numpy/scipy only).

  per frequency f:  G_f = Re(R_f^H N^-1 R_f)            cosmo_sampler.py:267-284
                    b_f = Re(R_f^H (N^-1 d + N^-1/2 w_n)) + (S^-1/2 w_s)_f   :303-349
  operator:         x -> S^-1 x + [G_f x_f]_f, S diagonal in 3D Fourier space   :197-222
The gains g_i conj(g_j) are folded into the rows of R first (:258-265); unlike the
point-source head the data are used as they are (no model subtraction).
"""
import numpy as np
from scipy.integrate import cumulative_trapezoid
from scipy.interpolate import interp1d

C_KMS = 299792.458            # cosmo_sampler.py:11
FREQ_21CM = 1420.405751768    # cosmo_sampler.py:12, MHz


def background(h=0.69, omega_m=0.31):
    """H(z) [km/s/Mpc] and comoving distance r(z) [Mpc] of flat LCDM as quadratic
    interpolants on z in {0} U logspace(-4, 2)  (cosmo_sampler.py:15-42)."""
    z = np.concatenate(([0.0], np.logspace(-4.0, 2.0)))
    hz = 100.0 * h * np.sqrt(omega_m * (1.0 + z) ** 3.0 + (1.0 - omega_m))
    hz_fn = interp1d(z, hz, kind="quadratic")
    r = cumulative_trapezoid(C_KMS / hz_fn(z), z, initial=0.0)
    return hz_fn, interp1d(z, r, kind="quadratic")


def field_grid(ra_bounds, dec_bounds, ra_ngrid, dec_ngrid):
    """Flattened meshgrid of the field's sample points, degrees in, radians out
    (cosmo_sampler.py:45-74)."""
    ra = np.linspace(min(ra_bounds), max(ra_bounds), ra_ngrid)
    dec = np.linspace(min(dec_bounds), max(dec_bounds), dec_ngrid)
    g_ra, g_dec = np.meshgrid(ra, dec)
    return np.deg2rad(g_ra.flatten()), np.deg2rad(g_dec.flatten())


def fourier_modes(x, y, freqs, **cosmo):
    """Comoving wavenumbers [1/Mpc] in FFT order for the grid axes x, y [deg] and
    freqs [MHz]  (cosmo_sampler.py:77-117)."""
    hz_fn, r_fn = background(**cosmo)
    zc = FREQ_21CM / (0.5 * (freqs.min() + freqs.max())) - 1.0
    r = r_fn(zc)
    dx = np.deg2rad(x[1] - x[0]) * r
    dy = np.deg2rad(y[1] - y[0]) * r
    dnu = (C_KMS * (1.0 + zc) ** 2.0 / hz_fn(zc)) * (freqs[1] - freqs[0]) / FREQ_21CM
    two_pi = 2.0 * np.pi
    return (two_pi * np.fft.fftfreq(n=x.size, d=dx), two_pi * np.fft.fftfreq(n=y.size, d=dy),
            two_pi * np.fft.fftfreq(n=freqs.size, d=dnu))


def pspec_on_grid(kbins, pspec, x, y, freqs, **cosmo):
    """Band powers P(|k|) looked up (nearest bin) on the (freq, x, y) Fourier grid
    (cosmo_sampler.py:120-164)."""
    assert np.ndim(x) == np.ndim(y) == np.ndim(freqs) == 1
    assert kbins.size == pspec.size
    kx, ky, knu = fourier_modes(x, y, freqs, **cosmo)
    knu3, kx3, ky3 = np.meshgrid(knu, kx, ky, indexing="ij")
    k = np.sqrt(kx3 ** 2.0 + ky3 ** 2.0 + knu3 ** 2.0)
    lookup = interp1d(kbins, pspec, kind="nearest", fill_value=(pspec[0], pspec[-1]))
    return lookup(k)


def apply_S(x, pspec, exponent=1):
    """FFT^-1( pspec^exponent FFT(x) ).real on a (Nf, Nx, Ny) cube (cosmo_sampler.py:167-194)."""
    assert x.ndim == 3
    return np.fft.ifftn(pspec ** exponent * np.fft.fftn(x)).real


def apply_lhs_operator(x, lhs_Ninv_operator, pspec3d):
    """(S^-1 + X^H N^-1 X) x with the likelihood part precomputed per frequency
    (cosmo_sampler.py:197-222).  Returns (Nf, Npix)."""
    nf = x.shape[0]
    y = apply_S(x, pspec3d, exponent=-1.0).reshape((nf, -1))
    for i in range(nf):
        y[i] += lhs_Ninv_operator[i] @ x[i].flatten()
    return y


def precompute(ants, antpairs, freqs, freq_chunk, proj, data, inv_noise_var, gains, pspec3d,
               omega_n=None, omega_s=None, realisation=True):
    """Single-process form of ``precompute_mpi`` (cosmo_sampler.py:225-351) with injectable
    fluctuation vectors; drawn from the global RNG in the reference's order (omega_n: Re then
    Im, :308-312; then omega_s, :346) when None.  Returns (linear_op (Nf_all, Npix, Npix),
    linear_rhs (Nf_all, Npix))."""
    assert data.shape == (len(antpairs), freq_chunk.size, proj.shape[2])
    assert data.shape == inv_noise_var.shape
    ants = np.atleast_1d(ants)
    projg = proj.copy()
    for k, (a1, a2) in enumerate(antpairs):
        i1 = np.where(ants == a1)[0][0]
        i2 = np.where(ants == a2)[0][0]
        projg[k] *= gains[i1, :, :, np.newaxis] * gains[i2, :, :, np.newaxis].conj()
    npix = projg.shape[-1]
    rt = np.sqrt(inv_noise_var[..., np.newaxis])
    v_re, v_im = projg.real * rt, projg.imag * rt
    op = np.zeros((freqs.size, npix, npix))
    for j in range(freq_chunk.size):
        i = np.where(freqs == freq_chunk[j])[0]
        a = v_re[:, j].reshape((-1, npix))
        c = v_im[:, j].reshape((-1, npix))
        op[i] = a.T @ a + c.T @ c
    sw = 1.0 if realisation else 0.0
    if omega_n is None:
        omega_n = sw * (1.0 * np.random.randn(*data.shape)
                        + 1.0j * np.random.randn(*data.shape)) / np.sqrt(2.0)
    y = data * inv_noise_var + omega_n * np.sqrt(inv_noise_var)
    b = np.zeros((freqs.size, npix))
    for j in range(freq_chunk.size):
        i = np.where(freqs == freq_chunk[j])[0]
        pj = projg[:, j].reshape((-1, npix))
        b[i] = pj.T.real @ y[:, j].real.flatten() + pj.T.imag @ y[:, j].imag.flatten()
    if omega_s is None:
        omega_s = sw * np.random.randn(*pspec3d.shape)
    bs = apply_S(omega_s, pspec3d, exponent=-0.5)
    for i in range(b.shape[0]):
        b[i] += bs[i].flatten()
    return op, b
