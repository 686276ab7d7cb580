"""
Oracle (test infrastructure): linearised per-antenna gain GCR operator,
restating ``hydra/gain_sampler.py`` (PINNED by tests/golden/gain_*.npz from the
reference functions; BASELINE config 4).

The sparse (Nbl x Nant) projectors of the reference (A_real: +1,+1; A_imag:
+1,-1 at the two antennas of a baseline, gain_sampler.py:51-78) are written here
as index gathers / scatter-adds over (ant1, ant2).
"""
import numpy as np


def bl_indices(ants, antpairs):
    ants = list(ants)
    pos = {a: ants.index(a) for a in ants}
    i1 = np.array([pos[a] for a, _ in antpairs], dtype=np.int64)
    i2 = np.array([pos[b] for _, b in antpairs], dtype=np.int64)
    return i1, i2


def apply_proj(x, i1, i2, model_vis):
    """(x_i + conj(x_j)) * model_vis   (gain_sampler.py:81-106)."""
    re = 1.0 * (x.real[i1] + x.real[i2])
    im = 1.0 * (x.imag[i1] - x.imag[i2])
    v = np.zeros(model_vis.shape, dtype=np.complex128)
    v[:, :, :] = re + 1.0j * im
    v *= model_vis
    return v


def apply_proj_conj(v, i1, i2, model_vis, gain_shape):
    """Adjoint: scatter-add of v conj(model_vis)   (gain_sampler.py:109-142)."""
    vv = v * model_vis.conj()
    g_re = np.zeros(gain_shape, dtype=np.float64)
    g_im = np.zeros(gain_shape, dtype=np.float64)
    np.add.at(g_re, i1, vv.real)
    np.add.at(g_re, i2, vv.real)
    np.add.at(g_im, i1, vv.imag)
    np.add.at(g_im, i2, -vv.imag)
    return 1.0 * g_re + 1.0j * g_im


def apply_operator(x, inv_noise_var, pspec_sqrt, i1, i2, model_vis, Fbasis):
    """x + S^1/2 F^H A^H N^-1 A F S^1/2 x   (gain_sampler.py:234-301, one rank)."""
    nmodes, nf, nt = Fbasis.shape
    nants = x.size // (2 * nmodes)
    assert inv_noise_var.shape == model_vis.shape
    xre = pspec_sqrt[np.newaxis, :] * x[: x.size // 2].reshape((nants, nmodes))
    xim = pspec_sqrt[np.newaxis, :] * x[x.size // 2:].reshape((nants, nmodes))
    FFc = Fbasis.conj().reshape((nmodes, -1))
    sqrtSx = np.zeros((nants, nf, nt), dtype=np.complex128)
    for k in range(nants):
        sqrtSx[k] = np.tensordot(Fbasis, xre[k] + 1.0j * xim[k], axes=((0,), (0,)))
    y = apply_proj_conj(apply_proj(sqrtSx, i1, i2, model_vis) * inv_noise_var,
                        i1, i2, model_vis, (nants, nf, nt))
    yy = np.zeros((nants, nmodes), dtype=np.complex128)
    for k in range(nants):
        yy[k] = pspec_sqrt * np.tensordot(FFc, y[k].flatten(), axes=((1,), (0,)))
    return x + np.concatenate((yy.real.flatten(), yy.imag.flatten()))


def construct_rhs(resid, inv_noise_var, pspec_sqrt, i1, i2, model_vis, Fbasis,
                  omega_y, omega_r):
    """S^1/2 F^H A^H (N^-1 r + N^-1/2 w_r) + w_y  (gain_sampler.py:145-231, one
    rank, with the two fluctuation terms injected: ``omega_y`` (Nant, Nmodes)
    complex, ``omega_r`` like ``resid``)."""
    nmodes = Fbasis.shape[0]
    nants = omega_y.shape[0]
    _, nf, nt = resid.shape
    yy = apply_proj_conj(resid * inv_noise_var + omega_r * np.sqrt(inv_noise_var),
                         i1, i2, model_vis, (nants, nf, nt))
    FFc = Fbasis.conj().reshape((nmodes, -1))
    my_y = np.zeros_like(omega_y)
    for k in range(nants):
        my_y[k] = pspec_sqrt * np.tensordot(FFc, yy[k].flatten(), axes=((1,), (0,)))
    tot = omega_y + my_y
    return np.concatenate((tot.flatten().real, tot.flatten().imag))


def draw_gain_omegas(seed, nants, nmodes, resid_shape, realisation=True):
    """Replay of the reference's draws (gain_sampler.py:163,176-183,195-199)."""
    np.random.seed(seed)
    s = 1.0 if realisation else 0.0
    omega_y = s * (1.0 * np.random.randn(nants, nmodes)
                   + 1.0j * np.random.randn(nants, nmodes)) / np.sqrt(2.0)
    omega_r = s * (1.0 * np.random.randn(*resid_shape)
                   + 1.0j * np.random.randn(*resid_shape)) / np.sqrt(2.0)
    return omega_y, omega_r
