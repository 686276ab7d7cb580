"""
Oracle (test infrastructure): point-source GCR system, restating
``hydra/ptsrc_sampler.py`` (PINNED by tests/golden/ptsrc_*.npz, generated from
the reference's own ``precompute_mpi``).

  A = 1 + S^1/2 Re(R^H N^-1 R) S^1/2          ptsrc_sampler.py:53-94
  b = S^1/2 Re(R^H (N^-1 r + N^-1/2 w_n)) + w_a   :97-138
  r = d - R 1                                   :104-106
with the gains g_i conj(g_j) folded into the rows of R first (:53-59).
"""
import numpy as np

from . import response


def calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, beams,
                       latitude=-0.5361913261514378):
    """ptsrc_sampler.py:143-215.  ``freqs`` in MHz.
    Returns (Nbl, Nf, Nt, Nsrc) complex128."""
    vis = response.simulate_vis_per_source(
        ants=ant_pos, fluxes=fluxes, ra=ra, dec=dec, freqs=freqs * 1e6,
        lsts=times, beams=beams, latitude=latitude)
    names = list(ant_pos.keys())
    out = np.zeros((len(antpairs), freqs.size, times.size, ra.size), dtype=np.complex128)
    for k, (a1, a2) in enumerate(antpairs):
        out[k] = vis[:, :, names.index(a1), names.index(a2), :]
    return out


def draw_omegas(data_shape, nsrcs, realisation=True):
    """Draw (omega_n, omega_a) from the *global* legacy numpy RNG in the
    reference's order (ptsrc_sampler.py:109-116 then :138): randn(shape) for
    Re, randn(shape) for Im, scaled by 1/sqrt 2; then randn(Ns)."""
    switch = 1.0 if realisation else 0.0
    omega_n = switch * (1.0 * np.random.randn(*data_shape)
                        + 1.0j * np.random.randn(*data_shape)) / np.sqrt(2.0)
    omega_a = switch * np.random.randn(nsrcs)
    return omega_n, omega_a


def apply_gains_to_proj(ants, antpairs, proj, gains):
    """Rows of R times g_i conj(g_j)  (ptsrc_sampler.py:48-59)."""
    ants = np.atleast_1d(ants)
    out = proj.copy()
    for k, (a1, a2) in enumerate(antpairs):
        i1 = np.where(ants == a1)[0][0]
        i2 = np.where(ants == a2)[0][0]
        out[k] *= gains[i1, :, :, np.newaxis] * gains[i2, :, :, np.newaxis].conj()
    return out


def gram(projg, inv_noise_var):
    """Re(R^H N^-1 R) as two real products (ptsrc_sampler.py:62-75)."""
    nsrcs = projg.shape[-1]
    rt = np.sqrt(inv_noise_var[..., np.newaxis])
    v_re = (projg.real * rt).reshape((-1, nsrcs))
    v_im = (projg.imag * rt).reshape((-1, nsrcs))
    return v_re.T @ v_re + v_im.T @ v_im


def precompute(ants, antpairs, proj, data, inv_noise_var, gains, amp_prior_std,
               omega_n=None, omega_a=None, realisation=True):
    """Single-process ``precompute_mpi(comm=None, ...)`` with injectable
    fluctuation vectors.  If ``omega_n``/``omega_a`` are None they are drawn from
    the global RNG in the reference's order.  Returns (A, b)."""
    assert data.shape == proj.shape[:3]
    assert data.shape == inv_noise_var.shape
    nsrcs = proj.shape[-1]
    projg = apply_gains_to_proj(ants, antpairs, proj, gains)
    G = gram(projg, inv_noise_var)
    A = np.eye(nsrcs) + np.diag(amp_prior_std) @ G @ np.diag(amp_prior_std)

    R2 = projg.reshape((-1, nsrcs))
    resid = data.copy() - (R2 @ np.ones_like(amp_prior_std)).reshape(data.shape)
    if omega_n is None:
        switch = 1.0 if realisation else 0.0
        omega_n = switch * (1.0 * np.random.randn(*resid.shape)
                            + 1.0j * np.random.randn(*resid.shape)) / np.sqrt(2.0)
    y = ((resid * inv_noise_var) + (omega_n * np.sqrt(inv_noise_var))).flatten()
    b = amp_prior_std * (R2.T.real @ y.real + R2.T.imag @ y.imag)
    if omega_a is None:
        switch = 1.0 if realisation else 0.0
        omega_a = switch * np.random.randn(nsrcs)
    b = b + omega_a
    return A, b


def dense_solve(A, b):
    """The reference's default solve for this head (example.py:944)."""
    import scipy.linalg
    return scipy.linalg.solve(A, b, assume_a="her")
