"""
Cosmological-field head on MI355X -- drop-in for ``hydra.cosmo_sampler``
(reference: hydra/cosmo_sampler.py).

    precompute_mpi(...)      -> (G (Nf,Npix,Npix), b (Nf,Npix))            cosmo_sampler.py:225-351
        G_f = Re(R_f^H N^-1 R_f),  b_f = Re(R_f^H (N^-1 d + N^-1/2 w_n)) + (S^-1/2 w_s)_f
    apply_lhs_operator(x, G, pspec3d) = S^-1 x + [G_f x_f]_f                 cosmo_sampler.py:197-222
    apply_S(x, pspec, exponent)       = FFT^-1(pspec^exponent FFT x).real    cosmo_sampler.py:167-194

The field is sampled on a (freq, RA, Dec) grid whose pixels enter the simulator as unit-flux
point sources (example.py:545-556), so the projection operator is the point-source one and may
be passed in factored form (``ptsrc_sampler.calc_proj_operator(..., materialize=False)``).  The
per-frequency Grams run on the same FP64 matrix-core kernels as the point-source head, one
channel per launch; the likelihood term of the operator is one batched matrix-vector kernel
(``hydra_gemv_batched``); the prior term is a 3D FFT (rocFFT through ``torch.fft`` -- library
plumbing, not a hand-written kernel).

Extensions that do not change the reference call signatures: ``comm=None`` runs serially (the
reference dereferences ``comm`` unconditionally, cosmo_sampler.py:250); ``omega_n=``/``omega_s=``
inject the fluctuation vectors, otherwise drawn from the global numpy RNG in the reference's
order (:308-312, then :346 on the root).
"""
import numpy as np
import torch
from scipy.integrate import cumulative_trapezoid
from scipy.interpolate import interp1d

from . import _dev, _tables
from .ptsrc_sampler import FactoredProj, _timed_gram

C = 299792.458               # speed of light, km/s        (cosmo_sampler.py:11)
FREQ_21CM = 1420.405751768   # MHz                         (cosmo_sampler.py:12)


# ----------------------------------------------------------------------------- host set-up

def calculate_cosmo_fns(h=0.69, omega_m=0.31):
    """Interpolation functions ``Hz(z)`` [km/s/Mpc] and comoving angular diameter distance
    ``dAc(z)`` [Mpc] of a flat LambdaCDM background (cosmo_sampler.py:15-42)."""
    zz = np.concatenate(([0.0], np.logspace(-4.0, 2.0)))
    hub = 100.0 * h * np.sqrt(omega_m * (1.0 + zz) ** 3.0 + (1.0 - omega_m))
    Hz = interp1d(zz, hub, kind="quadratic")
    dist = cumulative_trapezoid(C / Hz(zz), zz, initial=0.0)
    return Hz, interp1d(zz, dist, kind="quadratic")


def make_cosmo_field_grid(args):
    """RA, Dec (radians, flattened meshgrid) of the field's sample points from the driver's
    ``cosmo_field_{ra,dec}_{bounds,ngrid}`` settings in degrees (cosmo_sampler.py:45-74)."""
    ra = np.linspace(min(args.cosmo_field_ra_bounds), max(args.cosmo_field_ra_bounds),
                     args.cosmo_field_ra_ngrid)
    dec = np.linspace(min(args.cosmo_field_dec_bounds), max(args.cosmo_field_dec_bounds),
                      args.cosmo_field_dec_ngrid)
    ra_grid, dec_grid = np.meshgrid(ra, dec)
    return np.deg2rad(ra_grid.flatten()), np.deg2rad(dec_grid.flatten())


def comoving_fourier_modes(x, y, freqs, **cosmo_params):
    """Comoving wavenumbers ``kx, ky, knu`` [1/Mpc], FFT ordering, of a grid with angular axes
    ``x, y`` [deg, flat sky] and ``freqs`` [MHz] (cosmo_sampler.py:77-117)."""
    Hz, dAc = calculate_cosmo_fns(**cosmo_params)
    zc = FREQ_21CM / (0.5 * (freqs.min() + freqs.max())) - 1.0
    dist = dAc(zc)
    step_x = np.deg2rad(x[1] - x[0]) * dist
    step_y = np.deg2rad(y[1] - y[0]) * dist
    step_nu = (C * (1.0 + zc) ** 2.0 / Hz(zc)) * (freqs[1] - freqs[0]) / FREQ_21CM
    return (2.0 * np.pi * np.fft.fftfreq(n=x.size, d=step_x),
            2.0 * np.pi * np.fft.fftfreq(n=y.size, d=step_y),
            2.0 * np.pi * np.fft.fftfreq(n=freqs.size, d=step_nu))


def calculate_pspec_on_grid(kbins, pspec, x, y, freqs, **cosmo_params):
    """Binned 1D power spectrum ``P(|k|)`` evaluated (nearest bin) on the 3D Fourier grid of
    shape (Nfreqs, Nx, Ny) (cosmo_sampler.py:120-164)."""
    assert len(np.shape(x)) == len(np.shape(y)) == len(np.shape(freqs)) == 1, \
        "x, y, and freqs coordinate arrays must be 1D"
    assert kbins.size == pspec.size, "kbins and pspec must have the same length"
    kx, ky, knu = comoving_fourier_modes(x, y, freqs, **cosmo_params)
    knu3d, kx3d, ky3d = np.meshgrid(knu, kx, ky, indexing="ij")
    kmag = np.sqrt(kx3d ** 2.0 + ky3d ** 2.0 + knu3d ** 2.0)
    return interp1d(kbins, pspec, kind="nearest", fill_value=(pspec[0], pspec[-1]))(kmag)


# ----------------------------------------------------------------------------- operator

def _half_spectrum(pspec, exponent):
    """pspec**exponent on the half spectrum of a real transform, symmetrised: for real x,
    Re IFFT(P FFT(x)) = IFFT(((P(k) + P(-k)) / 2) FFT(x)), so the reference's ``.real`` of a full
    complex transform (cosmo_sampler.py:190-194) is one real-to-complex and one complex-to-real
    transform with that filter -- half the FFT work and three kernels fewer per application.
    Cached on the (GPU) tensor until it is modified in place."""
    key = (float(exponent), pspec._version)
    cache = getattr(pspec, "_hydra_half", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    P = pspec ** exponent
    Pm = torch.roll(torch.flip(P, dims=(0, 1, 2)), shifts=(1, 1, 1), dims=(0, 1, 2))   # P(-k)
    half = (0.5 * (P + Pm))[..., : P.shape[2] // 2 + 1].contiguous()
    try:
        pspec._hydra_half = (key, half)
    except AttributeError:
        pass
    return half


def _apply_S_dev(x, pspec, exponent):
    return torch.fft.irfftn(_half_spectrum(pspec, exponent) * torch.fft.rfftn(x), s=tuple(x.shape))


def apply_S(x, pspec, exponent=1):
    """Prior covariance (to a power) applied in Fourier space, where it is diagonal:
    ``FFT^-1(pspec**exponent * FFT(x)).real`` on a (Nfreqs, Nx, Ny) cube
    (cosmo_sampler.py:167-194)."""
    assert len(x.shape) == 3, "Input vector x must be a 3D array with shape (Nfreqs, Nx, Ny)"
    return _dev.like(x, _apply_S_dev(_dev.f64(x), _dev.f64(pspec), exponent))


def apply_lhs_operator(x, lhs_Ninv_operator, pspec3d):
    """``(S^-1 + X^dagger N^-1 X) x`` for a field ``x`` (Nfreqs, Nx, Ny): prior term in Fourier
    space plus the precomputed per-frequency likelihood blocks ``lhs_Ninv_operator``
    (Nfreqs, Npix, Npix).  Returns (Nfreqs, Npix) (cosmo_sampler.py:197-222)."""
    xd = _dev.f64(x)
    nf = xd.shape[0]
    npix = xd.shape[1] * xd.shape[2]
    G = _dev.f64(lhs_Ninv_operator)
    assert tuple(G.shape) == (nf, npix, npix)
    y = _apply_S_dev(xd, _dev.f64(pspec3d), -1.0).reshape(nf, npix)
    # the blocks are symmetric bit for bit (the Gram kernels mirror the lower triangle), which
    # lets the product read half of each; any other stack of matrices gets the plain product
    if getattr(lhs_Ninv_operator, "hydra_symmetric", None) is None and _dev.is_torch(lhs_Ninv_operator):
        lhs_Ninv_operator.hydra_symmetric = bool(torch.equal(G, G.transpose(1, 2)))
    if getattr(lhs_Ninv_operator, "hydra_symmetric", False):
        # the piece workspace lives with the operator: one allocation, not one per application
        ws = getattr(lhs_Ninv_operator, "_hydra_symv_ws", None)
        if ws is None or ws.device != G.device:
            ws = _dev.workspace(_dev._lib.query("hydra_symv_workspace_bytes", npix, nf))
            try:
                lhs_Ninv_operator._hydra_symv_ws = ws
            except AttributeError:
                pass
        _dev.call("hydra_symv_batched", _dev.ptr(G), npix, nf, _dev.ptr(xd), _dev.ptr(y), 1,
                  _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
    else:
        _dev.call("hydra_gemv_batched", _dev.ptr(G), npix, nf, _dev.ptr(xd), _dev.ptr(y), 1,
                  _dev.stream_ptr())
    return _dev.like(x, y)


# ----------------------------------------------------------------------------- linear system

def _draw_omega_n(shape, realisation):
    sw = 1.0 if realisation else 0.0      # cosmo_sampler.py:304-312: Re then Im
    return sw * (1.0 * np.random.randn(*shape) + 1.0j * np.random.randn(*shape)) / np.sqrt(2.0)


def precompute_mpi(comm, ants, antpairs, freqs, freq_chunk, time_chunk, proj_chunk, data_chunk,
                   inv_noise_var_chunk, gain_chunk, pspec3d, realisation=True, omega_n=None,
                   omega_s=None):
    """Per-frequency likelihood blocks and right-hand side of the cosmological-field GCR system
    (cosmo_sampler.py:225-351).  ``freqs`` are all channels of the field grid, ``freq_chunk``
    those held by this rank (rows of the other channels stay zero until the sum over ranks).
    ``proj_chunk`` (Nbl, Nf_chunk, Nt, Npix) complex128 or a ``FactoredProj``.  Only the
    root's return value is the full system."""
    myid = comm.Get_rank() if comm is not None else 0
    nbl = len(antpairs)
    freqs_h = np.asarray(freqs.cpu() if _dev.is_torch(freqs) else freqs, dtype=np.float64)
    fchunk_h = np.atleast_1d(np.asarray(
        freq_chunk.cpu() if _dev.is_torch(freq_chunk) else freq_chunk, dtype=np.float64))
    nfc, nt = fchunk_h.size, int(np.size(time_chunk))
    assert tuple(data_chunk.shape) == (nbl, nfc, nt)
    assert tuple(data_chunk.shape) == tuple(inv_noise_var_chunk.shape)
    factored = isinstance(proj_chunk, FactoredProj)
    npix = int(proj_chunk.shape[-1])
    assert tuple(proj_chunk.shape[:3]) == (nbl, nfc, nt)
    ref = data_chunk

    bl_i_h, bl_j_h = _tables.baseline_indices(ants, antpairs)
    bl_i, bl_j = _dev.i32(bl_i_h), _dev.i32(bl_j_h)
    data = _dev.c128(data_chunk)
    w = _dev.f64(inv_noise_var_chunk)
    gains = _dev.c128(gain_chunk)
    assert gains.shape[1:] == (nfc, nt)
    nant = gains.shape[0]
    st = _dev.stream_ptr()
    if factored and not (np.array_equal(bl_i_h, proj_chunk.bl_i_host)
                         and np.array_equal(bl_j_h, proj_chunk.bl_j_host)):
        raise ValueError("antpairs do not match the baselines of the factored operator")

    if omega_n is None:
        omega_n = _draw_omega_n((nbl, nfc, nt), realisation)
    om_n = _dev.c128(omega_n)
    assert tuple(om_n.shape) == (nbl, nfc, nt)

    # z = conj(g_i conj g_j) (d N^-1 + omega_n N^-1/2): the gains sit on the rows of R (:258-265)
    gg = _dev.empty((nbl, nfc, nt), torch.complex128)
    _dev.call("hydra_row_factors", _dev.ptr(gains), _dev.ptr(w), _dev.ptr(bl_i), _dev.ptr(bl_j),
              nbl, nfc, nt, _dev.ptr(gg), _dev.ptr(None), st)
    z = _dev.empty((nbl, nfc, nt), torch.complex128)
    _dev.call("hydra_rhs_rows", _dev.ptr(data), _dev.ptr(None), _dev.ptr(w), _dev.ptr(om_n),
              _dev.ptr(gg), nbl * nfc * nt, _dev.ptr(z), st)

    op = _dev.zeros((freqs_h.size, npix, npix), torch.float64)
    b = _dev.zeros((freqs_h.size, npix), torch.float64)
    if not factored:
        R_all = _dev.c128(proj_chunk)
    # one analysis of the noise weights for all channels (cached with the weight tensor)
    plans = proj_chunk.channel_plans(w) if factored and proj_chunk.complete \
        and proj_chunk.allow_uniform else None
    for j in range(nfc):
        hit = np.nonzero(freqs_h == fchunk_h[j])[0]          # :279, :325
        if hit.size == 0:
            continue
        w_j = w[:, j:j + 1, :].contiguous()
        g_j = gains[:, j:j + 1, :].contiguous()
        z_j = z[:, j:j + 1, :].contiguous()
        if factored:
            sub = proj_chunk.channel(j)
            G_j = _timed_gram(lambda: sub.gram(g_j, w_j, plan=None if plans is None else plans[j]))
            proj_chunk.last_gram_kernel = sub.last_gram_kernel
            b_j = sub.adjoint(z_j)
        else:
            R_j = R_all[:, j].reshape(nbl * nt, npix).contiguous()
            rows = _dev.empty((nbl, 1, nt), torch.complex128)
            _dev.call("hydra_row_factors", _dev.ptr(g_j), _dev.ptr(w_j), _dev.ptr(bl_i),
                      _dev.ptr(bl_j), nbl, 1, nt, _dev.ptr(None), _dev.ptr(rows), st)
            ws = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", nbl * nt, npix))
            G_j = _dev.empty((npix, npix), torch.float64)
            _dev.call("hydra_gram", _dev.ptr(R_j), _dev.ptr(rows), nbl * nt, npix, _dev.ptr(None),
                      _dev.ptr(None), _dev.ptr(G_j), _dev.ptr(ws), ws.numel(), st)
            ws = _dev.workspace(_dev._lib.query("hydra_adjoint_workspace_bytes", nbl * nt, npix))
            b_j = _dev.empty((npix,), torch.float64)
            _dev.call("hydra_adjoint_R", _dev.ptr(R_j), nbl * nt, npix, _dev.ptr(z_j),
                      _dev.ptr(None), _dev.ptr(None), _dev.ptr(b_j), _dev.ptr(ws), ws.numel(), st)
        for i in hit:
            op[i] = G_j
            b[i] = b_j

    if comm is not None:
        comm.reduce_(op, root=0)        # comm.Reduce, cosmo_sampler.py:295-299,339-340
        comm.reduce_(b, root=0)
    if myid == 0:
        p3 = _dev.f64(pspec3d)
        if omega_s is None:
            omega_s = (1.0 if realisation else 0.0) * np.random.randn(*p3.shape)   # :346
        b += _apply_S_dev(_dev.f64(omega_s), p3, -0.5).reshape(b.shape[0], -1)
    else:
        op = torch.zeros((1, 1, 1), dtype=torch.float64, device=op.device)
        b = torch.zeros((1, 1), dtype=torch.float64, device=b.device)
    return _dev.like(ref, op), _dev.like(ref, b)
