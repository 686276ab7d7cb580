"""
Spherical-harmonic diffuse head on MI355X -- drop-in for the GCR functions of
``hydra.sh_sampler`` (reference: hydra/sh_sampler.py).

    vis_proj_operator_no_rot(...) -> (R (Nvis, Nm), autos, ell, m)   sh_sampler.py:65-196
    construct_rhs_no_rot[_mpi]     b = R^T (N^-1 d + N^-1/2 w_n) + S^-1 a_0 + S^-1/2 w_a   :370-448
    apply_lhs_no_rot[_mpi]         (R^T N^-1 R + S^-1) a                                    :389-494
    get_em_ell_idx, alms2healpy, healpy2alms                                               :28-62,199-266

``R^T y`` for complex R and the real parameter vector means Re(R)^T Re(y) + Im(R)^T Im(y)
= Re(R^H y).  The reference evaluates the left-hand side left-to-right, i.e. it builds the
(Nm, Nm) Gram on every CG iteration; here the same operator is applied as R^H (N^-1 (R a))
(two passes over R, HBM bound), and ``lhs_matrix`` builds the Gram once on the FP64 matrix
cores for callers that iterate many times.  The GSM helpers of the reference module need pygdsm
data files and are not provided; ``sample_cl`` is a stub there.
"""
import os

import numpy as np
import torch

from . import _dev
from . import vis_simulator as _vs


def get_em_ell_idx(lmax):
    """(ems, ells, idx) of the real parameter vector: all real parts in (m, l) order, then the
    imaginary parts for m >= 1 (sh_sampler.py:28-62)."""
    ems, ells = [], []
    for first_m in (0, 1):
        for em in range(first_m, lmax + 1):
            for ell in range(em, lmax + 1):
                ems.append(em)
                ells.append(ell)
    return ems, ells, list(range(len(ems)))


def _pairs(names, include_autos, autos_only):
    cross, autos = [], []
    for i in names:                       # dict keys double as list indices (sh_sampler.py:137-150)
        for j in names:
            if include_autos:
                if j >= i:
                    cross.append((names[i], names[j]))
            elif autos_only:
                if j == i:
                    cross.append((names[i], names[j]))
            else:
                if j == i:
                    autos.append((names[i], names[j]))
                if j > i:
                    cross.append((names[i], names[j]))
    return cross, autos


def vis_proj_operator_no_rot(freqs, lsts, beams, ant_pos, lmax, nside,
                             latitude=-0.5361913261514378, include_autos=False,
                             autos_only=False, ref_freq=100.0, spectral_idx=0.0,
                             as_torch=False, cross_only=False):
    """Visibility response to each real spherical-harmonic parameter (sh_sampler.py:65-196).
    ``freqs`` in MHz.  Returns (vis_response_2D (Nbl*Nf*Nt, Nm), autos_2D, ell, m), or without
    ``autos_2D`` when ``include_autos`` / ``autos_only``.  ``as_torch=True`` keeps the
    operators on the GPU; ``cross_only=True`` skips the auto-correlation operator and returns
    (vis_response_2D, ell, m)."""
    freqs = np.asarray(freqs, dtype=np.float64)
    lsts = np.atleast_1d(np.asarray(lsts, dtype=np.float64))
    ell, m = _vs.getlm(lmax)
    names = list(ant_pos.keys())
    cross, autos = _pairs(names, include_autos, autos_only)
    V = _vs.sh_voltages(nside, ant_pos, freqs * 1e6, lsts, beams, latitude)
    ymap = _vs.sh_mode_maps(lmax, nside, 1.0, packed=True)
    nm = ymap.shape[1]
    spec = (freqs / ref_freq) ** spectral_idx

    def project(pairs, scale):
        bi = np.array([names.index(p[0]) for p in pairs], dtype=np.int32)
        bj = np.array([names.index(p[1]) for p in pairs], dtype=np.int32)
        out = _vs.sh_project(V, bi, bj, ymap, scale_f=scale)
        return out.reshape(len(pairs) * freqs.size * lsts.size, nm)

    resp = project(cross, spec)
    conv = (lambda t: t) if as_torch else (lambda t: t.cpu().numpy())
    if cross_only:
        return conv(resp), ell, m
    if not autos_only and not include_autos:
        au = project(autos, None)          # the reference applies no spectral factor to autos
        return conv(resp), conv(au), ell, m
    return conv(resp), ell, m


def radiometer_eq(auto_visibilities, ants, delta_time, delta_freq, Nnights=1, include_autos=False):
    """Noise variance of every baseline from the auto-correlations (sh_sampler.py:497-522):
    ``V_ii V_jj / (Nnights * delta_time * delta_freq)`` for the pairs i < j (i <= j with
    ``include_autos``) in antenna order, each a block of ``len(auto_visibilities) // len(ants)``
    samples; the blocks are concatenated.  ``ants`` are the block indices 0 .. Nants-1."""
    autos = np.asarray(auto_visibilities.cpu() if _dev.is_torch(auto_visibilities)
                       else auto_visibilities)
    ants = [int(a) for a in ants]
    indx = autos.shape[0] // len(ants)
    blocks = autos[:indx * (max(ants) + 1)].reshape(max(ants) + 1, indx) if ants else autos[:0]
    first = np.array([i for i in ants for j in ants if (j >= i if include_autos else j > i)],
                     dtype=np.int64)
    second = np.array([j for i in ants for j in ants if (j >= i if include_autos else j > i)],
                      dtype=np.int64)
    if first.size == 0:
        return np.empty((0,))
    return ((blocks[first] * blocks[second]) / (Nnights * delta_time * delta_freq)).reshape(-1)


def alms2healpy(alms, lmax):
    """[Re; Im(m>=1)] -> complex healpy-ordered a_lm (sh_sampler.py:199-232)."""
    alms = np.asarray(alms)
    if alms.ndim == 1:
        split = int((np.size(alms) + (lmax + 1)) / 2)
        return alms[:split] + 1.0j * np.concatenate((np.zeros(lmax + 1), alms[split:]))
    if alms.ndim == 2:
        return np.array([alms2healpy(a, lmax) for a in alms])
    raise ValueError("alms array must either have shape (Nmodes,) or (Nmaps, Nmodes)")


def healpy2alms(healpy_modes):
    """Complex healpy-ordered a_lm -> [Re; Im(m>=1)] (sh_sampler.py:235-266)."""
    healpy_modes = np.asarray(healpy_modes)
    if healpy_modes.ndim == 1:
        lmax = (-3.0 + np.sqrt(1.0 + 8.0 * healpy_modes.size)) / 2.0   # hp.Alm.getlmax
        lmax = int(lmax) if lmax == np.floor(lmax) else -1
        return np.concatenate((healpy_modes.real, healpy_modes.imag[(lmax + 1):]))
    if healpy_modes.ndim == 2:
        return np.array([healpy2alms(a) for a in healpy_modes])
    raise ValueError("Input array must have shape (Nmodes,) or (Nmaps, Nmodes).")


#: when set to a list, (start, end) HIP events recorded on the launch stream around each
#: right-hand-side pass over the response operator are appended -- bench.py's roofline
RHS_EVENTS = None


def _adjoint(R, z, scale=None, add=None):
    m, nm = R.shape
    ws = _dev.workspace(_dev._lib.query("hydra_adjoint_workspace_bytes", m, nm))
    out = _dev.empty((nm,), torch.float64)
    ev = None
    if RHS_EVENTS is not None:
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    _dev.call("hydra_adjoint_R", _dev.ptr(R), m, nm, _dev.ptr(z), _dev.ptr(scale), _dev.ptr(add),
              _dev.ptr(out), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
    if ev is not None:
        ev[1].record()
        RHS_EVENTS.append(ev)
    return out


def _data_term(data, inv_noise_var, omega_n, R):
    """Re(R^H (N^-1 d + N^-1/2 w_n))  (sh_sampler.py:376-381, 423-429)."""
    m = R.shape[0]
    d = _dev.c128(data).reshape(-1)
    w = _dev.f64(inv_noise_var).reshape(-1)
    om = _dev.c128(omega_n).reshape(-1)
    assert d.numel() == m and w.numel() == m and om.numel() == m
    z = _dev.empty((m,), torch.complex128)
    _dev.call("hydra_rhs_rows", _dev.ptr(d), _dev.ptr(None), _dev.ptr(w), _dev.ptr(om),
              _dev.ptr(None), m, _dev.ptr(z), _dev.stream_ptr())
    return _adjoint(R, z)


def construct_rhs_no_rot(data, inv_noise_var, inv_prior_var, omega_0, omega_1, a_0,
                         vis_response):
    """Right-hand side of the SH GCR system (sh_sampler.py:370-386)."""
    R = _dev.c128(vis_response)
    ip = _dev.f64(inv_prior_var)
    term = _data_term(data, inv_noise_var, omega_1, R)
    rhs = term + (ip * _dev.f64(a_0) + torch.sqrt(ip) * _dev.f64(omega_0))
    return _dev.like(vis_response, rhs)


def construct_rhs_no_rot_mpi(comm, data, inv_noise_var, inv_prior_var, omega_a, omega_n, a_0,
                             vis_response, all_ranks=False):
    """Right-hand side from data split over ranks (sh_sampler.py:405-448): ``omega_a`` is
    taken from the root (non-root buffers are overwritten, as in the reference), the data
    terms are summed over ranks, and only the root returns the full vector.
    ``all_ranks=True`` (an addition): the sum is an all-reduce and every rank returns the full
    vector -- what a CG with ``LhsOperator`` needs."""
    myid = comm.Get_rank() if comm is not None else 0
    if comm is not None:
        if myid != 0:
            omega_a *= 0.0
        comm.Bcast(omega_a, root=0)
    R = _dev.c128(vis_response)
    term = _data_term(data, inv_noise_var, omega_n, R)
    if comm is not None and all_ranks:
        comm.allreduce_(term)
        myid = 0
    elif comm is not None:
        comm.reduce_(term, root=0)
        comm.barrier()
    if myid == 0:
        ip = _dev.f64(inv_prior_var)
        rhs = term + ip * _dev.f64(a_0) + torch.sqrt(ip) * _dev.f64(omega_a)
        return _dev.like(vis_response, rhs)
    z = torch.zeros_like(_dev.f64(a_0))
    return _dev.like(vis_response, z)


def _noise_term(a, inv_noise_var, R):
    """Re(R^H N^-1 R a): one pass for y = R a, one for the adjoint."""
    m, nm = R.shape
    w = _dev.f64(inv_noise_var).reshape(-1)
    assert w.numel() == m
    y = _dev.empty((m,), torch.complex128)
    _dev.call("hydra_apply_R", _dev.ptr(R), m, nm, _dev.ptr(a), _dev.ptr(None), _dev.ptr(y),
              _dev.stream_ptr())
    _dev.call("hydra_scale_rows", _dev.ptr(y), _dev.ptr(w), m, _dev.ptr(y), _dev.stream_ptr())
    return _adjoint(R, y)


def apply_lhs_no_rot(a_cr, inv_noise_var, inv_prior_var, vis_response):
    """(R^T N^-1 R + S^-1) a  (sh_sampler.py:389-402)."""
    R = _dev.c128(vis_response)
    a = _dev.f64(a_cr)
    out = _noise_term(a, inv_noise_var, R) + _dev.f64(inv_prior_var) * a
    return _dev.like(a_cr, out)


def apply_lhs_no_rot_mpi(comm, a_cr, inv_noise_var, inv_prior_var, vis_response):
    """Left-hand side with the data split over ranks (sh_sampler.py:451-494): ``a_cr`` is
    taken from the root, the noise terms are summed over ranks; the root returns the result,
    other ranks zeros."""
    myid = comm.Get_rank() if comm is not None else 0
    if comm is not None:
        if myid != 0:
            a_cr *= 0.0
        comm.Bcast(a_cr, root=0)
    R = _dev.c128(vis_response)
    a = _dev.f64(a_cr)
    tot = _noise_term(a, inv_noise_var, R)
    if comm is not None:
        comm.reduce_(tot, root=0)
    if myid == 0:
        return _dev.like(a_cr, tot + _dev.f64(inv_prior_var) * a)
    return _dev.like(a_cr, torch.zeros_like(a))


class LhsOperator:
    """(R^T N^-1 R + S^-1) a with the rows of R split over the ranks of ``comm``, for
    ``linear_solver.cg(None, b, linear_op=op)`` WITHOUT ``comm``: each application is two passes
    over the rank's rows and ONE all-reduce of the (Nm,) result, after which every rank holds
    the same vector and runs the same fused CG step on it.  That one collective replaces, per
    iteration, the Bcast + Reduce of ``apply_lhs_no_rot_mpi`` (sh_sampler.py:465,485) and the
    finished-flag and ``pvec`` broadcasts of the root-driven ``cg`` (linear_solver.py:473,496);
    the CG scalars need no message at all (every rank computes them from identical data)."""

    def __init__(self, comm, inv_noise_var, inv_prior_var, vis_response):
        self.comm = comm
        self.R = _dev.c128(vis_response)
        self.w = _dev.f64(inv_noise_var).reshape(-1)
        self.ip = _dev.f64(inv_prior_var)
        assert self.w.numel() == self.R.shape[0] and self.ip.numel() == self.R.shape[1]

    def __call__(self, a_cr):
        a = _dev.f64(a_cr)
        tot = _noise_term(a, self.w, self.R)
        if self.comm is not None:
            self.comm.allreduce_(tot)
        return _dev.like(a_cr, tot + self.ip * a)


#: form Y^T G Y with torch's GEMM (a library call) instead of this package's matrix-core kernel
#: (HYDRA_SH_OWN_GEMM=0 selects the library call, for comparison)
PROJECT_WITH_LIBRARY_GEMM = os.environ.get("HYDRA_SH_OWN_GEMM", "1") == "0"


def _project_sym(G, ymap):
    """Lower triangle of Y^T G Y for a symmetric G (Nv, Nv) and the mode maps Y (Nv, Nm): the two
    products of the closed-form left-hand side.  Both run on the FP64 matrix cores through
    ``hydra_chol_update_rect`` (C[i][j] -= sum_k PA[i][k] PB[j][k], the update kernel of the dense
    solve: row-major operands with k along the rows, which is how Y^T and the symmetric G lie):
    C1 = -(Y^T G), then A = -(C1 Y) on and below the diagonal.  k is padded to the kernel's 16."""
    if PROJECT_WITH_LIBRARY_GEMM:
        return ymap.T @ (G @ ymap)
    nv, nm = ymap.shape
    nvp = -(-nv // 16) * 16
    dev = G.device
    yt = torch.zeros((nm, nvp), dtype=torch.float64, device=dev)
    yt[:, :nv] = ymap.T
    if nvp != nv or not G.is_contiguous():
        gp = torch.zeros((nv, nvp), dtype=torch.float64, device=dev)
        gp[:, :nv] = G
    else:
        gp = G
    st = _dev.stream_ptr()
    c1 = torch.zeros((nm, nvp), dtype=torch.float64, device=dev)
    _dev.call("hydra_chol_update_rect", _dev.ptr(c1), nvp, _dev.ptr(yt), nvp, _dev.ptr(gp), nvp,
              nm, nv, nv, nvp, st)
    A = torch.zeros((nm, nm), dtype=torch.float64, device=dev)
    _dev.call("hydra_chol_update_rect", _dev.ptr(A), nm, _dev.ptr(c1), nvp, _dev.ptr(yt), nvp,
              nm, nm, 0, nvp, st)
    return A


class StreamedSHOperator:
    """The SH response operator of ``vis_proj_operator_no_rot`` WITHOUT its (Nvis, Nm) matrix in
    memory -- for time axes whose operator does not fit HBM (HERA-127, lmax 40, 128 channels:
    27.5 GB per LST; 40 LSTs = 1.1 TB).  R = diag(spec) P Y factors into the per-pixel operator
    P[(b,f,t),p] = conj(V_i,p) V_j,p (pixel voltages V, vis_simulator.py:657-673) and the real mode
    maps Y (Npix, Nm) (:702-730), so

        R^T z        = Y^T Re(P^H (spec z))        one pass over V (``FactoredProj.adjoint``) + a GEMV
        R a          = spec P (Y a)                a GEMV + one pass over V (``FactoredProj.apply``)
        Re(R^H W R)  = sum over LST blocks of the Gram of the block's operator, regenerated on
                       the matrix cores block by block (``hydra_sh_project_tiled`` + ``hydra_gram``)

    -- nothing of size (Nvis, Nm) persists.  ``time_block`` LSTs are projected at a time.  The
    right-hand side and left-hand side equal ``construct_rhs_no_rot`` / ``lhs_matrix`` on the
    materialised operator up to summation order (sh_sampler.py:370-402)."""

    def __init__(self, freqs, lsts, beams, ant_pos, lmax, nside,
                 latitude=-0.5361913261514378, ref_freq=100.0, spectral_idx=0.0, time_block=2):
        from .ptsrc_sampler import FactoredProj
        self.freqs = np.asarray(freqs, dtype=np.float64)
        self.lsts = np.atleast_1d(np.asarray(lsts, dtype=np.float64))
        self.ell, self.m = _vs.getlm(lmax)
        names = list(ant_pos.keys())
        cross, _ = _pairs(names, False, False)
        self.bl_i = np.array([names.index(p[0]) for p in cross], dtype=np.int32)
        self.bl_j = np.array([names.index(p[1]) for p in cross], dtype=np.int32)
        self.nbl, self.nf, self.nt = len(cross), self.freqs.size, self.lsts.size
        self.ymap = _vs.sh_mode_maps(lmax, nside, 1.0, packed=True)          # (Npix, Nm)
        self.ymap_t = self.ymap.T.contiguous()                               # (Nm, Npix)
        self.npix, self.nm = self.ymap.shape
        # the spectral factor (f / ref)^beta of sh_sampler.py:164-168 rides on the pixel fluxes:
        # V carries its square root, conj(V_i) V_j the factor itself
        self.spec = (self.freqs / ref_freq) ** spectral_idx
        self.time_block = max(1, int(time_block))
        self._args = (nside, ant_pos, beams, latitude)
        # pixel voltages of all LSTs stay resident: (Nf, Nt, Nant, Npix) c128, (Nant-1)/2 x Nm/Npix
        # times smaller than the operator
        self.V = _vs.sh_voltages(nside, ant_pos, self.freqs * 1e6, self.lsts, beams, latitude,
                                 flux_f=self.spec)
        self.fac = FactoredProj(self.V, self.bl_i, self.bl_j)
        self.shape = (self.nbl * self.nf * self.nt, self.nm)

    def _block_response(self, t0, t1):
        """The operator of LSTs [t0, t1) as a matrix, rows in (b, f, t) order of that block."""
        nside, ant_pos, beams, latitude = self._args
        Vb = _vs.sh_voltages(nside, ant_pos, self.freqs * 1e6, self.lsts[t0:t1], beams, latitude,
                             flux_f=self.spec)
        out = _vs.sh_project(Vb, self.bl_i, self.bl_j, self.ymap)
        return out.reshape(self.nbl * self.nf * (t1 - t0), self.nm)

    #: bytes of scaled voltages the closed-form left-hand side keeps at a time (a frequency chunk)
    LHS_CHUNK_BYTES = 16 << 30

    def _lhs_closed_form(self, w):
        """Re(R^H N^-1 R) for weights ``w`` (Nbl, Nf, Nt) that do not depend on the baseline -- or
        do so only through flagged / down-weighted samples -- through pixel space:
        R = P Y, so Re(R^H W R) = Y^T Re(P^H W P) Y, and for a complete array the pixel Gram
        Re(P^H W P) has the closed form in the antenna voltages of csrc/gram.hip
        (``gram_uniform_kernel``: a rank-Nant update per (f,t) where the operator's Gram is a
        rank-Nbl one) -- 3.8 s -> 0.5 s at config 3, no block of the operator is ever formed.
        Only the pixels above the horizon at some LST of the run enter it.

        Flags (sh_sampler.py:389-402 accepts any ``inv_noise_var``; real data is always flagged):
        Re(P^H W P) = G_closed(w_ref) - sum over flagged (b,f,t) of (w_ref - w) conj(P_b) P_b with
        w_ref the common weight of the (f,t) sample -- the correction of the point-source head
        (``FactoredProj.gram``: ``hydra_flagged_rows`` + ``hydra_gram`` + ``hydra_gram_subtract``)
        in pixel space, one frequency chunk at a time so the flagged rows of a chunk (rows x
        visible pixels) stay a few GB.  Returns None when some chunk's weights are neither."""
        nf, nt, nant, npix = self.V.shape
        # pixels that are above the horizon at some LST of the run (the others have zero voltage
        # at every (f,t): exact zeros in the Gram); the Gram's cost goes with their number squared
        if getattr(self, "_visible", None) is None:
            mask = torch.zeros(npix, dtype=torch.bool, device=self.V.device)
            for j in range(nf):
                mask |= (self.V[j] != 0).any(dim=0).any(dim=0)
            self._visible = mask.nonzero().squeeze(1)
        idx = self._visible if self._visible.numel() < 0.9 * npix else None
        nv = npix if idx is None else int(idx.numel())
        ymap = self.ymap if idx is None else self.ymap.index_select(0, idx)
        per_chan = nt * (nant + 1) * nv * 16 * (1 if idx is None else 2)
        step = max(1, int(self.LHS_CHUNK_BYTES // per_chan))
        Gp = None
        self.last_lhs_flagged_rows = 0
        for j0 in range(0, nf, step):
            j1 = min(nf, j0 + step)
            sub = self.fac.channels(j0, j1)
            if idx is not None:
                sub.V = self.V[j0:j1].index_select(3, idx)
                sub.shape = tuple(sub.shape[:3]) + (nv,)
            wc = w[:, j0:j1].contiguous()
            plan = sub._noise_plan(wc)
            if plan[0] == "general":
                return None
            G = sub.gram(None, wc, plan=plan)
            self.last_lhs_flagged_rows += int(getattr(sub, "last_gram_flagged_rows", 0))
            Gp = G if Gp is None else Gp.add_(G)
            del sub
        self.fac.last_gram_kernel = "gram_uniform_kernel"
        A = _project_sym(Gp, ymap)
        # only the lower triangle is formed; the solvers read half of a matrix marked symmetric
        return torch.tril(A) + torch.tril(A, -1).T

    def lhs_matrix(self, inv_noise_var, inv_prior_var, comm=None, closed_form=True):
        """A = Re(R^H N^-1 R) + diag(S^-1) (``lhs_matrix`` of the module on an operator that is
        never whole): in closed form through pixel space when the array is complete and the noise
        weights do not depend on the baseline (``closed_form=False`` forces the other path),
        otherwise accumulated over LST blocks of the regenerated operator."""
        w = _dev.f64(inv_noise_var).reshape(self.nbl, self.nf, self.nt)
        ip = _dev.f64(inv_prior_var)
        A = None
        if closed_form and self.fac.complete and self.fac.allow_uniform and self.V.shape[2] > 1:
            A = self._lhs_closed_form(w)
        if A is None:
            A = self._lhs_blocks(w)
        if comm is not None:
            comm.allreduce_sym_(A)
        _dev.call("hydra_prior_epilogue", _dev.ptr(A), self.nm, _dev.ptr(None), _dev.ptr(ip),
                  _dev.stream_ptr())
        A.hydra_symmetric = True
        return A

    def _lhs_blocks(self, w):
        self.fac.last_gram_kernel = "gram_kernel"
        A = torch.zeros((self.nm, self.nm), dtype=torch.float64, device=w.device)
        G = _dev.empty((self.nm, self.nm), torch.float64)
        for t0 in range(0, self.nt, self.time_block):
            t1 = min(self.nt, t0 + self.time_block)
            R = self._block_response(t0, t1)
            wb = w[:, :, t0:t1].contiguous().reshape(-1)
            rows = torch.complex(torch.sqrt(wb), torch.zeros_like(wb))
            ws = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", R.shape[0], self.nm))
            _dev.call("hydra_gram", _dev.ptr(R), _dev.ptr(rows), R.shape[0], self.nm,
                      _dev.ptr(None), _dev.ptr(None), _dev.ptr(G), _dev.ptr(ws), ws.numel(),
                      _dev.stream_ptr())
            A += G
            del R, ws
        return A

    def _rows(self, x):
        return _dev.c128(x).reshape(self.nbl, self.nf, self.nt)

    def adjoint(self, z):
        """Re(R^H z) (Nm,) for z like the visibilities."""
        t_pix = self.fac.adjoint(self._rows(z).contiguous())
        out = _dev.empty((self.nm,), torch.float64)
        _dev.call("hydra_gemv_rows", _dev.ptr(self.ymap_t), self.nm, self.npix, _dev.ptr(t_pix),
                  _dev.ptr(out), 0, _dev.stream_ptr())
        return out

    def apply(self, a):
        """R a (Nbl, Nf, Nt) complex -- the model update of the driver (example.py:1065)."""
        sky = _dev.empty((self.npix,), torch.float64)
        ad = _dev.f64(a)
        _dev.call("hydra_gemv_rows", _dev.ptr(self.ymap), self.npix, self.nm, _dev.ptr(ad),
                  _dev.ptr(sky), 0, _dev.stream_ptr())
        return self.fac.apply(sky)

    def construct_rhs(self, data, inv_noise_var, inv_prior_var, omega_0, omega_1, a_0, comm=None):
        """``construct_rhs_no_rot`` (sh_sampler.py:370-386); with ``comm`` the data terms of the
        ranks' chunks are all-reduced and every rank returns the full vector."""
        d, w, om = self._rows(data), _dev.f64(inv_noise_var).reshape(-1), self._rows(omega_1)
        m = d.numel()
        z = _dev.empty((m,), torch.complex128)
        _dev.call("hydra_rhs_rows", _dev.ptr(d), _dev.ptr(None), _dev.ptr(w), _dev.ptr(om),
                  _dev.ptr(None), m, _dev.ptr(z), _dev.stream_ptr())
        term = self.adjoint(z)
        if comm is not None:
            comm.allreduce_(term)
        ip = _dev.f64(inv_prior_var)
        return term + (ip * _dev.f64(a_0) + torch.sqrt(ip) * _dev.f64(omega_0))

    def apply_lhs(self, a_cr, inv_noise_var, inv_prior_var):
        """(R^T N^-1 R + S^-1) a, matrix-free (``apply_lhs_no_rot``, sh_sampler.py:389-402)."""
        a = _dev.f64(a_cr)
        y = self.apply(a)
        w = _dev.f64(inv_noise_var).reshape(-1)
        _dev.call("hydra_scale_rows", _dev.ptr(y), _dev.ptr(w), y.numel(), _dev.ptr(y),
                  _dev.stream_ptr())
        return self.adjoint(y) + _dev.f64(inv_prior_var) * a


def lhs_matrix(inv_noise_var, inv_prior_var, vis_response, comm=None):
    """The SH left-hand-side operator as an explicit matrix
    A = Re(R^H N^-1 R) + diag(S^-1), built once on the FP64 matrix cores (csrc/gram.hip), so
    that ``linear_solver.cg(A, b)`` replaces the per-iteration Gram of the reference
    (sh_sampler.py:393-398).  With ``comm`` the Grams of the ranks' chunks are summed."""
    R = _dev.c128(vis_response)
    m, nm = R.shape
    w = _dev.f64(inv_noise_var).reshape(-1)
    rows = torch.complex(torch.sqrt(w), torch.zeros_like(w))
    ws = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", m, nm))
    A = _dev.empty((nm, nm), torch.float64)
    ip = _dev.f64(inv_prior_var)
    _dev.call("hydra_gram", _dev.ptr(R), _dev.ptr(rows), m, nm, _dev.ptr(None),
              _dev.ptr(ip if comm is None else None), _dev.ptr(A), _dev.ptr(ws), ws.numel(),
              _dev.stream_ptr())
    if comm is not None:
        comm.allreduce_sym_(A)
        _dev.call("hydra_prior_epilogue", _dev.ptr(A), nm, _dev.ptr(None), _dev.ptr(ip),
                  _dev.stream_ptr())
    if _dev.is_torch(vis_response):
        A.hydra_symmetric = True      # mirrored exactly by the Gram finish kernel
    return _dev.like(vis_response, A)
