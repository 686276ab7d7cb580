"""
Point-source amplitude head on MI355X -- drop-in for ``hydra.ptsrc_sampler``
(reference: hydra/ptsrc_sampler.py).

    calc_proj_operator(...)  -> R (Nbl, Nf, Nt, Nsrc)          ptsrc_sampler.py:143-215
    precompute_mpi(...)      -> (A, b) of the GCR system        ptsrc_sampler.py:17-140
        A = 1 + S^1/2 Re(R^H N^-1 R) S^1/2
        b = S^1/2 Re(R^H (N^-1 r + N^-1/2 w_n)) + w_a,  r = d - R 1

Extensions that do not change the reference call signatures:
  * ``calc_proj_operator(..., materialize=False)`` returns a ``FactoredProj`` (antenna
    voltages + baseline tables) that ``precompute_mpi`` accepts as ``proj_chunk``; the
    operator is then generated inside the kernels and never stored (at BASELINE config 2
    the materialised R would be 273 GB).
  * ``precompute_mpi(..., omega_n=, omega_a=)`` inject the fluctuation vectors; when
    omitted they are drawn from the global numpy RNG in the reference's order
    (ptsrc_sampler.py:109-116,138) so a seeded run reproduces the reference.
"""
import numpy as np
import torch

from . import _dev, _tables
from .vis_simulator import voltages_for


#: when set to a list, ``precompute_mpi`` appends a (start, end) pair of HIP events (recorded on
#: the stream the kernels run on) around each Gram launch -- used by bench.py for the roofline
GRAM_EVENTS = None


def _timed_gram(fn):
    """Run ``fn`` (one Gram entry point); with GRAM_EVENTS set, have the library record a pair
    of HIP events on the launch stream right around the main Gram kernel
    (hydra_profile_events), so the helper kernels of the entry point are not in the bracket."""
    if GRAM_EVENTS is None:
        return fn()
    import ctypes
    start = torch.cuda.Event(enable_timing=True)
    end = torch.cuda.Event(enable_timing=True)
    start.record()          # creates the underlying hipEvent_t handles
    end.record()
    _dev.call("hydra_profile_events", ctypes.c_void_p(start.cuda_event),
              ctypes.c_void_p(end.cuda_event))
    try:
        out = fn()
    finally:
        _dev.call("hydra_profile_events", ctypes.c_void_p(0), ctypes.c_void_p(0))
    GRAM_EVENTS.append((start, end))
    return out


class FactoredProj:
    """The projection operator in factored form: R[(b,f,t),s] = conj(V[f,t,i_b,s]) V[f,t,j_b,s].

    ``V`` (Nf,Nt,Nant,Ns) complex128 GPU tensor; ``bl_i``/``bl_j`` antenna index per baseline.
    ``shape`` is the shape the materialised operator would have."""

    def __init__(self, V, bl_i, bl_j):
        self.V = V
        self.bl_i_host = np.ascontiguousarray(bl_i, dtype=np.int32)
        self.bl_j_host = np.ascontiguousarray(bl_j, dtype=np.int32)
        self.bl_i = _dev.i32(self.bl_i_host)
        self.bl_j = _dev.i32(self.bl_j_host)
        nf, nt, nant, ns = V.shape
        self.shape = (self.bl_i_host.size, nf, nt, ns)
        self.seg_host, self.rowtab_host = _tables.build_segments(bl_i, bl_j, nant)
        self.seg = _dev.i32(self.seg_host)
        self.rowtab = _dev.i32(self.rowtab_host)
        #: every antenna pair i<j exactly once -> the closed-form Gram applies (csrc/gram.hip)
        self.complete = _tables.is_complete_array(self.bl_i_host, self.bl_j_host, nant)
        #: every pair listed once with i <= j: R^H z may run in antenna space for any array size
        key = self.bl_i_host.astype(np.int64) * nant + self.bl_j_host
        self.unique_upper = bool(np.all(self.bl_i_host <= self.bl_j_host)
                                 and np.unique(key).size == key.size)
        #: set False to force the general (per-baseline weights) Gram kernel
        self.allow_uniform = True
        #: set False to force the pair-loop R^H z for arrays of more than 64 antennas
        self.antspace_adjoint = True
        #: |V[f,t,a,s]| independent of the antenna (set by the simulator for a single beam model)
        self.same_beam = bool(getattr(V, "hydra_same_beam", False))

    def channel(self, j):
        """The operator of frequency channel ``j`` alone (a view of V, tables shared) -- for
        heads whose channels do not talk to each other (cosmo_sampler.py:276-284)."""
        return self.channels(j, j + 1)

    def channels(self, j0, j1):
        """The operator of frequency channels [j0, j1) (a view of V, tables shared)."""
        sub = object.__new__(FactoredProj)
        sub.__dict__.update(self.__dict__)
        sub.__dict__.pop("_plan_cache", None)
        sub.__dict__.pop("_chan_plan_cache", None)
        sub.V = self.V[j0:j1]
        sub.shape = (self.shape[0], j1 - j0) + tuple(self.shape[2:])
        return sub

    @property
    def nbytes_materialised(self):
        return int(np.prod(self.shape)) * 16

    def materialize(self):
        """R (Nbl,Nf,Nt,Ns) complex128 GPU tensor (csrc/response.hip)."""
        nf, nt, nant, ns = self.V.shape
        nbl = self.shape[0]
        R = _dev.empty(self.shape, torch.complex128)
        _dev.call("hydra_response", _dev.ptr(self.V), nf, nt, nant, ns, _dev.ptr(self.bl_i),
                  _dev.ptr(self.bl_j), nbl, _dev.ptr(R), _dev.stream_ptr())
        return R

    def apply(self, x, rowmul=None):
        """(Nbl,Nf,Nt) = rowmul * (R @ x) without materialising R."""
        nf, nt, nant, ns = self.V.shape
        nbl = self.shape[0]
        out = _dev.empty((nbl, nf, nt), torch.complex128)
        xd = None if x is None else _dev.f64(x)          # None = a vector of ones
        ws = _dev.workspace(_dev._lib.query("hydra_apply_V_workspace_bytes", nf, nt, nant, ns))
        _dev.call("hydra_apply_V", _dev.ptr(self.V), nf, nt, nant, ns, _dev.ptr(self.bl_i),
                  _dev.ptr(self.bl_j), nbl, _dev.ptr(xd), _dev.ptr(rowmul),
                  _dev.ptr(out), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        return out

    def adjoint(self, z, scale=None, add=None):
        """(Ns,) = scale * Re(R^H z) + add without materialising R."""
        nf, nt, nant, ns = self.V.shape
        nbl = self.shape[0]
        out = _dev.empty((ns,), torch.float64)
        if (64 < nant <= 1024 and self.unique_upper and self.antspace_adjoint
                and nant * ns * 16 < (1 << 32)):
            # more than 64 antennas: antenna space on the matrix cores (csrc/gemv.hip,
            # adjoint_V_big_kernel; adjoint_V_sb_kernel beyond 160) instead of the pair loop on
            # the vector units
            nbytes = _dev._lib.query("hydra_adjoint_V_antspace_workspace_bytes", nf, nt, nant, nbl,
                                     ns)
            ws = _dev.workspace(nbytes)
            _dev.call("hydra_adjoint_V_antspace", _dev.ptr(self.V), nf, nt, nant, ns,
                      _dev.ptr(self.bl_i), _dev.ptr(self.bl_j), nbl, _dev.ptr(z), _dev.ptr(scale),
                      _dev.ptr(add), _dev.ptr(out), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
            return out
        nbytes = _dev._lib.query("hydra_adjoint_V_workspace_bytes", nf, nt, nbl, ns)
        ws = _dev.workspace(nbytes)
        _dev.call("hydra_adjoint_V", _dev.ptr(self.V), nf, nt, nant, ns, _dev.ptr(self.bl_i),
                  _dev.ptr(self.bl_j), nbl, _dev.ptr(z), _dev.ptr(scale), _dev.ptr(add),
                  _dev.ptr(out), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        return out

    #: the closed form + flagged-row correction is taken while the deviating samples are at most
    #: this fraction of all rows and their materialised rows fit this many bytes
    FLAG_MAX_FRACTION = 0.10
    FLAG_MAX_BYTES = 24 << 30

    def _noise_plan(self, inv_noise):
        """How the noise weights enter the Gram: ("uniform", w_ft) -- the same for all baselines
        of an (f,t) sample; ("flagged", w_ref, idx) -- a few samples (flags) below the common
        weight w_ref[f,t], idx their sorted flat indices; ("general",) otherwise.
        The analysis reads one counter back to the host, so it is done once per weight tensor and
        cached (key: storage address + torch's in-place version counter, the tensor kept alive);
        the Gibbs loop passes the same inv_noise_var every iteration (example.py:878-889).
        The version counter only sees torch operations: a weight tensor rewritten through its raw
        pointer (a kernel of one's own, this package's C ABI) must be followed by
        ``invalidate_noise_plan()``, or be passed as a new tensor."""
        nf, nt, nant, ns = self.V.shape
        nbl = self.shape[0]
        key = (inv_noise.data_ptr(), inv_noise._version, tuple(inv_noise.shape))
        hit = getattr(self, "_plan_cache", None)
        if hit is not None and hit[0] == key and hit[1] is inv_noise:
            return hit[2]
        m = nbl * nf * nt
        cap = int(min(self.FLAG_MAX_FRACTION * m, self.FLAG_MAX_BYTES // (16 * ns)))
        w_ref = _dev.empty((nf * nt,), torch.float64)
        idx = torch.empty(max(cap, 1), dtype=torch.int64, device=w_ref.device)
        count = torch.zeros(1, dtype=torch.int64, device=w_ref.device)
        _dev.call("hydra_noise_deviations", _dev.ptr(inv_noise), nbl, nf, nt, _dev.ptr(w_ref),
                  _dev.ptr(idx), cap, _dev.ptr(count), _dev.stream_ptr())
        n_dev = int(count.item())
        if n_dev == 0:
            plan = ("uniform", w_ref)
        elif n_dev <= cap:
            plan = ("flagged", w_ref, torch.sort(idx[:n_dev]).values.contiguous())
        else:
            plan = ("general",)
        self._plan_cache = (key, inv_noise, plan)
        return plan

    def invalidate_noise_plan(self):
        """Forget the cached analysis of the noise weights (see ``_noise_plan``)."""
        self.__dict__.pop("_plan_cache", None)
        self.__dict__.pop("_chan_plan_cache", None)

    def channel_plans(self, inv_noise):
        """``_noise_plan`` of every frequency channel taken alone (for ``channel(j).gram(...,
        plan=plans[j])``), derived from ONE analysis of the whole weight tensor and cached with
        it -- a head that builds one Gram per channel (cosmo_sampler.py:276-284) would otherwise
        read a counter back per channel and sample."""
        plan = self._noise_plan(inv_noise)
        hit = getattr(self, "_chan_plan_cache", None)
        if hit is not None and hit[0] is plan:
            return hit[1]
        nf, nt = self.V.shape[0], self.V.shape[1]
        if plan[0] == "general":
            plans = [plan] * nf
        elif plan[0] == "uniform":
            plans = [("uniform", plan[1][j * nt:(j + 1) * nt]) for j in range(nf)]
        else:
            idx = plan[2]
            b, rem = idx // (nf * nt), idx % (nf * nt)
            f, t = rem // nt, rem % nt
            plans = []
            for j in range(nf):
                sel = f == j
                wr = plan[1][j * nt:(j + 1) * nt]
                ij = (b[sel] * nt + t[sel]).contiguous()
                plans.append(("flagged", wr, ij) if ij.numel() else ("uniform", wr))
        self._chan_plan_cache = (plan, plans)
        return plans

    def gram(self, gains, inv_noise, prior_std=None, add_diag=None, plan=None):
        """Re(R'^H N^-1 R') on the FP64 matrix cores, R' = diag(g_i conj g_j) R.

        General form: rank-Nbl update per (f,t) with per-baseline weights (gram_fused_kernel).
        If the array is complete and the weights do not depend on the baseline, the exact
        closed form in the antenna voltages is used instead (gram_uniform_kernel); if a few
        samples are flagged (weight below the common one of their (f,t)), the closed form is
        corrected by a Gram over the materialised rows of those samples."""
        nf, nt, nant, ns = self.V.shape
        nbl = self.shape[0]
        st = _dev.stream_ptr()
        if self.complete and self.allow_uniform and nant > 1:
            if plan is None:
                plan = ("uniform", None) if inv_noise is None else self._noise_plan(inv_noise)
            if plan[0] != "general":
                flagged = plan[0] == "flagged"
                nbytes = _dev._lib.query("hydra_gram_uniform_workspace_bytes", nf, nt, nant, ns)
                ws = _dev.workspace(nbytes)
                G = _dev.empty((ns, ns), torch.float64)
                _dev.call("hydra_gram_uniform", _dev.ptr(self.V), nf, nt, nant, ns,
                          _dev.ptr(gains), _dev.ptr(plan[1]), int(self.same_beam),
                          _dev.ptr(None if flagged else prior_std),
                          _dev.ptr(None if flagged else add_diag), _dev.ptr(G), _dev.ptr(ws),
                          ws.numel(), st)
                self.last_gram_kernel = "gram_uniform_kernel"
                self.last_gram_flagged_rows = 0
                if flagged:
                    del ws
                    # a timing bracket set by _timed_gram stays on the closed-form kernel
                    _dev.call("hydra_profile_events", _dev.ptr(None), _dev.ptr(None))
                    idx = plan[2]
                    nrows = int(idx.numel())
                    R = _dev.empty((nrows, ns), torch.complex128)
                    rows = _dev.empty((nrows,), torch.complex128)
                    _dev.call("hydra_flagged_rows", _dev.ptr(self.V), nf, nt, nant, ns,
                              _dev.ptr(gains), _dev.ptr(inv_noise), _dev.ptr(plan[1]),
                              _dev.ptr(self.bl_i), _dev.ptr(self.bl_j), nbl, _dev.ptr(idx), nrows,
                              _dev.ptr(R), _dev.ptr(rows), st)
                    ws2 = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", nrows, ns))
                    C = _dev.empty((ns, ns), torch.float64)
                    _dev.call("hydra_gram", _dev.ptr(R), _dev.ptr(rows), nrows, ns, _dev.ptr(None),
                              _dev.ptr(None), _dev.ptr(C), _dev.ptr(ws2), ws2.numel(), st)
                    _dev.call("hydra_gram_subtract", _dev.ptr(G), _dev.ptr(C), ns,
                              _dev.ptr(prior_std), _dev.ptr(add_diag), st)
                    self.last_gram_flagged_rows = nrows
                return G
        self.last_gram_kernel = "gram_fused_kernel"
        nbytes = _dev._lib.query("hydra_gram_fused_workspace_bytes", nf, nt, nbl, ns)
        ws = _dev.workspace(nbytes)
        G = _dev.empty((ns, ns), torch.float64)
        _dev.call("hydra_gram_fused", _dev.ptr(self.V), nf, nt, nant, ns, _dev.ptr(gains),
                  _dev.ptr(inv_noise), nbl, _dev.host_ptr(self.seg_host),
                  _dev.host_ptr(self.rowtab_host), _dev.ptr(self.seg), _dev.ptr(self.rowtab),
                  int(self.seg_host.shape[0]), int(self.rowtab_host.shape[0]),
                  _dev.ptr(prior_std), _dev.ptr(add_diag), _dev.ptr(G), _dev.ptr(ws), ws.numel(),
                  st)
        return G


class SystemCache:
    """The point-source GCR operator kept between Gibbs samples.

    ``A = 1 + S^1/2 Re(R^H N^-1 R) S^1/2`` and the model ``g_i g_j^* (R 1)`` depend on the gains,
    the noise weights and the prior (ptsrc_sampler.py:48-106) -- not on the data or the
    fluctuation draws -- yet the reference's driver rebuilds them in every iteration
    (example.py:878).  Pass one ``SystemCache`` as ``precompute_mpi(..., cache=)``: while gains,
    ``inv_noise_var`` and ``amp_prior_std`` hold the values of the last call (compared by
    content on the GPU: the driver forms a new gain tensor every iteration, example.py:868-889),
    a sample costs the right-hand side alone, and ``cache.solve(b)`` keeps the Cholesky factor as
    well, so the dense solve of example.py:944 becomes two triangular sweeps.  Both return bit
    for bit what the rebuilding path returns (``tests/test_gpu_parity.py``).

    After the first ``solve`` the LOWER triangle of ``A`` holds the factor ``L`` (in place: the
    100 000-source operator is 80 GB); the strict upper triangle and ``diag`` still hold the
    operator, and ``matvec`` multiplies by it in either state."""

    def __init__(self):
        self.hits = self.misses = 0
        self.clear()

    def clear(self):
        self.key = None
        self.A = self.diag = self.gg = self.model = None
        self.factor = None
        self.rows = False            # A is this rank's row block (precompute_mpi(shard_rows=True))
        self.comm = None

    def _match(self, comm, proj, gains, w, sigma, shard_rows):
        k = self.key
        hit = (k is not None and k[0] is proj and k[4] == bool(shard_rows)
               and k[5] == getattr(proj, "_version", None)
               and k[1].shape == gains.shape and k[2].shape == w.shape
               and k[3].shape == sigma.shape
               and torch.equal(k[1], gains) and torch.equal(k[2], w) and torch.equal(k[3], sigma))
        if comm is not None:         # every rank takes the same branch: its collectives differ
            hit = comm.allreduce(1.0 if hit else 0.0, op="min") == 1.0
        if hit:
            self.hits += 1
        else:
            self.misses += 1
            self.clear()             # frees the old operator before the new one is built
        return hit

    def _store(self, comm, proj, gains, w, sigma, shard_rows, A, gg, model, keep_A):
        self.key = (proj, gains.clone(), w.clone(), sigma.clone(), bool(shard_rows),
                    getattr(proj, "_version", None))
        self.A = A if keep_A else None
        self.gg, self.model = gg, model
        self.rows, self.comm = bool(shard_rows), comm
        self.factor = self.diag = None

    @property
    def factored(self):
        return self.factor is not None

    def solve(self, bvec):
        """x = A^-1 b: ``linear_solver.solve_dense(A, b)`` (``solve_dense_rows`` for an operator
        sharded by rows) with the factorisation done on the first call after a rebuild only."""
        from . import linear_solver
        assert self.A is not None, "SystemCache.solve: no operator (call precompute_mpi(cache=) first)"
        if self.factor is None:
            if self.rows:
                lay = self.comm.row_layout(self.A.shape[1])
                rows = self.comm._rows_dev(lay, lay.rank, self.A.device)
                self.diag = self.A[torch.arange(rows.numel(), device=rows.device), rows].clone()
                self.factor = linear_solver.factor_dense_rows(self.comm, self.A, overwrite_a=True)
            else:
                self.diag = self.A.diagonal().clone()
                self.factor = linear_solver.cho_factor(self.A, overwrite_a=True,
                                                       inverse_factor=False)
                self.A.hydra_symmetric = False      # its lower triangle is L from here on
        if self.rows:
            return self.factor.solve(bvec)
        return linear_solver.cho_solve(self.factor, bvec)

    def matvec(self, x):
        """A x, whether or not ``solve`` has overwritten the lower triangle with the factor (then
        from the strict upper triangle, its mirror image and the saved diagonal:
        ``hydra_symv_upper``).  The driver's residual check; every rank gets the whole product."""
        xd = _dev.f64(x).contiguous()
        n = self.A.shape[1]
        y = torch.zeros(n, dtype=torch.float64, device=xd.device)
        nloc = self.A.shape[0]
        if self.rows:
            lay = self.comm.row_layout(n)
            rows = self.comm._rows_dev(lay, lay.rank, xd.device)
        else:
            rows = torch.arange(n, dtype=torch.int64, device=xd.device)
        if self.factor is None or (not self.rows and self.factor.L is not self.A):
            if nloc:
                part = torch.empty(nloc, dtype=torch.float64, device=xd.device)
                _dev.call("hydra_gemv_rows", _dev.ptr(self.A), nloc, n, _dev.ptr(xd), _dev.ptr(part), 0,
                          _dev.stream_ptr())
                y[rows] = part
        elif nloc:
            rows = rows.contiguous()
            _dev.call("hydra_symv_upper", _dev.ptr(self.A), n, nloc, n, _dev.ptr(rows),
                      _dev.ptr(self.diag), _dev.ptr(xd), _dev.ptr(y), _dev.stream_ptr())
        if self.rows:
            self.comm.allreduce_(y)
        return y


def calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, beams,
                       latitude=-0.5361913261514378, materialize=True):
    """Visibility vector of each point source per (baseline, frequency, time): the projection
    operator from source amplitudes to visibilities, gains not included
    (ptsrc_sampler.py:143-215).  ``freqs`` in MHz, ``times`` = LSTs in rad.

    Returns (Nbl, Nf, Nt, Nsrc) complex128 (numpy, or a GPU tensor if ``fluxes`` is one), or a
    ``FactoredProj`` when ``materialize=False``."""
    freqs_h = np.asarray(freqs.cpu() if _dev.is_torch(freqs) else freqs, dtype=np.float64)
    V = voltages_for(ant_pos, fluxes, ra, dec, freqs_h * 1e6, times, beams, latitude)
    names = list(ant_pos.keys())
    bl_i = np.array([names.index(bl[0]) for bl in antpairs], dtype=np.int32)
    bl_j = np.array([names.index(bl[1]) for bl in antpairs], dtype=np.int32)
    # The reference only fills vis[..., i, j>=i, :] (vis_simulator.py:281-289), so a pair listed
    # with idx1 > idx2 picks up zeros (ptsrc_sampler.py:209-213).
    flipped = bl_i > bl_j
    proj = FactoredProj(V, bl_i, bl_j)
    if not materialize:
        if flipped.any():
            raise ValueError("antpairs must list each pair in antenna-dict order (idx1 <= idx2) "
                             "for the factored operator")
        return proj
    R = proj.materialize()
    if flipped.any():
        R[torch.as_tensor(np.nonzero(flipped)[0], device=R.device)] = 0
    return _dev.like(fluxes, R)


def _draw_omega_n(shape, realisation):
    # ptsrc_sampler.py:109-116: Re then Im from the global RNG
    sw = 1.0 if realisation else 0.0
    return sw * (1.0 * np.random.randn(*shape) + 1.0j * np.random.randn(*shape)) / np.sqrt(2.0)


def precompute_mpi(comm, ants, antpairs, freq_chunk, time_chunk, proj_chunk, data_chunk,
                   inv_noise_var_chunk, gain_chunk, amp_prior_std, realisation=True,
                   omega_n=None, omega_a=None, shard_rows=False, cache=None, replicate=False):
    """Linear operator and right-hand side of the point-source GCR system
    (ptsrc_sampler.py:17-140).  With ``comm`` (``hydra_amd._dist.Comm``) every rank passes its
    (freq, time) chunk and the partial Gram / RHS are summed over ranks; only the root's
    return value is the full system, as in the reference.

    ``shard_rows=True`` (with ``comm``): the sum leaves the operator SHARDED BY ROWS instead --
    every rank returns its rows ``comm.row_layout(Ns).rows()`` of ``A`` (block-cyclic, see
    ``_dist.RowLayout``) and the full ``b``; solve with ``linear_solver.solve_dense_rows`` or
    ``linear_solver.cg(None, b, linear_op=linear_solver.RowShardedOperator(comm, A_rows, Ns))``.
    For source counts whose matrix one GPU holds but should not multiply alone (config 5).

    ``replicate=True`` (with ``comm``; SURVEY 8(e), small systems): the sums are all-reduces and
    EVERY rank returns the full system, bit for bit the same, so that each solves it for itself and
    no broadcast of the solution follows (``cg(comm=...)`` of the reference also leaves x on every
    rank, linear_solver.py:506-508).  The root's omega_a joins its partial right-hand side before
    the sum.

    ``cache`` (a ``SystemCache``): keep the operator between calls and rebuild it only when the
    gains, the noise weights or the prior have changed; the returned ``A`` is then the cache's
    (do not modify it; after ``cache.solve`` its lower triangle holds the Cholesky factor)."""
    myid = comm.Get_rank() if comm is not None else 0
    bl_i_h, bl_j_h = _tables.baseline_indices(ants, antpairs)
    nbl = len(antpairs)
    nf, nt = int(np.size(freq_chunk)), int(np.size(time_chunk))
    assert tuple(data_chunk.shape) == (nbl, nf, nt)
    assert tuple(data_chunk.shape) == tuple(inv_noise_var_chunk.shape)
    factored = isinstance(proj_chunk, FactoredProj)
    ns = int(proj_chunk.shape[-1])
    assert tuple(proj_chunk.shape[:3]) == (nbl, nf, nt)
    ref = data_chunk

    data = _dev.c128(data_chunk)
    w = _dev.f64(inv_noise_var_chunk)
    gains = _dev.c128(gain_chunk)
    sigma = _dev.f64(amp_prior_std)
    assert gains.shape[1:] == (nf, nt)
    nant = gains.shape[0]
    assert max(bl_i_h.max(), bl_j_h.max()) < nant
    bl_i, bl_j = _dev.i32(bl_i_h), _dev.i32(bl_j_h)
    st = _dev.stream_ptr()
    m = nbl * nf * nt

    # fluctuation vectors: injected, or the reference's draws in the reference's order
    if omega_n is None:
        omega_n = _draw_omega_n((nbl, nf, nt), realisation)
    om_n = _dev.c128(omega_n)
    assert tuple(om_n.shape) == (nbl, nf, nt)

    hit = cache is not None and cache._match(comm, proj_chunk, gains, w, sigma, shard_rows)
    prior_here = None if comm is not None else sigma
    A = None
    if hit:
        gg, model = cache.gg, cache.model
        if not factored:
            R = _dev.c128(proj_chunk).reshape(m, ns)
    elif factored:
        gg = _dev.empty((nbl, nf, nt), torch.complex128)
        if not (np.array_equal(bl_i_h, proj_chunk.bl_i_host)
                and np.array_equal(bl_j_h, proj_chunk.bl_j_host)):
            raise ValueError("antpairs do not match the baselines of the factored operator")
        _dev.call("hydra_row_factors", _dev.ptr(gains), _dev.ptr(w), _dev.ptr(bl_i),
                  _dev.ptr(bl_j), nbl, nf, nt, _dev.ptr(gg), _dev.ptr(None), st)
        A = _timed_gram(lambda: proj_chunk.gram(gains, w, prior_std=prior_here))
        model = proj_chunk.apply(None, rowmul=gg)        # R 1, ptsrc_sampler.py:104-106
    else:
        gg = _dev.empty((nbl, nf, nt), torch.complex128)
        R = _dev.c128(proj_chunk).reshape(m, ns)
        rows = _dev.empty((nbl, nf, nt), torch.complex128)
        _dev.call("hydra_row_factors", _dev.ptr(gains), _dev.ptr(w), _dev.ptr(bl_i),
                  _dev.ptr(bl_j), nbl, nf, nt, _dev.ptr(gg), _dev.ptr(rows), st)
        ws = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", m, ns))
        A = _dev.empty((ns, ns), torch.float64)
        _timed_gram(lambda: _dev.call(
            "hydra_gram", _dev.ptr(R), _dev.ptr(rows), m, ns, _dev.ptr(prior_here),
            _dev.ptr(None), _dev.ptr(A), _dev.ptr(ws), ws.numel(), st))
        model = _dev.empty((nbl, nf, nt), torch.complex128)
        ones = torch.ones(ns, dtype=torch.float64, device=data.device)
        _dev.call("hydra_apply_R", _dev.ptr(R), m, ns, _dev.ptr(ones), _dev.ptr(gg),
                  _dev.ptr(model), st)
        del ws, rows

    # z = conj(gg) ((d - model) N^-1 + omega_n N^-1/2)          ptsrc_sampler.py:104-123
    z = _dev.empty((nbl, nf, nt), torch.complex128)
    _dev.call("hydra_rhs_rows", _dev.ptr(data), _dev.ptr(model), _dev.ptr(w), _dev.ptr(om_n),
              _dev.ptr(gg), m, _dev.ptr(z), st)

    # omega_a is drawn after omega_n, on the root only       ptsrc_sampler.py:137-138
    om_a = None
    if myid == 0:
        if omega_a is None:
            omega_a = (1.0 if realisation else 0.0) * np.random.randn(ns)
        om_a = _dev.f64(omega_a)
    assert not (replicate and shard_rows), "replicate and shard_rows exclude each other"
    replicate = bool(replicate and comm is not None)
    add_here = om_a if comm is None else None
    if factored:
        b = proj_chunk.adjoint(z, scale=sigma, add=add_here)
    else:
        ws = _dev.workspace(_dev._lib.query("hydra_adjoint_workspace_bytes", m, ns))
        b = _dev.empty((ns,), torch.float64)
        _dev.call("hydra_adjoint_R", _dev.ptr(R), m, ns, _dev.ptr(z), _dev.ptr(sigma),
                  _dev.ptr(add_here), _dev.ptr(b), _dev.ptr(ws), ws.numel(), st)

    if comm is not None and shard_rows:
        # block row k of the sum goes to rank k; prior and identity on the rank's rows
        # (ptsrc_sampler.py:91-94); omega_a is the root's draw on every rank
        comm.allreduce_(b)
        if not hit:
            A = comm.reduce_rows_(A)
            lay = comm.row_layout(ns)
            rows = comm._rows_dev(lay, lay.rank, A.device)
            if rows.numel():
                A.mul_(sigma[None, :]).mul_(sigma[rows, None])
                A[torch.arange(rows.numel(), device=A.device), rows] += 1.0
        oa = om_a if myid == 0 else torch.zeros(ns, dtype=torch.float64, device=b.device)
        comm.bcast_(oa, root=0)
        b += oa
    elif replicate:
        # the same sums as all-reduces: the full system on every rank (the packed lower triangle on
        # the wire, mirrored on arrival; omega_a rides in the root's term of b)
        if myid == 0:
            b += om_a
        if not hit:
            comm.allreduce_sym_(A)
            _dev.call("hydra_prior_epilogue", _dev.ptr(A), ns, _dev.ptr(sigma), _dev.ptr(None), st)
        comm.allreduce_(b)
    elif comm is not None:
        # sum over ranks (comm.Reduce at ptsrc_sampler.py:86,132), then prior + omega_a on root
        if not hit:
            comm.reduce_sym_(A, root=0)      # lower triangle on the wire, mirrored on arrival
        comm.reduce_(b, root=0)
        if myid == 0:
            if not hit:
                _dev.call("hydra_prior_epilogue", _dev.ptr(A), ns, _dev.ptr(sigma), _dev.ptr(None),
                          st)
            b += om_a
        else:
            A = torch.zeros((1, 1), dtype=torch.float64, device=b.device)
            b = torch.zeros((1,), dtype=torch.float64, device=b.device)
    if cache is not None:
        if hit:
            if cache.A is not None:
                A = cache.A
        else:
            keep_A = shard_rows or replicate or myid == 0
            cache._store(comm, proj_chunk, gains, w, sigma, shard_rows, A, gg, model, keep_A)
    if (_dev.is_torch(ref) and not shard_rows and A.shape[0] == A.shape[1]
            and not (cache is not None and cache.factored)):
        A.hydra_symmetric = True      # mirrored exactly by the Gram finish / unpack kernels
    return _dev.like(ref, A), _dev.like(ref, b)


def apply_proj(proj, x):
    """Model visibilities ``(proj.reshape(-1, Ns) @ x).reshape(Nbl, Nf, Nt)`` -- the model
    update of the driver (example.py:974-976) for either operator form."""
    if isinstance(proj, FactoredProj):
        return proj.apply(x)
    nbl, nf, nt, ns = proj.shape
    R = _dev.c128(proj).reshape(-1, ns)
    out = _dev.empty((nbl, nf, nt), torch.complex128)
    xd = _dev.f64(x)
    _dev.call("hydra_apply_R", _dev.ptr(R), nbl * nf * nt, ns, _dev.ptr(xd),
              _dev.ptr(None), _dev.ptr(out), _dev.stream_ptr())
    return _dev.like(proj, out)
