"""
Linearised per-antenna gain head on MI355X -- drop-in for ``hydra.gain_sampler``
(reference: hydra/gain_sampler.py).

    proj_operator(ants, antpairs)          -> (A_real, A_imag)          gain_sampler.py:51-78
    apply_proj(x, A_real, A_imag, mv)      (x_i + conj x_j) * mv         :81-106
    apply_proj_conj(v, A_real, A_imag, mv, gain_shape)                   :109-142
    construct_rhs_mpi(comm, resid, ...)    right-hand side               :145-231
    apply_operator_mpi(comm, x, ...)       x + S^1/2 F^H A^H N^-1 A F S^1/2 x   :234-301

``proj_operator`` returns scipy DOK matrices exactly like the reference (callers keep them
and hand them back); the kernels use the equivalent index form (baseline -> antenna pair,
antenna -> adjacent baselines), derived once per matrix pair and cached.

``GainOperator`` is the device-resident form of the same operator for a CG solve: the real
weight W = |model_vis|^2 N^-1, the Fourier basis and the adjacency stay in HBM and
``apply(X)`` takes a batch of vectors (BASELINE config 4: "batched CG matvecs") -- one pass
over W serves up to 8 vectors.
"""
import numpy as np
import torch

from . import _dev

_INDEX_CACHE = {}          # (id(A_real), id(A_imag)) -> (A_real, A_imag, _Index); bounded, see below
_INDEX_CACHE_MAX = 8


def _remember(A_real, A_imag, idx):
    """Keep the index form of a projector pair; the cache holds the matrices alive (so that an id
    is never reused while its entry exists) and drops the oldest entry beyond a few pairs --
    every ``proj_operator`` call used to pin its matrices, device tables and an Nant^2 pair table
    for the life of the process."""
    while len(_INDEX_CACHE) >= _INDEX_CACHE_MAX:
        _INDEX_CACHE.pop(next(iter(_INDEX_CACHE)))
    _INDEX_CACHE[(id(A_real), id(A_imag))] = (A_real, A_imag, idx)


def proj_operator(ants, antpairs):
    """Sparse projectors from antenna gain fluctuations to baselines, one acting on the real
    part and one on the imaginary part (gain_sampler.py:51-78)."""
    from scipy.sparse import dok_matrix
    nvis, nants = len(antpairs), len(ants)
    A_real = dok_matrix((nvis, nants), dtype=int)
    A_imag = dok_matrix((nvis, nants), dtype=int)
    ants = list(ants)
    ant_idx = {ant: ants.index(ant) for ant in ants}
    for i, (ant1, ant2) in enumerate(antpairs):
        A_real[i, ant_idx[ant1]] += 1
        A_real[i, ant_idx[ant2]] += 1
        A_imag[i, ant_idx[ant1]] += 1
        A_imag[i, ant_idx[ant2]] += -1
    i1 = np.array([ant_idx[p[0]] for p in antpairs], dtype=np.int32)
    i2 = np.array([ant_idx[p[1]] for p in antpairs], dtype=np.int32)
    _remember(A_real, A_imag, _Index(i1, i2, nants))
    return A_real, A_imag


class _Index:
    """Index form of (A_real, A_imag): antenna pair per baseline and the CSR adjacency
    antenna -> (baseline, other antenna, role), on the host and on the device."""

    def __init__(self, i1, i2, nants):
        self.i1, self.i2, self.nants = i1, i2, int(nants)
        nbl = i1.size
        ant = np.concatenate([i1, i2]).astype(np.int64)
        bl = np.concatenate([np.arange(nbl), np.arange(nbl)])
        other = np.concatenate([i2, i1])
        role = np.concatenate([np.zeros(nbl, np.int32), np.ones(nbl, np.int32)])
        order = np.lexsort((role, bl, ant))
        self.adj_ptr = np.zeros(self.nants + 1, dtype=np.int32)
        self.adj_ptr[1:] = np.cumsum(np.bincount(ant, minlength=self.nants))
        self.adj_b = np.ascontiguousarray(bl[order], dtype=np.int32)
        self.adj_o = np.ascontiguousarray(other[order], dtype=np.int32)
        self.adj_role = np.ascontiguousarray(role[order], dtype=np.int32)
        self._dev = None
        # dense pair -> baseline table for the LDS-tiled operator kernel (None if a pair repeats)
        lo, hi = np.minimum(i1, i2).astype(np.int64), np.maximum(i1, i2).astype(np.int64)
        self.pair_table = None
        if np.unique(lo * self.nants + hi).size == nbl:
            tab = np.full((self.nants, self.nants), -1, dtype=np.int32)
            tab[lo, hi] = np.arange(nbl, dtype=np.int32)
            tab[hi, lo] = np.arange(nbl, dtype=np.int32)
            self.pair_table = np.ascontiguousarray(tab)
        self._pair_dev = None

    def pair_dev(self):
        if self._pair_dev is None and self.pair_table is not None:
            self._pair_dev = _dev.i32(self.pair_table)
        return self._pair_dev

    def dev(self):
        if self._dev is None:
            self._dev = tuple(_dev.i32(a) for a in (self.i1, self.i2, self.adj_ptr, self.adj_b,
                                                    self.adj_o, self.adj_role))
        return self._dev


def _index_of(A_real, A_imag):
    hit = _INDEX_CACHE.get((id(A_real), id(A_imag)))
    if hit is not None and hit[0] is A_real and hit[1] is A_imag:
        return hit[2]
    # matrices not made by proj_operator (dense arrays or other sparse formats): recover the pairs
    Ar = np.asarray(A_real.todense() if hasattr(A_real, "todense") else A_real)
    Ai = np.asarray(A_imag.todense() if hasattr(A_imag, "todense") else A_imag)
    nvis, nants = Ar.shape
    i1 = np.empty(nvis, dtype=np.int32)
    i2 = np.empty(nvis, dtype=np.int32)
    for k in range(nvis):
        plus, minus = np.nonzero(Ai[k] > 0)[0], np.nonzero(Ai[k] < 0)[0]
        if plus.size == 1 and minus.size == 1:
            i1[k], i2[k] = plus[0], minus[0]
        else:                                   # auto-correlation: A_real = 2, A_imag = 0
            cols = np.nonzero(Ar[k])[0]
            assert cols.size == 1 and Ar[k, cols[0]] == 2, "not a baseline projector"
            i1[k] = i2[k] = cols[0]
    idx = _Index(i1, i2, nants)
    _remember(A_real, A_imag, idx)
    return idx


def apply_proj(x, A_real, A_imag, model_vis):
    """(x_i + conj(x_j)) * model_vis for gain fluctuations ``x`` (Nants, Nfreqs, Ntimes)
    (gain_sampler.py:81-106)."""
    idx = _index_of(A_real, A_imag)
    i1, i2 = idx.dev()[:2]
    xd, mv = _dev.c128(x), _dev.c128(model_vis)
    nbl = mv.shape[0]
    nft = int(np.prod(mv.shape[1:]))
    assert xd.shape[0] == idx.nants and int(np.prod(xd.shape[1:])) == nft
    out = _dev.empty(mv.shape, torch.complex128)
    _dev.call("hydra_gain_proj", _dev.ptr(xd), _dev.ptr(i1), _dev.ptr(i2), nbl, _dev.ptr(mv), nft,
              _dev.ptr(out), _dev.stream_ptr())
    return _dev.like(x, out)


def _proj_conj_dev(v, idx, mv, nft):
    _, _, adj_ptr, adj_b, _, adj_role = idx.dev()
    y = _dev.empty((idx.nants, nft), torch.complex128)
    _dev.call("hydra_gain_proj_conj", _dev.ptr(v), _dev.ptr(mv), idx.nants, nft,
              _dev.host_ptr(idx.adj_ptr), _dev.ptr(adj_ptr), _dev.ptr(adj_b), _dev.ptr(adj_role),
              int(idx.adj_b.size), _dev.ptr(y), _dev.stream_ptr())
    return y


def apply_proj_conj(v, A_real, A_imag, model_vis, gain_shape):
    """Conjugate transpose of ``apply_proj`` (gain_sampler.py:109-142): (Nants, Nfreqs, Ntimes)."""
    idx = _index_of(A_real, A_imag)
    vd, mv = _dev.c128(v), _dev.c128(model_vis)
    assert tuple(vd.shape) == tuple(mv.shape)
    nants, n1, n2 = gain_shape
    assert nants == idx.nants and n1 * n2 == int(np.prod(mv.shape[1:]))
    y = _proj_conj_dev(vd, idx, mv, n1 * n2)
    return _dev.like(v, y.reshape(nants, n1, n2))


class GainOperator:
    """Device-resident gain GCR operator  x -> x + S^1/2 F^H A^H N^-1 A F S^1/2 x.

    Built once per Gibbs iteration from (inv_noise_var, pspec_sqrt, projectors, model_vis,
    Fbasis); ``apply(X)`` maps a vector (2*Nants*Nmodes,) or a batch (nvec, 2*Nants*Nmodes).
    ``local(X)`` returns only this rank's S^1/2 F^H A^H N^-1 A F S^1/2 X (to be summed over
    ranks before the identity is added, gain_sampler.py:296-301)."""

    def __init__(self, inv_noise_var, pspec_sqrt, A_real, A_imag, model_vis, Fbasis):
        self.idx = _index_of(A_real, A_imag)
        mv = _dev.c128(model_vis)
        w = _dev.f64(inv_noise_var)
        assert tuple(w.shape) == tuple(mv.shape)
        self.F = _dev.c128(Fbasis)
        self.nmodes = int(self.F.shape[0])
        self.nft = int(np.prod(self.F.shape[1:]))
        self.nbl = int(mv.shape[0])
        assert int(np.prod(mv.shape[1:])) == self.nft
        self.s = _dev.f64(pspec_sqrt)
        assert tuple(self.s.shape) == (self.nmodes,)
        self._mv, self._w, self._W = mv, w, None
        #: batches of vectors go through the matrix-core pair kernel when the pair list allows
        self.batch_mfma = True
        self.n = 2 * self.idx.nants * self.nmodes
        #: use the LDS-tiled pair kernel when the pair list has no duplicates
        self.tiled = True
        #: single vectors go through the streamed pair kernel (every weight read once per
        #: application) when the pair list has neither duplicates nor auto-correlations
        nants = self.idx.nants
        count = _dev._lib.query("hydra_gain_stream_weight_count", nants, self.nft)
        self.stream = bool(self.idx.pair_table is not None and count > 0
                           and not np.any(self.idx.i1 == self.idx.i2))
        self.Wt = self.sw = None
        if self.stream:
            st = _dev.stream_ptr()
            self.Wt = _dev.empty((count,), torch.float64)
            _dev.call("hydra_gain_weights_stream", _dev.ptr(mv), _dev.ptr(w), self.nbl, nants,
                      self.nft, _dev.ptr(self.idx.pair_dev()), _dev.ptr(self.Wt), st)
            ones = torch.ones((nants, self.nft), dtype=torch.complex128, device=mv.device)
            y1 = _dev.empty((nants, self.nft), torch.complex128)
            _dev.call("hydra_gain_pairs_stream", _dev.ptr(ones), nants, self.nft,
                      _dev.ptr(self.Wt), _dev.ptr(None), _dev.ptr(y1), st)
            self.sw = y1.real.contiguous()

    @classmethod
    def _from_stream(cls, idx, w, s, F, Wt, sw, model):
        """Operator around stream-layout weights that ``hydra_gain_prepare_stream`` has already
        written; ``model = (model_vis, gains)`` is kept for the by-baseline weights of the
        batched kernels, built only if asked for."""
        self = cls.__new__(cls)
        self.idx, self.F, self.s = idx, F, s
        self.nmodes = int(F.shape[0])
        self.nft = int(np.prod(F.shape[1:]))
        self.nbl = int(w.shape[0])
        self._mv, self._w, self._W, self._model = None, w, None, model
        self.batch_mfma = True
        self.n = 2 * idx.nants * self.nmodes
        self.tiled, self.stream = True, True
        self.Wt, self.sw = Wt, sw
        return self

    @property
    def W(self):
        """The fused weight |model_vis|^2 N^-1 as (Nbl, Nft) (gain_sampler.py:278-284), for the
        pair kernels that read it by baseline; built on first use."""
        if self._W is None and self._mv is None:
            from . import utils
            mv, gg = self._model
            pairs = list(zip(self.idx.i1.tolist(), self.idx.i2.tolist()))
            self._mv = utils.apply_gains(mv, gg, np.arange(self.idx.nants), pairs, inline=False)
        if self._W is None:
            self._W = _dev.empty((self.nbl, self.nft), torch.float64)
            _dev.call("hydra_gain_weights", _dev.ptr(self._mv), _dev.ptr(self._w),
                      self.nbl * self.nft, _dev.ptr(self._W), _dev.stream_ptr())
        return self._W

    @property
    def Wb(self):
        """The fused weights in the 16 x 16 antenna-block layout of the matrix-core batch kernel
        (csrc/gain.hip, gain_pairs_mfma_kernel), or None if the pair list has duplicates /
        auto-correlations or the array does not fit; built on first use (with ``sw``, the sum of
        the weights over each antenna's pairs, if the streamed form has not left it)."""
        if getattr(self, "_Wb", None) is None:
            self._Wb = False
            nants = self.idx.nants
            count = _dev._lib.query("hydra_gain_blocks_weight_count", nants, self.nft)
            if (self.idx.pair_table is not None and count > 0
                    and not np.any(self.idx.i1 == self.idx.i2)):
                st = _dev.stream_ptr()
                Wb = _dev.empty((count,), torch.float64)
                if self._mv is not None:
                    mv, gg = self._mv, None
                else:
                    mv, gg = (_dev.c128(t) for t in self._model)
                _dev.call("hydra_gain_weights_blocks", _dev.ptr(mv), _dev.ptr(gg), _dev.ptr(self._w),
                          self.nbl, nants, self.nft, _dev.ptr(self.idx.pair_dev()), _dev.ptr(Wb), st)
                if self.sw is None:
                    ones = torch.ones((1, nants, self.nft), dtype=torch.complex128, device=Wb.device)
                    y1 = _dev.empty((1, nants, self.nft), torch.complex128)
                    _dev.call("hydra_gain_pairs_mfma", _dev.ptr(ones), 1, nants, self.nft,
                              _dev.ptr(Wb), _dev.ptr(None), _dev.ptr(y1), st)
                    self.sw = y1[0].real.contiguous()
                self._Wb = Wb
        return self._Wb if self._Wb is not False else None

    def local(self, X, add=None):
        X = _dev.f64(X)
        single = X.dim() == 1
        Xb = X.reshape(1, -1) if single else X
        nvec = Xb.shape[0]
        assert Xb.shape[1] == self.n
        nants, nft = self.idx.nants, self.nft
        _, _, adj_ptr, adj_b, adj_o, _ = self.idx.dev()
        st = _dev.stream_ptr()
        addb = None if add is None else _dev.f64(add).reshape(nvec, self.n)
        if nvec > 1 and self.batch_mfma and self.Wb is not None:
            # a batch: per sample the pair sums are a symmetric matrix times 2 nvec real columns --
            # on the FP64 matrix cores, every weight read once per 8 vectors, the fields of a pass
            # in a sample-major layout between the three kernels (csrc/gain.hip)
            out = _dev.empty((nvec, self.n), torch.float64)
            ws = _dev.workspace(_dev._lib.query("hydra_gain_operator_blocks_workspace_bytes", nants,
                                                nft))
            _dev.call("hydra_gain_operator_blocks", _dev.ptr(Xb), nvec, nants, self.nmodes, nft,
                      _dev.ptr(self.F), _dev.ptr(self.s), _dev.ptr(self.Wb), _dev.ptr(self.sw),
                      _dev.ptr(addb), _dev.ptr(out), _dev.ptr(ws), ws.numel(), st)
            return out
        g = _dev.empty((nvec, nants, nft), torch.complex128)
        _dev.call("hydra_gain_basis_fwd", _dev.ptr(Xb), nvec, nants, self.nmodes, nft,
                  _dev.ptr(self.F), _dev.ptr(self.s), _dev.ptr(g), st)
        y = _dev.empty((nvec, nants, nft), torch.complex128)
        pairb = self.idx.pair_dev() if self.tiled else None
        if self.stream and nvec == 1:
            _dev.call("hydra_gain_pairs_stream", _dev.ptr(g), nants, nft, _dev.ptr(self.Wt),
                      _dev.ptr(self.sw), _dev.ptr(y), st)
        elif pairb is not None:
            _dev.call("hydra_gain_pairs_tiled", _dev.ptr(g), nvec, nants, nft, _dev.ptr(self.W),
                      self.nbl, _dev.ptr(pairb), _dev.ptr(y), st)
        else:
            _dev.call("hydra_gain_pairs", _dev.ptr(g), nvec, nants, nft, _dev.ptr(self.W),
                      self.nbl, _dev.host_ptr(self.idx.adj_ptr), _dev.ptr(adj_ptr),
                      _dev.ptr(adj_b), _dev.ptr(adj_o), int(self.idx.adj_b.size), _dev.ptr(y), st)
        out = _dev.empty((nvec, self.n), torch.float64)
        _dev.call("hydra_gain_basis_adj", _dev.ptr(y), nvec, nants, self.nmodes, nft,
                  _dev.ptr(self.F), _dev.ptr(self.s), _dev.ptr(addb), _dev.ptr(out), st)
        return out[0] if single else out

    def apply(self, X):
        Xd = _dev.f64(X)
        return _dev.like(X, self.local(Xd, add=Xd))


def _rhs_draws(comm, myid, shape, nants, nmodes, realisation, seed, omega_y, omega_r):
    """The fluctuation terms of the right-hand side in the reference's order
    (gain_sampler.py:163-199): ``np.random.seed(seed)``, the root draws w_y (Nants, Nmodes) and
    broadcasts it, then every rank draws w_r like the residual -- unless injected."""
    sw = 1.0 if realisation else 0.0
    np.random.seed(seed)
    if omega_y is None:
        if myid == 0:
            omega_y = sw * (1.0 * np.random.randn(nants, nmodes)
                            + 1.0j * np.random.randn(nants, nmodes)) / np.sqrt(2.0)
        else:
            omega_y = np.zeros((nants, nmodes), dtype=np.complex128)
    b = np.array(omega_y, dtype=np.complex128)
    if comm is not None:
        comm.Bcast(b, root=0)
    if omega_r is None:
        omega_r = sw * (1.0 * np.random.randn(*shape)
                        + 1.0j * np.random.randn(*shape)) / np.sqrt(2.0)
    return b, omega_r


def _rhs_finish(comm, yy, b, F, s, nants, nmodes, nft):
    """[Re; Im] of  w_y + S^1/2 F^H yy, the visibility term summed over ranks."""
    my_y = _dev.empty((1, 2 * nants * nmodes), torch.float64)
    _dev.call("hydra_gain_basis_adj", _dev.ptr(yy), 1, nants, nmodes, nft, _dev.ptr(F),
              _dev.ptr(s), _dev.ptr(None), _dev.ptr(my_y), _dev.stream_ptr())
    if comm is not None:
        comm.allreduce_(my_y)
    return my_y[0] + _dev.f64(np.concatenate((b.flatten().real, b.flatten().imag)))


def construct_rhs_mpi(comm, resid, inv_noise_var, pspec_sqrt, A_real, A_imag, model_vis, Fbasis,
                      realisation=True, seed=None, omega_y=None, omega_r=None):
    """Right-hand side of the gain GCR system (gain_sampler.py:145-231):
    [Re; Im] of  w_y + S^1/2 F^H A^H (N^-1 r + N^-1/2 w_r), summed over ranks.

    The fluctuation terms are drawn as in the reference -- ``np.random.seed(seed)``, the root
    draws w_y (Nants, Nmodes) first, then every rank draws w_r like ``resid`` (:163-199) --
    unless injected through ``omega_y`` / ``omega_r``.  ``comm=None`` runs a single rank."""
    myid = comm.Get_rank() if comm is not None else 0
    idx = _index_of(A_real, A_imag)
    rd = _dev.c128(resid)
    nvis, nfreqs, ntimes = rd.shape
    F = _dev.c128(Fbasis)
    nmodes, nants = int(F.shape[0]), idx.nants
    s = _dev.f64(pspec_sqrt)
    assert tuple(s.shape) == (nmodes,)
    nft = nfreqs * ntimes
    assert int(np.prod(F.shape[1:])) == nft
    b, omega_r = _rhs_draws(comm, myid, (nvis, nfreqs, ntimes), nants, nmodes, realisation, seed,
                            omega_y, omega_r)

    w = _dev.f64(inv_noise_var)
    mv = _dev.c128(model_vis)
    om = _dev.c128(omega_r)
    m = nvis * nft
    z = _dev.empty((nvis, nfreqs, ntimes), torch.complex128)
    st = _dev.stream_ptr()
    _dev.call("hydra_rhs_rows", _dev.ptr(rd), _dev.ptr(None), _dev.ptr(w), _dev.ptr(om),
              _dev.ptr(None), m, _dev.ptr(z), st)
    yy = _proj_conj_dev(z, idx, mv, nft)
    bvec = _rhs_finish(comm, yy, b, F, s, nants, nmodes, nft)
    return _dev.like(resid, bvec)


def prepare_sample(comm, data, model_vis, gains, inv_noise_var, pspec_sqrt, A_real, A_imag, Fbasis,
                   realisation=True, seed=None, omega_y=None, omega_r=None):
    """Right-hand side AND operator of one gain GCR sample from the calibrated-model inputs, i.e.
    the driver's sequence (example.py:690-745)

        ggv  = utils.apply_gains(model_vis, gains, ants, antpairs)
        b    = construct_rhs_mpi(comm, data - ggv, inv_noise_var, pspec_sqrt, A_real, A_imag, ggv,
                                 Fbasis, ...)
        op   = GainOperator(inv_noise_var, pspec_sqrt, A_real, A_imag, ggv, Fbasis)

    with ``gains`` (Nants, Nfreqs, Ntimes) indexed like the projectors' columns.  Returns
    ``(b, op)``.  When the streamed pair kernel applies (no repeated pair, no auto-correlation)
    and every baseline is listed as (lower, higher) antenna index, all of it is ONE pass over the
    visibilities (``hydra_gain_prepare_stream``: 14 GB instead of 55 GB at HERA-331 x 64 x 64);
    otherwise the sequence above runs as written.  Same draws, same results."""
    from . import utils
    myid = comm.Get_rank() if comm is not None else 0
    idx = _index_of(A_real, A_imag)
    dd, mv, gg = _dev.c128(data), _dev.c128(model_vis), _dev.c128(gains)
    w = _dev.f64(inv_noise_var)
    nvis, nfreqs, ntimes = dd.shape
    assert tuple(mv.shape) == tuple(dd.shape) == tuple(w.shape)
    F = _dev.c128(Fbasis)
    nmodes, nants = int(F.shape[0]), idx.nants
    nft = nfreqs * ntimes
    if tuple(gg.shape[1:]) != (nfreqs, ntimes):
        gg = gg.expand(gg.shape[0], nfreqs, ntimes).contiguous()
    assert tuple(gg.shape) == (nants, nfreqs, ntimes) and int(np.prod(F.shape[1:])) == nft
    count = _dev._lib.query("hydra_gain_stream_weight_count", nants, nft)
    fused = bool(idx.pair_table is not None and count > 0 and np.all(idx.i1 < idx.i2))
    if not fused:
        ants = np.arange(nants)
        pairs = list(zip(idx.i1.tolist(), idx.i2.tolist()))
        ggv = utils.apply_gains(mv, gg, ants, pairs, inline=False)
        b = construct_rhs_mpi(comm, dd - ggv, w, pspec_sqrt, A_real, A_imag, ggv, Fbasis,
                              realisation=realisation, seed=seed, omega_y=omega_y, omega_r=omega_r)
        return _dev.like(data, b), GainOperator(w, pspec_sqrt, A_real, A_imag, ggv, Fbasis)

    s = _dev.f64(pspec_sqrt)
    assert tuple(s.shape) == (nmodes,)
    b, omega_r = _rhs_draws(comm, myid, (nvis, nfreqs, ntimes), nants, nmodes, realisation, seed,
                            omega_y, omega_r)
    om = _dev.c128(omega_r)
    Wt = _dev.empty((count,), torch.float64)
    sw = _dev.empty((nants, nft), torch.float64)
    yy = _dev.empty((nants, nft), torch.complex128)
    _dev.call("hydra_gain_prepare_stream", _dev.ptr(dd), _dev.ptr(mv), _dev.ptr(gg), _dev.ptr(w),
              _dev.ptr(om), nvis, nants, nft, _dev.ptr(idx.pair_dev()), _dev.ptr(Wt), _dev.ptr(sw),
              _dev.ptr(yy), _dev.stream_ptr())
    bvec = _rhs_finish(comm, yy, b, F, s, nants, nmodes, nft)
    op = GainOperator._from_stream(idx, w, s, F, Wt, sw, model=(mv, gg))
    return _dev.like(data, bvec), op


def apply_operator_mpi(comm, x, inv_noise_var, pspec_sqrt, A_real, A_imag, model_vis, Fbasis):
    """Gain GCR operator applied to ``x`` (gain_sampler.py:234-301); the root's ``x`` is used
    on every rank and the data-dependent term is summed over ranks.  For repeated
    application build a ``GainOperator`` once instead."""
    op = GainOperator(inv_noise_var, pspec_sqrt, A_real, A_imag, model_vis, Fbasis)
    xd = _dev.f64(x).clone()
    if comm is not None:
        comm.bcast_(xd, root=0)
    tot = op.local(xd)
    if comm is not None:
        comm.allreduce_(tot)
    return _dev.like(x, _dev.f64(x) + tot)
