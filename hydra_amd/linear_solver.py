"""
Linear solvers of the GCR system on MI355X -- drop-in for ``hydra.linear_solver``
(reference: hydra/linear_solver.py).

``cg`` keeps the reference's algorithm (linear_solver.py:378-509): no
preconditioner, convergence tested before each iteration (element-wise by
default), ``x`` returned after ``maxiters`` regardless.  The iteration state
(x, r, p, r.r, done flag, iteration count) lives on the GPU and one fused kernel
per iteration does both dot products, the three vector updates and the test
(csrc/solver.hip); a converged solve turns later enqueued steps into no-ops, so
the host only reads the flag every ``check_every`` iterations.

``cg_mpi`` / ``setup_mpi_blocks`` / ``collect_linear_sys_blocks`` / ``matvec_mpi``
keep the reference's 2-D block layout over ranks (linear_solver.py:11-375) on
``hydra_amd._dist.Comm`` (torch.distributed: RCCL on GPUs, gloo on CPUs).
"""
import ctypes

import numpy as np
import torch

from . import _dev

CHECK_EVERY = 16

#: iterations the last CONVERGED operator-form solve with a given operator, size and tolerance
#: needed.  The converged flag lives on the GPU and reading it stalls the queue, so iterations
#: are enqueued in blocks and the steps enqueued past convergence are no-ops -- but their operator
#: applications are not: with blocks of 16, config 4's gain solve (24 iterations, 0.41 ms each)
#: ran 32 applications, 18 % of a sample.  Consecutive Gibbs samples need about the same number
#: of iterations, so the first block is the last solve's count and the following ones are short.
#: The schedule only decides when the flag is read: the iterates and the result do not depend on
#: it.  It does decide how many times the operator is CALLED, which matters when the operator
#: holds a collective: the history is therefore kept per operator (``_history_key``), and an
#: operator that names its communicator (``.comm``) has its prediction agreed across the ranks.
_ITER_HISTORY = {}

#: longest first block: a prediction is a number of operator applications enqueued blind
FIRST_BLOCK_MAX = 8 * CHECK_EVERY


def _history_key(linear_op, explicit=None):
    """What identifies an operator between solves: the caller's ``history_key``, the operator's
    ``hydra_history_key`` attribute, or the code object behind the callable (a closure rebuilt
    every Gibbs iteration at one call site keeps its key; two different call sites never share
    one, so a root-only solve cannot change how often a collective operator is applied)."""
    if explicit is not None:
        return explicit
    key = getattr(linear_op, "hydra_history_key", None)
    if key is not None:
        return key
    fn = getattr(linear_op, "__func__", linear_op)            # bound method -> function
    code = getattr(fn, "__code__", None)
    if code is None:                                          # callable object
        call = getattr(type(linear_op), "__call__", None)
        code = getattr(call, "__code__", None)
    return code


def _predicted_iters(key, linear_op):
    """The last converged solve's count for this key; agreed with the other ranks (the root's
    value) when the operator names the communicator its collectives run on."""
    pred = int(_ITER_HISTORY.get(key, 0)) if key[-1] is not None else 0
    comm = getattr(linear_op, "comm", None)
    if comm is not None and comm.Get_size() > 1:
        t = torch.tensor([pred], dtype=torch.int64, device=_dev.device())
        comm.bcast_(t, root=0)
        pred = int(t.item())
    return pred


def _block_sizes(pred, as_torch):
    """Iterations to enqueue before each read of the converged flag."""
    if not as_torch:                 # a numpy operator synchronises every call anyway
        while True:
            yield 1
    if pred > 0:
        yield min(pred, FIRST_BLOCK_MAX)
        while True:
            yield max(2, pred // 16)
    while True:
        yield CHECK_EVERY


def _record_iters(key, st, iters):
    """Only a solve that set the converged flag is a prediction for the next one (a run into
    ``maxiters`` would enqueue that many applications blind)."""
    if key[-1] is not None and float(st[1].item()) != 0.0:
        _ITER_HISTORY[key] = iters


def mark_symmetric(A):
    """Tag a matrix as exactly symmetric (A == A.T bit for bit), as the Gram finish kernels leave
    it: ``cg`` then reads half of it per product without checking.  Returns ``A``."""
    if isinstance(A, torch.Tensor):
        A.hydra_symmetric = True
    return A


def _is_symmetric(A):
    """Exact symmetry of a square GPU matrix.  Matrices built by this package carry a tag
    (``mark_symmetric``); anything else is compared with its transpose in row blocks, so that no
    n^2 temporary is made (the 100 000-source operator is 80 GB)."""
    if getattr(A, "hydra_symmetric", False):
        return True
    n = A.shape[0]
    step = max(1, min(n, (1 << 26) // max(n, 1)))          # <= 64 Mi entries per comparison
    for i in range(0, n, step):
        if not torch.equal(A[i:i + step], A[:, i:i + step].T):
            return False
    return True


def _dense_cg(Amat, bvec, maxiters, abs_tol, use_norm_tol, x0):
    A = _dev.f64(Amat)
    b = _dev.f64(bvec)
    n = b.numel()
    assert tuple(A.shape) == (n, n)
    x = _dev.f64(x0).clone() if x0 is not None else torch.zeros_like(b)
    ws = _dev.workspace(_dev._lib.query("hydra_cg_workspace_bytes", n))
    import ctypes
    iters = ctypes.c_int(0)
    # the GCR operators are symmetric by construction and tagged so by the functions that build
    # them; an untagged matrix is checked, so that an unsymmetric Amat still gets A @ p
    symmetric = n >= 4096 and (getattr(Amat, "hydra_symmetric", False) or _is_symmetric(A))
    _dev.call("hydra_cg_solve_dense", _dev.ptr(A), _dev.ptr(b), _dev.ptr(x), n, int(maxiters),
              float(abs_tol), int(bool(use_norm_tol)), int(symmetric), CHECK_EVERY,
              ctypes.byref(iters),
              _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
    return x, iters.value


def _op_cg(bvec, maxiters, abs_tol, use_norm_tol, x0, linear_op, history_key=None):
    """CG with a caller-supplied operator.  The operator is called with the same kind of array
    as ``bvec`` (numpy or GPU tensor) and may return either."""
    as_torch = _dev.is_torch(bvec)
    b = _dev.f64(bvec)
    n = b.numel()
    x = _dev.f64(x0).clone() if x0 is not None else torch.zeros_like(b)
    r = torch.empty_like(b)
    p = torch.empty_like(b)
    st = _dev.zeros((_dev._lib.query("hydra_cg_state_len"),), torch.float64)
    sp = _dev.stream_ptr()

    def op(v):
        out = linear_op(v if as_torch else v.cpu().numpy())
        return _dev.f64(out)

    # r = b - A x0 (linear_solver.py:433-446); without a starting guess x0 = 0 and A x0 = 0 for any
    # linear operator, so that application (one of 25 at config 4) is not made
    ax0 = op(x) if x0 is not None else torch.zeros_like(b)
    _dev.call("hydra_cg_init", n, _dev.ptr(b), _dev.ptr(ax0), _dev.ptr(r), _dev.ptr(p),
              _dev.ptr(st), float(abs_tol), int(bool(use_norm_tol)), sp)
    it = 0
    key = (n, float(abs_tol), bool(use_norm_tol), _history_key(linear_op, history_key))
    blocks = _block_sizes(_predicted_iters(key, linear_op), as_torch)
    while True:
        if float(st[1].item()) != 0.0 or it >= maxiters:
            break
        for _ in range(min(next(blocks), maxiters - it)):
            ap = op(p)
            _dev.call("hydra_cg_step", n, _dev.ptr(ap), _dev.ptr(x), _dev.ptr(r), _dev.ptr(p),
                      _dev.ptr(st), float(abs_tol), int(bool(use_norm_tol)), sp)
            it += 1
    iters = int(st[2].item())
    _record_iters(key, st, iters)
    return x, iters


def _op_cg_root_driven(bvec, maxiters, abs_tol, use_norm_tol, x0, linear_op, comm,
                       history_key=None):
    """The reference's ``cg(..., linear_op=, comm=)`` (linear_solver.py:441-499): ``linear_op``
    is COLLECTIVE (every rank must call it the same number of times; it synchronises its own
    input) and only the root's return value is valid -- e.g. ``apply_lhs_no_rot_mpi``, which
    returns zeros elsewhere (sh_sampler.py:405-494).  The root drives: after every step its
    direction vector and CG state (r.r, done flag, iteration count) are broadcast -- ONE
    broadcast of p and the state together, where the reference sends the finished flag and
    ``pvec`` separately (:466-476, :495-496) -- so every rank sees the root's flag and stops
    after the same number of operator calls.  Non-root ranks run the same fused step on their
    (meaningless) operator result; their x is replaced by the root's at the end."""
    as_torch = _dev.is_torch(bvec)
    b = _dev.f64(bvec)
    n = b.numel()
    x = _dev.f64(x0).clone() if x0 is not None else torch.zeros_like(b)
    r = torch.empty_like(b)
    nst = _dev._lib.query("hydra_cg_state_len")
    ps = torch.zeros(n + nst, dtype=torch.float64, device=b.device)   # p and the state: one message
    p, st = ps[:n], ps[n:]
    sp = _dev.stream_ptr()

    def op(v):
        out = linear_op(v if as_torch else v.cpu().numpy())
        return _dev.f64(out)

    _dev.call("hydra_cg_init", n, _dev.ptr(b), _dev.ptr(op(x)), _dev.ptr(r), _dev.ptr(p),
              _dev.ptr(st), float(abs_tol), int(bool(use_norm_tol)), sp)
    comm.bcast_(ps, root=0)
    it = 0
    # every rank sees the root's flag and iteration count, but a rank may have recorded other
    # solves under the same key: the root's prediction is the one all of them use
    key = (n, float(abs_tol), bool(use_norm_tol), ("root", _history_key(linear_op, history_key)))
    pred = int(_ITER_HISTORY.get(key, 0))
    if comm.Get_size() > 1:
        t = torch.tensor([pred], dtype=torch.int64, device=b.device)
        comm.bcast_(t, root=0)
        pred = int(t.item())
    blocks = _block_sizes(pred, as_torch)
    while True:
        if float(st[1].item()) != 0.0 or it >= maxiters:      # the root's flag, on every rank
            break
        for _ in range(min(next(blocks), maxiters - it)):
            ap = op(p)
            _dev.call("hydra_cg_step", n, _dev.ptr(ap), _dev.ptr(x), _dev.ptr(r), _dev.ptr(p),
                      _dev.ptr(st), float(abs_tol), int(bool(use_norm_tol)), sp)
            comm.bcast_(ps, root=0)
            it += 1
    iters = int(st[2].item())
    _record_iters(key, st, iters)
    return x, iters


def cg(Amat, bvec, maxiters=1000, abs_tol=1e-8, use_norm_tol=False, x0=None, linear_op=None,
       comm=None, return_iters=False, history_key=None):
    """Conjugate-gradient solve of ``Amat x = bvec`` (linear_solver.py:378-509).

    ``linear_op`` (callable) replaces ``Amat`` when given.  With ``comm`` the solve is driven by
    the root as in the reference: only the root's operator results (or matrix) count, and its
    solution is broadcast at the end, so all ranks return the same ``x`` (:441-508).  An
    operator that already returns the summed result on EVERY rank (an all-reduce inside, e.g.
    ``RowShardedOperator`` or ``sh_sampler.LhsOperator``) needs no ``comm`` here: every rank then
    runs the same iteration on identical data, one collective per iteration instead of three.
    Such an operator should carry its communicator as ``.comm`` (both of the above do): the
    number of iterations enqueued between reads of the converged flag is learned from the last
    solve with the same operator (``history_key``, default: the operator's code), and with
    ``.comm`` the ranks agree on it, so that every rank applies the operator equally often.
    """
    if x0 is not None:
        assert tuple(x0.shape) == tuple(bvec.shape), (
            "Initial guess x0 has a different shape to bvec")
        assert x0.dtype == bvec.dtype, "Initial guess x0 has a different type to bvec"
    if linear_op is not None and comm is not None:
        x, iters = _op_cg_root_driven(bvec, maxiters, abs_tol, use_norm_tol, x0, linear_op, comm,
                                      history_key)
    elif linear_op is not None:
        x, iters = _op_cg(bvec, maxiters, abs_tol, use_norm_tol, x0, linear_op, history_key)
    elif comm is None or comm.Get_rank() == 0:
        x, iters = _dense_cg(Amat, bvec, maxiters, abs_tol, use_norm_tol, x0)
    else:       # only the root's matrix counts (:441-443); its solution arrives below
        x, iters = torch.zeros_like(_dev.f64(bvec)), 0
    if comm is not None:
        comm.barrier()
        comm.bcast_(x, root=0)
    out = _dev.like(bvec, x)
    return (out, iters) if return_iters else out


BLOCKED_MIN_N = 6144     # hydra_chol_solve switches to 256-wide panels + MFMA updates from here


def solve_dense(Amat, bvec, overwrite_a=False):
    """Dense fallback: ``scipy.linalg.solve(A, b, assume_a='her')`` of the driver
    (example.py:944) for the symmetric positive-definite GCR operator, by blocked Cholesky
    on the GPU (csrc/chol.hip; from 6144 unknowns on: 256-wide panels with the trailing update on
    the FP64 matrix cores -- the 100 000-source operator of config 5 is an 80 GB matrix).
    ``overwrite_a=True`` (GPU tensor input only) factors in place instead of on a copy.
    Raises ``np.linalg.LinAlgError`` if A is not positive definite."""
    A = _dev.f64(Amat)
    b = _dev.f64(bvec).clone()
    n = b.numel()
    assert tuple(A.shape) == (n, n)
    if n >= BLOCKED_MIN_N and n % 2 == 1:
        # the large-matrix path moves rows by 16-byte LDS-DMA: pad to even n with a unit pivot
        Ap = torch.empty((n + 1, n + 1), dtype=torch.float64, device=A.device)
        Ap[:n, :n] = A
        Ap[n, :] = 0.0
        Ap[:n, n] = 0.0
        Ap[n, n] = 1.0
        bp = torch.zeros(n + 1, dtype=torch.float64, device=A.device)
        bp[:n] = b
        return _dev.like(bvec, solve_dense(Ap, bp, overwrite_a=True)[:n].clone())
    if not (overwrite_a and A is Amat):
        A = A.clone()
    info = _dev.zeros((1,), torch.int32)
    _dev.call("hydra_chol_solve", _dev.ptr(A), _dev.ptr(b), n, _dev.ptr(info), _dev.stream_ptr())
    if int(info.item()) != 0:
        raise np.linalg.LinAlgError("matrix is not positive definite")
    return _dev.like(bvec, b)


#: up to this size ``cho_factor`` also forms L^-1 (twice n^2 doubles: itself and its transpose)
INVERSE_FACTOR_MAX_N = 4096


class CholeskyFactor:
    """What ``cho_factor`` returns: ``L`` (lower triangle of the padded matrix), the system size
    ``n`` and, for small systems, ``linv`` = L^-1 and ``linv_t`` = L^-T as dense matrices."""

    def __init__(self, L, n, linv=None, linv_t=None):
        self.L, self.n, self.linv, self.linv_t = L, n, linv, linv_t


def cho_factor(Amat, overwrite_a=False, inverse_factor=None):
    """Cholesky factor of a symmetric positive-definite operator, kept on the GPU for
    ``cho_solve`` (the names follow scipy.linalg).  For systems whose matrix does not change from
    sample to sample -- the SH head's ``R^T N^-1 R + S^-1`` (sh_sampler.py:389-402) -- one
    factorisation replaces the CG the reference runs per sample (example.py:1014-1052) by two
    triangular solves; the result is the exact solution the CG converges to.

    ``inverse_factor`` (default: for n <= 4096): also invert the factor once, so that a solve is
    two matrix-vector products x = L^-T (L^-1 b) at the memory rate instead of two dependent
    substitution chains (1681 unknowns: 0.75 ms -> 0.02 ms per right-hand side)."""
    A = _dev.f64(Amat)
    n = A.shape[0]
    assert tuple(A.shape) == (n, n)
    if n >= BLOCKED_MIN_N and n % 2 == 1:
        Ap = torch.zeros((n + 1, n + 1), dtype=torch.float64, device=A.device)
        Ap[:n, :n] = A
        Ap[n, n] = 1.0
        A, overwrite_a = Ap, True
    elif not (overwrite_a and A is Amat):
        A = A.clone()
    info = _dev.zeros((1,), torch.int32)
    _dev.call("hydra_chol_solve", _dev.ptr(A), _dev.ptr(None), A.shape[0], _dev.ptr(info),
              _dev.stream_ptr())
    if int(info.item()) != 0:
        raise np.linalg.LinAlgError("matrix is not positive definite")
    if inverse_factor is None:
        inverse_factor = n <= INVERSE_FACTOR_MAX_N
    if not inverse_factor:
        return CholeskyFactor(A, n)
    # X = L^-1 (set-up, once per matrix), column block by column block (csrc/chol.hip)
    linv = torch.empty((n, n), dtype=torch.float64, device=A.device)
    _dev.call("hydra_tri_inverse", _dev.ptr(A), A.shape[1], n, _dev.ptr(linv), n, _dev.stream_ptr())
    return CholeskyFactor(A, n, linv, linv.T.contiguous())


def _gemv(M, v):
    y = torch.empty(M.shape[0], dtype=torch.float64, device=M.device)
    _dev.call("hydra_gemv_rows", _dev.ptr(M), M.shape[0], M.shape[1], _dev.ptr(v), _dev.ptr(y), 0,
              _dev.stream_ptr())
    return y


def cho_solve(factor, bvec):
    """x = A^-1 b from ``cho_factor``'s result."""
    L, n = factor.L, factor.n
    if factor.linv is not None:
        b = _dev.f64(bvec).reshape(-1).contiguous()
        return _dev.like(bvec, _gemv(factor.linv_t, _gemv(factor.linv, b)))
    np_ = L.shape[0]
    b = torch.zeros(np_, dtype=torch.float64, device=L.device)
    b[:n] = _dev.f64(bvec).reshape(-1)
    _dev.call("hydra_chol_apply", _dev.ptr(L), _dev.ptr(b), np_, _dev.stream_ptr())
    return _dev.like(bvec, b[:n] if np_ != n else b)


# ------------------------------------------------------------------ row-sharded operator

class RowShardedOperator:
    """A dense operator sharded by rows over the ranks of ``comm`` (one process per GPU): rank k
    keeps the rows ``comm.row_layout(n).rows(k)`` (block-cyclic, see ``_dist.RowLayout``) as
    ``A_rows`` (n_k, n).  Calling it with a full vector returns the full product on every rank:
    the local row-block product (``hydra_gemv_rows``, HBM-bound on 8 n_k n bytes) followed by one
    all-gather of n doubles over RCCL.  Pass it as ``cg(None, b, linear_op=op)``: every rank then
    runs the same fused CG steps on identical data and returns the same x, as ``cg(comm=...)``
    does in the reference (linear_solver.py:506-508).  This is the 1-D analogue of the
    reference's 2-D block layout (``cg_mpi``, linear_solver.py:238-375) for matrices too large for
    one GPU's patience -- the 80 GB operator of 100 000 sources costs 7 ms per product on one GPU,
    1/size of that here."""

    def __init__(self, comm, A_rows, n):
        self.comm = comm
        self.A = _dev.f64(A_rows)
        self.n = int(n)
        nloc = comm.row_layout(self.n).nloc()
        assert tuple(self.A.shape) == (nloc, self.n), (tuple(self.A.shape), (nloc, self.n))
        assert self.A.is_contiguous()
        self.y = torch.empty(max(nloc, 1), dtype=torch.float64, device=self.A.device)

    def __call__(self, v):
        vd = _dev.f64(v)
        assert vd.numel() == self.n
        nloc = self.A.shape[0]
        if nloc > 0:
            _dev.call("hydra_gemv_rows", _dev.ptr(self.A), nloc, self.n, _dev.ptr(vd),
                      _dev.ptr(self.y), 0, _dev.stream_ptr())
        return _dev.like(v, self.comm.allgather_rows(self.y[:nloc], self.n))


#: columns per trailing update of the row-sharded factorisation (csrc/chol.hip: HYDRA_CHOL_SPW)
SUPER_PANEL = 2048
PANEL = 256


def _addr(t, row, col):
    """Device address of element (row, col) of a contiguous 2-D float64 tensor."""
    return t.data_ptr() + (row * t.shape[1] + col) * 8


class RowShardedFactor:
    """Cholesky factor of an operator sharded by rows over the ranks of ``comm`` (what
    ``factor_dense_rows`` returns): a rank's rows of L in the lower triangle of its rows ``A`` of
    the operator (``comm.row_layout(n)``); what lies above the diagonal is still the operator.
    ``solve(b)`` runs the two triangular sweeps block by block, each diagonal block by its owner
    (``hydra_trsv_block``), the rest of L by everyone on their own rows (``hydra_gemv_sub``)."""

    def __init__(self, comm, A, n):
        self.comm, self.A, self.n = comm, A, n
        self.lay = comm.row_layout(n)

    def solve(self, bvec):
        comm, A, n, lay = self.comm, self.A, self.n, self.lay
        rank, st = lay.rank, _dev.stream_ptr()
        multi = lay.size > 1
        b = _dev.f64(bvec)
        assert b.numel() == n
        nloc = lay.nloc()
        # L y = b.  y_loc: this rank's entries in local order; a solved block goes to everyone,
        # then every rank takes it out of its rows below
        rows_d = comm._rows_dev(lay, rank, b.device)
        y_loc = b[rows_d].contiguous() if nloc else torch.zeros(1, dtype=torch.float64, device=b.device)
        z = torch.empty(n, dtype=torch.float64, device=b.device)
        for blk in range(lay.nblocks):
            g0, g1 = lay.block_range(blk)
            o, l0 = lay.owner(blk), lay.local_start(blk)
            zb = z[g0:g1]
            if rank == o:
                zb.copy_(y_loc[l0:l0 + g1 - g0])
                _dev.call("hydra_trsv_block", _addr(A, l0, g0), n, g1 - g0, _dev.ptr(zb), 0, st)
            if multi:
                comm.bcast_(zb, root=o)
            lb = lay.local_index(g1)
            if nloc > lb:
                _dev.call("hydra_gemv_sub", _addr(A, lb, g0), n, nloc - lb, g1 - g0, _dev.ptr(zb),
                          y_loc.data_ptr() + lb * 8, 0, st)
        # L^T x = y, from the last block up.  Every rank holds all of z; the owner of a block
        # solves it and takes its rows out of everything above
        c = torch.empty(n, dtype=torch.float64, device=b.device)
        for blk in range(lay.nblocks - 1, -1, -1):
            g0, g1 = lay.block_range(blk)
            o, l0 = lay.owner(blk), lay.local_start(blk)
            zb = z[g0:g1]
            if rank == o:
                _dev.call("hydra_trsv_block", _addr(A, l0, g0), n, g1 - g0, _dev.ptr(zb), 1, st)
                if g0 > 0:
                    c[:g0].zero_()
                    _dev.call("hydra_gemv_sub", _addr(A, l0, 0), n, g1 - g0, g0, _dev.ptr(zb),
                              _dev.ptr(c), 1, st)
            if multi:
                comm.bcast_(zb, root=o)
                if g0 > 0:
                    comm.bcast_(c[:g0], root=o)
            if g0 > 0:
                z[:g0] += c[:g0]
        return _dev.like(bvec, z)


def solve_dense_rows(comm, A_rows, bvec, overwrite_a=False):
    """``factor_dense_rows(comm, A_rows).solve(bvec)``: the dense solve of the driver
    (example.py:940-947) for an operator sharded by rows; x (n,) on every rank."""
    n = int(_dev.f64(bvec).numel())
    return factor_dense_rows(comm, A_rows, overwrite_a=overwrite_a, n=n).solve(bvec)


def factor_dense_rows(comm, A_rows, overwrite_a=False, n=None):
    """The factorisation half of the dense solve of the driver
    (``scipy.linalg.solve(A, b, assume_a='her')``, example.py:940-947) for an operator sharded by
    rows over the ranks of ``comm`` -- the layout ``precompute_mpi(..., shard_rows=True)`` leaves it
    in: rank k holds the rows ``comm.row_layout(n).rows(k)`` (block-cyclic) as ``A_rows`` (n_k, n).
    Returns a ``RowShardedFactor`` whose ``solve(b)`` (b (n,), the same on every rank) gives x (n,)
    on every rank; a caller whose operator does not change between samples factors once and
    solves per sample.

    Right-looking Cholesky, every rank working on its own rows, every step a kernel of this
    library (csrc/chol.hip).  Columns are finished 256 at a time: the rank that owns the
    256 x 256 diagonal block sends it to all, and every rank runs the one-GPU routine's 32-wide
    panel steps (``hydra_chol_panel``) on that block stacked on ITS rows of the panel -- the block
    is factored redundantly (8 short steps), the rows are solved against it -- in a buffer that is
    then all-gathered (n x 256 doubles on every rank).  As in the one-GPU routine ("super-panels")
    the matrix right of a SUPER-panel of 2048 columns is updated once, with K = 2048, on the FP64
    matrix cores (``hydra_chol_update_rect``: one round trip of C per 2048 columns instead of per
    256); inside the super-panel the columns still to be factored take a rank-256 update from each
    finished panel.  Look-ahead: of that big update the columns of the next super-panel are done
    first and the rest runs on a second stream, so the next super-panel's chain of short panel
    kernels, broadcasts and all-gathers is hidden under it.  With block-cyclic rows every rank's share of the rows below the current column
    is the same at every stage (contiguous blocks left the last rank 2 - 1/size times the mean
    work).  n must be even.  Raises ``np.linalg.LinAlgError`` if A is not positive definite."""
    A = _dev.f64(A_rows)
    if not (overwrite_a and A is A_rows):
        A = A.clone()
    n = int(A.shape[1]) if n is None else int(n)
    lay = comm.row_layout(n)
    size, rank, B = lay.size, lay.rank, lay.block
    nloc, maxloc = lay.nloc(), max(lay.maxloc, 1)
    assert tuple(A.shape) == (nloc, n) and A.is_contiguous(), (tuple(A.shape), (nloc, n))
    assert n % 2 == 0, "factor_dense_rows: n must be even (rows are moved in 16-byte pieces)"
    dev, st = A.device, _dev.stream_ptr()
    multi = size > 1
    info = _dev.zeros((1,), torch.int32)
    bad = _dev.zeros((1,), torch.int32)
    # A rank's rows of the current panel, in local order, below PANEL spare rows: the diagonal block
    # goes right above the first row that is below it, so [block; rows below] is one array
    mine_ext = torch.zeros((PANEL + maxloc, PANEL), dtype=torch.float64, device=dev)
    mine = mine_ext[PANEL:]
    panel = torch.empty((size * maxloc, PANEL), dtype=torch.float64, device=dev) if multi else mine
    # two gathered super-panels: the matrix far to the right is still taking the last one's update
    # (on the side stream) while the next one is being filled
    spanels = [torch.empty((size * maxloc, SUPER_PANEL), dtype=torch.float64, device=dev)
               for _ in range(2 if n > 2 * SUPER_PANEL else 1)] if n > SUPER_PANEL else None
    low = torch.ones((PANEL, PANEL), dtype=torch.bool, device=dev).tril()
    main = torch.cuda.current_stream(dev)
    side = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)] if n > 2 * SUPER_PANEL else None

    def update(c0, c1, k_lo, k_hi, src, ldb, streams=None):
        """My rows below column c0 of columns [c0, c1) take the rank-(k_hi - k_lo) update
        C[i][j] -= sum_k A[i][k] L[j][k], k in [k_lo, k_hi): PA = my rows, PB = the gathered rows
        of ``src`` (leading dimension ldb; rank o's local row l at o * maxloc + l), one column
        block at a time -- the rows of a block sit together in ``src``."""
        if c1 <= c0 or nloc == 0:
            return
        for it, cb in enumerate(range(c0 // B, (c1 - 1) // B + 1)):
            # column blocks write different columns: on two streams in turn, so that the last,
            # partly filled round of workgroups of one call runs beside the first of the next
            st = _dev.stream_ptr() if streams is None else \
                ctypes.c_void_p(streams[it % len(streams)].cuda_stream)
            g0, g1 = lay.block_range(cb)
            j0, j1 = max(c0, g0), min(c1, g1)
            o, l0 = lay.owner(cb), lay.local_start(cb)
            pb_ptr = src.data_ptr() + ((o * maxloc + l0 + (j0 - g0)) * ldb) * 8
            if o == rank:
                # my rows from the diagonal on: inside the block those left of column j <= row,
                # and the rows of my later blocks follow in local order, where that holds for
                # every column -- one trapezoid
                li = l0 + (j0 - g0)
                _dev.call("hydra_chol_update_rect", _addr(A, li, j0), n, _addr(A, li, k_lo), n,
                          pb_ptr, ldb, nloc - li, j1 - j0, 0, k_hi - k_lo, st)
            else:              # my rows in later blocks: the whole rectangle
                lb = lay.local_index(g1)
                if nloc > lb:
                    _dev.call("hydra_chol_update_rect", _addr(A, lb, j0), n, _addr(A, lb, k_lo), n,
                              pb_ptr, ldb, nloc - lb, j1 - j0, n, k_hi - k_lo, st)

    for isp, k0 in enumerate(range(0, n, SUPER_PANEL)):
        sp_end = min(k0 + SUPER_PANEL, n)
        spanel = spanels[isp % len(spanels)] if spanels is not None else None
        for k in range(k0, sp_end, PANEL):
            pb = min(PANEL, n - k)
            kend = k + pb
            blk = k // B
            o, l0 = lay.owner(blk), lay.local_start(blk) + (k - blk * B)
            lb = lay.local_index(kend)            # my first row below the diagonal block
            nbelow = nloc - lb
            # [diagonal block; my rows of the panel below it], contiguous with leading dimension pb
            if pb == PANEL:
                P = mine_ext[lb:lb + PANEL + nbelow]
            else:                                  # the last, narrower panel has no rows below
                P = torch.empty((pb, pb), dtype=torch.float64, device=dev)
            if rank == o:
                P[:pb].copy_(A[l0:l0 + pb, k:kend])
            if multi:
                comm.bcast_(P[:pb], root=o)
            if nbelow > 0:
                P[pb:].copy_(A[lb:, k:kend])
            _dev.call("hydra_chol_panel", _dev.ptr(P), pb, pb + nbelow, pb, _dev.ptr(info), st)
            bad |= info
            if rank == o:
                # the factor goes below the diagonal only: what is above it stays the operator
                # (ptsrc_sampler.SystemCache.matvec reads it after the factorisation)
                blkv = A[l0:l0 + pb, k:kend]
                blkv.copy_(torch.where(low[:pb, :pb], P[:pb], blkv))
            if nbelow > 0:
                A[lb:, k:kend].copy_(P[pb:])
            if n - kend <= 0:
                break
            # the whole panel on every rank (rank o's local row l at o * maxloc + l)
            if multi:
                comm.allgather_into_(panel, mine)
            if spanel is not None and n > sp_end:
                spanel[:, k - k0:kend - k0].copy_(panel)
            # the columns of this super-panel that are still to be factored
            update(kend, sp_end, k, kend, panel, PANEL)
        if n > sp_end:
            # everything right of the super-panel, once, with K = its width -- the columns of the
            # NEXT super-panel first, the rest on the side stream: the next super-panel's panel
            # steps, broadcasts and all-gathers (short kernels and transfers in a dependent chain)
            # then run under this update instead of after it.  Same kernels on the same elements
            # in the same order, so the factor does not depend on the split.
            nxt = min(sp_end + SUPER_PANEL, n)
            for sd in side or ():
                main.wait_stream(sd)         # the columns [sp_end, nxt) have taken the last update
            update(sp_end, nxt, k0, sp_end, spanel, SUPER_PANEL)
            if n > nxt:
                for sd in side or ():
                    sd.wait_stream(main)     # the gathered super-panel is complete
                update(nxt, n, k0, sp_end, spanel, SUPER_PANEL, streams=side)
    for sd in side or ():
        main.wait_stream(sd)
    if multi:
        comm.allreduce_(bad, op="max")
    if int(bad.item()) != 0:
        raise np.linalg.LinAlgError("matrix is not positive definite")
    return RowShardedFactor(comm, A, n)


# ------------------------------------------------------------------ 2-D block CG over ranks

def matvec_mpi(comm_row, mat_block, vec_block):
    """Row-block matrix-vector product summed over the ranks of a block row
    (linear_solver.py:11-49)."""
    A = _dev.f64(mat_block)
    v = _dev.f64(vec_block)
    y = torch.empty(A.shape[0], dtype=torch.float64, device=A.device)
    _gemv_rect(A, v, y)
    comm_row.allreduce_(y)
    return _dev.like(vec_block, y)


def _gemv_rect(A, v, y):
    n = A.shape[0]
    assert A.shape[0] == A.shape[1] == v.numel(), "blocks are square (linear_solver.py:132-134)"
    _dev.call("hydra_gemv", _dev.ptr(A), n, _dev.ptr(v), _dev.ptr(y), _dev.stream_ptr())


def setup_mpi_blocks(comm, matrix_shape, split=1):
    """Block layout and sub-communicators (linear_solver.py:52-137): returns
    ((active, row, col, diag) communicators, block_map, block_shape)."""
    assert len(matrix_shape) == 2, "'matrix_shape' must be a tuple with 2 entries"
    assert matrix_shape[0] == matrix_shape[1], "Only square matrices are currently supported"
    myid, nworkers = comm.Get_rank(), comm.Get_size()
    nblocks = split * split
    assert nworkers >= nblocks, "Specified more blocks than workers"
    workers = np.arange(nblocks).reshape((split, split))
    block_map = {int(w): (int(w) // split, int(w) % split) for w in workers.flatten()}
    # sub-groups must be created collectively by every rank (torch.distributed.new_group)
    comm_active = comm.Create(workers.flatten().tolist())
    rows = [comm.Create(workers[i, :].tolist()) for i in range(split)]
    cols = [comm.Create(workers[:, j].tolist()) for j in range(split)]
    comm_diag = comm.Create(np.diag(workers).tolist())
    if myid >= nblocks:
        return None, {}, {}
    myrow, mycol = block_map[myid]
    block_rows = int(np.ceil(matrix_shape[0] / split))
    block_cols = int(np.ceil(matrix_shape[1] / split))
    assert block_rows == block_cols, "Current implementation assumes that blocks are square"
    comms = (comm_active, rows[myrow], cols[mycol], comm_diag if myrow == mycol else None)
    return comms, block_map, (block_rows, block_cols)


def collect_linear_sys_blocks(comm, block_map, block_shape, Amat=None, bvec=None,
                              verbose=False):
    """Root slices (A, b) into zero-padded blocks and sends one to each active rank; the b
    block is indexed by the block COLUMN (linear_solver.py:140-235)."""
    myid = comm.Get_rank()
    used = sorted(block_map.keys())
    if myid not in used:
        return None, None
    br, bc = block_shape
    my_A = np.zeros((br, bc), dtype=np.float64)
    my_b = np.zeros((br,), dtype=np.float64)
    if myid == 0:
        Amat = np.asarray(Amat.cpu() if _dev.is_torch(Amat) else Amat)
        bvec = np.asarray(bvec.cpu() if _dev.is_torch(bvec) else bvec)
        for w in used:
            wrow, wcol = block_map[w]
            ii, jj = wrow * br, wcol * bc
            iip, jjp = min(ii + br, Amat.shape[0]), min(jj + bc, Amat.shape[1])
            Ab = np.zeros_like(my_A)
            bb = np.zeros_like(my_b)
            Ab[: iip - ii, : jjp - jj] = Amat[ii:iip, jj:jjp]
            bb[: jjp - jj] = bvec[jj:jjp]
            if w == 0:
                my_A, my_b = Ab, bb
            else:
                comm.Send(Ab, dest=w)
                comm.Send(bb, dest=w)
        if verbose:
            print("All send operations completed.")
    else:
        comm.Recv(my_A, source=0)
        comm.Recv(my_b, source=0)
        if verbose:
            print("Worker %d finished receive" % myid)
    return my_A, my_b


def cg_mpi(comm_groups, Amat_block, bvec_block, vec_size, block_map, maxiters=1000,
           abs_tol=1e-8):
    """Block-distributed CG (linear_solver.py:238-375): x0 = 0, one block of A per rank,
    row all-reduce for the product, column broadcast of p, scalar all-reduces over the
    diagonal ranks.  Returns the full solution on diagonal ranks, None elsewhere."""
    if comm_groups is None:
        return None
    comm_active, comm_row, comm_col, comm_diag = comm_groups
    myid = comm_active.Get_rank()
    myrow, mycol = block_map[myid]
    A = _dev.f64(Amat_block)
    r = _dev.f64(bvec_block).clone()
    # The reference's r_block / pvec_block start as views of bvec_block (linear_solver.py:300-301).
    # Diagonal ranks rebind them every iteration; off-diagonal ranks only ever write pvec_block
    # in place (:324-325), so there r_block IS the p received from the diagonal, and that is
    # what their convergence test (:309) sees.  Keep that aliasing: without it off-diagonal
    # ranks never report convergence and the iteration runs on into 0/0.
    p = r if myrow != mycol else r.clone()
    x = torch.zeros_like(r)
    y = torch.empty_like(r)
    niter = 0
    while niter < maxiters:
        conv = torch.all(torch.abs(r) < abs_tol).to(torch.int32).reshape(1)
        comm_active.allreduce_(conv, op="min")
        if int(conv.item()) != 0:
            break
        if myrow != mycol:
            p.zero_()
        comm_col.bcast_(p, root=mycol)      # group-rank of the diagonal member of this column
        _gemv_rect(A, p, y)
        comm_row.allreduce_(y)
        if myrow == mycol:
            sc = torch.stack([torch.dot(r, r), torch.dot(p, y)])
            comm_diag.allreduce_(sc)
            alpha = sc[0] / sc[1]
            x = x + alpha * p
            r = r - alpha * y
            rn = torch.dot(r, r).reshape(1)
            comm_diag.allreduce_(rn)
            p = r + (rn[0] / sc[0]) * p
        comm_active.barrier()
        niter += 1
    x_all = None
    if myrow == mycol:
        parts = comm_diag.allgather(x)
        x_all = torch.cat(parts)[:vec_size]
        x_all = _dev.like(bvec_block, x_all)
    comm_active.barrier()
    return x_all
