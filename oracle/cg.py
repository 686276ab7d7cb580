"""
Oracle (test infrastructure): the reference's conjugate-gradient solver,
restating ``hydra/linear_solver.py:378-509`` (serial branch, comm=None) and the
2-D block layout of ``:52-235`` (PINNED: golden cg_*.npz from the reference and
the reference's own test ``hydra/tests/test_linear_solver.py:9-54``).

Algorithm: x = x0 or 0; r = b - A x; p = r; while it < maxiters:
test convergence FIRST (all |r_i| < tol, or ||r||_2 < tol); rr = r.r;
Ap = A p; alpha = rr / p.Ap; x += alpha p; r -= alpha Ap; beta = r.r / rr;
p = r + beta p.  No preconditioner, no restart, no status returned.
"""
import numpy as np


def cg(Amat, bvec, maxiters=1000, abs_tol=1e-8, use_norm_tol=False, x0=None,
       linear_op=None, return_iters=False):
    if linear_op is None:
        def linear_op(v):
            return Amat @ v
    if x0 is None:
        x = np.zeros_like(bvec)
    else:
        assert x0.shape == bvec.shape, "Initial guess x0 has a different shape to bvec"
        assert x0.dtype == bvec.dtype, "Initial guess x0 has a different type to bvec"
        x = x0.copy()
    r = bvec - linear_op(x)
    p = r
    it = 0
    while it < maxiters:
        if use_norm_tol:
            done = np.linalg.norm(r) < abs_tol
        else:
            done = np.all(np.abs(r) < abs_tol)
        if done:
            break
        rr = np.dot(r.T, r)
        Ap = linear_op(p)
        alpha = rr / (p.T @ Ap)
        x = x + alpha * p
        r = r - alpha * Ap
        beta = np.dot(r.T, r) / rr
        p = r + beta * p
        it += 1
    if return_iters:
        return x, it
    return x


def block_layout(n, split):
    """Block geometry of ``setup_mpi_blocks`` (linear_solver.py:101-131):
    worker w owns block (w // split, w % split) of size ceil(n/split)."""
    nb = int(np.ceil(n / split))
    block_map = {w: (w // split, w % split) for w in range(split * split)}
    return block_map, (nb, nb)


def slice_blocks(Amat, bvec, block_map, block_shape):
    """What ``collect_linear_sys_blocks`` hands each worker
    (linear_solver.py:184-223): zero-padded A block, and the b block indexed by
    the block COLUMN."""
    br, bc = block_shape
    out = {}
    for w, (row, col) in block_map.items():
        ii, jj = row * br, col * bc
        iip, jjp = min(ii + br, Amat.shape[0]), min(jj + bc, Amat.shape[1])
        Ab = np.zeros((br, bc), dtype=bvec.dtype)
        bb = np.zeros((br,), dtype=bvec.dtype)
        Ab[: iip - ii, : jjp - jj] = Amat[ii:iip, jj:jjp]
        bb[: jjp - jj] = bvec[jj:jjp]
        out[w] = (Ab, bb)
    return out


def cg_blocked(Amat, bvec, split, maxiters=1000, abs_tol=1e-8):
    """Serial emulation of ``cg_mpi`` (linear_solver.py:238-375): same block
    partition, same reduction grouping (row-sum of block GEMVs, per-block dot
    products summed over the diagonal), x0 = 0."""
    n = bvec.size
    block_map, (nb, _) = block_layout(n, split)
    blocks = slice_blocks(Amat, bvec, block_map, (nb, nb))
    x = [np.zeros(nb, dtype=bvec.dtype) for _ in range(split)]
    r = [blocks[d * split + d][1].copy() for d in range(split)]
    p = [rb.copy() for rb in r]
    it = 0
    while it < maxiters:
        if all(np.all(np.abs(rb) < abs_tol) for rb in r):
            break
        Ap = []
        for row in range(split):
            acc = np.zeros(nb, dtype=bvec.dtype)
            for col in range(split):
                acc = acc + blocks[row * split + col][0] @ p[col]
            Ap.append(acc)
        rr = sum(np.dot(rb.T, rb) for rb in r)
        pAp = sum(np.dot(pb.T, ab) for pb, ab in zip(p, Ap))
        alpha = rr / pAp
        x = [xb + alpha * pb for xb, pb in zip(x, p)]
        r = [rb - alpha * ab for rb, ab in zip(r, Ap)]
        rr_new = sum(np.dot(rb.T, rb) for rb in r)
        beta = rr_new / rr
        p = [rb + beta * pb for rb, pb in zip(r, p)]
        it += 1
    return np.concatenate(x)[:n]
