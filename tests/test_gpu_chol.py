"""
Dense fallback solve (example.py:944, scipy.linalg.solve(assume_a='her')) on the GPU:
hydra_chol_solve (32-wide steps) and hydra_chol_solve_blocked (256-wide panels, trailing update
on the FP64 matrix cores) against scipy on the same SPD systems -- sizes around the panel and
tile edges, ill-scaled diagonals, the not-positive-definite flag, and the automatic switch
(incl. the odd-n padding of linear_solver.solve_dense).
"""
import numpy as np
import pytest
import scipy.linalg
import torch

from hydra_amd import _dev, _lib
from hydra_amd import linear_solver as ls

pytestmark = pytest.mark.gpu


def _spd(n, seed, cond_scale=False):
    rng = np.random.RandomState(seed)
    m = min(n, 300)
    B = rng.randn(n, m)
    A = B @ B.T / m + np.eye(n)              # GCR shape: identity + low-rank-ish Gram
    if cond_scale:                           # prior-preconditioned operators span decades
        d = 10.0 ** rng.uniform(-2, 2, n)
        A = d[:, None] * A * d[None, :]
    return A, rng.randn(n)


def _blocked(A, b):
    Ad = torch.from_numpy(A).cuda()
    bd = torch.from_numpy(b).cuda()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _dev.call("hydra_chol_solve_blocked", _dev.ptr(Ad), _dev.ptr(bd), A.shape[0], _dev.ptr(info),
              _dev.stream_ptr())
    torch.cuda.synchronize()
    return Ad.cpu().numpy(), bd.cpu().numpy(), int(info.item())


@pytest.mark.parametrize("n", [2, 30, 32, 34, 254, 256, 258, 384, 386, 512, 640, 1000, 1282, 2050])
def test_blocked_cholesky_matches_scipy(n):
    A, b = _spd(n, n)
    L, x, info = _blocked(A, b)
    assert info == 0
    ref = scipy.linalg.solve(A, b, assume_a="her")
    assert np.max(np.abs(x - ref)) <= 1e-11 * np.max(np.abs(ref))
    Lref = np.linalg.cholesky(A)
    assert np.max(np.abs(np.tril(L) - Lref)) <= 1e-12 * np.max(np.abs(Lref))
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)


def test_blocked_cholesky_ill_scaled_and_factor_only():
    n = 1536
    A, b = _spd(n, 7, cond_scale=True)
    L, x, info = _blocked(A, b)
    ref = scipy.linalg.solve(A, b, assume_a="her")
    assert info == 0
    assert np.linalg.norm(A @ x - b) <= 1e-10 * np.linalg.norm(b)     # north_star gate is 1e-8
    assert np.max(np.abs(x - ref) / (np.abs(ref) + 1e-300)) < 1e-6
    # factor only (b == NULL): same L
    Ad = torch.from_numpy(A).cuda()
    info_d = torch.zeros(1, dtype=torch.int32, device="cuda")
    _dev.call("hydra_chol_solve_blocked", _dev.ptr(Ad), None, n, _dev.ptr(info_d), _dev.stream_ptr())
    assert np.array_equal(np.tril(Ad.cpu().numpy()), np.tril(L))


def test_blocked_cholesky_flags_indefinite_and_rejects_odd_n():
    A, b = _spd(600, 3)
    A[450, 450] = -1.0
    _, _, info = _blocked(A, b)
    assert info == 1
    buf = torch.zeros(9, dtype=torch.float64, device="cuda")
    info_d = torch.zeros(1, dtype=torch.int32, device="cuda")
    with pytest.raises(_lib.HydraError, match="n even"):
        _dev.call("hydra_chol_solve_blocked", _dev.ptr(buf), _dev.ptr(buf), 3, _dev.ptr(info_d),
                  _dev.stream_ptr())
    with pytest.raises(np.linalg.LinAlgError):
        ls.solve_dense(A, b)


@pytest.mark.parametrize("n", [6144, 6401])
def test_solve_dense_takes_the_blocked_path_for_large_systems(n):
    """even n: hydra_chol_solve switches by itself; odd n: solve_dense pads to n+1."""
    A, b = _spd(n, n)
    x = ls.solve_dense(A, b)
    assert np.linalg.norm(A @ x - b) <= 1e-12 * np.linalg.norm(b)
    ref = scipy.linalg.solve(A, b, assume_a="pos")
    assert np.max(np.abs(x - ref)) <= 1e-11 * np.max(np.abs(ref))
    # GPU tensors in -> GPU tensor out, input matrix untouched unless overwrite_a
    Ad, bd = torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda()
    xd = ls.solve_dense(Ad, bd)
    assert xd.is_cuda and torch.equal(Ad.cpu(), torch.from_numpy(A))
    assert np.array_equal(xd.cpu().numpy(), x)
