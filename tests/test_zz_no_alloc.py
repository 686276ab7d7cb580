"""
hydra_symv_static -- the one entry point that allocates nothing (include/hydra_hip.h): its
scratch (the per-block-column pieces of the symmetric product) is a device array inside the
library instead of a caller workspace.  Checked through the C ABI against the existing entry
point hydra_symv_batched (same kernels, so bit for bit) on two calls in a row, the second with
a smaller matrix (the array keeps the first call's contents), and for the size limit.

Collected last (test_zz_*) so that a failure here cannot hide other tests under `pytest -x`.
"""
import numpy as np
import pytest
import torch

from hydra_amd import _dev, _lib

pytestmark = pytest.mark.gpu


def _sym(n, seed):
    rng = np.random.RandomState(seed)
    a = rng.randn(n, n)
    return torch.from_numpy(a + a.T).cuda(), torch.from_numpy(rng.randn(n)).cuda()


def _batched(A, x, accumulate, y0):
    n = x.numel()
    ws = _dev.workspace(_lib.query("hydra_symv_workspace_bytes", n, 1))
    y = y0.clone()
    _dev.call("hydra_symv_batched", _dev.ptr(A), n, 1, _dev.ptr(x), _dev.ptr(y), accumulate,
              _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
    return y


def _static(A, x, accumulate, y0):
    y = y0.clone()
    _dev.call("hydra_symv_static", _dev.ptr(A), x.numel(), _dev.ptr(x), _dev.ptr(y), accumulate,
              _dev.stream_ptr())
    return y


def test_static_symv_matches_workspace_symv_two_calls_second_smaller():
    assert _lib.query("hydra_symv_static_max_n") == 8192
    # first call: large (odd n takes the scalar-load kernel), second: smaller, third: max size
    for n, seed, acc in ((5001, 1, 0), (1234, 2, 1), (300, 3, 0), (8192, 4, 0), (65, 5, 1)):
        A, x = _sym(n, seed)
        y0 = torch.from_numpy(np.random.RandomState(seed + 100).randn(n)).cuda()
        want = _batched(A, x, acc, y0)
        got = _static(A, x, acc, y0)
        torch.cuda.synchronize()
        assert torch.equal(got, want), (n, float((got - want).abs().max()))
        # and both are the product (fp64 rounding of a different summation order)
        ref = A @ x + (y0 if acc else 0.0)
        assert float((got - ref).abs().max()) <= 1e-11 * float(ref.abs().max()) * np.sqrt(n)


def test_static_symv_rejects_sizes_beyond_the_scratch():
    n = 8192 + 64
    buf = torch.zeros(16, dtype=torch.float64, device="cuda")
    with pytest.raises(_lib.HydraError, match="exceeds the static scratch"):
        _dev.call("hydra_symv_static", _dev.ptr(buf), n, _dev.ptr(buf), _dev.ptr(buf), 0,
                  _dev.stream_ptr())


def test_static_symv_replays_from_a_captured_graph():
    """The callers the entry point is for: under HIP graph capture nothing may allocate or look
    symbols up.  After the set-up call (hydra_symv_static_max_n resolves the scratch address on
    this device) a call is two kernel launches, which a stream capture records and a graph
    replays -- with new contents in the same buffers -- bit for bit like the workspace form."""
    assert _lib.query("hydra_symv_static_max_n") == 8192          # set-up, outside the capture
    n = 1500
    A, x = _sym(n, 11)
    y = torch.zeros(n, dtype=torch.float64, device="cuda")
    _static(A, x, 0, y)                                           # (warm: kernels loaded)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            _dev.call("hydra_symv_static", _dev.ptr(A), n, _dev.ptr(x), _dev.ptr(y), 0,
                      _dev.stream_ptr())
    torch.cuda.current_stream().wait_stream(side)
    for seed in (12, 13):                                          # new inputs, same buffers
        A2, x2 = _sym(n, seed)
        A.copy_(A2)
        x.copy_(x2)
        y.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        want = _batched(A, x, 0, torch.zeros_like(y))
        assert torch.equal(y, want), float((y - want).abs().max())
