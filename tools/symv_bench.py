"""Times the symmetric half-read product (hydra_symv_batched) against the full-read one
(hydra_gemv_batched) and checks them against each other.
usage: python tools/symv_bench.py [n:batch ...]      e.g. 1024:64 16384:1 65536:1"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import torch

import hydra_amd  # noqa: F401
from hydra_amd import _dev


def run(n, nb, reps=10):
    torch.manual_seed(0)
    A = torch.empty((nb, n, n), dtype=torch.float64, device="cuda")
    for q in range(nb):                      # symmetric, built in place (no second n x n copy)
        A[q].normal_()
        for i0 in range(0, n, 2048):         # mirror the lower triangle, a strip of rows at a time
            i1 = min(i0 + 2048, n)
            blk = A[q, i0:i1, i0:i1]
            blk.copy_(torch.tril(blk) + torch.tril(blk, -1).T)
            if i1 < n:
                A[q, i0:i1, i1:] = A[q, i1:, i0:i1].T
    x = torch.randn((nb, n), dtype=torch.float64, device="cuda")
    y0 = torch.empty_like(x)
    y1 = torch.empty_like(x)
    ws = _dev.workspace(_dev._lib.query("hydra_symv_workspace_bytes", n, nb))

    def full():
        _dev.call("hydra_gemv_batched", _dev.ptr(A), n, nb, _dev.ptr(x), _dev.ptr(y0), 0, _dev.stream_ptr())

    def half():
        _dev.call("hydra_symv_batched", _dev.ptr(A), n, nb, _dev.ptr(x), _dev.ptr(y1), 0, _dev.ptr(ws),
                  ws.numel(), _dev.stream_ptr())

    nr = max(1, n // 8)          # a rank's row block at 8 GPUs (first matrix of the stack)
    y2 = torch.empty(nr, dtype=torch.float64, device="cuda")

    def rows():
        _dev.call("hydra_gemv_rows", _dev.ptr(A), nr, n, _dev.ptr(x), _dev.ptr(y2), 0, _dev.stream_ptr())

    res = {}
    for name, fn in (("full", full), ("half", half), ("rows", rows)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps
    err = float((y0 - y1).abs().max() / y0.abs().max())
    gb = nb * n * n * 8 / 1e9
    print(f"n={n} x{nb}: full {res['full'] * 1e3:.3f} ms ({gb / res['full'] / 1e3:.2f} TB/s), "
          f"half {res['half'] * 1e3:.3f} ms ({gb / 2 / res['half'] / 1e3:.2f} TB/s on half the bytes), "
          f"rows/8 {res['rows'] * 1e3:.3f} ms ({gb / nb / 8 / res['rows'] / 1e3:.2f} TB/s), "
          f"max rel diff {err:.1e}", flush=True)
    assert err < 1e-12
    assert float((y2 - y0[0, :nr]).abs().max() / y0.abs().max()) < 1e-12


if __name__ == "__main__":
    for a in sys.argv[1:] or ["1024:64", "2000:1", "16384:1", "65536:1"]:
        n, nb = a.split(":")
        run(int(n), int(nb))
