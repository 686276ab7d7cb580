"""Times hydra_adjoint_V / hydra_apply_V (antenna-space kernels) for a few array sizes.
usage: python tools/adjoint_bench.py [nant ...]   (HYDRA_HIP_LIBRARY selects a library variant)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

import hydra_amd  # noqa: F401
from hydra_amd import _dev


def run(nant, ns=3000, nf=30, nt=20, reps=5):
    rng = np.random.RandomState(0)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    bl_i = _dev.i32(np.array([p[0] for p in pairs], dtype=np.int32))
    bl_j = _dev.i32(np.array([p[1] for p in pairs], dtype=np.int32))
    nbl = len(pairs)
    V = _dev.c128(rng.randn(nf, nt, nant, ns) + 1j * rng.randn(nf, nt, nant, ns))
    z = _dev.c128(rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt))
    x = _dev.f64(rng.randn(ns))
    out = _dev.empty((ns,), torch.float64)
    vis = _dev.empty((nbl, nf, nt), torch.complex128)
    ws = _dev.workspace(_dev._lib.query("hydra_adjoint_V_workspace_bytes", nf, nt, nbl, ns))
    ws2 = _dev.workspace(_dev._lib.query("hydra_apply_V_workspace_bytes", nf, nt, nant, ns))

    def adj():
        _dev.call("hydra_adjoint_V", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(bl_i), _dev.ptr(bl_j),
                  nbl, _dev.ptr(z), 0, 0, _dev.ptr(out), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())

    big = 64 < nant <= 160
    if big:
        ws3 = _dev.workspace(_dev._lib.query("hydra_adjoint_V_antspace_workspace_bytes", nf, nt, nant,
                                             nbl, ns))

    def adj_big():
        _dev.call("hydra_adjoint_V_antspace", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(bl_i),
                  _dev.ptr(bl_j), nbl, _dev.ptr(z), 0, 0, _dev.ptr(out), _dev.ptr(ws3), ws3.numel(),
                  _dev.stream_ptr())

    def app():
        _dev.call("hydra_apply_V", _dev.ptr(V), nf, nt, nant, ns, _dev.ptr(bl_i), _dev.ptr(bl_j),
                  nbl, _dev.ptr(x), 0, _dev.ptr(vis), _dev.ptr(ws2), ws2.numel(), _dev.stream_ptr())

    res = {}
    for name, fn in (("adjoint", adj_big if big else adj), ("apply", app)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / reps * 1e3
    gb = nf * nt * nant * ns * 16 / 1e9
    print(f"nant={nant} ns={ns} nft={nf * nt}: adjoint {res['adjoint']:.2f} ms, apply {res['apply']:.2f} ms "
          f"(V = {gb:.2f} GB: {gb / res['adjoint'] * 1e3:.0f} / {gb / res['apply'] * 1e3:.0f} GB/s)",
          flush=True)


def trace():
    """per-wave timeline of adjoint_V_big_kernel (stamped build: make -C hydra_amd/csrc stamps,
    HYDRA_HIP_LIBRARY=hydra_amd/libhydra_hip_stamps.so)"""
    import ctypes
    from hydra_amd import _lib
    lib = _lib.load()
    if not hasattr(lib, "hydra_debug_adj_trace"):
        return
    tr = (ctypes.c_ulonglong * 64)()
    if lib.hydra_debug_adj_trace(tr) != 0:
        return
    t = np.array(list(tr), dtype=np.int64).reshape(8, 8)[:5, :4]
    t0 = t[:, 0].min()
    print("adjoint_V_big_kernel, workgroup 0, third (f,t): cycles from the first barrier arrival")
    print("  wave  at-barrier  past-barrier  first-row-block-done  all-done")
    for w in range(5):
        r = t[w] - t0
        print(f"  {w:4d}  {r[0]:10d}  {r[1]:12d}  {r[2]:20d}  {r[3]:8d}")


if __name__ == "__main__":
    # sizes: ADJ_NS / ADJ_NF / ADJ_NT (config 3 at one LST block: 12288 / 128 / 1)
    kw = dict(ns=int(os.environ.get("ADJ_NS", 3000)), nf=int(os.environ.get("ADJ_NF", 30)),
              nt=int(os.environ.get("ADJ_NT", 20)))
    for n in [int(a) for a in sys.argv[1:]] or [40, 56, 64]:
        run(n, **kw)
    trace()
