"""Time hydra_gain_pairs_stream alone at config-4 size: python tools/gain_stream_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hydra_amd import _dev, gain_sampler as gs  # noqa: E402

nant, nft = 331, 4096
pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
nbl = len(pairs)
idx = gs._Index(np.array([p[0] for p in pairs], np.int32), np.array([p[1] for p in pairs], np.int32), nant)
dev = torch.device("cuda")
mv = torch.randn((nbl, nft), dtype=torch.complex128, device=dev)
w = 1.0 + torch.rand((nbl, nft), dtype=torch.float64, device=dev)
g = torch.randn((nant, nft), dtype=torch.complex128, device=dev)
y = torch.empty_like(g)
sw = torch.rand((nant, nft), dtype=torch.float64, device=dev)
st = _dev.stream_ptr()
ref = None
for _ in [0]:
    ft = nb = 0
    count = _dev._lib.query("hydra_gain_stream_weight_count", nant, nft)
    if count == 0:
        print(ft, nb, "does not fit")
        continue
    Wt = _dev.empty((count,), torch.float64)
    _dev.call("hydra_gain_weights_stream", _dev.ptr(mv), _dev.ptr(w), nbl, nant, nft,
              _dev.ptr(idx.pair_dev()), _dev.ptr(Wt), st)
    _dev.call("hydra_gain_pairs_stream", _dev.ptr(g), nant, nft, _dev.ptr(Wt), _dev.ptr(sw), _dev.ptr(y), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        _dev.call("hydra_gain_pairs_stream", _dev.ptr(g), nant, nft, _dev.ptr(Wt), _dev.ptr(sw), _dev.ptr(y), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    if ref is None:
        ref = y.clone()
    err = float((y - ref).abs().max() / ref.abs().max())
    print(f"{ms:.4f} ms  {8e-9 * nbl * nft / (ms * 1e-3):.0f} GB/s on 8M  relerr vs first {err:.1e}")
    del Wt
