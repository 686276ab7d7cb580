"""
Device-memory plumbing between numpy / torch arrays and the C ABI.

PyTorch is used only to own HBM buffers and streams; every arithmetic step of the
hot path runs in libhydra_hip.so.  There is no CPU fallback: ``device()`` raises
when no GPU is visible.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_NP2TORCH = {
    np.dtype(np.float64): torch.float64,
    np.dtype(np.complex128): torch.complex128,
    np.dtype(np.int32): torch.int32,
    np.dtype(np.uint8): torch.uint8,
}


_have_gpu = False


def device():
    global _have_gpu
    if not _have_gpu:      # checked until it succeeds once; the iteration loops call this a lot
        if not torch.cuda.is_available():
            raise RuntimeError(
                "hydra_amd needs an AMD GPU (MI355X / gfx950): no HIP device is visible and the "
                "package has no CPU fallback.")
        _have_gpu = True
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def is_torch(x):
    return isinstance(x, torch.Tensor)


def to_dev(x, dtype):
    """Contiguous tensor of ``dtype`` on the current GPU from numpy / torch / sequence."""
    if isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == dtype and x.is_contiguous():
        return x           # already what the kernels take (the common case inside solver loops)
    dev = device()
    if isinstance(x, torch.Tensor):
        t = x.to(device=dev, dtype=dtype)
    else:
        arr = np.ascontiguousarray(np.asarray(x), dtype={torch.float64: np.float64,
                                                         torch.complex128: np.complex128,
                                                         torch.int32: np.int32,
                                                         torch.uint8: np.uint8}[dtype])
        t = torch.from_numpy(arr).to(dev)
    return t.contiguous()


def f64(x):
    return to_dev(x, torch.float64)


def c128(x):
    return to_dev(x, torch.complex128)


def i32(x):
    return to_dev(x, torch.int32)


def empty(shape, dtype):
    return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=device())


def zeros(shape, dtype):
    return torch.zeros(tuple(int(s) for s in shape), dtype=dtype, device=device())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    assert t.is_cuda and t.is_contiguous()
    return ctypes.c_void_p(t.data_ptr())


def host_ptr(arr):
    """Pointer to a contiguous numpy array (kept alive by the caller)."""
    return ctypes.c_void_p(arr.ctypes.data)


def workspace(nbytes):
    nbytes = max(int(nbytes), 16)
    return torch.empty(nbytes, dtype=torch.uint8, device=device())


def like(ref, t):
    """Return ``t`` (a GPU tensor) in the same kind as the caller's input ``ref``:
    torch tensors stay on the GPU, anything else gets a numpy array."""
    if isinstance(ref, torch.Tensor):
        return t
    return t.cpu().numpy()


def call(name, *args):
    return _lib.call(name, *args)
