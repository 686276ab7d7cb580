"""
``Comm``: the subset of the mpi4py communicator API that Hydra's GCR path uses
(Get_rank / Get_size / barrier / Bcast / Reduce / Allreduce / Send / Recv / Create),
on ``torch.distributed`` -- RCCL over xGMI when the process group is "nccl" (one
process per MI355X), gloo for CPU-only runs and tests.

Reference call sites replaced: ptsrc_sampler.py:86,132 (Reduce of Gram / RHS),
sh_sampler.py:420,439,465,485 (Bcast + Reduce per CG iteration),
linear_solver.py:48,121-124,222-230,313-367,473-508.

Buffers may be numpy arrays (mpi4py style, modified in place) or torch tensors.  With the
nccl backend host buffers are staged through the rank's GPU; with gloo GPU tensors are
staged through the host.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

SUM = "sum"
MIN = "min"
MAX = "max"
_OPS = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment
    (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT) and bind this process to
    its GPU.  Returns a world ``Comm`` (or None for a single-process run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return None
    if backend is None:
        # HYDRA_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsals on a 1-GPU box)
        backend = os.environ.get("HYDRA_DIST_BACKEND") or \
            ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend)
    return Comm()


class Comm:
    def __init__(self, group=None, ranks=None):
        self.group = group                     # None = world
        self.ranks = ranks                     # global ranks of the members (None = all)
        self.backend = dist.get_backend(group) if (group is not None or dist.is_initialized()) \
            else None

    # ---- mpi4py-style queries
    def Get_rank(self):
        return dist.get_rank(self.group)

    def Get_size(self):
        return dist.get_world_size(self.group)

    def barrier(self):
        dist.barrier(group=self.group)

    Barrier = barrier

    def _global(self, group_rank):
        return group_rank if self.ranks is None else self.ranks[group_rank]

    def Create(self, ranks):
        """Sub-communicator of the given (global) ranks; collective over the whole world, like
        ``comm.Create(comm.group.Incl(ranks))`` (linear_solver.py:121-124).  Returns None on
        ranks that are not members."""
        ranks = [int(r) for r in ranks]
        grp = dist.new_group(ranks=ranks)
        if dist.get_rank() not in ranks:
            return None
        return Comm(group=grp, ranks=ranks)

    # ---- staging
    def _stage(self, buf):
        """-> (tensor usable by the backend, write-back function)."""
        if isinstance(buf, np.ndarray):
            t = torch.from_numpy(buf)          # shares memory with the caller's array
            if self.backend == "nccl":
                d = t.cuda()
                return d, lambda: t.copy_(d.cpu())
            return t, lambda: None
        if buf.is_cuda and self.backend != "nccl":
            h = buf.cpu()
            return h, lambda: buf.copy_(h)
        if not buf.is_cuda and self.backend == "nccl":
            d = buf.cuda()
            return d, lambda: buf.copy_(d.cpu())
        return buf, lambda: None

    @staticmethod
    def _as_real(t):
        return torch.view_as_real(t) if t.is_complex() else t

    # ---- in-place tensor collectives (used by hydra_amd itself)
    def allreduce_(self, t, op="sum"):
        s, back = self._stage(t)
        dist.all_reduce(self._as_real(s), op=_OPS[op], group=self.group)
        back()
        return t

    def reduce_(self, t, root=0, op="sum"):
        s, back = self._stage(t)
        dist.reduce(self._as_real(s), dst=self._global(root), op=_OPS[op], group=self.group)
        if self.Get_rank() == root:
            back()
        return t

    def bcast_(self, t, root=0):
        s, back = self._stage(t)
        dist.broadcast(self._as_real(s), src=self._global(root), group=self.group)
        back()
        return t

    def allgather(self, t):
        s, _ = self._stage(t)
        parts = [torch.empty_like(s) for _ in range(self.Get_size())]
        dist.all_gather(parts, s, group=self.group)
        return [p.to(t.device) for p in parts] if isinstance(t, torch.Tensor) else parts

    # ---- row sharding of a dense operator (large-N solve)
    def row_range(self, n, rank=None):
        """Rows [r0, r1) of an n-row matrix owned by ``rank`` (default: this rank): blocks of
        ceil(n / size) rows, the last one shorter (possibly empty)."""
        size = self.Get_size()
        rank = self.Get_rank() if rank is None else rank
        per = -(-n // size)
        return min(rank * per, n), min((rank + 1) * per, n)

    def reduce_rows_(self, A):
        """Sum the (n, ncols) partial matrices of all ranks block row by block row, each block
        to the rank that owns it.  Returns this rank's rows; the rows of ``A`` are left in an
        unspecified state.

        RCCL: ONE reduce-scatter (``reduce_scatter_tensor``) when the rows divide evenly over the
        ranks -- every GPU then sends and receives on all of its xGMI links at once, where a
        loop of ``size`` rooted reduces serialises them (0.13 s against 0.9 s for
        the 80 GB Gram of config 5 on 8 GPUs).  The result is a new (n/size, ncols) tensor: the
        collective's output may not alias its input.  Otherwise (gloo, or a ragged last block)
        one in-place reduce per block on views of ``A``."""
        n = A.shape[0]
        size = self.Get_size()
        if self.backend == "nccl" and A.is_cuda and A.is_contiguous() and n % size == 0:
            out = torch.empty((n // size,) + tuple(A.shape[1:]), dtype=A.dtype, device=A.device)
            dist.reduce_scatter_tensor(out, A, op=dist.ReduceOp.SUM, group=self.group)
            return out
        for k in range(size):
            r0, r1 = self.row_range(n, k)
            if r1 > r0:
                self.reduce_(A[r0:r1], root=k)
        r0, r1 = self.row_range(n)
        return A[r0:r1]

    def _sym_packed(self, A):
        from . import _dev
        n = A.shape[0]
        assert A.is_cuda and A.is_contiguous() and tuple(A.shape) == (n, n)
        packed = torch.empty(n * (n + 1) // 2, dtype=torch.float64, device=A.device)
        _dev.call("hydra_pack_lower", _dev.ptr(A), n, _dev.ptr(packed), _dev.stream_ptr())
        return packed

    def _sym_unpack(self, packed, A):
        from . import _dev
        _dev.call("hydra_unpack_lower", _dev.ptr(packed), A.shape[0], _dev.ptr(A), _dev.stream_ptr())
        A.hydra_symmetric = True
        return A

    def reduce_sym_(self, A, root=0):
        """Sum-reduce an exactly symmetric GPU matrix to ``root`` with only its lower triangle on
        the wire (half the bytes of ``reduce_``); the root's ``A`` comes back mirrored, hence
        symmetric bit for bit whatever order the collective summed in."""
        packed = self._sym_packed(A)
        self.reduce_(packed, root=root)
        if self.Get_rank() == root:
            self._sym_unpack(packed, A)
        return A

    def allreduce_sym_(self, A):
        """All-reduce of an exactly symmetric GPU matrix through its packed lower triangle."""
        packed = self._sym_packed(A)
        self.allreduce_(packed)
        return self._sym_unpack(packed, A)

    def allgather_into_(self, out, part):
        """``out`` (size * m, ...) <- the ranks' equally shaped ``part`` (m, ...) in rank order:
        one ``all_gather_into_tensor`` on RCCL, no concatenation copy."""
        if self.backend == "nccl" and out.is_cuda and out.is_contiguous() and part.is_contiguous():
            dist.all_gather_into_tensor(out, part, group=self.group)
            return out
        s, _ = self._stage(part)
        parts = [torch.empty_like(s) for _ in range(self.Get_size())]
        dist.all_gather(parts, s, group=self.group)
        m = part.shape[0]
        for k, p in enumerate(parts):
            out[k * m:(k + 1) * m].copy_(p)
        return out

    def allgather_rows(self, part, n):
        """Concatenate the ranks' row-block vectors (lengths as in ``row_range``) into the full
        length-n vector on every rank."""
        size = self.Get_size()
        per = -(-n // size)
        pad = torch.zeros(per, dtype=part.dtype, device=part.device)
        pad[: part.numel()] = part
        s, _ = self._stage(pad)
        out = torch.empty(size * per, dtype=s.dtype, device=s.device)
        if self.backend == "nccl":
            dist.all_gather_into_tensor(out, s, group=self.group)
        else:
            dist.all_gather(list(out.split(per)), s, group=self.group)
        return out[:n].to(part.device)

    # ---- mpi4py-style buffer collectives
    def Bcast(self, buf, root=0):
        self.bcast_(buf, root)

    def bcast(self, obj, root=0):
        box = [obj]
        dist.broadcast_object_list(box, src=self._global(root), group=self.group)
        return box[0]

    def Reduce(self, sendbuf, recvbuf, op=SUM, root=0):
        tmp = np.array(sendbuf, copy=True) if isinstance(sendbuf, np.ndarray) else sendbuf.clone()
        self.reduce_(tmp, root, op if isinstance(op, str) else "sum")
        if self.Get_rank() == root:
            if isinstance(recvbuf, np.ndarray):
                recvbuf[...] = np.asarray(tmp).reshape(recvbuf.shape)
            else:
                recvbuf.copy_(tmp.reshape(recvbuf.shape))

    def Allreduce(self, sendbuf, recvbuf, op=SUM):
        tmp = np.array(sendbuf, copy=True) if isinstance(sendbuf, np.ndarray) else sendbuf.clone()
        self.allreduce_(tmp, op if isinstance(op, str) else "sum")
        if isinstance(recvbuf, np.ndarray):
            recvbuf[...] = np.asarray(tmp).reshape(recvbuf.shape)
        else:
            recvbuf.copy_(tmp.reshape(recvbuf.shape))

    def allreduce(self, value, op=SUM):
        t = torch.tensor([float(value)], dtype=torch.float64)
        self.allreduce_(t, op if isinstance(op, str) else "sum")
        return t.item()

    def Send(self, buf, dest):
        s, _ = self._stage(np.ascontiguousarray(buf) if isinstance(buf, np.ndarray) else buf)
        dist.send(self._as_real(s), dst=self._global(dest), group=self.group)

    def Recv(self, buf, source):
        s, back = self._stage(buf)
        dist.recv(self._as_real(s), src=self._global(source), group=self.group)
        back()

    def Allgather(self, sendbuf, recvbuf):
        parts = self.allgather(torch.from_numpy(np.ascontiguousarray(sendbuf))
                               if isinstance(sendbuf, np.ndarray) else sendbuf)
        flat = torch.cat([p.reshape(-1).cpu() for p in parts])
        if isinstance(recvbuf, np.ndarray):
            recvbuf[...] = flat.numpy().reshape(recvbuf.shape)
        else:
            recvbuf.copy_(flat.reshape(recvbuf.shape))
