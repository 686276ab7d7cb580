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


#: rows per block of the block-cyclic row layout of a sharded operator (= the super-panel width of
#: the row-sharded factorisation, a multiple of its 256-wide panels)
ROW_BLOCK = 2048


class RowLayout:
    """Block-cyclic rows of an n-row matrix over ``size`` ranks: global row block q (``block``
    rows, the last one shorter) belongs to rank q % size, and a rank keeps its blocks one after
    the other in ascending order.  A right-looking factorisation works on the rows below the
    current column: with contiguous row blocks the last rank had 2 - 1/size times the mean work
    and the first ones ran dry early; cyclic blocks give every rank the same share at every
    stage.  ``block`` is a multiple of 256 (a 256-wide diagonal block never straddles two owners):
    2048 when every rank gets at least four such blocks, smaller for small matrices."""

    def __init__(self, n, size, rank, block=None):
        self.n, self.size, self.rank = int(n), int(size), int(rank)
        if block is None:
            block = 256 * max(1, min(ROW_BLOCK // 256, self.n // (self.size * 4 * 256)))
        assert block % 256 == 0
        self.block = int(block)
        self.nblocks = -(-self.n // self.block)
        self._rows = {}

    def owner(self, blk):
        return blk % self.size

    def block_range(self, blk):
        """Global rows [g0, g1) of block ``blk``."""
        return blk * self.block, min((blk + 1) * self.block, self.n)

    def local_start(self, blk):
        """First local row of block ``blk`` on its owner (all of an owner's earlier blocks are full)."""
        return (blk // self.size) * self.block

    def blocks_of(self, rank=None):
        return range(self.rank if rank is None else rank, self.nblocks, self.size)

    def nloc(self, rank=None):
        return sum(g1 - g0 for g0, g1 in map(self.block_range, self.blocks_of(rank)))

    @property
    def maxloc(self):
        return max(self.nloc(k) for k in range(self.size))

    def rows(self, rank=None):
        """Global indices of a rank's rows in local order (numpy int64, ascending)."""
        rank = self.rank if rank is None else rank
        if rank not in self._rows:
            parts = [np.arange(*self.block_range(b), dtype=np.int64) for b in self.blocks_of(rank)]
            self._rows[rank] = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        return self._rows[rank]

    def local_index(self, g):
        """Number of this rank's rows with global index < g."""
        return int(np.searchsorted(self.rows(), g))


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
    def row_layout(self, n, block=None):
        """The block-cyclic row layout of an n-row operator over this communicator's ranks."""
        key = (int(n), block)
        cache = self.__dict__.setdefault("_layouts", {})
        if key not in cache:
            cache[key] = RowLayout(n, self.Get_size(), self.Get_rank(), block)
        return cache[key]

    def reduce_rows_(self, A):
        """Sum the (n, ncols) partial matrices of all ranks, each row block to the rank that owns
        it in ``row_layout(n)``.  Returns this rank's rows (nloc, ncols) as a new tensor; the rows of
        ``A`` are left in an unspecified state.

        RCCL: one reduce-scatter (``reduce_scatter_tensor``) per ROUND of ``size`` consecutive full
        blocks -- a round is contiguous in ``A`` and its k-th piece belongs to rank k -- so every GPU
        sends and receives on all of its xGMI links at once, where a loop of rooted reduces
        serialises them (0.13 s against 0.9 s for the 80 GB Gram of config 5 on 8
        GPUs); the collective writes straight into the rank's rows.  Otherwise (gloo, or the ragged
        last round) one rooted reduce per block."""
        n = A.shape[0]
        lay = self.row_layout(n)
        size, rank, B = lay.size, lay.rank, lay.block
        out = torch.empty((lay.nloc(),) + tuple(A.shape[1:]), dtype=A.dtype, device=A.device)
        fast = self.backend == "nccl" and A.is_cuda and A.is_contiguous()
        for rd in range(-(-lay.nblocks // size)):
            b0 = rd * size
            full = (b0 + size) * B <= n
            if fast and full:
                dist.reduce_scatter_tensor(out[rd * B:(rd + 1) * B], A[b0 * B:(b0 + size) * B],
                                           op=dist.ReduceOp.SUM, group=self.group)
                continue
            for blk in range(b0, min(b0 + size, lay.nblocks)):
                g0, g1 = lay.block_range(blk)
                self.reduce_(A[g0:g1], root=lay.owner(blk))
                if lay.owner(blk) == rank:
                    l0 = lay.local_start(blk)
                    out[l0:l0 + g1 - g0].copy_(A[g0:g1])
        return out

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
        """The full length-n vector, on every rank, from the ranks' pieces in ``row_layout(n)``
        (``part``: this rank's rows in local order)."""
        lay = self.row_layout(n)
        size, per = lay.size, max(lay.maxloc, 1)
        pad = torch.zeros(per, dtype=part.dtype, device=part.device)
        pad[: part.numel()] = part
        s, _ = self._stage(pad)
        got = torch.empty(size * per, dtype=s.dtype, device=s.device)
        if self.backend == "nccl":
            dist.all_gather_into_tensor(got, s, group=self.group)
        else:
            dist.all_gather(list(got.split(per)), s, group=self.group)
        got = got.to(part.device)
        out = torch.empty(n, dtype=part.dtype, device=part.device)
        for k in range(size):
            idx = self._rows_dev(lay, k, part.device)
            if idx.numel():
                out[idx] = got[k * per:k * per + idx.numel()]
        return out

    def _rows_dev(self, lay, rank, device):
        cache = self.__dict__.setdefault("_rows_dev_cache", {})
        key = (lay.n, lay.block, rank, str(device))
        if key not in cache:
            cache[key] = torch.as_tensor(lay.rows(rank), device=device)
        return cache[key]

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
