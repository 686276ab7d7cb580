"""Time linear_solver.solve_dense_rows (row-sharded Cholesky) on the ranks it is launched with:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/chol_rows_bench.py [n ...]
(one rank on a 1-GPU box: the orchestration overhead against hydra_chol_solve_blocked)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from hydra_amd import _dev, _dist, linear_solver

torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29531")):   # started bare: one rank
    os.environ.setdefault(key, val)
dist.init_process_group(backend=os.environ.get("HYDRA_DIST_BACKEND", "nccl"))
comm = _dist.Comm()
rank, size = comm.Get_rank(), comm.Get_size()
for n in [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [16384, 32768]:
    lay = comm.row_layout(n)
    mine = torch.as_tensor(lay.rows(), device="cuda")
    g = torch.Generator(device="cuda").manual_seed(n)
    B = torch.randn(n, 512, dtype=torch.float64, device="cuda", generator=g)
    rows = (B[mine] @ B.T) / 512
    rows[torch.arange(mine.numel(), device="cuda"), mine] += 1.0
    b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    ref = rows.clone()
    torch.cuda.synchronize(); comm.barrier()
    t0 = time.perf_counter()
    x = linear_solver.solve_dense_rows(comm, rows, b, overwrite_a=True)
    torch.cuda.synchronize(); comm.barrier()
    dt = time.perf_counter() - t0
    res = ref @ x - b[mine]
    r2 = torch.sum(res * res).reshape(1)
    comm.allreduce_(r2)
    if rank == 0:
        fl = n ** 3 / 3
        print(f"n={n} ranks={size}: {dt*1e3:.1f} ms = {fl/dt/1e12:.1f} TFLOP/s, rel residual "
              f"{float(torch.sqrt(r2) / torch.linalg.norm(b)):.2e}", flush=True)
    del rows, ref
    if size == 1 and "--compare" in sys.argv:      # the in-place one-GPU routine on the same matrix
        A = (B @ B.T) / 512
        A.diagonal().add_(1.0)
        b1 = b.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        linear_solver.solve_dense(A, b1, overwrite_a=True)
        torch.cuda.synchronize()
        d1 = time.perf_counter() - t0
        print(f"n={n} one-GPU in-place routine: {d1*1e3:.1f} ms; row-sharded form on one rank "
              f"{dt / d1:.2f} x", flush=True)
        del A
    del B
dist.destroy_process_group()
