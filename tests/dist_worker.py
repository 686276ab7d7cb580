"""
Worker run under torch.distributed.run by tests/test_dist.py (gloo backend, 127.0.0.1).

    python -m torch.distributed.run --nproc-per-node N tests/dist_worker.py <case>

Cases "comm" and "blocks" need no GPU; "ptsrc", "blshard", "rowshard", "cosmo" and "cgmpi" run every rank on GPU 0 of the box
(collectives staged through the host by hydra_amd._dist.Comm) and compare with a single-process
run of the same path.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from hydra_amd import _dist  # noqa: E402


def case_comm(comm):
    rank, size = comm.Get_rank(), comm.Get_size()
    # Bcast (numpy, in place) -- sh_sampler.py:420,465
    buf = np.arange(5, dtype=np.float64) * (rank + 1)
    comm.Bcast(buf, root=0)
    assert np.array_equal(buf, np.arange(5.0))
    # Reduce (SUM to root) -- ptsrc_sampler.py:86,132
    send = np.full((3, 3), rank + 1.0)
    recv = np.zeros((3, 3))
    comm.Reduce(send, recv, op=_dist.SUM, root=0)
    if rank == 0:
        assert np.all(recv == sum(range(1, size + 1)))
    else:
        assert np.all(recv == 0.0)
    assert np.all(send == rank + 1.0), "send buffer must not be modified"
    # Allreduce, complex -- gain_sampler.py:226
    z = np.full(4, (rank + 1) * (1 + 2j))
    out = np.zeros(4, dtype=complex)
    comm.Allreduce(z, out, op=_dist.SUM)
    assert np.allclose(out, sum(range(1, size + 1)) * (1 + 2j))
    # scalar allreduce -- linear_solver.py:335
    assert comm.allreduce(rank + 1.0, op=_dist.SUM) == sum(range(1, size + 1))
    # in-place tensor collectives used by hydra_amd itself
    t = torch.full((4,), float(rank))
    comm.allreduce_(t, op="max")
    assert torch.all(t == size - 1)
    t = torch.full((4,), float(rank + 1))
    comm.reduce_(t, root=0)
    if rank == 0:
        assert torch.all(t == sum(range(1, size + 1)))
    # Send / Recv -- linear_solver.py:222-230
    if size >= 2:
        if rank == 0:
            comm.Send(np.arange(6, dtype=np.float64).reshape(2, 3), dest=1)
        elif rank == 1:
            got = np.zeros((2, 3))
            comm.Recv(got, source=0)
            assert np.array_equal(got, np.arange(6.0).reshape(2, 3))
    # sub-communicator -- linear_solver.py:121-124
    sub = comm.Create([0, size - 1])
    if rank in (0, size - 1):
        assert sub is not None and sub.Get_size() == len({0, size - 1})
        v = np.array([float(rank)])
        sub.Bcast(v, root=sub.Get_size() - 1)       # group-rank root
        assert v[0] == size - 1
    else:
        assert sub is None
    parts = comm.allgather(torch.tensor([float(rank)]))
    assert [float(p.item()) for p in parts] == [float(r) for r in range(size)]
    obj = comm.bcast({"k": rank}, root=0)
    assert obj == {"k": 0}
    # row sharding of a dense operator: block row k of the sum lands on rank k, vectors come back
    # whole (n not a multiple of the world size; a rank may own no row at all)
    for n in (7, size - 1 if size > 1 else 1):
        ranges = [comm.row_range(n, k) for k in range(size)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(ranges[k][1] == ranges[k + 1][0] for k in range(size - 1))
        A = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3) * (rank + 1)
        mine = comm.reduce_rows_(A)
        r0, r1 = comm.row_range(n)
        want = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3)[r0:r1] * sum(range(1, size + 1))
        assert mine.shape == want.shape and torch.equal(mine, want)
        full = comm.allgather_rows(torch.arange(r0, r1, dtype=torch.float64), n)
        assert torch.equal(full, torch.arange(n, dtype=torch.float64))
    comm.barrier()


def case_blocks(comm):
    """setup_mpi_blocks / collect_linear_sys_blocks (linear_solver.py:52-235) against the
    oracle's block slicing; world size 4, split 2."""
    from hydra_amd import linear_solver
    from oracle import cg as ocg
    rank = comm.Get_rank()
    n, split = 11, 2
    rng = np.random.RandomState(3)
    A = rng.randn(n, n)
    A = A @ A.T + n * np.eye(n)
    b = rng.randn(n)
    comms, block_map, block_shape = linear_solver.setup_mpi_blocks(comm, (n, n), split=split)
    ref_map, ref_shape = ocg.block_layout(n, split)
    assert block_map == ref_map and tuple(block_shape) == tuple(ref_shape)
    my_A, my_b = linear_solver.collect_linear_sys_blocks(
        comm, block_map, block_shape, A if rank == 0 else None, b if rank == 0 else None)
    blocks = ocg.slice_blocks(A, b, ref_map, ref_shape)
    assert np.array_equal(my_A, blocks[rank][0]) and np.array_equal(my_b, blocks[rank][1])
    active, row, col, diag = comms
    myrow, mycol = block_map[rank]
    assert row.Get_size() == split and col.Get_size() == split
    assert (diag is not None) == (myrow == mycol)
    comm.barrier()


def _ptsrc_inputs():
    from hydra_amd import example, utils
    np.random.seed(7)
    ant_pos = utils.build_hex_array(hex_spec=(3, 4), d=14.6)
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    ns, nf, nt = 40, 4, 3
    ra, dec, amps = example.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fluxes = utils.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    rng = np.random.RandomState(11)
    nbl = len(antpairs)
    gains = ((1.0 + 1.0j) + 0.05 * (rng.randn(nant, 1, 1) + 1j * rng.randn(nant, 1, 1))) \
        * np.ones((nant, nf, nt))
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    inv_noise = 300.0 + 100.0 * rng.rand(nbl, nf, nt)
    om_n = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    om_a = rng.randn(ns)
    prior = 0.1 + 0.05 * rng.rand(ns)
    return dict(ant_pos=ant_pos, antpairs=antpairs, ra=ra, dec=dec, freqs=freqs, times=times,
                fluxes=fluxes, gains=gains, data=data, inv_noise=inv_noise, om_n=om_n, om_a=om_a,
                prior=prior, ants=np.arange(nant))


def case_ptsrc(comm):
    """Frequency-sharded precompute_mpi over 2 ranks == single-process result (the Reduce of
    ptsrc_sampler.py:86,132), for the factored and the materialised operator."""
    import hydra_amd as hy
    rank, size = comm.Get_rank(), comm.Get_size()
    p = _ptsrc_inputs()
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in p["ant_pos"]]
    fidx = np.array_split(np.arange(p["freqs"].size), size)[rank]
    for materialize in (False, True):
        full = hy.ptsrc_sampler.calc_proj_operator(p["ra"], p["dec"], p["fluxes"], p["ant_pos"],
                                                   p["antpairs"], p["freqs"], p["times"], beams,
                                                   materialize=materialize)
        A1, b1 = hy.ptsrc_sampler.precompute_mpi(
            None, p["ants"], p["antpairs"], p["freqs"], p["times"], full, p["data"],
            p["inv_noise"], p["gains"], p["prior"], omega_n=p["om_n"], omega_a=p["om_a"])
        mine = hy.ptsrc_sampler.calc_proj_operator(
            p["ra"], p["dec"], p["fluxes"][:, fidx], p["ant_pos"], p["antpairs"],
            p["freqs"][fidx], p["times"], beams, materialize=materialize)
        A2, b2 = hy.ptsrc_sampler.precompute_mpi(
            comm, p["ants"], p["antpairs"], p["freqs"][fidx], p["times"], mine,
            p["data"][:, fidx], p["inv_noise"][:, fidx], p["gains"][:, fidx], p["prior"],
            omega_n=p["om_n"][:, fidx], omega_a=p["om_a"])
        if rank == 0:
            ea = np.max(np.abs(A2 - A1)) / np.max(np.abs(A1))
            eb = np.max(np.abs(b2 - b1)) / np.max(np.abs(b1))
            assert ea < 1e-12 and eb < 1e-12, (materialize, ea, eb)
            x = hy.linear_solver.cg(A2, b2, maxiters=500, abs_tol=1e-10, comm=comm)
        else:
            x = hy.linear_solver.cg(np.eye(p["prior"].size), np.zeros(p["prior"].size),
                                    maxiters=1, comm=comm)
        x_ref = np.linalg.solve(A1, b1)
        assert np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref)) < 1e-8   # every rank holds x
    comm.barrier()


def case_blshard(comm):
    """Baseline-sharded precompute_mpi (rank r owns antpairs[r::size], all channels): the
    all-reduced Gram / RHS equal the single-process system (BASELINE config 5's partitioning)."""
    import hydra_amd as hy
    rank, size = comm.Get_rank(), comm.Get_size()
    p = _ptsrc_inputs()
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in p["ant_pos"]]
    full = hy.ptsrc_sampler.calc_proj_operator(p["ra"], p["dec"], p["fluxes"], p["ant_pos"],
                                               p["antpairs"], p["freqs"], p["times"], beams,
                                               materialize=False)
    A1, b1 = hy.ptsrc_sampler.precompute_mpi(
        None, p["ants"], p["antpairs"], p["freqs"], p["times"], full, p["data"], p["inv_noise"],
        p["gains"], p["prior"], omega_n=p["om_n"], omega_a=p["om_a"])
    sel = np.arange(len(p["antpairs"]))[rank::size]
    mypairs = [p["antpairs"][k] for k in sel]
    mine = hy.ptsrc_sampler.calc_proj_operator(p["ra"], p["dec"], p["fluxes"], p["ant_pos"],
                                               mypairs, p["freqs"], p["times"], beams,
                                               materialize=False)
    assert not mine.complete
    A2, b2 = hy.ptsrc_sampler.precompute_mpi(
        comm, p["ants"], mypairs, p["freqs"], p["times"], mine, p["data"][sel],
        p["inv_noise"][sel], p["gains"], p["prior"], omega_n=p["om_n"][sel], omega_a=p["om_a"])
    if rank == 0:
        ea = np.max(np.abs(A2 - A1)) / np.max(np.abs(A1))
        eb = np.max(np.abs(b2 - b1)) / np.max(np.abs(b1))
        assert ea < 1e-12 and eb < 1e-12, (ea, eb)
    comm.barrier()


def case_cgmpi(comm):
    """cg_mpi on a 2x2 block grid (linear_solver.py:238-375) == dense solve."""
    from hydra_amd import linear_solver
    rank = comm.Get_rank()
    n, split = 203, 2
    rng = np.random.RandomState(5)
    A = rng.randn(n, n)
    A = np.eye(n) + 0.1 * 0.5 * (A + A.T) / np.sqrt(n)
    b = rng.randn(n)
    comms, block_map, block_shape = linear_solver.setup_mpi_blocks(comm, (n, n), split=split)
    my_A, my_b = linear_solver.collect_linear_sys_blocks(
        comm, block_map, block_shape, A if rank == 0 else None, b if rank == 0 else None)
    x = linear_solver.cg_mpi(comms, my_A, my_b, n, block_map, maxiters=500, abs_tol=1e-10)
    myrow, mycol = block_map[rank]
    if myrow == mycol:
        x_ref = np.linalg.solve(A, b)
        assert x.shape == (n,)
        assert np.max(np.abs(x - x_ref)) / np.max(np.abs(x_ref)) < 1e-8
    else:
        assert x is None
    comm.barrier()


def case_rowshard(comm):
    """precompute_mpi(shard_rows=True) + RowShardedOperator CG over the ranks == the
    single-process system and its solution, on every rank (large-N solve)."""
    import hydra_amd as hy
    rank, size = comm.Get_rank(), comm.Get_size()
    p = _ptsrc_inputs()
    ns = p["prior"].size
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in p["ant_pos"]]
    fidx = np.array_split(np.arange(p["freqs"].size), size)[rank]
    full = hy.ptsrc_sampler.calc_proj_operator(p["ra"], p["dec"], p["fluxes"], p["ant_pos"],
                                               p["antpairs"], p["freqs"], p["times"], beams,
                                               materialize=False)
    A1, b1 = hy.ptsrc_sampler.precompute_mpi(
        None, p["ants"], p["antpairs"], p["freqs"], p["times"], full, p["data"],
        p["inv_noise"], p["gains"], p["prior"], omega_n=p["om_n"], omega_a=p["om_a"])
    mine = hy.ptsrc_sampler.calc_proj_operator(
        p["ra"], p["dec"], p["fluxes"][:, fidx], p["ant_pos"], p["antpairs"],
        p["freqs"][fidx], p["times"], beams, materialize=False)
    dev = torch.device("cuda")
    A_rows, b2 = hy.ptsrc_sampler.precompute_mpi(
        comm, p["ants"], p["antpairs"], p["freqs"][fidx], p["times"], mine,
        torch.as_tensor(p["data"][:, fidx], device=dev), p["inv_noise"][:, fidx], p["gains"][:, fidx],
        p["prior"], omega_n=p["om_n"][:, fidx], omega_a=p["om_a"] if rank == 0 else None,
        shard_rows=True)
    r0, r1 = comm.row_range(ns)
    assert tuple(A_rows.shape) == (r1 - r0, ns)
    ea = np.max(np.abs(A_rows.cpu().numpy() - A1[r0:r1])) / np.max(np.abs(A1))
    eb = np.max(np.abs(b2.cpu().numpy() - b1)) / np.max(np.abs(b1))
    assert ea < 1e-12 and eb < 1e-12, (ea, eb)
    op = hy.linear_solver.RowShardedOperator(comm, A_rows, ns)
    v = torch.as_tensor(np.random.RandomState(3).randn(ns), device=dev)
    assert np.max(np.abs(op(v).cpu().numpy() - A1 @ v.cpu().numpy())) / np.max(np.abs(A1)) < 1e-12
    x, iters = hy.linear_solver.cg(None, b2, maxiters=500, abs_tol=1e-10, linear_op=op,
                                   return_iters=True)
    x_ref = np.linalg.solve(A1, b1)
    assert 0 < iters < 500
    assert np.max(np.abs(x.cpu().numpy() - x_ref)) / np.max(np.abs(x_ref)) < 1e-8
    comm.barrier()


def case_cosmo(comm):
    """Frequency-sharded cosmo_sampler.precompute_mpi (each rank holds a slice of the field's
    channels; blocks of the other channels stay zero until comm.Reduce, cosmo_sampler.py:295-299,
    339-340) == the single-process system built from golden inputs of the reference."""
    import hydra_amd as hy
    rank, size = comm.Get_rank(), comm.Get_size()
    g = np.load(os.path.join(ROOT, "tests", "golden", "cosmo.npz"))
    ap = [tuple(int(x) for x in pr) for pr in g["antpairs"]]
    fidx = np.array_split(np.arange(g["freqs"].size), size)[rank]
    op, rhs = hy.cosmo_sampler.precompute_mpi(
        comm, g["ants"], ap, g["freqs"], g["freqs"][fidx], g["times"], g["proj"][:, fidx],
        g["data"][:, fidx], g["inv_noise"][:, fidx], g["gains"][:, fidx], g["pspec3d"],
        omega_n=g["omega_n"][:, fidx], omega_s=g["omega_s"])
    if rank == 0:
        ea = np.max(np.abs(op - g["op"])) / np.max(np.abs(g["op"]))
        eb = np.max(np.abs(rhs - g["rhs"])) / np.max(np.abs(g["rhs"]))
        assert ea < 1e-10 and eb < 1e-10, (ea, eb)
    else:
        assert op.shape == (1, 1, 1) and rhs.shape == (1, 1)     # dummies off the root
    comm.barrier()


if __name__ == "__main__":
    case = sys.argv[1]
    logdir = sys.argv[2] if len(sys.argv) > 2 else None
    comm = _dist.init_from_env(backend="gloo")
    assert comm is not None, "run under torch.distributed.run with WORLD_SIZE > 1"
    try:
        {"comm": case_comm, "blocks": case_blocks, "ptsrc": case_ptsrc,
         "blshard": case_blshard, "cgmpi": case_cgmpi, "cosmo": case_cosmo,
         "rowshard": case_rowshard}[case](comm)
    except BaseException:
        import traceback
        if logdir:
            with open(os.path.join(logdir, f"rank{os.environ.get('RANK')}.err"), "w") as fh:
                traceback.print_exc(file=fh)
        raise
    import torch.distributed as dist
    dist.destroy_process_group()
    print(f"rank {os.environ.get('RANK')} ok: {case}")
