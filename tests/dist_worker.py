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
    # row sharding of a dense operator: block-cyclic rows (256-row blocks at these sizes), every
    # block of the sum lands on its owner, vectors come back whole (a ragged last block; a rank
    # may own no row at all)
    for n in (7, 300, 256 * size + 40, 256 * (2 * size + 1)):
        lay = comm.row_layout(n)
        allrows = np.concatenate([lay.rows(k) for k in range(size)])
        assert np.array_equal(np.sort(allrows), np.arange(n)) and lay.block % 256 == 0
        assert all(lay.owner(b) == b % size for b in range(lay.nblocks))
        mine_rows = lay.rows()
        assert mine_rows.size == lay.nloc() and np.all(np.diff(mine_rows) > 0)
        for b in lay.blocks_of():
            g0, g1 = lay.block_range(b)
            assert np.array_equal(mine_rows[lay.local_start(b):lay.local_start(b) + g1 - g0], np.arange(g0, g1))
        assert lay.local_index(n) == lay.nloc() and lay.local_index(0) == 0
        A = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3) * (rank + 1)
        mine = comm.reduce_rows_(A)
        want = torch.arange(n * 3, dtype=torch.float64).reshape(n, 3)[mine_rows] * sum(range(1, size + 1))
        assert mine.shape == want.shape and torch.equal(mine, want)
        full = comm.allgather_rows(torch.as_tensor(mine_rows, dtype=torch.float64), n)
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
        # replicate=True (small systems): all-reduces, the whole system on EVERY rank,
        # the same bits everywhere, so each rank solves for itself and nothing is broadcast
        A3, b3 = hy.ptsrc_sampler.precompute_mpi(
            comm, p["ants"], p["antpairs"], p["freqs"][fidx], p["times"], mine,
            p["data"][:, fidx], p["inv_noise"][:, fidx], p["gains"][:, fidx], p["prior"],
            omega_n=p["om_n"][:, fidx], omega_a=p["om_a"], replicate=True)
        assert np.max(np.abs(A3 - A1)) / np.max(np.abs(A1)) < 1e-12
        assert np.max(np.abs(b3 - b1)) / np.max(np.abs(b1)) < 1e-12
        assert np.array_equal(A3, A3.T)
        for arr in (A3, b3):
            t = torch.as_tensor(np.ascontiguousarray(arr), device="cuda")
            parts = comm.allgather(t)
            assert all(torch.equal(q, parts[0]) for q in parts)
        x3 = hy.linear_solver.solve_dense(A3, b3)
        assert np.max(np.abs(x3 - x_ref)) / np.max(np.abs(x_ref)) < 1e-10
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
    myrows = comm.row_layout(ns).rows()
    assert tuple(A_rows.shape) == (myrows.size, ns)
    ea = (np.max(np.abs(A_rows.cpu().numpy() - A1[myrows])) / np.max(np.abs(A1))) if myrows.size else 0.0
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
    # SystemCache over the ranks, rows sharded and rooted: one build, then reuse; every sample
    # equal to the single-process system's solution with the same fluctuation vectors
    for rows_mode in (True, False):
        cache = hy.ptsrc_sampler.SystemCache()
        for k in range(3):
            rng = np.random.RandomState(50 + k)
            om_n = (rng.randn(*p["om_n"].shape) + 1j * rng.randn(*p["om_n"].shape)) / np.sqrt(2)
            om_a = rng.randn(ns)
            _, bk = hy.ptsrc_sampler.precompute_mpi(
                None, p["ants"], p["antpairs"], p["freqs"], p["times"], full, p["data"],
                p["inv_noise"], p["gains"], p["prior"], omega_n=om_n, omega_a=om_a)
            _, bc = hy.ptsrc_sampler.precompute_mpi(
                comm, p["ants"], p["antpairs"], p["freqs"][fidx], p["times"], mine,
                torch.as_tensor(p["data"][:, fidx], device=dev), p["inv_noise"][:, fidx],
                p["gains"][:, fidx], p["prior"], omega_n=om_n[:, fidx],
                omega_a=om_a if rank == 0 else None, shard_rows=rows_mode, cache=cache)
            xk = np.linalg.solve(A1, bk)
            if rows_mode or rank == 0:
                xc = cache.solve(bc)
                assert np.max(np.abs(xc.cpu().numpy() - xk)) / np.max(np.abs(xk)) < 1e-10, (rows_mode, k)
                if rows_mode:
                    yk = cache.matvec(xc).cpu().numpy()
                    assert np.max(np.abs(yk - bk)) / np.max(np.abs(bk)) < 1e-10
        assert (cache.misses, cache.hits) == (1, 2), (cache.misses, cache.hits)
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


def case_shmpi(comm):
    """SH head with the visibility rows split over ranks: construct_rhs_no_rot_mpi,
    apply_lhs_no_rot_mpi, lhs_matrix(comm=) and the reference's own solve pattern
    cg(None, b, linear_op=apply_lhs_no_rot_mpi(comm, ...), x0=, comm=comm) (example.py:1014-1052;
    sh_sampler.py:405-494; linear_solver.py:441-508) == the single-process system and solution;
    plus the one-collective-per-iteration form (LhsOperator, every rank iterates)."""
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    rank, size = comm.Get_rank(), comm.Get_size()
    rng = np.random.RandomState(21)
    m, nm = 600, 37
    R = (rng.randn(m, nm) + 1j * rng.randn(m, nm)) * 0.05
    w = 300.0 + 100.0 * rng.rand(m)
    ip = 50.0 + 50.0 * rng.rand(nm)
    d = rng.randn(m) + 1j * rng.randn(m)
    om_n = (rng.randn(m) + 1j * rng.randn(m)) / np.sqrt(2)
    om_a = rng.randn(nm)
    a0 = 0.1 * rng.randn(nm)
    x0 = 0.01 * rng.randn(nm)
    b1 = sh_sampler.construct_rhs_no_rot(d, w, ip, om_a, om_n, a0, R)
    A1 = sh_sampler.lhs_matrix(w, ip, R)
    x_ref = np.linalg.solve(A1, b1)
    rows = np.array_split(np.arange(m), size)[rank]
    Rk, wk, dk, onk = R[rows], w[rows], d[rows], om_n[rows]
    # right-hand side: root's omega_a counts, non-root buffers are overwritten (:417-420)
    oa = om_a.copy() if rank == 0 else np.full(nm, 5.0)     # (no rank-dependent draws from rng)
    b2 = sh_sampler.construct_rhs_no_rot_mpi(comm, dk, wk, ip, oa, onk, a0, Rk)
    assert np.array_equal(oa, om_a)
    if rank == 0:
        assert np.max(np.abs(b2 - b1)) <= 1e-12 * np.max(np.abs(b1))
    else:
        assert np.all(b2 == 0.0)
    # operator: root's vector counts
    v = rng.randn(nm)
    vk = v.copy() if rank == 0 else np.full(nm, 7.0)
    y2 = sh_sampler.apply_lhs_no_rot_mpi(comm, vk, wk, ip, Rk)
    assert np.array_equal(vk, v)
    if rank == 0:
        assert np.max(np.abs(y2 - A1 @ v)) <= 1e-12 * np.max(np.abs(A1 @ v))
    else:
        assert np.all(y2 == 0.0)
    A2 = sh_sampler.lhs_matrix(wk, ip, Rk, comm=comm)
    assert np.max(np.abs(A2 - A1)) <= 1e-12 * np.max(np.abs(A1)) and np.array_equal(A2, A2.T)
    # the reference's solve, verbatim: collective operator valid on the root only, comm=comm
    bk = b2 if rank == 0 else np.zeros(nm)
    x0k = x0.copy()
    x, iters = hy.linear_solver.cg(
        None, bk, maxiters=300, abs_tol=1e-11, x0=x0k, comm=comm, return_iters=True,
        linear_op=lambda a: sh_sampler.apply_lhs_no_rot_mpi(comm, a, wk, ip, Rk))
    assert 0 < iters < 300
    assert np.max(np.abs(x - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))      # on EVERY rank
    # GPU-tensor operands: the flag is read every CHECK_EVERY iterations, steps past convergence
    # are no-ops on every rank
    dev = torch.device("cuda")
    Rk_d, wk_d, ip_d = (torch.as_tensor(t, device=dev) for t in (Rk, wk, ip))
    xt = hy.linear_solver.cg(
        None, torch.as_tensor(bk, device=dev), maxiters=300, abs_tol=1e-11,
        x0=torch.as_tensor(x0, device=dev), comm=comm,
        linear_op=lambda a: sh_sampler.apply_lhs_no_rot_mpi(comm, a, wk_d, ip_d, Rk_d))
    assert np.max(np.abs(xt.cpu().numpy() - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
    # one collective per iteration: every rank iterates on identical data, no comm in cg
    b3 = sh_sampler.construct_rhs_no_rot_mpi(comm, dk, wk, ip, om_a.copy(), onk, a0, Rk,
                                             all_ranks=True)
    assert np.max(np.abs(b3 - b1)) <= 1e-12 * np.max(np.abs(b1))
    op = sh_sampler.LhsOperator(comm, wk, ip, Rk)
    x3, it3 = hy.linear_solver.cg(None, b3, maxiters=300, abs_tol=1e-11, x0=x0, linear_op=op,
                                  return_iters=True)
    assert it3 == iters
    assert np.max(np.abs(x3 - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
    # dense matrix + comm: only the root's matrix counts, every rank gets the solution
    Ak = A2 if rank == 0 else np.zeros((nm, nm))
    x4 = hy.linear_solver.cg(Ak, bk, maxiters=300, abs_tol=1e-11, comm=comm)
    assert np.max(np.abs(x4 - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
    comm.barrier()


def case_shstream(comm):
    """StreamedSHOperator with the frequency channels split over ranks (bench.py's c3t40 / c3f on
    N GPUs): lhs_matrix(comm=) -- closed form through pixel space and block-accumulated -- and
    construct_rhs(comm=) on every rank == the single-process operator over all channels; the
    factor-once solve on the replicated matrix."""
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    rank, size = comm.Get_rank(), comm.Get_size()
    ant_pos = hy.utils.build_hex_array(hex_spec=(2, 3), d=14.6)
    freqs, lsts = np.linspace(100.0, 120.0, 4), np.linspace(0.2, 0.5, 3)
    lmax, nside = 5, 8
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    full = sh_sampler.StreamedSHOperator(freqs, lsts, beams, ant_pos, lmax, nside, time_block=2)
    nbl, nf, nt, nm = full.nbl, full.nf, full.nt, full.nm
    rng = np.random.RandomState(31)
    ip = 50.0 + 50.0 * rng.rand(nm)
    d = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    om1 = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    om0, a0 = rng.randn(nm), 0.1 * rng.randn(nm)
    fidx = np.array_split(np.arange(nf), size)[rank]
    mine = sh_sampler.StreamedSHOperator(freqs[fidx], lsts, beams, ant_pos, lmax, nside, time_block=2)
    for uniform in (True, False):
        w = (300.0 + 200.0 * rng.rand(1 if uniform else nbl, nf, nt)) * np.ones((nbl, 1, 1))
        A1 = full.lhs_matrix(w.reshape(-1), ip).cpu().numpy()
        b1 = full.construct_rhs(d.reshape(-1), w.reshape(-1), ip, om0, om1.reshape(-1), a0).cpu().numpy()
        wk = np.ascontiguousarray(w[:, fidx]).reshape(-1)
        A2 = mine.lhs_matrix(wk, ip, comm=comm)
        assert (mine.fac.last_gram_kernel == "gram_uniform_kernel") == uniform
        b2 = mine.construct_rhs(np.ascontiguousarray(d[:, fidx]).reshape(-1), wk, ip, om0,
                                np.ascontiguousarray(om1[:, fidx]).reshape(-1), a0, comm=comm)
        A2n, b2n = A2.cpu().numpy(), b2.cpu().numpy()
        assert np.max(np.abs(A2n - A1)) <= 1e-11 * np.max(np.abs(A1)) and np.array_equal(A2n, A2n.T)
        assert np.max(np.abs(b2n - b1)) <= 1e-11 * np.max(np.abs(b1))
        x = hy.linear_solver.cho_solve(hy.linear_solver.cho_factor(A2), b2).cpu().numpy()
        assert np.max(np.abs(x - np.linalg.solve(A1, b1))) <= 1e-9 * np.max(np.abs(x))
    comm.barrier()


def case_gainmpi(comm):
    """Gain head with the frequency channels split over ranks: construct_rhs_mpi and
    apply_operator_mpi (gain_sampler.py:145-301, Bcast + Allreduce per call) and the CG with one
    all-reduce of the operator result per iteration (the driver's pattern, example.py:700-722)
    == the single-process right-hand side, operator and solution."""
    import hydra_amd as hy
    from hydra_amd import gain_sampler as gs
    rank, size = comm.Get_rank(), comm.Get_size()
    rng = np.random.RandomState(31)
    nant, nf, nt = 6, 4, 5
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    freqs, times = np.linspace(100.0, 120.0, nf), np.linspace(0.2, 0.5, nt)
    Fb, _, _ = hy.utils.partial_fourier_basis_2d_from_nmax(freqs, times, 1, 1, 20.0, 0.3)
    nmodes = Fb.shape[0]
    ps = 0.05 + 0.05 * rng.rand(nmodes)
    shape = (nbl, nf, nt)
    mv = rng.randn(*shape) + 1j * rng.randn(*shape)
    w = 300.0 + 100.0 * rng.rand(*shape)
    resid = rng.randn(*shape) + 1j * rng.randn(*shape)
    om_y = (rng.randn(nant, nmodes) + 1j * rng.randn(nant, nmodes)) / np.sqrt(2)
    om_r = (rng.randn(*shape) + 1j * rng.randn(*shape)) / np.sqrt(2)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    b1 = gs.construct_rhs_mpi(None, resid, w, ps, A_real, A_imag, mv, Fb, omega_y=om_y, omega_r=om_r)
    n = 2 * nant * nmodes
    v = rng.randn(n)
    y1 = gs.apply_operator_mpi(None, v, w, ps, A_real, A_imag, mv, Fb)
    op1 = gs.GainOperator(w, ps, A_real, A_imag, mv, Fb)
    x_ref = hy.linear_solver.cg(None, b1, linear_op=op1.apply, use_norm_tol=True, abs_tol=1e-11,
                                maxiters=500)
    f = np.array_split(np.arange(nf), size)[rank]
    mvk, wk, rk, ork, Fk = mv[:, f], w[:, f], resid[:, f], om_r[:, f], np.ascontiguousarray(Fb[:, f])
    oy = om_y if rank == 0 else np.zeros_like(om_y)
    b2 = gs.construct_rhs_mpi(comm, rk, wk, ps, A_real, A_imag, mvk, Fk, omega_y=oy, omega_r=ork)
    assert np.max(np.abs(b2 - b1)) <= 1e-12 * np.max(np.abs(b1))          # Allreduce: every rank
    vk = v if rank == 0 else np.zeros(n)
    y2 = gs.apply_operator_mpi(comm, vk, wk, ps, A_real, A_imag, mvk, Fk)
    if rank == 0:
        assert np.max(np.abs(y2 - y1)) <= 1e-12 * np.max(np.abs(y1))
    opk = gs.GainOperator(wk, ps, A_real, A_imag, mvk, Fk)

    def lin(x):
        t = opk.local(x)
        comm.allreduce_(t)
        return x + t
    dev = torch.device("cuda")
    x2, iters = hy.linear_solver.cg(None, torch.as_tensor(b2, device=dev), linear_op=lin,
                                    use_norm_tol=True, abs_tol=1e-11, maxiters=500, return_iters=True)
    assert 0 < iters < 500
    assert np.max(np.abs(x2.cpu().numpy() - x_ref)) <= 1e-8 * np.max(np.abs(x_ref))
    # the one-pass form on each rank's channels: same right-hand side on every rank, same operator
    gains = 1.0 + 0.2 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    data = rng.randn(*shape) + 1j * rng.randn(*shape)
    ggv = hy.utils.apply_gains(mv, gains, ants, antpairs, inline=False)
    b3 = gs.construct_rhs_mpi(None, data - ggv, w, ps, A_real, A_imag, ggv, Fb, omega_y=om_y,
                              omega_r=om_r)
    op3 = gs.GainOperator(w, ps, A_real, A_imag, ggv, Fb)
    b4, op4 = gs.prepare_sample(comm, data[:, f], mvk, gains[:, f], wk, ps, A_real, A_imag, Fk,
                                omega_y=oy, omega_r=ork)
    assert op4._mv is None                                   # hydra_gain_prepare_stream ran
    assert np.max(np.abs(b4 - b3)) <= 1e-12 * np.max(np.abs(b3))
    t = op4.local(torch.as_tensor(v, device=dev))
    comm.allreduce_(t)
    y4 = (torch.as_tensor(v, device=dev) + t).cpu().numpy()
    y3 = op3.apply(v)
    assert np.max(np.abs(y4 - y3)) <= 1e-12 * np.max(np.abs(y3))
    comm.barrier()


def case_driver(comm):
    """example.py on several ranks (gloo, one GPU): the reference's freq x time grid of chunks for
    a square worker count (example.py:161-174) with --solver mpicg (:899-938), against the same
    run on one rank reassembled from the per-worker files; frequency-only chunks otherwise."""
    import tempfile
    sys.path.insert(0, ROOT)
    import example
    rank, size = comm.Get_rank(), comm.Get_size()
    base = comm.bcast(tempfile.mkdtemp(prefix="hydra_driver_") if rank == 0 else None, root=0)
    common = ["--hex-array", "3", "4", "--Nptsrc", "12", "--Ntimes", "6", "--Nfreqs", "8",
              "--Niters", "2", "--seed", "5", "--ptsrc", "--gains"]
    out = os.path.join(base, "multi")
    solver = ["--solver", "mpicg"] if int(round(size ** 0.5)) ** 2 == size else []
    assert example.main(common + solver + ["--output-dir", out]) == 0
    comm.barrier()
    if rank == 0:
        fi, ti = [], []
        ngrid = int(round(size ** 0.5))
        square = ngrid * ngrid == size
        shapes = [np.load(os.path.join(out, "sim_data_chunk_w%04d.npy" % w)).shape for w in range(size)]
        nbl = shapes[0][0]
        if square:
            assert all(s == (nbl, 8 // ngrid, 6 // ngrid) for s in shapes), shapes
        else:
            assert sum(s[1] for s in shapes) == 8 and all(s[2] == 6 for s in shapes), shapes
        for n in range(2):
            x = np.load(os.path.join(out, "src_amp_%05d.npy" % n))
            g = np.load(os.path.join(out, "delta_g_amps_%05d.npy" % n))
            assert x.shape == (12,) and np.all(np.isfinite(x)) and np.max(np.abs(x)) < 0.5
            assert g.shape == (10, 4) and np.all(np.isfinite(g))
        # the catalogue and gain draws are the root's and identical on every layout
        amps = np.load(os.path.join(out, "ptsrc_amps0.npy"))
        assert amps.shape == (12,)
    comm.barrier()


def case_symreduce(comm):
    """Packed-lower-triangle reduce / all-reduce of a symmetric matrix (hydra_pack_lower /
    hydra_unpack_lower around the collective) == the full-square sum, exactly symmetric."""
    rank, size = comm.Get_rank(), comm.Get_size()
    dev = torch.device("cuda")
    for n in (1, 63, 64, 65, 200):
        rng = np.random.RandomState(100 + n)
        mats = [rng.randn(n, n) for _ in range(size)]
        mats = [a + a.T for a in mats]
        want = sum(mats)
        A = torch.as_tensor(mats[rank], device=dev).contiguous()
        comm.reduce_sym_(A, root=0)
        if rank == 0:
            got = A.cpu().numpy()
            assert np.max(np.abs(got - want)) <= 1e-13 * max(np.max(np.abs(want)), 1.0)
            assert np.array_equal(got, got.T)
        B = torch.as_tensor(mats[rank], device=dev).contiguous()
        comm.allreduce_sym_(B)
        got = B.cpu().numpy()
        assert np.max(np.abs(got - want)) <= 1e-13 * max(np.max(np.abs(want)), 1.0)
        assert np.array_equal(got, got.T) and getattr(B, "hydra_symmetric", False)
    comm.barrier()


def case_cholrows(comm):
    """linear_solver.solve_dense_rows: Cholesky of a matrix sharded by rows over the ranks
    (example.py:940-947 for an operator that stays sharded) against numpy's solve; sizes around
    the 256-wide panels and with row blocks that cut through panels."""
    from hydra_amd import linear_solver
    rank, size = comm.Get_rank(), comm.Get_size()
    dev = torch.device("cuda")
    # (n, super-panel width): the default 2048 on both sides of its edge, and narrow super-panels
    # so that several trailing updates with K = 512 follow one another at a small size (beyond two
    # super-panels the far part of an update runs on a side stream under the next panels, and the
    # two gathered super-panel buffers take turns: 2100 / 512 reuses the first one)
    for n, spw in ((2, 2048), (254, 2048), (256, 2048), (258, 2048), (700, 2048), (1026, 2048),
                   (1026, 512), (1300, 512), (2100, 512), (2306, 2048)):
        linear_solver.SUPER_PANEL = spw
        rng = np.random.RandomState(n)
        B = rng.randn(n, n + 5)
        A = B @ B.T / n + np.eye(n)
        b = rng.randn(n)
        want = np.linalg.solve(A, b)
        mine = comm.row_layout(n).rows()
        rows = torch.as_tensor(A[mine], device=dev).contiguous()
        x = linear_solver.solve_dense_rows(comm, rows, torch.as_tensor(b, device=dev))
        got = x.cpu().numpy()
        assert np.max(np.abs(got - want)) <= 1e-11 * np.max(np.abs(want)), (n, np.max(np.abs(got - want)))
        assert torch.equal(rows, torch.as_tensor(A[mine], device=dev))      # input untouched
        # factor once, solve twice; what lies above the diagonal is still the operator
        fac = linear_solver.factor_dense_rows(comm, rows)
        for seed in (1, 2):
            b2 = np.random.RandomState(seed).randn(n)
            got2 = fac.solve(torch.as_tensor(b2, device=dev)).cpu().numpy()
            want2 = np.linalg.solve(A, b2)
            assert np.max(np.abs(got2 - want2)) <= 1e-11 * np.max(np.abs(want2)), (n, seed)
        if mine.size:
            up = np.triu(np.ones((n, n), dtype=bool), 1)[mine]
            assert np.array_equal(fac.A.cpu().numpy()[up], A[mine][up])
            L = np.linalg.cholesky(A)
            lo = np.tril(np.ones((n, n), dtype=bool))[mine]
            assert np.max(np.abs(fac.A.cpu().numpy()[lo] - L[mine][lo])) <= 1e-11 * np.max(np.abs(L))
    linear_solver.SUPER_PANEL = 2048
    # not positive definite: every rank raises
    A = -np.eye(300)
    mine = comm.row_layout(300).rows()
    try:
        linear_solver.solve_dense_rows(comm, torch.as_tensor(A[mine], device=dev).contiguous(),
                                       torch.ones(300, dtype=torch.float64, device=dev))
        raise AssertionError("no error for an indefinite matrix")
    except np.linalg.LinAlgError:
        pass
    comm.barrier()


def case_cghist(comm):
    """cg(linear_op=) with a collective operator when the ranks' iteration histories differ
    (ADVICE round 2): rank 0 first solves, alone, an easier system of the same size and tolerance
    through the SAME operator code; the collective solve that follows must still apply the
    operator equally often on every rank (a surplus all-reduce on one rank would hang) and give
    the single-process answer."""
    from hydra_amd import linear_solver
    rank, size = comm.Get_rank(), comm.Get_size()
    dev = torch.device("cuda")
    rng = np.random.RandomState(5)
    n = 96
    M = rng.randn(4 * n, n)
    A = M.T @ M / n + np.eye(n)
    b = rng.randn(n)
    rows = np.array_split(np.arange(4 * n), size)[rank]
    Mk = torch.as_tensor(M[rows], device=dev)
    calls = [0]

    class Op:
        """(M^T M / n + 1) v with the rows of M split over the ranks: one all-reduce per call."""
        def __init__(self, comm, scale):
            self.comm, self.scale = comm, scale

        def __call__(self, v):
            calls[0] += 1
            t = Mk.T @ (Mk @ v) / n * self.scale
            if self.comm is not None:
                self.comm.allreduce_(t)
            return t + v

    linear_solver._ITER_HISTORY.clear()
    bd = torch.as_tensor(b, device=dev)
    if rank == 0:      # a local solve, same operator code / size / tolerance, far fewer iterations
        x_loc, it_loc = linear_solver.cg(None, bd, linear_op=Op(None, 1e-3), abs_tol=1e-10,
                                         return_iters=True)
        assert 0 < it_loc < 10
    comm.barrier()
    for _ in range(2):     # the second solve runs on the learned schedule
        calls[0] = 0
        x, iters = linear_solver.cg(None, bd, linear_op=Op(comm, 1.0), abs_tol=1e-10,
                                    return_iters=True)
        c = torch.tensor([float(calls[0])])
        lo, hi = c.clone(), c.clone()
        comm.allreduce_(lo, op="min")
        comm.allreduce_(hi, op="max")
        assert lo.item() == hi.item() == calls[0], (lo.item(), hi.item(), calls[0])
        assert np.abs(x.cpu().numpy() - np.linalg.solve(A, b)).max() < 1e-8
    assert iters <= calls[0] <= iters + max(2, iters // 16)
    # the root-driven form (comm=): histories differ the same way, the root's prediction is used
    if rank == 0:
        linear_solver.cg(None, bd, linear_op=lambda v: v, comm=None, abs_tol=1e-10)
    Ad = torch.as_tensor(A, device=dev)
    for _ in range(2):
        xr = linear_solver.cg(None, bd, linear_op=lambda v: Ad @ v, comm=comm, abs_tol=1e-10)
        assert np.abs(xr.cpu().numpy() - np.linalg.solve(A, b)).max() < 1e-8
    comm.barrier()


def case_nccl1(comm):
    """Every collective of _dist.Comm on RCCL itself, with the one rank a 1-GPU box can hold:
    the nccl-only branches (GPU-resident numpy staging, reduce_scatter_tensor,
    all_gather_into_tensor, sub-groups) run with real communicators and kernels, so the first
    multi-GPU launch is not also the first RCCL call."""
    assert comm.backend == "nccl" and comm.Get_size() == 1
    dev = torch.device("cuda")
    x = torch.arange(12, dtype=torch.float64, device=dev).reshape(3, 4)
    assert torch.equal(comm.allreduce_(x.clone()), x)
    assert torch.equal(comm.allreduce_(x.clone(), op="max"), x)
    assert torch.equal(comm.reduce_(x.clone(), root=0), x)
    assert torch.equal(comm.bcast_(x.clone(), root=0), x)
    z = torch.randn(5, dtype=torch.complex128, device=dev)
    assert torch.equal(comm.allreduce_(z.clone()), z)
    h = np.arange(6.0)
    comm.allreduce_(h)                               # numpy buffer staged through the GPU
    assert np.array_equal(h, np.arange(6.0))
    hb = np.arange(4.0) + 1j
    comm.Bcast(hb, root=0)
    assert np.array_equal(hb, np.arange(4.0) + 1j)
    parts = comm.allgather(x)
    assert len(parts) == 1 and torch.equal(parts[0], x)
    A = torch.randn(8, 8, dtype=torch.float64, device=dev)
    A = (A + A.T).contiguous()
    rows = comm.reduce_rows_(A.clone())              # reduce_scatter_tensor
    assert torch.equal(rows, A)
    assert torch.equal(comm.reduce_sym_(A.clone(), root=0), A)
    assert torch.equal(comm.allreduce_sym_(A.clone()), A)
    v = torch.arange(7, dtype=torch.float64, device=dev)
    assert torch.equal(comm.allgather_rows(v, 7)[:7], v)
    sub = comm.Create([0])
    assert sub is not None and sub.Get_size() == 1
    assert torch.equal(sub.allreduce_(x.clone()), x)
    # the root-driven CG of the SH head (example.py:1014-1052) over RCCL
    from hydra_amd import linear_solver
    rng = np.random.RandomState(0)
    M = rng.randn(40, 12)
    Ad = torch.as_tensor(M.T @ M + np.eye(12), device=dev)
    b = torch.as_tensor(rng.randn(12), device=dev)
    xs = linear_solver.cg(None, b, linear_op=lambda p: Ad @ p, comm=comm, abs_tol=1e-12)
    assert float(torch.linalg.norm(Ad @ xs - b)) < 1e-9
    case_cholrows(comm)
    comm.barrier()


if __name__ == "__main__":
    case = sys.argv[1]
    logdir = sys.argv[2] if len(sys.argv) > 2 else None
    if case == "nccl1":
        import torch.distributed as dist
        torch.cuda.set_device(0)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
        comm = _dist.Comm()
    else:
        comm = _dist.init_from_env(backend="gloo")
    assert comm is not None, "run under torch.distributed.run with WORLD_SIZE > 1"
    try:
        {"comm": case_comm, "blocks": case_blocks, "ptsrc": case_ptsrc,
         "blshard": case_blshard, "cgmpi": case_cgmpi, "cosmo": case_cosmo,
         "rowshard": case_rowshard, "shmpi": case_shmpi, "gainmpi": case_gainmpi,
         "symreduce": case_symreduce, "shstream": case_shstream, "driver": case_driver, "nccl1": case_nccl1,
         "cholrows": case_cholrows, "cghist": case_cghist}[case](comm)
    except BaseException:
        import traceback
        if logdir:
            with open(os.path.join(logdir, f"rank{os.environ.get('RANK')}.err"), "w") as fh:
                traceback.print_exc(file=fh)
        raise
    import torch.distributed as dist
    dist.destroy_process_group()
    print(f"rank {os.environ.get('RANK')} ok: {case}")
