"""
Multi-process tests of the N > 1 path (one process per rank, gloo on 127.0.0.1):
hydra_amd._dist.Comm (the mpi4py subset the reference uses: ptsrc_sampler.py:86,132;
sh_sampler.py:420-485; linear_solver.py:48-367) and the block layout on the CPU; the
frequency-sharded point-source system and cg_mpi with every rank on the one GPU of the box.
"""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(case, nproc, timeout=300):
    import glob
    import tempfile
    logdir = tempfile.mkdtemp(prefix=f"hydra_dist_{case}_")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_worker.py"), case, logdir]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    errs = "\n".join(f"--- {os.path.basename(f)}\n{open(f).read()}"
                     for f in sorted(glob.glob(os.path.join(logdir, "*.err"))))
    assert res.returncode == 0, errs + "\n" + res.stdout[-1500:] + "\n" + res.stderr[-1500:]
    assert res.stdout.count(f"ok: {case}") == nproc, res.stdout[-2000:]


def test_comm_collectives_world2():
    _run("comm", 2)


def test_comm_collectives_world3():
    _run("comm", 3)


def test_block_layout_world4():
    _run("blocks", 4)


@pytest.mark.gpu
def test_freq_sharded_ptsrc_system_world2():
    _run("ptsrc", 2)


@pytest.mark.gpu
def test_baseline_sharded_ptsrc_system_world3():
    _run("blshard", 3)


@pytest.mark.gpu
def test_row_sharded_ptsrc_solve_world3():
    _run("rowshard", 3)


@pytest.mark.gpu
def test_freq_sharded_cosmo_system_world3():
    _run("cosmo", 3)


@pytest.mark.gpu
def test_cg_mpi_world4():
    _run("cgmpi", 4)


@pytest.mark.gpu
def test_sh_head_rows_split_over_ranks_and_root_driven_cg_world2():
    _run("shmpi", 2)


@pytest.mark.gpu
def test_sh_head_rows_split_over_ranks_world3():
    _run("shmpi", 3)


@pytest.mark.gpu
def test_streamed_sh_operator_channels_split_over_ranks_world2():
    _run("shstream", 2)


@pytest.mark.gpu
def test_gain_head_channels_split_over_ranks_world2():
    _run("gainmpi", 2)


@pytest.mark.gpu
def test_symmetric_packed_reduce_world3():
    _run("symreduce", 3)


@pytest.mark.gpu
def test_driver_on_a_2x2_chunk_grid_with_mpicg_world4():
    _run("driver", 4)


@pytest.mark.gpu
def test_driver_with_frequency_chunks_world2():
    _run("driver", 2)


@pytest.mark.gpu
def test_rccl_collectives_world1():
    """The nccl branches of _dist.Comm on a real RCCL communicator (one rank: a 1-GPU box)."""
    _run("nccl1", 1)


@pytest.mark.gpu
def test_cg_flag_schedule_agrees_across_ranks_with_different_histories_world2():
    _run("cghist", 2)


@pytest.mark.gpu
def test_row_sharded_cholesky_world2():
    _run("cholrows", 2)


@pytest.mark.gpu
def test_row_sharded_cholesky_world3():
    _run("cholrows", 3)


def _bench(extra, env_extra=None, timeout=300):
    import json
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    res = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py")] + extra,
                         env=env, capture_output=True, text=True, timeout=timeout)
    assert res.returncode == 0, res.stdout[-1500:] + "\n" + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[-2000:]   # stdout: ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_started_bare_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no torchrun environment spawns its ranks itself (child
    torchrun, before any GPU call) and prints ONE line that agrees with the single-GPU line on
    the workload (the reference's `mpirun -n N python example.py`, example.py:161-174)."""
    common = ["--config", "c1", "--steps", "2", "--warmup", "1", "--no-cpu"]
    one = _bench(["--gpus", "1"] + common)
    two = _bench(["--gpus", "2"] + common, {"HYDRA_DIST_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["rows_M"] == one["config"]["rows_M"] == 45 * 16 * 20
    assert two["config"]["workload"] == one["config"]["workload"]
    for line in (one, two):
        assert line["value"] > 0 and line["sample_rel_residual"] < 1e-8
        assert line["steps"] == 2 and line["warmup"] == 1
    # under torchrun's environment the same file is a rank and does not spawn again
    assert "freq-sharded x2" in two["config"]["parallelism"]
