#!/usr/bin/env python
"""
bench.py -- GCR samples/s of the point-source head at BASELINE config 2
("ptsrc GCR: HERA-37, 2000 sources, 64 freqs, 200 times, complex128 on 1 MI355X").

One step = one GCR sample of the point-source head, everything the reference does per Gibbs
iteration for this head (example.py:856-984): fresh fluctuation vectors, GCR operator
A = 1 + S^1/2 Re(R^H N^-1 R) S^1/2 and right-hand side from the current gains / data
(ptsrc_sampler.precompute_mpi), CG solve to the reference tolerance, broadcast of the sample.
The operator is held in factored form (antenna voltages, 15 GB), so R (273 GB) is never
materialised; all inputs are resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|mid] [--no-cpu]

With N > 1 (launched by torch.distributed.run) the frequency channels are sharded over ranks,
partial Gram / RHS are summed with RCCL (one reduce of Ns^2 + Ns doubles per sample), the root
solves and broadcasts: total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0 (see the contract in the task description) with
  roofline     -- the Gram kernel (FP64 MFMA bound): algorithmic flops 2 M Ns (Ns+1) per launch
                  over its HIP-event duration, against the 78.6 TFLOP/s FP64 matrix peak;
  cpu_baseline -- the numpy/scipy oracle (= the reference's CPU algorithm) timed on a bounded
                  slice of the same workload on this host, scaled linearly in M.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TFLOPS = 78.6      # MI355X FP64 matrix (vendor datasheet)

CONFIGS = {
    # name: (hex_spec, Nsrc, Nf, Nt)
    "c2": ((4, 7), 2000, 64, 200),
    "mid": ((4, 7), 1000, 16, 50),
    "c1": ((3, 4), 100, 16, 20),
}


def synth_inputs(hex_spec, ns, nf, nt):
    """Synthetic inputs of SURVEY section 8(d): hex array (14.6 m), random catalogue seed 0,
    100-120 MHz, LST 0.2-0.5 rad, Gaussian beam D = 14 m, gains (1+1j) + 0.05 N(0,1),
    sigma_noise = 0.05, prior std 0.1."""
    from hydra_amd import example, utils
    np.random.seed(0)
    ant_pos = utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    ra, dec, amps = example.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4),
                                                            (-1.0, 2.0))
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fluxes = utils.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    g0 = (1.0 + 1.0j) + 0.05 * (np.random.randn(nant) + 1j * np.random.randn(nant))
    x_true = 0.1 * np.random.randn(ns)
    return dict(ant_pos=ant_pos, antpairs=antpairs, ra=ra, dec=dec, freqs=freqs, times=times,
                fluxes=fluxes, g0=g0, x_true=x_true, sigma_noise=0.05, prior=0.1 * np.ones(ns))


def pmc_traffic(config):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of
    this same command (profiles/pmc_traffic.json; tools/gpu_check.sh collects them), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            return json.load(fh)[config]["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def cpu_baseline(inp, nf_s, nt_s, full_rows):
    """Time the oracle (numpy/scipy restatement of the reference) on an (nf_s x nt_s) slice of
    the workload and scale linearly in the number of visibility rows."""
    from oracle import beams as obeams, gcr_ptsrc, helpers
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = int(os.environ.get("OMP_NUM_THREADS", "1"))
    ant_pos, antpairs = inp["ant_pos"], inp["antpairs"]
    nant, nbl, ns = len(ant_pos), len(antpairs), inp["ra"].size
    freqs, times = inp["freqs"][:nf_s], inp["times"][:nt_s]
    fl = inp["fluxes"][:, :nf_s]
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    t0 = time.time()
    proj = gcr_ptsrc.calc_proj_operator(inp["ra"], inp["dec"], fl, ant_pos, antpairs, freqs, times,
                                        bo)
    t_resp = time.time() - t0
    gains = inp["g0"][:, None, None] * np.ones((nant, nf_s, nt_s))
    ants = np.arange(nant)
    data = helpers.apply_gains((proj.reshape(-1, ns) @ (1 + inp["x_true"])).reshape(
        nbl, nf_s, nt_s), gains, ants, antpairs)
    inv_noise = np.full(data.shape, 1.0 / inp["sigma_noise"] ** 2)
    np.random.seed(12345)
    t0 = time.time()
    A, b = gcr_ptsrc.precompute(ants, antpairs, proj, data, inv_noise, gains, inp["prior"])
    t_pre = time.time() - t0
    t0 = time.time()
    gcr_ptsrc.dense_solve(A, b)
    t_solve = time.time() - t0
    rows = nbl * nf_s * nt_s
    scale = full_rows / rows
    t_sample_full = t_pre * scale + t_solve       # the response is precomputed once, not per sample
    return {
        "value": 1.0 / t_sample_full, "unit": "samples/s", "cores": int(threads), "kind": "port",
        "sample": (f"oracle (numpy/scipy restatement of ptsrc_sampler.precompute_mpi + "
                   f"scipy.linalg.solve) on a {nf_s}x{nt_s} (freq x time) slice = {rows} of "
                   f"{full_rows} rows, {ns} sources; precompute {t_pre:.2f}s scaled linearly in "
                   f"rows + solve {t_solve:.3f}s; response build {t_resp:.2f}s not counted"),
        "gram_gflops": 4.0 * rows * ns * ns / t_pre / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cg-tol", type=float, default=1e-8)
    ap.add_argument("--solver", default="dense", choices=["dense", "cg"],
                    help="dense = blocked Cholesky (the reference's default for this head, "
                         "example.py:944); cg = linear_solver.cg")
    args = ap.parse_args()

    import torch
    import hydra_amd as hy
    from hydra_amd import _dev, _dist, ptsrc_sampler

    comm = _dist.init_from_env()
    rank = comm.Get_rank() if comm is not None else 0
    world = comm.Get_size() if comm is not None else 1
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if comm is None:
        torch.cuda.set_device(0)
    dev = _dev.device()

    hex_spec, ns, nf, nt = CONFIGS[args.config]
    inp = synth_inputs(hex_spec, ns, nf, nt)
    ant_pos, antpairs = inp["ant_pos"], inp["antpairs"]
    nant, nbl = len(ant_pos), len(antpairs)
    ants = np.arange(nant)
    fidx = np.array_split(np.arange(nf), world)[rank]          # frequency sharding
    nf_loc = fidx.size
    freqs_loc = inp["freqs"][fidx]
    times = inp["times"]
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]

    # ---- set-up (untimed): factored operator, gains, data, noise weights -- all in HBM
    fac = ptsrc_sampler.calc_proj_operator(inp["ra"], inp["dec"], inp["fluxes"][:, fidx], ant_pos,
                                           antpairs, freqs_loc, times, beams, materialize=False)
    gains = torch.as_tensor(inp["g0"], device=dev)[:, None, None].expand(nant, nf_loc, nt).contiguous()
    gg = _dev.empty((nbl, nf_loc, nt), torch.complex128)
    _dev.call("hydra_row_factors", _dev.ptr(gains), _dev.ptr(None), _dev.ptr(fac.bl_i),
              _dev.ptr(fac.bl_j), nbl, nf_loc, nt, _dev.ptr(gg), _dev.ptr(None), _dev.stream_ptr())
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    data = fac.apply(torch.as_tensor(1.0 + inp["x_true"], device=dev), rowmul=gg)
    data += inp["sigma_noise"] * torch.randn(data.shape, dtype=torch.complex128, device=dev,
                                             generator=gen)
    inv_noise = torch.full((nbl, nf_loc, nt), 1.0 / inp["sigma_noise"] ** 2, dtype=torch.float64,
                           device=dev)
    prior = torch.as_tensor(inp["prior"], device=dev)
    del gg
    m_loc = nbl * nf_loc * nt
    m_all = nbl * nf * nt

    def one_sample():
        om_n = torch.randn((nbl, nf_loc, nt), dtype=torch.complex128, device=dev, generator=gen)
        om_a = torch.randn(ns, dtype=torch.float64, device=dev, generator=gen)
        A, b = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                            inv_noise, gains, prior, omega_n=om_n, omega_a=om_a)
        if rank == 0 and args.solver == "cg":
            x, iters = hy.linear_solver.cg(A, b, maxiters=1000, abs_tol=args.cg_tol,
                                           return_iters=True)
        elif rank == 0:
            x, iters = hy.linear_solver.solve_dense(A, b), 0
        else:
            x, iters = torch.zeros(ns, dtype=torch.float64, device=dev), 0
        if comm is not None:
            comm.bcast_(x, root=0)
        return A, b, x, iters

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_sample()
    ptsrc_sampler.GRAM_EVENTS = []
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        A, b, x, iters = one_sample()
    sync()
    elapsed = time.perf_counter() - t0
    events = ptsrc_sampler.GRAM_EVENTS
    ptsrc_sampler.GRAM_EVENTS = None
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if comm is not None:
        comm.allreduce_(t_all, op="max")
    elapsed = float(t_all.item())
    gram_ms = float(np.mean([s.elapsed_time(e) for s, e in events])) if events else float("nan")

    if rank != 0:
        return
    # sanity of the last sample (outside the timed region)
    resid = float((torch.linalg.norm(A @ x - b) / torch.linalg.norm(b)).item())
    gram_flops = 2.0 * m_loc * ns * (ns + 1)          # algorithmic, per launch (this rank's rows)
    achieved = gram_flops / (gram_ms * 1e-3) / 1e12
    out = {
        "metric": "GCR samples/sec (ptsrc head, whole job)",
        "value": args.steps / elapsed,
        "unit": "samples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config {args.config}: ptsrc GCR, hex{hex_spec} = {nant} antennas "
                         f"/ {nbl} baselines, {ns} sources, {nf} freqs x {nt} times, complex128, "
                         f"factored operator (R never materialised)"),
            "rows_M": m_all, "solver": args.solver, "cg_abs_tol": args.cg_tol,
            "cg_iters_last": int(iters),
            "parallelism": f"freq-sharded x{world}" if world > 1 else "single GPU",
        },
        "gram_tflops_whole_job": world * achieved,
        "sample_rel_residual": resid,
        "roofline": {
            "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic(args.config) if world == 1 else None,
            "kernel": "gram_fused_kernel", "ms_per_launch": gram_ms,
            "flops_per_launch": gram_flops,
        },
    }
    if not args.no_cpu:
        nf_s, nt_s = (2, 10) if args.config == "c2" else (min(nf, 2), min(nt, 10))
        out["cpu_baseline"] = cpu_baseline(inp, nf_s, nt_s, m_all)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
