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
    # point-source head -- name: (hex_spec, Nsrc, Nf, Nt)
    "c2": ((4, 7), 2000, 64, 200),
    "mid": ((4, 7), 1000, 16, 50),
    "c1": ((3, 4), 100, 16, 20),
    # c5s = BASELINE config 5 scaled to one box-minute: the 350-antenna layout (331-element hex +
    # 19 outriggers on a 300 m ring), gridded beam with bicubic interpolation,
    # baseline-sharded Gram + all-reduce; 8192 sources instead of 100 000, 8 channels x 1 time
    "c5s": ((11, 21), 8192, 8, 1),
    # c5 = BASELINE config 5: 350 antennas, 100 000 sources (80 GB Gram), gridded beam, 128
    # channels x 1 time on 8 GPUs = 16 CHANNELS PER GPU: `--gpus N` runs 16 N channels (one GPU: its
    # share of the node's job; 8 GPUs: the whole of config 5), so this config scales WEAKLY;
    # solved densely as the reference's driver does whatever --solver says (example.py:940-947):
    # 256-wide-panel Cholesky on the matrix cores (csrc/chol.hip), row-sharded over the ranks on
    # more than one GPU; --solver cg gives the reference's unpreconditioned CG instead
    "c5": ((11, 21), 100000, 16, 1),
    # c5q = a quick look at the same system: 2 channels (what rounds 1-2 called c5)
    "c5q": ((11, 21), 100000, 2, 1),
}
PER_GPU_CHANNELS = ("c5",)       # Nf above is per GPU: the job has Nf x n_gpus channels
#: up to this many sources a multi-GPU job all-reduces the system and every rank solves it
#: ("small N"); above it the root solves, or the operator stays sharded by rows
REPLICATE_MAX_N = 20000
SH_CONFIGS = {
    # spherical-harmonic head -- name: (hex_spec, lmax, nside, Nf, Nt)
    # c3 = BASELINE config 3 (HERA-127, lmax 40, 128 freqs); Nt is left open by BASELINE.json and
    # is set so that the response operator (Nbl*Nf*Nt x Nm complex128 = 27.5 GB per time) of all
    # 128 channels stays resident in one GPU's 288 GB of HBM.
    "c3": ((7, 13), 40, 32, 128, 4),
    "c3s": ((4, 7), 10, 8, 8, 3),
    # SURVEY 8(d)'s C3: 40 LSTs.  The operator would be 1.1 TB: sh_sampler.StreamedSHOperator keeps
    # the pixel voltages (128 GB) instead and never forms it (right-hand side through the factored
    # per-pixel operator, left-hand-side matrix accumulated over LST blocks at set-up)
    "c3t40": ((7, 13), 40, 32, 128, 40),
    "c3st": ((4, 7), 10, 8, 8, 5),
    # c3's size through the same factored path: no operator in memory although it would fit
    "c3f": ((7, 13), 40, 32, 128, 4),
}
SH_STREAMED = ("c3t40", "c3st", "c3f")
GAIN_CONFIGS = {
    # linearised gain head -- name: (hex_spec, Nf, Nt, nmaxfreq, nmaxtime, batch)
    # c4 = BASELINE config 4 (HERA-331 per-antenna perturbations, batched CG matvecs)
    "c4": ((11, 21), 64, 64, 2, 2, 8),
    "c4s": ((4, 7), 16, 12, 2, 2, 8),
}


COSMO_CONFIGS = {
    # cosmological-field head -- name: (hex_spec, RA grid, Dec grid, Nf, Nt, sigma)
    # the c2 array and (f,t) grid with a 32 x 32 field grid: 64 per-channel Grams of 1024^2,
    # CG on the (S^-1 + blockdiag G_f) operator over 65 536 unknowns.  The field is faint: the
    # noise level (in units of the unit-flux pixels) is set so that the reference's
    # unpreconditioned CG converges within its 1000-iteration default (linear_solver.py:378).
    "cosmo": ((4, 7), 32, 32, 64, 200, 10.0),
    "cosmos": ((3, 4), 8, 6, 8, 10, 2.0),
}


BEAM_CONFIGS = {
    # beam-head consumers of the generator -- name: (hex_spec, Nsrc, Nf, Nt, nmax, mmax)
    # Fourier-Bessel basis of nmax radial x (2 mmax + 1) azimuthal modes
    "beam": ((4, 7), 500, 16, 30, 8, 2),
    "beams": ((3, 4), 60, 3, 4, 3, 1),
}


def synth_inputs(hex_spec, ns, nf, nt, outriggers=0):
    """Synthetic inputs of SURVEY section 8(d): hex array (14.6 m), random catalogue seed 0,
    100-120 MHz, LST 0.2-0.5 rad, Gaussian beam D = 14 m, gains (1+1j) + 0.05 N(0,1),
    sigma_noise = 0.05, prior std 0.1."""
    from hydra_amd import example, utils
    np.random.seed(0)
    ant_pos = utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    if outriggers:
        n0 = len(ant_pos)
        for k in range(outriggers):
            ang = 2.0 * np.pi * k / outriggers
            ant_pos[n0 + k] = np.array([300.0 * np.cos(ang), 300.0 * np.sin(ang), 0.0])
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


def pmc_traffic(config, kernel=None):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes of
    this same command (profiles/pmc_traffic.json; tools/gpu_check.sh collects them), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as fh:
            entry = json.load(fh)[config]
        if kernel is not None and kernel in entry:
            entry = entry[kernel]
        elif kernel is not None and entry.get("kernel") != kernel:
            return None
        return entry["traffic_bytes"]
    except (OSError, KeyError, ValueError):
        return None


def _blas_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        return int(os.environ.get("OMP_NUM_THREADS", "1"))


def cpu_baseline(inp, nf_s, nt_s, full_rows, single_thread=False):
    """Time the oracle (numpy/scipy restatement of the reference) on an (nf_s x nt_s) slice of
    the workload and scale linearly in the number of visibility rows.  ``single_thread`` runs
    the same slice with the BLAS pool limited to one thread (SURVEY 8(d) asks for both)."""
    if single_thread:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):
            out = cpu_baseline(inp, nf_s, nt_s, full_rows)
        out["cores"] = 1
        return out
    from oracle import beams as obeams, gcr_ptsrc, helpers
    threads = _blas_threads()
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


def sh_cpu_baseline(rows, full_rows, nm):
    """The oracle's SH head (numpy restatement of sh_sampler.construct_rhs_no_rot + cg with the
    reference's per-iteration apply_lhs_no_rot) on ``rows`` visibility rows, scaled linearly in
    rows; the reference forms the (Nm,Nm) Gram inside EVERY CG iteration (sh_sampler.py:393-398)."""
    from oracle import gcr_sh
    threads = _blas_threads()
    # a random operator of the right shape stands in for the response (building the real one on
    # the CPU takes minutes and is not part of a sample); the arithmetic timed is identical
    rng = np.random.RandomState(0)
    R = (rng.randn(rows, nm) + 1j * rng.randn(rows, nm)) * 1e-3
    w = np.full(rows, 400.0)
    ip = np.full(nm, 100.0)
    d = rng.randn(rows) + 1j * rng.randn(rows)
    om1 = (rng.randn(rows) + 1j * rng.randn(rows)) / np.sqrt(2)
    om0 = rng.randn(nm)
    t0 = time.time()
    b = gcr_sh.construct_rhs_no_rot(d, w, ip, om0, om1, np.zeros(nm), R)
    t_rhs = time.time() - t0
    t0 = time.time()
    gcr_sh.apply_lhs_no_rot(b, w, ip, R)
    t_lhs = time.time() - t0
    scale = full_rows / rows
    return {"rhs_s": t_rhs * scale, "lhs_apply_s": t_lhs * scale, "rows": rows, "cores": int(threads)}


def main_sh_streamed(args, comm, rank, world, dev):
    """The SH head with the response operator never in memory (sh_sampler.StreamedSHOperator):
    per sample the right-hand side b = Y^T Re(P^H (N^-1 d + N^-1/2 w_n)) + prior terms through the
    factored per-pixel operator (one pass over the pixel voltages, HBM / FP64 bound) and a CG
    solve with the left-hand-side matrix accumulated over LST blocks at set-up."""
    import torch
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    hex_spec, lmax, nside, nf, nt = SH_CONFIGS[args.config]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    nbl = nant * (nant - 1) // 2
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fidx = np.array_split(np.arange(nf), world)[rank]
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    op = sh_sampler.StreamedSHOperator(freqs[fidx], times, beams, ant_pos, lmax, nside, time_block=1)
    sync()
    t_volt = time.perf_counter() - t0
    m_loc, nm = op.shape
    m_all = nbl * nf * nt
    gen = torch.Generator(device=dev).manual_seed(4321 + rank)
    gen_a = torch.Generator(device=dev).manual_seed(99)
    prior_std, sigma_noise = 0.1, 0.05
    inv_prior = torch.full((nm,), 1.0 / prior_std ** 2, dtype=torch.float64, device=dev)
    inv_noise = torch.full((m_loc,), 1.0 / sigma_noise ** 2, dtype=torch.float64, device=dev)
    if args.flags > 0.0:      # flagged samples: the closed-form left-hand side takes its correction
        inv_noise[torch.rand(m_loc, device=dev, generator=gen) < args.flags] = 0.0
    a_true = prior_std * torch.randn(nm, dtype=torch.float64, device=dev, generator=gen_a)
    data = op.apply(a_true).reshape(-1)
    data += sigma_noise * torch.randn(m_loc, dtype=torch.complex128, device=dev, generator=gen)
    a_0 = torch.zeros(nm, dtype=torch.float64, device=dev)
    sync()
    t0 = time.perf_counter()
    A = op.lhs_matrix(inv_noise, inv_prior, comm=comm)
    sync()
    t_lhs = time.perf_counter() - t0
    # --solver dense: the left-hand side does not change between samples, so it is factored once
    # here and a sample costs two triangular solves (the reference runs its CG per sample)
    factor = hy.linear_solver.cho_factor(A) if args.solver == "dense" else None

    def one_sample(x0):
        om_a = torch.randn(nm, dtype=torch.float64, device=dev, generator=gen_a)
        om_n = torch.randn(m_loc, dtype=torch.complex128, device=dev, generator=gen)
        b = op.construct_rhs(data, inv_noise, inv_prior, om_a, om_n, a_0, comm=comm)
        if factor is not None:
            return b, hy.linear_solver.cho_solve(factor, b), 0
        x, iters = hy.linear_solver.cg(A, b, maxiters=1000, abs_tol=args.cg_tol, x0=x0,
                                       return_iters=True)      # replicated A: every rank solves
        return b, x, iters

    x = torch.zeros(nm, dtype=torch.float64, device=dev)
    for _ in range(args.warmup):
        b, x, iters = one_sample(x)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b, x, iters = one_sample(x)
    sync()
    elapsed = time.perf_counter() - t0
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if comm is not None:
        comm.allreduce_(t_all, op="max")
    elapsed = float(t_all.item())
    # the adjoint pass alone, HIP events on the launch stream
    z = torch.randn(m_loc, dtype=torch.complex128, device=dev, generator=gen)
    op.adjoint(z)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2):
        op.adjoint(z)
    e1.record()
    torch.cuda.synchronize()
    adj_ms = e0.elapsed_time(e1) / 2
    if rank != 0:
        return
    resid = float((torch.linalg.norm(A @ x - b) / torch.linalg.norm(b)).item())
    vbytes = 16.0 * op.V.numel() + 16.0 * m_loc            # V once + z
    flops = 8.0 * m_loc * op.npix
    out = {
        "metric": "GCR samples/sec (SH head, whole job)",
        "value": args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config 3 ({args.config}"
                         + (", SURVEY 8(d)'s 40 LSTs" if nt == 40 else "") + "): SH diffuse GCR, "
                         f"hex{hex_spec} = {nant} antennas / {nbl} baselines, lmax {lmax} ({nm} real "
                         f"modes), nside {nside}, {nf} freqs x {nt} times, complex128; the response "
                         f"operator ({16e-9 * m_all * nm:.0f} GB) is never formed: pixel voltages "
                         f"resident ({16e-9 * op.V.numel():.1f} GB per GPU), left-hand-side matrix "
                         f"accumulated over LST blocks at set-up"),
            "rows_M": m_all,
            "solver": "cg" if factor is None else "dense, factored once at set-up (cho_factor / cho_solve)",
            "cg_abs_tol": args.cg_tol, "cg_iters_last": int(iters),
            "parallelism": f"freq-sharded x{world}" if world > 1 else "single GPU",
        },
        "setup_voltages_s": t_volt, "setup_lhs_matrix_s": t_lhs,
        "setup_lhs_path": ("closed form through pixel space (baseline-independent weights, complete array)"
                           + (f" minus the Gram of {op.last_lhs_flagged_rows} flagged rows"
                              if getattr(op, "last_lhs_flagged_rows", 0) else "")
                           if op.fac.last_gram_kernel == "gram_uniform_kernel" else
                           "Gram of the regenerated operator, LST block by LST block"),
        "flagged_fraction": args.flags,
        "setup_lhs_tflops_operator_form": (2.0 * m_loc * op.npix * nm + 2.0 * m_loc * nm * (nm + 1)) / t_lhs / 1e12,
        "sample_rel_residual": resid,
        "roofline": ({
            "bound": "mfma", "achieved": flops / (adj_ms * 1e-3) / 1e12, "peak": 78.6,
            "unit": "TFLOP/s", "frac": flops / (adj_ms * 1e-3) / 1e12 / 78.6,
            "traffic": pmc_traffic(args.config, "adjoint_V_big_kernel") if world == 1 else None,
            "kernel": "adjoint_V_big_kernel (R^H z in antenna space, W = Zc V on FP64 MFMA) + Y^T GEMV",
            "ms_per_launch": adj_ms, "flops_per_launch": flops,
            "hbm_GBps": vbytes / (adj_ms * 1e-3) / 1e9,
            "note": ("algorithmic flops = 8 per (baseline, f, t, pixel) complex multiply-add; the "
                     "kernel executes 3 real 16x16x4 products per complex tile over the upper "
                     "triangle of 16-antenna blocks"),
        } if (64 < nant <= 160 and op.fac.unique_upper and op.fac.antspace_adjoint) else {
            "bound": "hbm", "achieved": vbytes / (adj_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
            "frac": vbytes / (adj_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
            "kernel": "adjoint_V_kernel (R^H z through the factored per-pixel operator) + Y^T GEMV",
            "ms_per_launch": adj_ms, "bytes_per_launch": vbytes,
            "fp64_tflops": flops / (adj_ms * 1e-3) / 1e12,
        }),
        "cpu_baseline": None,
    }
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        cb = sh_cpu_baseline(min(m_all, 1000), m_all, nm)
        iters_ref = max(int(iters), 1)
        t_sample = cb["rhs_s"] + (iters_ref + 1) * cb["lhs_apply_s"]
        out["cpu_baseline"] = {
            "value": 1.0 / t_sample, "unit": "samples/s", "cores": cb["cores"], "kind": "port",
            "sample": (f"oracle (numpy restatement of sh_sampler.construct_rhs_no_rot and "
                       f"apply_lhs_no_rot) timed on {cb['rows']} of {m_all} rows x {nm} modes and "
                       f"scaled linearly in rows: rhs {cb['rhs_s']:.1f}s + ({iters_ref}+1 CG "
                       f"operator applications, each forming the Gram as the reference does) x "
                       f"{cb['lhs_apply_s']:.1f}s"),
        }
    print(json.dumps(out))


def main_sh(args, comm, rank, world, dev):
    """GCR samples/s of the spherical-harmonic head (example.py:990-1069): per sample fresh
    (w_a, w_n), right-hand side b = R^T (N^-1 d + N^-1/2 w_n) + S^-1 a_0 + S^-1/2 w_a (one pass
    over the response operator, HBM bound) and a CG solve of (R^T N^-1 R + S^-1) x = b to the
    reference tolerance.  The operator R (sh_sampler.vis_proj_operator_no_rot) and the matrix
    of the left-hand side (sh_sampler.lhs_matrix, FP64 MFMA) are built once before the Gibbs
    loop -- their build times are reported as setup_*; the reference recomputes the Gram inside
    every CG iteration (sh_sampler.py:393-398)."""
    import torch
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    hex_spec, lmax, nside, nf, nt = SH_CONFIGS[args.config]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    nbl = nant * (nant - 1) // 2
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fidx = np.array_split(np.arange(nf), world)[rank]
    nf_loc = fidx.size
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    R, ell, m = sh_sampler.vis_proj_operator_no_rot(freqs[fidx], times, beams, ant_pos, lmax, nside,
                                                    as_torch=True, cross_only=True)
    sync()
    t_resp = time.perf_counter() - t0
    m_loc, nm = R.shape
    m_all = nbl * nf * nt
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    prior_std = 0.1
    inv_prior = torch.full((nm,), 1.0 / prior_std ** 2, dtype=torch.float64, device=dev)
    sigma_noise = 0.05
    inv_noise = torch.full((m_loc,), 1.0 / sigma_noise ** 2, dtype=torch.float64, device=dev)
    gen_a = torch.Generator(device=dev)
    gen_a.manual_seed(99)                      # the same sky on every rank
    a_true = prior_std * torch.randn(nm, dtype=torch.float64, device=dev, generator=gen_a)
    data = torch.empty(m_loc, dtype=torch.complex128, device=dev)
    from hydra_amd import _dev as dv
    dv.call("hydra_apply_R", dv.ptr(R), m_loc, nm, dv.ptr(a_true), dv.ptr(None), dv.ptr(data),
            dv.stream_ptr())
    data += sigma_noise * torch.randn(m_loc, dtype=torch.complex128, device=dev, generator=gen)
    a_0 = torch.zeros(nm, dtype=torch.float64, device=dev)
    sync()
    t0 = time.perf_counter()
    A = sh_sampler.lhs_matrix(inv_noise, inv_prior, R, comm=comm)
    sync()
    t_gram = time.perf_counter() - t0
    factor = hy.linear_solver.cho_factor(A) if (args.solver == "dense" and rank == 0) else None

    def one_sample(x0):
        om_a = torch.randn(nm, dtype=torch.float64, device=dev, generator=gen_a)
        om_n = torch.randn(m_loc, dtype=torch.complex128, device=dev, generator=gen)
        if comm is None:
            b = sh_sampler.construct_rhs_no_rot(data, inv_noise, inv_prior, om_a, om_n, a_0, R)
        else:
            b = sh_sampler.construct_rhs_no_rot_mpi(comm, data, inv_noise, inv_prior, om_a, om_n,
                                                    a_0, R)
        if rank == 0 and factor is not None:
            x, iters = hy.linear_solver.cho_solve(factor, b), 0
        elif rank == 0:
            x, iters = hy.linear_solver.cg(A, b, maxiters=1000, abs_tol=args.cg_tol, x0=x0,
                                           return_iters=True)
        else:
            x, iters = torch.zeros(nm, dtype=torch.float64, device=dev), 0
        if comm is not None:
            comm.bcast_(x, root=0)
        return b, x, iters

    x = torch.zeros(nm, dtype=torch.float64, device=dev)
    for _ in range(args.warmup):
        b, x, iters = one_sample(x)
    sh_sampler.RHS_EVENTS = []
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b, x, iters = one_sample(x)
    sync()
    elapsed = time.perf_counter() - t0
    events = sh_sampler.RHS_EVENTS
    sh_sampler.RHS_EVENTS = None
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if comm is not None:
        comm.allreduce_(t_all, op="max")
    elapsed = float(t_all.item())
    adj_ms = float(np.mean([s.elapsed_time(e) for s, e in events])) if events else float("nan")
    if rank != 0:
        return
    resid = float((torch.linalg.norm(A @ x - b) / torch.linalg.norm(b)).item())
    adj_bytes = 16.0 * m_loc * nm + 16.0 * m_loc          # R once + z
    achieved = adj_bytes / (adj_ms * 1e-3) / 1e9
    out = {
        "metric": "GCR samples/sec (SH head, whole job)",
        "value": args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config {args.config}: SH diffuse GCR, hex{hex_spec} = {nant} "
                         f"antennas / {nbl} baselines, lmax {lmax} ({nm} real modes), nside {nside}, "
                         f"{nf} freqs x {nt} times, complex128, response operator resident in HBM "
                         f"({16e-9 * m_loc * nm:.1f} GB per GPU), left-hand-side matrix built once"),
            "rows_M": m_all,
            "solver": "cg" if args.solver != "dense" else "dense, factored once at set-up (cho_factor / cho_solve)",
            "cg_abs_tol": args.cg_tol, "cg_iters_last": int(iters),
            "parallelism": f"freq-sharded x{world}" if world > 1 else "single GPU",
        },
        "setup_response_s": t_resp, "setup_lhs_matrix_s": t_gram,
        "projection_tflops": 4.0 * m_loc * (12 * nside ** 2) * nm / t_resp / 1e12,
        "lhs_gram_tflops": 2.0 * m_loc * nm * (nm + 1) / t_gram / 1e12,
        "sample_rel_residual": resid,
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0, "traffic": pmc_traffic(args.config) if world == 1 else None,
            "kernel": "adjoint_R_kernel", "ms_per_launch": adj_ms, "bytes_per_launch": adj_bytes,
        },
    }
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        cb = sh_cpu_baseline(min(m_all, 1000), m_all, nm)
        iters_ref = max(int(iters), 1)
        t_sample = cb["rhs_s"] + (iters_ref + 1) * cb["lhs_apply_s"]
        out["cpu_baseline"] = {
            "value": 1.0 / t_sample, "unit": "samples/s", "cores": cb["cores"], "kind": "port",
            "sample": (f"oracle (numpy restatement of sh_sampler.construct_rhs_no_rot and "
                       f"apply_lhs_no_rot) timed on {cb['rows']} of {m_all} rows x {nm} modes and "
                       f"scaled linearly in rows: rhs {cb['rhs_s']:.1f}s + ({iters_ref}+1 CG "
                       f"operator applications, each forming the Gram as the reference does) x "
                       f"{cb['lhs_apply_s']:.1f}s"),
        }
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


def gain_cpu_baseline(ants, antpairs, Fb_all, nf, nt, iters, seed=0):
    """The oracle's gain head (numpy restatement of gain_sampler.construct_rhs_mpi and
    apply_operator_mpi) on a 2 x 8 (freq x time) slice of the workload; both are sums over
    (f,t) samples of per-sample work, so the time is scaled linearly in Nf*Nt.  One sample =
    one right-hand side + ``iters`` operator applications (the CG of the driver)."""
    from oracle import gcr_gain
    nf_s, nt_s = min(nf, 2), min(nt, 8)
    nant, nbl = len(ants), len(antpairs)
    i1, i2 = gcr_gain.bl_indices(ants, antpairs)
    rng = np.random.RandomState(seed)
    Fb = np.ascontiguousarray(Fb_all[:, :nf_s, :nt_s])
    nmodes = Fb.shape[0]
    shape = (nbl, nf_s, nt_s)
    mv = rng.randn(*shape) + 1j * rng.randn(*shape)
    w = np.full(shape, 400.0)
    ps = np.full(nmodes, 0.05)
    x = rng.randn(2 * nant * nmodes)
    om_y, om_r = gcr_gain.draw_gain_omegas(1, nant, nmodes, shape)
    t0 = time.perf_counter()
    gcr_gain.construct_rhs(mv, w, ps, i1, i2, mv, Fb, om_y, om_r)
    t_rhs = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(2):
        gcr_gain.apply_operator(x, w, ps, i1, i2, mv, Fb)
    t_op = (time.perf_counter() - t0) / 2
    scale = (nf * nt) / (nf_s * nt_s)
    total = scale * (t_rhs + max(int(iters), 1) * t_op)
    return {"value": 1.0 / total, "unit": "samples/s", "cores": _blas_threads(), "kind": "port",
            "sample": (f"oracle (numpy restatement of gain_sampler.construct_rhs_mpi + apply_operator_mpi) "
                       f"on a {nf_s}x{nt_s} of {nf}x{nt} (freq x time) slice, all {nbl} baselines: rhs "
                       f"{t_rhs:.2f}s + {int(iters)} CG operator applications x {t_op:.2f}s, scaled "
                       f"linearly in Nf*Nt")}


def main_gain(args, comm, rank, world, dev):
    """GCR samples/s of the linearised gain head (example.py:673-745): per sample the residual
    d - g g* V, the right-hand side (gain_sampler.construct_rhs_mpi) and a CG solve of
    x + S^1/2 F^H A^H N^-1 A F S^1/2 x = b with the matrix-free operator (use_norm_tol=True as in
    the driver).  Frequency channels are sharded over ranks; the operator result is all-reduced
    once per CG iteration.  Also times the batched operator (``batch`` vectors per pass over the
    weights) for the HBM roofline of the pair kernel."""
    import torch
    import hydra_amd as hy
    from hydra_amd import gain_sampler as gs
    hex_spec, nf, nt, nmaxf, nmaxt, batch = GAIN_CONFIGS[args.config]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fidx = np.array_split(np.arange(nf), world)[rank]
    nf_loc = fidx.size
    Fb_all, _, _ = hy.utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, nmaxf, nmaxt, freqs[-1] - freqs[0], times[-1] - times[0])
    nmodes = Fb_all.shape[0]
    Fb = torch.as_tensor(np.ascontiguousarray(Fb_all[:, fidx, :]), device=dev)
    ps = torch.full((nmodes,), 0.05, dtype=torch.float64, device=dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + rank)
    shape = (nbl, nf_loc, nt)
    model = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    gains = torch.as_tensor((1.0 + 1.0j) + 0.05 * (np.random.RandomState(2).randn(nant)
                            + 1j * np.random.RandomState(3).randn(nant)), device=dev)
    gains = gains[:, None, None].expand(nant, nf_loc, nt).contiguous()
    sigma = 0.05
    inv_noise = torch.full(shape, 1.0 / sigma ** 2, dtype=torch.float64, device=dev)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    n = 2 * nant * nmodes
    rng = np.random.RandomState(7)
    x_true = torch.as_tensor(rng.randn(n), device=dev)
    # data = g g* (1 + x_i + x_j*) V + noise, with x = F S^1/2 x_true per antenna
    ggv = hy.utils.apply_gains(model, gains, ants, antpairs, inline=False)
    op0 = gs.GainOperator(inv_noise, ps, A_real, A_imag, ggv, Fb)
    g_field = torch.empty((1, nant, nf_loc * nt), dtype=torch.complex128, device=dev)
    from hydra_amd import _dev as dv
    dv.call("hydra_gain_basis_fwd", dv.ptr(x_true.reshape(1, -1)), 1, nant, nmodes, nf_loc * nt,
            dv.ptr(Fb), dv.ptr(ps), dv.ptr(g_field), dv.stream_ptr())
    data = ggv + gs.apply_proj(g_field.reshape(nant, nf_loc, nt), A_real, A_imag, ggv)
    data = data + sigma * torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    del op0, g_field
    m_loc, m_all = nbl * nf_loc * nt, nbl * nf * nt

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    def one_sample(k):
        om_y = torch.randn((nant, nmodes), dtype=torch.complex128, device="cpu",
                           generator=torch.Generator().manual_seed(1000 + k)).numpy()
        om_r = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
        if args.gain_unfused:
            ggv_k = hy.utils.apply_gains(model, gains, ants, antpairs, inline=False)
            resid = data - ggv_k
            b = gs.construct_rhs_mpi(comm, resid, inv_noise, ps, A_real, A_imag, ggv_k, Fb,
                                     omega_y=om_y, omega_r=om_r)
            op = gs.GainOperator(inv_noise, ps, A_real, A_imag, ggv_k, Fb)
        else:
            # the same four steps in one pass over the visibilities (gain_sampler.prepare_sample)
            b, op = gs.prepare_sample(comm, data, model, gains, inv_noise, ps, A_real, A_imag, Fb,
                                      omega_y=om_y, omega_r=om_r)
        if comm is None:
            lin = op.apply
        else:
            def lin(v):
                t = op.local(v)
                comm.allreduce_(t)
                return v + t
            lin.comm = comm
        x, iters = hy.linear_solver.cg(None, b, linear_op=lin, use_norm_tol=True,
                                       abs_tol=args.cg_tol, maxiters=1000, return_iters=True)
        return op, b, x, iters

    for k in range(args.warmup):
        one_sample(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        op, b, x, iters = one_sample(100 + k)
    sync()
    elapsed = time.perf_counter() - t0
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if comm is not None:
        comm.allreduce_(t_all, op="max")
    elapsed = float(t_all.item())

    # operator alone: HIP events on the launch stream around `reps` applications, for one vector
    # (what the CG loop does) and for a batch
    def timed(arg, reps=10):
        op.local(arg)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            op.local(arg)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    X = torch.randn((batch, n), dtype=torch.float64, device=dev, generator=gen)
    one_ms = timed(X[0].contiguous())
    mv_ms = timed(X)
    # the pair kernel alone (the dominant kernel of an application), same events
    from hydra_amd import _dev as dv
    nft_loc = nf_loc * nt
    g1 = torch.randn((1, nant, nft_loc), dtype=torch.complex128, device=dev, generator=gen)
    y1 = torch.empty_like(g1)

    def pair_once():
        if op.stream:
            dv.call("hydra_gain_pairs_stream", dv.ptr(g1), nant, nft_loc, dv.ptr(op.Wt),
                    dv.ptr(op.sw), dv.ptr(y1), dv.stream_ptr())
        else:
            dv.call("hydra_gain_pairs_tiled", dv.ptr(g1), 1, nant, nft_loc, dv.ptr(op.W), nbl,
                    dv.ptr(op.idx.pair_dev()), dv.ptr(y1), dv.stream_ptr())
    pair_once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        pair_once()
    e1.record()
    torch.cuda.synchronize()
    pair_ms = e0.elapsed_time(e1) / 20
    pair_kernel = "gain_pairs_stream_kernel" if op.stream else "gain_pairs_tiled_kernel"
    # the one-pass preparation kernel alone (model, data, weights, omega read once; W written once)
    prep = None
    if not args.gain_unfused and op.stream and getattr(op, "_model", None) is not None:
        om_r = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
        sw1, yy1 = torch.empty_like(op.sw), torch.empty((nant, nft_loc), dtype=torch.complex128, device=dev)
        idx = gs._index_of(A_real, A_imag)

        def prep_once():
            dv.call("hydra_gain_prepare_stream", dv.ptr(data), dv.ptr(model), dv.ptr(gains),
                    dv.ptr(inv_noise), dv.ptr(om_r), nbl, nant, nft_loc, dv.ptr(idx.pair_dev()),
                    dv.ptr(op.Wt), dv.ptr(sw1), dv.ptr(yy1), dv.stream_ptr())
        prep_once()
        e0.record()
        for _ in range(5):
            prep_once()
        e1.record()
        torch.cuda.synchronize()
        prep_ms = e0.elapsed_time(e1) / 5
        prep_bytes = 64.0 * m_loc            # 16 + 16 + 8 + 16 read, 8 written per visibility
        prep = {"kernel": "gain_prepare_stream_kernel", "ms_per_launch": prep_ms,
                "bytes_per_launch": prep_bytes, "achieved": prep_bytes / (prep_ms * 1e-3) / 1e9,
                "peak": 8000.0, "unit": "GB/s", "frac": prep_bytes / (prep_ms * 1e-3) / 1e9 / 8000.0}
        del om_r, sw1, yy1
    if rank != 0:
        return
    lhs = x + op.local(x) if comm is None else None
    resid_rel = float((torch.linalg.norm(lhs - b) / torch.linalg.norm(b)).item()) if lhs is not None \
        else float("nan")
    alg_bytes = 8.0 * m_loc              # the fused real weight, read once
    achieved = alg_bytes / (pair_ms * 1e-3) / 1e9
    out = {
        "metric": "GCR samples/sec (gain head, whole job)",
        "value": args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"BASELINE config {args.config}: linearised gain GCR, hex{hex_spec} = {nant} "
                         f"antennas / {nbl} baselines, {nf} freqs x {nt} times, {nmodes} Fourier "
                         f"modes per antenna ({n} real unknowns), matrix-free CG, complex128"),
            "rows_M": m_all, "solver": "cg (use_norm_tol)", "cg_abs_tol": args.cg_tol,
            "cg_iters_last": int(iters),
            "parallelism": f"freq-sharded x{world}" if world > 1 else "single GPU",
        },
        "sample_rel_residual": resid_rel,
        "batched_matvec": {
            "vectors": batch, "ms": mv_ms, "matvecs_per_s": batch / (mv_ms * 1e-3),
            "kernel": ("gain_pairs_mfma_kernel (per sample: symmetric weight matrix x 16 real "
                       "columns on the FP64 matrix cores, weights in 16 x 16 antenna blocks, each "
                       "read once per 8 vectors) between gain_basis_fwd_blocks_kernel and "
                       "gain_basis_adj_mfma_kernel, fields sample-major"
                       if (op.batch_mfma and op.Wb is not None) else "gain_pairs_tiled_kernel"),
            "weights_GBps": alg_bytes / (mv_ms * 1e-3) / 1e9,
            "frac_of_hbm_on_8M_bytes": alg_bytes / (mv_ms * 1e-3) / 1e9 / 8000.0,
        },
        "roofline": {
            "bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
            "frac": achieved / 8000.0,
            "traffic": pmc_traffic(args.config, pair_kernel) if world == 1 else None,
            "kernel": pair_kernel, "ms_per_launch": pair_ms, "bytes_per_launch": alg_bytes,
            "note": "8 M bytes: the fused real weight |g g* V|^2 N^-1 of every visibility once",
        },
        "operator_application_ms": one_ms,
        "prepare_pass": prep,
        "cpu_baseline": None,
    }
    out["config"]["preparation"] = ("separate passes (apply_gains, residual, construct_rhs_mpi, "
                                    "GainOperator)" if args.gain_unfused else
                                    "one pass over the visibilities (gain_sampler.prepare_sample)")
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        out["cpu_baseline"] = gain_cpu_baseline(ants, antpairs, Fb_all, nf, nt, iters)
    print(json.dumps(out))


def main_cosmo(args, comm, rank, world, dev):
    """GCR samples/s of the cosmological-field head (example.py:751-851): per sample the
    per-channel likelihood blocks and right-hand side (cosmo_sampler.precompute_mpi, factored
    operator, frequency channels sharded over ranks, summed to the root) and the root's CG solve
    with the FFT prior + batched block matvec operator."""
    import torch
    import hydra_amd as hy
    from hydra_amd import cosmo_sampler as cs, ptsrc_sampler
    hex_spec, n_ra, n_dec, nf, nt, sigma = COSMO_CONFIGS[args.config]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fidx = np.array_split(np.arange(nf), world)[rank]
    nf_loc = fidx.size
    gargs = argparse.Namespace(cosmo_field_ra_bounds=(0.0, 60.0), cosmo_field_dec_bounds=(-40.0, -20.0),
                               cosmo_field_ra_ngrid=n_ra, cosmo_field_dec_ngrid=n_dec)
    g_ra, g_dec = cs.make_cosmo_field_grid(gargs)
    npix = g_ra.size
    gx, gy = np.unique(g_ra), np.unique(g_dec)
    kbins = np.linspace(0.0, 10.0, 25)
    pspec3d = cs.calculate_pspec_on_grid(kbins=kbins, pspec=0.1 * kbins + 1.0, x=gx, y=gy,
                                         freqs=freqs, h=0.69, omega_m=0.31)
    shape3 = pspec3d.shape
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    lat = -30.7215 * np.pi / 180.0
    fac = ptsrc_sampler.calc_proj_operator(g_ra, g_dec, np.ones((npix, nf_loc)), ant_pos, antpairs,
                                           freqs[fidx], times, beams, lat, materialize=False)
    g0 = (1.0 + 1.0j) + 0.05 * (np.random.RandomState(2).randn(nant)
                                + 1j * np.random.RandomState(3).randn(nant))
    gains = torch.as_tensor(g0, device=dev)[:, None, None].expand(nant, nf_loc, nt).contiguous()
    gen = torch.Generator(device=dev)
    gen.manual_seed(99 + rank)
    field = 0.3 * np.random.RandomState(5).randn(nf, npix)
    data = torch.empty((nbl, nf_loc, nt), dtype=torch.complex128, device=dev)
    for j in range(nf_loc):
        data[:, j:j + 1] = fac.channel(j).apply(torch.as_tensor(field[fidx[j]], device=dev))
    data = hy.utils.apply_gains(data, gains, ants, antpairs, inline=False)
    data = data + sigma * torch.randn(data.shape, dtype=torch.complex128, device=dev, generator=gen)
    inv_noise = torch.full((nbl, nf_loc, nt), 1.0 / sigma ** 2, dtype=torch.float64, device=dev)
    p3d = torch.as_tensor(pspec3d, device=dev)
    m_loc, m_all = nbl * nf_loc * nt, nbl * nf * nt

    def one_sample():
        om_n = torch.randn((nbl, nf_loc, nt), dtype=torch.complex128, device=dev, generator=gen)
        om_s = torch.randn(shape3, dtype=torch.float64, device=dev, generator=gen)
        op, rhs = cs.precompute_mpi(comm, ants, antpairs, freqs, freqs[fidx], times, fac, data,
                                    inv_noise, gains, p3d, omega_n=om_n, omega_s=om_s)
        x, iters = None, 0
        if rank == 0:
            x, iters = hy.linear_solver.cg(
                Amat=None, bvec=rhs.flatten(), maxiters=1000, abs_tol=args.cg_tol, return_iters=True,
                linear_op=lambda v: cs.apply_lhs_operator(v.reshape(shape3), op, p3d).flatten())
        return op, rhs, x, iters

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
        op, rhs, x, iters = one_sample()
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
    lhs = cs.apply_lhs_operator(x.reshape(shape3), op, p3d).flatten()
    resid = float((torch.linalg.norm(lhs - rhs.flatten()) / torch.linalg.norm(rhs)).item())
    # one launch = one channel's Gram (Nt samples): flop counts as in main()
    kernel = getattr(fac, "last_gram_kernel", "gram_fused_kernel")
    base_flops = 2.0 * (nbl * nt) * npix * (npix + 1)
    if kernel == "gram_uniform_kernel":
        per_elem = (6.0 * nant + 10.0) if fac.same_beam else (8.0 * nant + 8.0)
        gram_flops = nt * 0.5 * npix * (npix + 1) * per_elem
    else:
        gram_flops = base_flops
    achieved = gram_flops / (gram_ms * 1e-3) / 1e12
    out = {
        "metric": "GCR samples/sec (cosmo field head, whole job)",
        "value": args.steps / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"cosmological-field GCR ({args.config}): hex{hex_spec} = {nant} antennas / "
                         f"{nbl} baselines, {n_ra} x {n_dec} field grid = {npix} pixels per channel, "
                         f"{nf} freqs x {nt} times ({nf * npix} unknowns), noise sigma {sigma}, factored operator, "
                         f"per-channel Grams + CG with FFT prior, complex128"),
            "rows_M": m_all, "solver": "cg", "cg_abs_tol": args.cg_tol, "cg_iters_last": int(iters),
            "parallelism": f"freq-sharded x{world}" if world > 1 else "single GPU",
        },
        "gram_tflops_effective_baseline_form": world * base_flops / (gram_ms * 1e-3) / 1e12,
        "sample_rel_residual": resid,
        "roofline": {
            "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic(args.config, kernel) if world == 1 else None,
            "kernel": kernel + " (one channel per launch)", "ms_per_launch": gram_ms,
            "flops_per_launch": gram_flops,
        },
        "cpu_baseline": None,
    }
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        out["cpu_baseline"] = cosmo_cpu_baseline(ant_pos, antpairs, freqs, times, g_ra, g_dec, g0,
                                                 sigma, pspec3d, op.cpu().numpy(), int(iters), m_all)
    print(json.dumps(out))


def cosmo_cpu_baseline(ant_pos, antpairs, freqs, times, g_ra, g_dec, g0, sigma, pspec3d, op_full,
                       iters, full_rows):
    """The oracle (numpy restatement of cosmo_sampler.precompute_mpi + linear_solver.cg with
    apply_lhs_operator) on this host: precompute on a 1-channel x 10-time slice, scaled linearly
    in rows; the CG as iterations x one timed application of the full-size operator."""
    from oracle import beams as obeams, gcr_cosmo as oc, gcr_ptsrc as op_
    nant = len(ant_pos)
    nf_s, nt_s = 1, min(10, times.size)
    f_s, t_s = freqs[:nf_s], times[:nt_s]
    beams = [obeams.GaussianBeam(14.0) for _ in range(nant)]
    t0 = time.perf_counter()
    proj = op_.calc_proj_operator(g_ra, g_dec, np.ones((g_ra.size, nf_s)), ant_pos, antpairs, f_s, t_s,
                                  beams, -30.7215 * np.pi / 180.0)
    t_resp = time.perf_counter() - t0
    nbl = len(antpairs)
    rng = np.random.RandomState(0)
    gains = g0[:, None, None] * np.ones((nant, nf_s, nt_s))
    data = rng.randn(nbl, nf_s, nt_s) + 1j * rng.randn(nbl, nf_s, nt_s)
    inv_noise = np.full((nbl, nf_s, nt_s), 1.0 / sigma ** 2)
    t0 = time.perf_counter()
    oc.precompute(np.arange(nant), antpairs, freqs, f_s, proj, data, inv_noise, gains, pspec3d)
    t_pre = time.perf_counter() - t0
    rows = nbl * nf_s * nt_s
    xv = rng.randn(*pspec3d.shape)
    t0 = time.perf_counter()
    for _ in range(3):
        oc.apply_lhs_operator(xv, op_full, pspec3d)
    t_mv = (time.perf_counter() - t0) / 3
    total = t_pre * full_rows / rows + iters * t_mv
    return {"value": 1.0 / total, "unit": "samples/s", "cores": len(os.sched_getaffinity(0)),
            "kind": "port",
            "sample": (f"oracle precompute on a {nf_s}x{nt_s} (freq x time) slice = {rows} of "
                       f"{full_rows} rows ({t_pre:.2f}s, scaled linearly in rows) + {iters} CG "
                       f"iterations x one full-size operator application ({t_mv * 1e3:.1f} ms); "
                       f"response build {t_resp:.2f}s not counted")}


def main_beam(args, comm, rank, world, dev):
    """Operator builds/s of the beam head's simulator consumers (per_antenna_beam_example.py:
    118-260): one step = the sky contraction with the outer product of the Fourier-Bessel design
    matrix (get_bess_sky_contraction) + the coefficient-to-visibility operator of every antenna
    (get_bess_to_vis_from_contraction).  Frequency channels are independent and sharded over
    ranks with no collective (each rank's channels are its own output)."""
    import torch
    import hydra_amd as hy
    from hydra_amd import per_ant_beam_sampler as bs
    hex_spec, ns, nf, nt, nmax, mmax = BEAM_CONFIGS[args.config]
    inp = synth_inputs(hex_spec, ns, nf, nt)
    ant_pos = inp["ant_pos"]
    nant = len(ant_pos)
    fidx = np.array_split(np.arange(nf), world)[rank]
    nf_loc = fidx.size
    freqs_hz = inp["freqs"][fidx] * 1e6
    fluxes = inp["fluxes"][:, fidx]
    lsts = inp["times"]
    nm = np.repeat(np.arange(1, nmax + 1), 2 * mmax + 1)
    mm = np.tile(np.arange(-mmax, mmax + 1), nmax)
    nc = nm.size
    rng = np.random.RandomState(3)
    bess = bs.get_bess_matr(nm, mm, rng.uniform(0.0, 1.0, (nt, ns)), rng.uniform(0.0, 2 * np.pi, (nt, ns)))
    outer = torch.as_tensor(bs.get_bess_outer(bess), device=dev).contiguous()
    coeffs = torch.as_tensor(rng.randn(nc, nf_loc, nant, 1, 1) + 1j * rng.randn(nc, nf_loc, nant, 1, 1),
                             device=dev)
    lat = -30.7215 * np.pi / 180.0

    def one_step():
        bsc = bs.get_bess_sky_contraction(outer, ant_pos, fluxes, inp["ra"], inp["dec"], freqs_hz,
                                          lsts, latitude=lat)
        ops = [bs.get_bess_to_vis_from_contraction(bsc, coeffs, nant, a) for a in range(nant)]
        return bsc, ops

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bsc, ops = one_step()
    sync()
    elapsed = time.perf_counter() - t0
    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if comm is not None:
        comm.allreduce_(t_all, op="max")
    elapsed = float(t_all.item())
    # the contraction kernel alone, HIP events on the launch stream
    from hydra_amd import _dev as dv
    V = bs._uniform_voltages(ant_pos, fluxes, inp["ra"], inp["dec"], freqs_hz, lsts, lat)
    out = dv.empty((nf_loc, nt, nant, nant, nc * nc), torch.complex128)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        dv.call("hydra_bess_sky_contraction", dv.ptr(V), nf_loc, nt, nant, ns, dv.ptr(outer), nc * nc,
                dv.ptr(None), -1, dv.ptr(out), dv.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    k_ms = e0.elapsed_time(e1) / reps
    if rank != 0:
        return
    # work of the formulation that runs: three real products per complex one (6 flops per complex
    # multiply-add; the textbook count is 8)
    flops = 6.0 * nf_loc * nt * nant * nant * ns * nc * nc
    achieved = flops / (k_ms * 1e-3) / 1e12
    res = {
        "metric": "beam-head operator builds/sec (sky contraction + per-antenna operators, whole job)",
        "value": args.steps / elapsed, "unit": "builds/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"beam-head consumers ({args.config}): hex{hex_spec} = {nant} antennas, {ns} sources, "
                         f"{nf} freqs x {nt} times, {nc} Fourier-Bessel modes ({nc * nc} basis pairs), "
                         f"uniform-beam generator fused into the contraction, complex128"),
            "parallelism": f"freq-sharded x{world}, no collective" if world > 1 else "single GPU",
        },
        "roofline": {
            "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic(args.config, "bess_contract_kernel") if world == 1 else None,
            "kernel": "bess_contract_kernel (FP64 matrix cores, 3-product complex GEMM: 6 Nant^2 Ns Q flops per (f,t))",
            "ms_per_launch": k_ms, "flops_per_launch": flops,
        },
        "cpu_baseline": None,
    }
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        from oracle import gcr_beam as ob
        nf_s, nt_s = 1, min(nt, 2)
        t0 = time.perf_counter()
        ref = ob.get_bess_sky_contraction(bs.get_bess_outer(bess[:nt_s]), ant_pos, inp["fluxes"][:, :nf_s],
                                          inp["ra"], inp["dec"], inp["freqs"][:nf_s] * 1e6, lsts[:nt_s],
                                          latitude=lat)
        cf = coeffs[:, :nf_s].cpu().numpy()
        for a in range(nant):
            ob.get_bess_to_vis_from_contraction(ref, cf, nant, a)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = {
            "value": 1.0 / (dt * nf * nt / (nf_s * nt_s)), "unit": "builds/s",
            "cores": len(os.sched_getaffinity(0)), "kind": "port",
            "sample": (f"oracle (numpy restatement of get_bess_sky_contraction + "
                       f"get_bess_to_vis_from_contraction for all antennas) on {nf_s} x {nt_s} of "
                       f"{nf} x {nt} (freq x time) samples, {dt:.2f}s, scaled linearly"),
        }
        got = bsc[0, 0, 0, :nt_s].cpu().numpy()
        res["parity_vs_oracle_slice"] = float(np.abs(got - ref[0, 0, 0]).max() / np.abs(ref).max())
    print(json.dumps(res))


def self_launch(ngpus):
    """``python bench.py --gpus N`` without torchrun's environment: run the same command line
    under ``python -m torch.distributed.run`` (one rank per GPU, rendezvous on 127.0.0.1 at a free
    port) as a CHILD process -- this process has made no GPU call and makes none -- and exit with
    the child's status.  Rank 0's JSON line is passed on alone."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // ngpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={ngpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    # stdout of this command is ONE JSON line: pass rank 0's line on and send whatever else the
    # ranks or their libraries write there (gloo reports its connections on stdout) to stderr
    done = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in done.stdout.splitlines() if ln.strip()]
    result = [ln for ln in lines if ln.lstrip().startswith('{"metric"')]
    for ln in lines:
        if ln not in result:
            print(ln, file=sys.stderr)
    if result:
        print(result[-1])
    sys.stdout.flush()
    if done.returncode != 0:
        sys.exit(done.returncode)
    if not result:
        sys.exit("bench.py: the ranks printed no result line")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS) + sorted(SH_CONFIGS) + sorted(GAIN_CONFIGS)
                    + sorted(COSMO_CONFIGS) + sorted(BEAM_CONFIGS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--cg-tol", type=float, default=1e-8)
    ap.add_argument("--gram", default="auto", choices=["auto", "general"],
                    help="auto = closed-form antenna-space Gram when the array is complete and the "
                         "noise weights do not depend on the baseline (true for this workload), "
                         "general = always the per-baseline-weight kernel")
    ap.add_argument("--flags", type=float, default=0.0,
                    help="fraction of (baseline, freq, time) samples flagged (inverse noise variance "
                         "0) at random, plus one dead baseline: the closed-form Gram then takes its "
                         "flagged-row correction (ptsrc configs)")
    ap.add_argument("--gain-unfused", action="store_true",
                    help="gain head: run apply_gains / residual / construct_rhs_mpi / GainOperator "
                         "as separate passes instead of gain_sampler.prepare_sample")
    ap.add_argument("--no-general", action="store_true",
                    help="skip the extra launches of the general Gram kernel after the timed region")
    ap.add_argument("--no-fixed", action="store_true",
                    help="ptsrc configs: skip the gains-fixed leg (operator and factor reused)")
    ap.add_argument("--shard", default=None, choices=["freq", "baseline"],
                    help="how the visibility rows are split over GPUs (default: freq; c5s: baseline)")
    ap.add_argument("--solver", default=None, choices=["dense", "cg", "cg-rows", "dense-rows"],
                    help="dense = blocked Cholesky (the reference's default for this head, "
                         "example.py:944); cg = linear_solver.cg on the root; cg-rows = the operator "
                         "stays sharded by rows over the GPUs and every rank runs the CG with a "
                         "row-block product + all-gather per iteration; dense-rows = the same "
                         "layout factored in place by the row-sharded Cholesky "
                         "(linear_solver.solve_dense_rows; config 5 on > 1 GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started bare (`python bench.py --gpus N`): start the N ranks as a child torchrun, one
        # process per GPU as in the reference's `mpirun -n N python example.py` (example.py:161-174),
        # BEFORE anything here touches the GPU, and pass on its output and exit code
        return self_launch(args.gpus)

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
    if args.config in SH_STREAMED:
        return main_sh_streamed(args, comm, rank, world, dev)
    if args.config in SH_CONFIGS:
        return main_sh(args, comm, rank, world, dev)
    if args.config in GAIN_CONFIGS:
        return main_gain(args, comm, rank, world, dev)
    if args.config in COSMO_CONFIGS:
        return main_cosmo(args, comm, rank, world, dev)
    if args.config in BEAM_CONFIGS:
        return main_beam(args, comm, rank, world, dev)

    hex_spec, ns, nf, nt = CONFIGS[args.config]
    if args.config in PER_GPU_CHANNELS:
        nf *= world
    big = args.config in ("c5s", "c5", "c5q")
    huge = args.config in ("c5", "c5q")
    shard = args.shard or ("baseline" if args.config == "c5s" else "freq")
    if args.solver is None:
        # one GPU factors the operator (config 5: 80 GB, 5.7 s); with more GPUs config 5 keeps the
        # summed Gram sharded by rows and the ranks factor it together
        args.solver = "dense-rows" if (huge and world > 1) else "dense"
    if huge:
        args.no_cpu = True     # the CPU path cannot form or solve a 100 000^2 system in minutes
    if args.solver == "cg-rows" and world == 1:
        args.solver = "cg"
    if args.solver == "dense-rows" and world == 1:
        args.solver = "dense"
    inp = synth_inputs(hex_spec, ns, nf, nt, outriggers=19 if big else 0)
    ant_pos, antpairs_all = inp["ant_pos"], inp["antpairs"]
    nant = len(ant_pos)
    ants = np.arange(nant)
    if shard == "freq":
        fidx = np.array_split(np.arange(nf), world)[rank]      # frequency sharding
        antpairs = antpairs_all
    else:
        fidx = np.arange(nf)                                    # baseline sharding (round-robin)
        antpairs = antpairs_all[rank::world]
    nbl, nbl_all = len(antpairs), len(antpairs_all)
    nf_loc = fidx.size
    freqs_loc = inp["freqs"][fidx]
    times = inp["times"]
    if big:
        # gridded beam table (181 x 360 at 1 degree, 3 channels) with the Gaussian's shape
        za_g = np.linspace(0.0, np.pi, 181)
        az_g = np.arange(360) * np.pi / 180.0
        f_g = np.array([90e6, 105e6, 120e6, 135e6])
        sig = hy.beams.GaussianBeam(diameter=14.0).get_sigmas(f_g)
        tab = np.exp(-(za_g[None, :, None] / sig[:, None, None]) ** 2) * np.ones((1, 1, az_g.size)) \
            + 1e-3
        gb = hy.beams.GriddedBeam(f_g, za_g, az_g, tab)
        beams = [gb for _ in ant_pos]
    else:
        beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]

    # ---- set-up (untimed): factored operator, gains, data, noise weights -- all in HBM
    fac = ptsrc_sampler.calc_proj_operator(inp["ra"], inp["dec"], inp["fluxes"][:, fidx], ant_pos,
                                           antpairs, freqs_loc, times, beams, materialize=False)
    fac.allow_uniform = args.gram == "auto" and shard == "freq"
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
    if args.flags > 0.0:
        inv_noise[torch.rand(inv_noise.shape, device=dev, generator=gen) < args.flags] = 0.0
        inv_noise[nbl // 3] = 0.0
    prior = torch.as_tensor(inp["prior"], device=dev)
    del gg
    m_loc = nbl * nf_loc * nt
    m_all = nbl_all * nf * nt

    def one_sample():
        om_n = torch.randn((nbl, nf_loc, nt), dtype=torch.complex128, device=dev, generator=gen)
        om_a = torch.randn(ns, dtype=torch.float64, device=dev, generator=gen)
        if args.solver == "cg-rows":
            A, b = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                                inv_noise, gains, prior, omega_n=om_n, omega_a=om_a,
                                                shard_rows=True)
            op = hy.linear_solver.RowShardedOperator(comm, A, ns)
            x, iters = hy.linear_solver.cg(None, b, maxiters=1000, abs_tol=args.cg_tol,
                                           linear_op=op, return_iters=True)
            return op, b, x, iters
        if args.solver == "dense-rows":
            A, b = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                                inv_noise, gains, prior, omega_n=om_n, omega_a=om_a,
                                                shard_rows=True)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            keep = A.clone() if ns <= 20000 else None
            ev[0].record()
            x = hy.linear_solver.solve_dense_rows(comm, A, b, overwrite_a=True)
            ev[1].record()
            solve_events.append(ev)
            return keep, b, x, 0
        # small systems over several GPUs: the sums are all-reduces, every rank holds
        # the whole system and solves it for itself -- no broadcast of x, no rank waits for the root
        replicated = comm is not None and ns <= REPLICATE_MAX_N
        A, b = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                            inv_noise, gains, prior, omega_n=om_n, omega_a=om_a,
                                            replicate=replicated)
        solver_here = rank == 0 or replicated
        if solver_here and args.solver == "cg":
            x, iters = hy.linear_solver.cg(A, b, maxiters=1000, abs_tol=args.cg_tol,
                                           return_iters=True)
        elif solver_here:
            # A is rebuilt every sample, so the factorisation may overwrite it (no 80 GB copy);
            # keep what the residual check below needs
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            keep_A = A if ns <= 20000 else None
            ev[0].record()
            x = hy.linear_solver.solve_dense(A.clone() if keep_A is not None else A, b,
                                             overwrite_a=True)
            ev[1].record()
            solve_events.append(ev)
            iters = 0
            if keep_A is None:
                A = None
        else:
            x, iters = torch.zeros(ns, dtype=torch.float64, device=dev), 0
        if comm is not None and not replicated:
            comm.bcast_(x, root=0)
        return A, b, x, iters

    def sync():
        if comm is not None:
            comm.barrier()
        torch.cuda.synchronize()

    solve_events = []
    for _ in range(args.warmup):
        one_sample()
    ptsrc_sampler.GRAM_EVENTS = []
    solve_events.clear()
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
    solve_ms = float(np.mean([s.elapsed_time(e) for s, e in solve_events])) if solve_events \
        else float("nan")

    # ---- the same sample while the gains are NOT being sampled (ptsrc_sampler.SystemCache): the
    # operator and its Cholesky factor depend on gains, noise weights and prior only, so a run
    # that keeps the gains fixed (example.py without --gains) pays for them once and a sample is
    # the right-hand side + two triangular sweeps.  Reported beside `value`, never as `value`:
    # the reference rebuilds the operator in every iteration (example.py:878).
    fixed = None
    if args.solver in ("dense", "dense-rows") and not args.no_fixed:
        if huge:
            A = None                      # (consumed by the in-place factorisation anyway)
        rows_mode = args.solver == "dense-rows"
        cache = ptsrc_sampler.SystemCache()

        def fixed_sample():
            om_n = torch.randn((nbl, nf_loc, nt), dtype=torch.complex128, device=dev, generator=gen)
            om_a = torch.randn(ns, dtype=torch.float64, device=dev, generator=gen)
            repl = comm is not None and not rows_mode and ns <= REPLICATE_MAX_N
            _, b_c = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                                  inv_noise, gains, prior, omega_n=om_n,
                                                  omega_a=om_a, shard_rows=rows_mode, cache=cache,
                                                  replicate=repl)
            if rows_mode or repl or rank == 0:
                x_c = cache.solve(b_c)
            else:
                x_c = torch.zeros(ns, dtype=torch.float64, device=dev)
            if comm is not None and not rows_mode and not repl:
                comm.bcast_(x_c, root=0)
            return b_c, x_c

        sync()
        t0 = time.perf_counter()
        fixed_sample()                    # builds the operator and factors it
        sync()
        t_first = time.perf_counter() - t0
        fixed_sample()
        nfix = max(args.steps, 5)
        sync()
        t0 = time.perf_counter()
        for _ in range(nfix):
            b_f, x_f = fixed_sample()
        sync()
        t_fix = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if comm is not None:
            comm.allreduce_(t_fix, op="max")
        t_fix = float(t_fix.item())
        assert cache.misses == 1 and cache.hits == nfix + 1, (cache.misses, cache.hits)
        r_f = None
        if rows_mode or rank == 0:
            r_f = float((torch.linalg.norm(cache.matvec(x_f) - b_f) / torch.linalg.norm(b_f)).item())
        fixed = {"value": nfix / t_fix, "unit": "samples/s", "ms_per_step": 1e3 * t_fix / nfix,
                 "steps": nfix, "first_sample_s": t_first, "sample_rel_residual": r_f,
                 "what": ("gains, noise weights and prior unchanged between samples: operator and "
                          "Cholesky factor kept (ptsrc_sampler.SystemCache), per sample the "
                          "right-hand side (R^H z) + two triangular sweeps; x is bit-identical to "
                          "the rebuilding path (tests/test_gpu_parity.py)")}

    # sanity of the last sample (outside the timed region; the row-sharded product is collective).
    # The in-place factorisation of the 80 GB operator consumed it: the cache of the gains-fixed leg
    # holds the same operator (same inputs; b and x belong to the last sample's fluctuation
    # draws, which only enter b), else rebuild it
    if huge and fixed is not None:
        Ax = cache.matvec(x) if (args.solver == "dense-rows" or rank == 0) else None
    else:
        if args.solver == "dense-rows":
            if A is None:
                A, _ = ptsrc_sampler.precompute_mpi(comm, ants, antpairs, freqs_loc, times, fac, data,
                                                    inv_noise, gains, prior, realisation=False,
                                                    shard_rows=True)
            A = hy.linear_solver.RowShardedOperator(comm, A, ns)
        elif A is None and rank == 0:
            A, _ = ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs_loc, times, fac, data,
                                                inv_noise, gains, prior, realisation=False)
        Ax = A(x) if args.solver in ("cg-rows", "dense-rows") else (A @ x if rank == 0 else None)
    if rank != 0:
        return
    resid = float((torch.linalg.norm(Ax - b) / torch.linalg.norm(b)).item())
    del Ax
    # Algorithmic flops per launch of the kernel that ran (this rank's rows):
    #   gram_fused_kernel   (baseline form):  2 M Ns (Ns+1)                  [SURVEY 8(d)]
    #   gram_uniform_kernel (antenna form, 3-product complex multiply):
    #       Nft Ns(Ns+1)/2 (8 Nant + 8)   [3 real FMAs for T and 1 for the q term per antenna
    #       and output element, + 8 flops of square-accumulate]; with one beam model for the
    #       whole array the q term is rank one per (f,t): Nft Ns(Ns+1)/2 (6 Nant + 10)
    kernel = getattr(fac, "last_gram_kernel", "gram_fused_kernel")
    baseline_form_flops = 2.0 * m_loc * ns * (ns + 1)
    if kernel == "gram_uniform_kernel":
        per_elem = (6.0 * nant + 10.0) if fac.same_beam else (8.0 * nant + 8.0)
        gram_flops = nf_loc * nt * 0.5 * ns * (ns + 1) * per_elem
    else:
        gram_flops = baseline_form_flops
    achieved = gram_flops / (gram_ms * 1e-3) / 1e12
    effective = baseline_form_flops / (gram_ms * 1e-3) / 1e12
    roof_ran = {
        "bound": "mfma", "achieved": achieved, "peak": FP64_MATRIX_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": achieved / FP64_MATRIX_PEAK_TFLOPS,
        "traffic": pmc_traffic(args.config, kernel) if world == 1 else None,
        "kernel": kernel, "ms_per_launch": gram_ms, "flops_per_launch": gram_flops,
    }
    # SURVEY 8(d) defines the Gram rate on the baseline-form count 2 M Ns (Ns+1), which is what
    # gram_fused_kernel executes.  When the sample above took the closed form, time the general
    # kernel on the same inputs in this same run (a few launches, outside `value`) and make THAT
    # the roofline entry; the closed-form figures move to roofline_closed_form.
    roof_general, agree = None, None
    if kernel == "gram_uniform_kernel" and world == 1 and not big and not args.no_general:
        A_auto = fac.gram(gains, inv_noise, prior_std=prior)
        fac.allow_uniform = False
        ptsrc_sampler.GRAM_EVENTS = []
        for _ in range(2):
            A_gen = ptsrc_sampler._timed_gram(lambda: fac.gram(gains, inv_noise, prior_std=prior))
        torch.cuda.synchronize()
        gen_ms = float(np.mean([s_.elapsed_time(e_) for s_, e_ in ptsrc_sampler.GRAM_EVENTS]))
        ptsrc_sampler.GRAM_EVENTS = None
        fac.allow_uniform = True
        agree = float(((A_auto - A_gen).abs().max() / A_gen.abs().max()).item())
        del A_auto, A_gen
        g_ach = baseline_form_flops / (gen_ms * 1e-3) / 1e12
        roof_general = {
            "bound": "mfma", "achieved": g_ach, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": g_ach / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic(args.config, "gram_fused_kernel"),
            "kernel": "gram_fused_kernel", "ms_per_launch": gen_ms,
            "flops_per_launch": baseline_form_flops,
            "note": ("SURVEY 8(d) Gram: per-baseline-weight kernel, 2 M Ns (Ns+1) flops, timed on "
                     "this run's inputs outside `value`; `value` uses the closed-form kernel "
                     "(roofline_closed_form), which forms the same matrix with fewer operations"),
        }
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
                         f"/ {nbl_all} baselines, {ns} sources, {nf} freqs x {nt} times, complex128, "
                         f"factored operator (R never materialised)"
                         + (", gridded beam (bicubic spline)" if big else "")),
            "rows_M": m_all, "solver": args.solver, "cg_abs_tol": args.cg_tol,
            "cg_iters_last": int(iters), "gram": args.gram,
            "flagged_fraction": args.flags,
            "flagged_rows_corrected": int(getattr(fac, "last_gram_flagged_rows", 0)),
            "parallelism": f"{shard}-sharded x{world}" if world > 1 else "single GPU",
            "solve_placement": ("one GPU" if world == 1 else
                                "operator sharded by rows over the GPUs" if args.solver in ("cg-rows", "dense-rows") else
                                "system all-reduced, solved on every GPU (no broadcast)" if ns <= REPLICATE_MAX_N else
                                "root solves, solution broadcast"),
        },
        # the metric's "Gram FP64 TFLOP/s" as an EFFECTIVE rate: baseline-form flops 2 M Ns (Ns+1)
        # over the Gram time, whole job.  Not a roofline figure: it exceeds the matrix peak when
        # the antenna-form kernel forms the same Gram with fewer operations (roofline* hold the
        # executed-work rates)
        "gram_tflops_effective_baseline_form": world * effective,
        "sample_rel_residual": resid,
        "roofline": roof_general if roof_general is not None else roof_ran,
    }
    if fixed is not None:
        out["gains_fixed"] = fixed
    if args.config in PER_GPU_CHANNELS:
        out["scaling"] = "weak"
        out["config"]["channels_per_gpu"] = nf // world
    if roof_general is not None:
        out["roofline_closed_form"] = roof_ran
        out["general_vs_closed_form_relerr"] = agree
    if solve_events:
        fl = ns ** 3 / 3.0
        s_ach = fl / (solve_ms * 1e-3) / 1e12
        solve_gpus = world if args.solver == "dense-rows" else 1
        out["dense_solve"] = {
            "kernel": ("row-sharded 256-wide-panel Cholesky over the ranks (solve_dense_rows: panel "
                       "all-gather + chol_syrk_mfma_kernel on each rank's rows)"
                       if args.solver == "dense-rows" else
                       "chol_syrk_mfma_kernel (256-wide-panel Cholesky, trailing update on the FP64 "
                       "matrix cores) + panel steps + triangular solves" if ns >= 6144 else
                       "32-wide blocked Cholesky (launch-chain bound at this size)"),
            "n": ns, "ms": solve_ms, "flops": fl, "bound": "mfma", "achieved": s_ach,
            "peak": FP64_MATRIX_PEAK_TFLOPS * solve_gpus, "unit": "TFLOP/s",
            "frac": s_ach / (FP64_MATRIX_PEAK_TFLOPS * solve_gpus), "gpus": solve_gpus,
        }
    if not args.no_cpu and world == 1:     # CPU leg: single-GPU runs only
        nf_s, nt_s = (2, 10) if args.config == "c2" else (min(nf, 2), min(nt, 10))
        if big:      # a 2000-baseline, one-channel slice (the full slice would need 16 GB)
            inp = dict(inp, antpairs=antpairs_all[:2000])
            nf_s, nt_s = 1, 1
        out["cpu_baseline"] = cpu_baseline(inp, nf_s, nt_s, m_all)
        out["cpu_baseline"]["blas_threads"] = out["cpu_baseline"]["cores"]
        out["cpu_baseline"]["host_cpus"] = len(os.sched_getaffinity(0))
        # the same path with one BLAS thread, on a smaller slice (bounded CPU time)
        out["cpu_baseline_single_thread"] = cpu_baseline(inp, 1, min(nt_s, 5), m_all,
                                                         single_thread=True)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
