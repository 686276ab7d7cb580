#!/usr/bin/env python
"""
Gibbs driver on MI355X: the counterpart of the reference's ``example.py`` for the heads that sit
on the GCR hot path -- point-source amplitudes (--ptsrc), linearised gains (--gains),
spherical-harmonic modes (--sh) and the cosmological field (--cosmo) -- with the reference's
full flag set (hydra_amd.config = hydra/config.py) and output files (ptsrc_amps0.npy,
sim_*_w%04d.npy, src_amp_%05d.npy, delta_g_amps_%05d.npy, sh_amp_%05d.npy; example.py:191-192,
405-408,740,953,1061).  The reference's cosmo step plots its first solution and exits
(example.py:829-841); here every sample is saved as cosmo_field_%05d.npy and, as there
(:850-851), does not feed back into the data model.

    python example.py --ptsrc --gains --Niters 10
    python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 example.py --ptsrc

One process per GPU, ``hydra_amd._dist.Comm`` where ``MPI.COMM_WORLD`` stood.  A square number of
workers is laid out on the reference's (frequency x time) grid of chunks (example.py:161-174,
utils.freqs_times_for_worker); any other count splits the frequency channels only.  All arrays
of the loop stay in HBM; the operator of the source head is kept in factored form.
Random numbers are drawn from numpy's global generator in the reference's order (catalogue,
gain levels, ``seed(1)`` + gain modes, noise; per head: the draws of the reference's functions),
so a seeded run can be replayed by the CPU oracle (tests/test_gpu_driver.py).

--solver cg | gmres | bicgstab solve the source head densely on the root, as the reference does
whatever the name (example.py:940-947); --solver mpicg distributes the operator in
sqrt(N) x sqrt(N) blocks and runs ``linear_solver.cg_mpi`` (:899-938).
--regions / --sim-diffuse-sky-model need a pygdsm sky model (not shipped; the region operator
itself is hydra_amd.region_sampler.calc_proj_operator); --beam, --cl and --pspec are heads
outside the GCR hot path and raise NotImplementedError, --stats raises it
as in the reference (example.py:1254-1255).
"""
import os
import time

import numpy as np
import torch

import hydra_amd as hydra
from hydra_amd import _dev, _dist
from hydra_amd.config import get_config  # noqa: F401  (re-exported: tests and callers use it)

SOLVERS = ("cg", "gmres", "bicgstab", "mpicg")


def worker_chunks(comm, freqs, times):
    """(freq idxs, time idxs, ngrid) of this rank: the reference's square grid of chunks
    (example.py:161-174) when the worker count is a square, else frequency chunks only."""
    nworkers = comm.Get_size() if comm is not None else 1
    ngrid = int(round(np.sqrt(nworkers)))
    if comm is None:
        return np.arange(freqs.size), np.arange(times.size), 1
    if ngrid * ngrid == nworkers:
        fi, ti, _ = hydra.freqs_times_for_worker(comm=comm, freqs=freqs, times=times,
                                                 fchunks=ngrid, tchunks=ngrid)
        return fi, ti, ngrid
    fi, ti, _ = hydra.freqs_times_for_worker(comm=comm, freqs=freqs, times=times,
                                             fchunks=nworkers, tchunks=1)
    return fi, ti, 0


def main(argv=None, on_head=None):
    """Run the sampler.  ``on_head(name, n, state)`` (optional) is called right before each head
    of iteration ``n`` draws its random numbers, with a dict of the arrays that head is about to
    use -- the tests replay the head on the CPU oracle from there."""
    args = get_config(argv)
    if args.solver_name not in SOLVERS:
        raise ValueError("Solver '%s' not recognised." % args.solver_name)
    for flag, name in ((args.sample_regions, "--regions"), (args.sample_beam, "--beam"),
                       (args.sample_sh_pspec, "--cl"), (args.sample_pspec, "--pspec")):
        if flag:
            raise NotImplementedError(
                "%s: this head is outside the GCR hot path of this package (SURVEY section 8); "
                "--regions needs a pygdsm sky model" % name)
    if args.sim_diffuse_sky_model != "none":
        raise NotImplementedError("--sim-diffuse-sky-model needs pygdsm sky-model data")
    if args.calculate_stats:
        raise NotImplementedError()                       # example.py:1254-1255
    comm = _dist.init_from_env()
    myid = comm.Get_rank() if comm is not None else 0
    nworkers = comm.Get_size() if comm is not None else 1
    if comm is None:
        torch.cuda.set_device(0)
    dev = _dev.device()
    np.random.seed(args.seed + myid)                      # example.py:139
    out = args.output_dir
    if myid == 0:
        os.makedirs(out, exist_ok=True)
    if comm is not None:
        comm.barrier()
    ftime = os.path.join(out, "timing.dat")

    def log(msg, colour=None):
        if myid == 0:
            hydra.status(None, msg, colour)

    def timing(n, task, dt):
        if args.save_timing_info and myid == 0:
            hydra.timing_info(ftime, n, task, dt, verbose=False)

    def tdev(x, dtype):
        return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype, device=dev)

    def notify(name, n, **state):
        if on_head is not None:
            on_head(name, n, state)

    # ---- array, this rank's (freq, time) chunk, catalogue (example.py:149-201)
    lat = np.deg2rad(args.latitude)
    freqs = np.linspace(min(args.freq_bounds), max(args.freq_bounds), args.Nfreqs)
    times = np.linspace(min(args.lst_bounds), max(args.lst_bounds), args.Ntimes)
    fidx, tidx, ngrid = worker_chunks(comm, freqs, times)
    freq_chunk, time_chunk = freqs[fidx], times[tidx]
    log("    Solver:  %s\n    Seed:    %d\n    Parallelisation: %03d workers, chunks %s"
        % (args.solver_name, args.seed, nworkers,
           "%d x %d (freq x time)" % (ngrid, ngrid) if ngrid else "%d x 1" % nworkers))
    ant_pos = hydra.build_hex_array(hex_spec=tuple(args.hex_array), d=14.6)
    ants = np.array(list(ant_pos.keys()))
    nants = len(ants)
    antpairs = [(i, j) for i in range(nants) for j in range(i, nants) if i != j]
    nbl, nf, nt = len(antpairs), freq_chunk.size, time_chunk.size
    Nptsrc = args.Nptsrc
    cat = np.zeros((3, Nptsrc))
    if myid == 0:
        cat[:] = hydra.example.generate_random_ptsrc_catalogue(
            Nptsrc, ra_bounds=args.ra_bounds, dec_bounds=args.dec_bounds,
            logflux_bounds=(-1.0, 2.0))
        np.save(os.path.join(out, "ptsrc_amps0"), cat[2])
        np.save(os.path.join(out, "ptsrc_coords0"), cat[:2])
    if comm is not None:
        comm.Bcast(cat, root=0)
    ra, dec, ptsrc_amps = cat
    fluxes_chunk = hydra.get_flux_from_ptsrc_amp(ptsrc_amps, freq_chunk, -2.7)
    beams = hydra.example.example_beams(args.beam_sim_type, nants)
    log("Simulation beam type: %s" % args.beam_sim_type, "y")

    # ---- operators (example.py:507-523, 597-624): factored source operator, SH response
    t0 = time.time()
    source_proj = hydra.ptsrc_sampler.calc_proj_operator(
        ra=ra, dec=dec, fluxes=fluxes_chunk, ant_pos=ant_pos, antpairs=antpairs,
        freqs=freq_chunk, times=time_chunk, beams=beams, latitude=lat, materialize=False)
    log("Source projection operator (factored) took %6.3f sec" % (time.time() - t0), "c")
    ones = torch.ones(Nptsrc, dtype=torch.float64, device=dev)
    model0_chunk = source_proj.apply(ones)                # the simulated sky = the catalogue
    sh_response = None
    if args.sample_sh:
        t0 = time.time()
        sh_response, _ell, _m = hydra.sh_sampler.vis_proj_operator_no_rot(
            freqs=freq_chunk, lsts=time_chunk, beams=beams, ant_pos=ant_pos, lmax=args.sh_lmax,
            nside=args.sh_nside, latitude=lat, ref_freq=args.sh_ref_freq,
            spectral_idx=args.sh_spectral_idx, as_torch=True, cross_only=True)
        nmodes_sh = sh_response.shape[1]
        sh_true = np.zeros(nmodes_sh)
        if myid == 0:
            sh_true[:] = args.sh_prior_std * np.random.randn(nmodes_sh)
            np.save(os.path.join(out, "sim_sh_amps0"), sh_true)
        if comm is not None:
            comm.Bcast(sh_true, root=0)
        sh_model0 = torch.empty((nbl, nf, nt), dtype=torch.complex128, device=dev)
        _dev.call("hydra_apply_R", _dev.ptr(sh_response), nbl * nf * nt, nmodes_sh,
                  _dev.ptr(tdev(sh_true, torch.float64)), _dev.ptr(None), _dev.ptr(sh_model0),
                  _dev.stream_ptr())
        model0_chunk = model0_chunk + sh_model0
        log("SH response operator took %6.3f sec" % (time.time() - t0), "c")

    cosmo_proj = None
    if args.sample_cosmo_field:                           # example.py:539-594
        t0 = time.time()
        cosmo_grid_ra, cosmo_grid_dec = hydra.cosmo_sampler.make_cosmo_field_grid(args)
        cosmo_proj = hydra.ptsrc_sampler.calc_proj_operator(
            ra=cosmo_grid_ra, dec=cosmo_grid_dec,
            fluxes=np.ones((cosmo_grid_ra.size, freq_chunk.size)), ant_pos=ant_pos,
            antpairs=antpairs, freqs=freq_chunk, times=time_chunk, beams=beams, latitude=lat,
            materialize=False)
        cosmo_x, cosmo_y = np.unique(cosmo_grid_ra), np.unique(cosmo_grid_dec)
        cosmo_pspec_kbins = np.linspace(0.0, 10, 25)
        cosmo_pspec_current = 0.1 * cosmo_pspec_kbins + np.ones(cosmo_pspec_kbins.size)
        cosmo_background_params = {"h": 0.69, "omega_m": 0.31}
        cosmo_soln_shape = (freqs.size, cosmo_x.size, cosmo_y.size)
        log("Cosmo field projection operator (factored) took %6.3f sec" % (time.time() - t0), "c")

    # ---- calibration source: the brightest near the zenith (example.py:286-312)
    calsrc = args.calsrc_std >= 0.0
    near = np.where(np.abs(dec - lat) * 180.0 / np.pi < args.calsrc_radius)[0]
    assert len(near) > 0, "No sources found within %d deg of the zenith" % args.calsrc_radius
    calsrc_idx = int(near[np.argmax(ptsrc_amps[near])])
    log("Calibration source: enabled %s, index %d, amplitude %6.3e" % (calsrc, calsrc_idx,
                                                                       ptsrc_amps[calsrc_idx]))

    # ---- gains and simulated data (example.py:322-408)
    Lfreq = (freqs[1] - freqs[0]) * freqs.size if freqs.size > 1 else 1.0
    Ltime = (times[1] - times[0]) * times.size if times.size > 1 else 1.0
    Fbasis_h, k_freq, k_time = hydra.partial_fourier_basis_2d_from_nmax(
        freqs=freq_chunk, times=time_chunk, nmaxfreq=args.gain_nmaxfreq,
        nmaxtime=args.gain_nmaxtime, Lfreq=Lfreq, Ltime=Ltime, freq0=freqs[0], time0=times[0],
        shape0=(freqs.size, times.size), positive_only=args.gain_only_positive_modes)
    ngm = k_freq.size
    np.save(os.path.join(out, "Fbasis_w%04d" % myid), Fbasis_h)
    gain0_level = np.zeros(nants, dtype=np.complex128)
    delta_g_amps0 = np.zeros((nants, ngm), dtype=np.complex128)
    prior_std_delta_g = args.sim_gain_amp_std * np.ones(ngm)
    prior_std_delta_g[0] *= 0.0                            # example.py:362
    if myid == 0:
        gain0_level[:] = 0.05 * (np.random.randn(nants) + 1.0j * np.random.randn(nants))
        np.random.seed(1)                                  # example.py:368
        delta_g_amps0[:] = prior_std_delta_g * (1.0 * np.random.randn(nants, ngm)
                                                + 1.0j * np.random.randn(nants, ngm))
        np.save(os.path.join(out, "sim_gain0_level"), gain0_level)
        np.save(os.path.join(out, "k_freq"), k_freq)
        np.save(os.path.join(out, "k_time"), k_time)
    if comm is not None:
        comm.Bcast(gain0_level, root=0)
        comm.Bcast(delta_g_amps0, root=0)
    Fbasis = tdev(Fbasis_h, torch.complex128)
    gains_chunk = tdev(((1.0 + 1.0j) + gain0_level)[:, None, None] * np.ones((nants, nf, nt)),
                       torch.complex128)
    delta_g_chunk0 = torch.tensordot(tdev(delta_g_amps0, torch.complex128), Fbasis, dims=([1], [0]))
    data_chunk = hydra.apply_gains(model0_chunk, gains_chunk * (1.0 + delta_g_chunk0), ants,
                                   antpairs, inline=False)
    noise_chunk = args.sigma_noise * np.sqrt(0.5) * (1.0 * np.random.randn(nbl, nf, nt)
                                                     + 1.0j * np.random.randn(nbl, nf, nt))
    data_chunk = data_chunk + tdev(noise_chunk, torch.complex128)
    del noise_chunk
    inv_noise_var_chunk = torch.full((nbl, nf, nt), 1.0 / args.sigma_noise ** 2,
                                     dtype=torch.float64, device=dev)
    np.save(os.path.join(out, "sim_model0_chunk_w%04d" % myid), model0_chunk.cpu().numpy())
    np.save(os.path.join(out, "sim_data_chunk_w%04d" % myid), data_chunk.cpu().numpy())
    np.save(os.path.join(out, "sim_delta_g0_chunk_w%04d" % myid), delta_g_chunk0.cpu().numpy())
    np.save(os.path.join(out, "sim_delta_g_amps0_w%04d" % myid), delta_g_amps0)
    log("Simulation step finished", "b")

    # ---- priors and starting state (example.py:410-460)
    amp_prior_h = args.ptsrc_amp_prior_level * np.ones(Nptsrc)
    if calsrc:
        amp_prior_h[calsrc_idx] = args.calsrc_std
    amp_prior_std = tdev(amp_prior_h, torch.float64)
    gain_pspec_sqrt = args.gain_prior_amp * np.ones(ngm)
    if args.gain_prior_zero_mode_std is not None:
        zero_mode = np.where(np.logical_and(k_freq == 0.0, k_time == 0.0))[0]
        gain_pspec_sqrt[zero_mode] = float(args.gain_prior_zero_mode_std)
        log("Gain zero-mode prior level: %6.4e" % args.gain_prior_zero_mode_std, "b")
    gain_pspec_sqrt_d = tdev(gain_pspec_sqrt, torch.float64)
    A_real = A_imag = None
    if args.sample_gains:
        A_real, A_imag = hydra.gain_sampler.proj_operator(ants, antpairs)
    current_delta_gain = torch.zeros_like(gains_chunk)
    zero_vis = torch.zeros((nbl, nf, nt), dtype=torch.complex128, device=dev)
    # components of the current visibility model, without any gain factor (:417-423).  A component
    # that is not being sampled stays at its simulated truth (the reference starts the total from
    # model0 but the components from 0, so an --sh run without --ptsrc would fit the point
    # sources with spherical harmonics)
    model_ptsrc, model_sh = source_proj.apply(ones), zero_vis
    current_model = model0_chunk.clone()
    x_soln = torch.zeros(Nptsrc, dtype=torch.float64, device=dev)
    mpicg_groups = None
    # the source head's operator and its Cholesky factor change only with the gains (and the noise
    # weights / prior): kept between iterations, rebuilt when the gain head has moved
    source_cache = hydra.ptsrc_sampler.SystemCache() if args.solver_name != "mpicg" else None
    if args.sample_sh:
        inv_prior_sh = torch.full((nmodes_sh,), 1.0 / args.sh_prior_std ** 2, dtype=torch.float64,
                                  device=dev)
        sh_current = torch.zeros(nmodes_sh, dtype=torch.float64, device=dev)
        a_0 = torch.zeros_like(sh_current)
        t0 = time.time()
        # the left-hand side does not change between iterations: built once on the matrix cores
        # (the reference forms it inside every CG iteration, sh_sampler.py:393-398)
        # (rebuilt below whenever the gains, which scale its noise weights, have changed)
        sh_lhs = None

    # ---- Gibbs loop (example.py:663-1322)
    for n in range(args.Niters):
        log("-" * 60 + "\n>>> Iteration %4d / %4d\n" % (n + 1, args.Niters) + "-" * 60)
        t_iter = time.time()

        if args.sample_gains:                             # (A) example.py:673-745
            t0 = time.time()
            if on_head is not None:
                ggv = hydra.apply_gains(current_model, gains_chunk, ants, antpairs, inline=False)
                notify("gains", n, resid=data_chunk - ggv, model_vis=ggv,
                       inv_noise_var=inv_noise_var_chunk, pspec_sqrt=gain_pspec_sqrt,
                       Fbasis=Fbasis_h, seed=100000 * myid + n, ants=ants, antpairs=antpairs)
                del ggv
            # apply_gains, the residual, construct_rhs_mpi and the operator weights
            # (example.py:690-745) in one pass over the visibilities
            bgain, op = hydra.gain_sampler.prepare_sample(
                comm, data_chunk, current_model, gains_chunk, inv_noise_var_chunk,
                gain_pspec_sqrt_d, A_real, A_imag, Fbasis, realisation=True,
                seed=100000 * myid + n)
            if comm is None:
                lin = op.apply
            else:
                def lin(v, op=op):        # one all-reduce per CG iteration, every rank iterates
                    t = op.local(v)
                    comm.allreduce_(t)
                    return v + t
                lin.comm = comm           # cg: the ranks agree on when the converged flag is read
            xgain = hydra.linear_solver.cg(Amat=None, bvec=bgain, linear_op=lin, use_norm_tol=True)
            half = xgain.numel() // 2
            xg = torch.complex(xgain[:half], xgain[half:]).reshape(nants, ngm) * gain_pspec_sqrt_d
            if myid == 0:
                np.save(os.path.join(out, "delta_g_amps_%05d" % n), xg.cpu().numpy())
            current_delta_gain = torch.tensordot(xg, Fbasis, dims=([1], [0]))
            timing(n, "(A) Gain sampler", time.time() - t0)

        if args.sample_cosmo_field:                       # (BBBB) example.py:751-851
            t0 = time.time()
            ggv = hydra.apply_gains(current_model, gains_chunk, ants, antpairs, inline=False)
            pspec3d = hydra.cosmo_sampler.calculate_pspec_on_grid(
                kbins=cosmo_pspec_kbins, pspec=cosmo_pspec_current, x=cosmo_x, y=cosmo_y,
                freqs=freqs, **cosmo_background_params)
            cosmo_op, cosmo_rhs = hydra.cosmo_sampler.precompute_mpi(
                comm, freqs=freqs, ants=ants, antpairs=antpairs, freq_chunk=freq_chunk,
                time_chunk=time_chunk, proj_chunk=cosmo_proj, data_chunk=data_chunk - ggv,
                inv_noise_var_chunk=inv_noise_var_chunk,
                gain_chunk=gains_chunk * (1.0 + current_delta_gain), pspec3d=pspec3d,
                realisation=True)
            if myid == 0:
                p3d = tdev(pspec3d, torch.float64)
                cosmo_soln = hydra.linear_solver.cg(
                    Amat=None, bvec=cosmo_rhs.flatten(),
                    linear_op=lambda v: hydra.cosmo_sampler.apply_lhs_operator(
                        v.reshape(cosmo_soln_shape), cosmo_op, p3d).flatten())
                np.save(os.path.join(out, "cosmo_field_%05d" % n),
                        cosmo_soln.reshape(cosmo_soln_shape).cpu().numpy())
            if comm is not None:
                comm.barrier()
            timing(n, "(BBBB) Cosmo field sampler", time.time() - t0)

        if args.sample_ptsrc:                             # (B) example.py:856-984
            t0 = time.time()
            g_now = gains_chunk * (1.0 + current_delta_gain)
            resid = data_chunk - hydra.apply_gains(model_sh, gains_chunk, ants, antpairs,
                                                   inline=False)
            notify("ptsrc", n, resid=resid, gains=g_now, inv_noise_var=inv_noise_var_chunk,
                   amp_prior_std=amp_prior_h, ra=ra, dec=dec, fluxes=fluxes_chunk, ant_pos=ant_pos,
                   antpairs=antpairs, ants=ants, freqs=freq_chunk, times=time_chunk, latitude=lat)
            source_op, source_rhs = hydra.ptsrc_sampler.precompute_mpi(
                comm, ants=ants, antpairs=antpairs, freq_chunk=freq_chunk, time_chunk=time_chunk,
                proj_chunk=source_proj, data_chunk=resid, inv_noise_var_chunk=inv_noise_var_chunk,
                gain_chunk=g_now, amp_prior_std=amp_prior_std, realisation=True,
                cache=source_cache)
            if args.solver_name == "mpicg" and comm is not None and ngrid > 0:
                # sqrt(N) x sqrt(N) blocks of the operator, one per worker (example.py:899-938)
                shape = comm.bcast(tuple(source_op.shape), root=0)
                if mpicg_groups is None:
                    mpicg_groups = hydra.linear_solver.setup_mpi_blocks(
                        comm, matrix_shape=shape, split=ngrid)
                comm_groups, block_map, block_shape = mpicg_groups
                my_A = my_b = None
                if comm_groups is not None:
                    my_A, my_b = hydra.linear_solver.collect_linear_sys_blocks(
                        comm_groups[0], block_map, block_shape, Amat=source_op, bvec=source_rhs)
                comm.barrier()
                xs = hydra.linear_solver.cg_mpi(comm_groups, my_A, my_b, shape[0], block_map)
                if myid == 0:
                    x_soln = _dev.f64(xs)
            elif myid == 0:
                if args.solver_name == "mpicg":           # a single worker: the same CG, x0 = 0
                    x_soln = hydra.linear_solver.cg(source_op, source_rhs)
                else:     # the reference's dense solve, whatever the solver's name (:940-947);
                    # factored on the first iteration after the gains changed, reused otherwise
                    x_soln = source_cache.solve(source_rhs)
            if myid == 0:
                x_soln = x_soln * amp_prior_std           # we solved for x = S^-1/2 s
                np.save(os.path.join(out, "src_amp_%05d" % n), x_soln.cpu().numpy())
            if comm is not None:
                comm.bcast_(x_soln, root=0)
            model_ptsrc = source_proj.apply(1.0 + x_soln)
            current_model = model_ptsrc + model_sh
            timing(n, "(B) Source sampler", time.time() - t0)

        if args.sample_sh:                                # (C) example.py:990-1069
            t0 = time.time()
            # The SH response carries no gains (sh_sampler.py:65-196) while the data do: fit the
            # gain-corrected residual d / (g_i g_j^*) - ptsrc model.  (The reference subtracts
            # g g^* x model from the data and fits the gain-free response to what is left,
            # example.py:1000-1013, which scales the modes by |g|^2 ~ 2.)
            g_now = gains_chunk * (1.0 + current_delta_gain)
            gg = hydra.apply_gains(torch.ones((nbl, nf, nt), dtype=torch.complex128, device=dev),
                                   g_now, ants, antpairs, inline=False)
            resid = data_chunk / gg - model_ptsrc
            # dividing the data by g_i g_j^* divides its noise variance by |g_i g_j|^2: both sides
            # of the system use N^-1 |g_i g_j|^2
            w_sh = inv_noise_var_chunk * (gg.real ** 2 + gg.imag ** 2)
            if sh_lhs is None or args.sample_gains:
                t1 = time.time()
                sh_lhs = hydra.sh_sampler.lhs_matrix(w_sh, inv_prior_sh, sh_response, comm=comm)
                log("SH left-hand-side matrix took %6.3f sec" % (time.time() - t1), "c")
            notify("sh", n, resid=resid, inv_noise_var=w_sh,
                   inv_prior_var=inv_prior_sh, a_0=a_0, x0=sh_current, vis_response=sh_response)
            omega_a = np.random.randn(nmodes_sh)           # example.py:1022-1025
            omega_n = (1.0 * np.random.randn(nbl, nf, nt)
                       + 1.0j * np.random.randn(nbl, nf, nt)) / np.sqrt(2.0)
            omega_n = tdev(omega_n, torch.complex128)
            if comm is None:
                bsh = hydra.sh_sampler.construct_rhs_no_rot(
                    resid.reshape(-1), w_sh, inv_prior_sh,
                    tdev(omega_a, torch.float64), omega_n, a_0, sh_response)
            else:
                bsh = hydra.sh_sampler.construct_rhs_no_rot_mpi(
                    comm, resid.reshape(-1), w_sh, inv_prior_sh, omega_a, omega_n,
                    a_0, sh_response)
                bsh = _dev.f64(bsh)
            if myid == 0:
                sh_current = hydra.linear_solver.cg(sh_lhs, bsh, maxiters=1000, abs_tol=1e-8,
                                                    use_norm_tol=False, x0=sh_current)
                np.save(os.path.join(out, "sh_amp_%05d" % n), sh_current.cpu().numpy())
            if comm is not None:
                comm.bcast_(sh_current, root=0)
            model_sh = torch.empty((nbl, nf, nt), dtype=torch.complex128, device=dev)
            _dev.call("hydra_apply_R", _dev.ptr(sh_response), nbl * nf * nt, nmodes_sh,
                      _dev.ptr(sh_current), _dev.ptr(None), _dev.ptr(model_sh), _dev.stream_ptr())
            current_model = model_ptsrc + model_sh
            timing(n, "(C) SH sampler", time.time() - t0)

        torch.cuda.synchronize()
        log("Iteration %d took %6.3f sec" % (n, time.time() - t_iter), "c")
    if comm is not None:
        comm.barrier()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
