#!/usr/bin/env python
"""
Gibbs driver on MI355X: the counterpart of the reference's ``example.py`` for the heads that sit
on the GCR hot path -- point-source amplitudes (--ptsrc), linearised gains (--gains),
spherical-harmonic modes (--sh) and the cosmological field (--cosmo) -- with the reference's flag
names (hydra/config.py) and output files (ptsrc_amps0.npy, sim_*_w%04d.npy, src_amp_%05d.npy,
delta_g_amps_%05d.npy, sh_amp_%05d.npy; example.py:191-192,405-408,740,953,1061).  The
reference's cosmo step plots its first solution and exits (example.py:829-841); here every
sample is saved as cosmo_field_%05d.npy and, as there (:850-851), does not feed back into the
data model.

    python example.py --ptsrc --gains --Niters 10
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 example.py --ptsrc

One process per GPU; the frequency channels are split over ranks (the reference splits a
freq x time grid over MPI ranks, example.py:161-174) and ``hydra_amd._dist.Comm`` stands where
``MPI.COMM_WORLD`` stood.  All arrays of the loop stay in HBM; the operator of the source head
is kept in factored form.  --regions needs a pygdsm sky model (not shipped; the region operator
itself is hydra_amd.region_sampler.calc_proj_operator); --beam and --vis are outside the hot
path.
"""
import argparse
import os
import time

import numpy as np
import torch

import hydra_amd as hydra
from hydra_amd import _dev, _dist


def get_config(argv=None):
    """The flags of hydra/config.py:5-589 that the three heads use, same names / dests."""
    p = argparse.ArgumentParser(description="Hydra GCR heads on MI355X")
    p.add_argument("--seed", type=int, default=0, dest="seed")
    p.add_argument("--ptsrc", action="store_true", dest="sample_ptsrc")
    p.add_argument("--gains", action="store_true", dest="sample_gains")
    p.add_argument("--sh", action="store_true", dest="sample_sh")
    p.add_argument("--cosmo", action="store_true", dest="sample_cosmo_field")
    p.add_argument("--timing", action="store_true", dest="save_timing_info")
    p.add_argument("--latitude", type=float, default=-30.7215, dest="latitude")
    p.add_argument("--hex-array", type=int, nargs=2, default=(3, 4), dest="hex_array")
    p.add_argument("--Nptsrc", type=int, default=100, dest="Nptsrc")
    p.add_argument("--Ntimes", type=int, default=30, dest="Ntimes")
    p.add_argument("--Nfreqs", type=int, default=60, dest="Nfreqs")
    p.add_argument("--Niters", type=int, default=100, dest="Niters")
    p.add_argument("--sigma-noise", type=float, default=0.05, dest="sigma_noise")
    p.add_argument("--solver", type=str, default="cg", dest="solver_name")
    p.add_argument("--output-dir", type=str, default="./output", dest="output_dir")
    p.add_argument("--beam-type", type=str, default="gaussian", dest="beam_type")
    p.add_argument("--ra-bounds", type=float, nargs=2, default=(0.0, 1.0), dest="ra_bounds")
    p.add_argument("--dec-bounds", type=float, nargs=2, default=(-0.6, 0.4), dest="dec_bounds")
    p.add_argument("--lst-bounds", type=float, nargs=2, default=(0.2, 0.5), dest="lst_bounds")
    p.add_argument("--freq-bounds", type=float, nargs=2, default=(100.0, 120.0), dest="freq_bounds")
    p.add_argument("--ptsrc-amp-prior-level", type=float, default=0.1, dest="ptsrc_amp_prior_level")
    p.add_argument("--gain-prior-amp", type=float, default=0.1, dest="gain_prior_amp")
    p.add_argument("--gain-nmax-freq", type=int, default=2, dest="gain_nmaxfreq")
    p.add_argument("--gain-nmax-time", type=int, default=2, dest="gain_nmaxtime")
    p.add_argument("--gain-only-positive-modes", type=bool, default=True,
                   dest="gain_only_positive_modes")
    p.add_argument("--sim-gain-amp-std", type=float, default=0.05, dest="sim_gain_amp_std")
    p.add_argument("--cosmo-ra-bounds", type=float, nargs=2, default=(0, 60),
                   dest="cosmo_field_ra_bounds")
    p.add_argument("--cosmo-dec-bounds", type=float, nargs=2, default=(-40.0, -20.0),
                   dest="cosmo_field_dec_bounds")
    p.add_argument("--cosmo-ra-ngrid", type=int, default=10, dest="cosmo_field_ra_ngrid")
    p.add_argument("--cosmo-dec-ngrid", type=int, default=10, dest="cosmo_field_dec_ngrid")
    p.add_argument("--sh-lmax", type=int, default=8, dest="sh_lmax")
    p.add_argument("--sh-nside", type=int, default=16, dest="sh_nside")
    p.add_argument("--sh-prior-std", type=float, default=0.1, dest="sh_prior_std")
    p.add_argument("--sh-ref-freq", type=float, default=100.0, dest="sh_ref_freq")
    p.add_argument("--sh-spectral-idx", type=float, default=0.0, dest="sh_spectral_idx")
    return p.parse_args(argv)


def main(argv=None):
    args = get_config(argv)
    comm = _dist.init_from_env()
    myid = comm.Get_rank() if comm is not None else 0
    nworkers = comm.Get_size() if comm is not None else 1
    if comm is None:
        torch.cuda.set_device(0)
    dev = _dev.device()
    np.random.seed(args.seed + myid)                     # example.py:139
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

    # ---- array, catalogue, this rank's channels (example.py:149-201)
    lat = np.deg2rad(args.latitude)
    freqs = np.linspace(args.freq_bounds[0], args.freq_bounds[1], args.Nfreqs)
    times = np.linspace(args.lst_bounds[0], args.lst_bounds[1], args.Ntimes)
    fidx = np.array_split(np.arange(args.Nfreqs), nworkers)[myid]
    freq_chunk, time_chunk = freqs[fidx], times
    ant_pos = hydra.build_hex_array(hex_spec=tuple(args.hex_array), d=14.6)
    ants = np.array(list(ant_pos.keys()))
    nants = len(ants)
    antpairs = [(i, j) for i in range(nants) for j in range(i + 1, nants)]
    nbl, nf, nt = len(antpairs), freq_chunk.size, time_chunk.size
    cat = np.zeros((3, args.Nptsrc))
    if myid == 0:
        cat[:] = hydra.example.generate_random_ptsrc_catalogue(
            args.Nptsrc, ra_bounds=args.ra_bounds, dec_bounds=args.dec_bounds,
            logflux_bounds=(-1.0, 2.0))
        np.save(os.path.join(out, "ptsrc_amps0"), cat[2])
        np.save(os.path.join(out, "ptsrc_coords0"), cat[:2])
    if comm is not None:
        comm.Bcast(cat, root=0)
    ra, dec, ptsrc_amps = cat
    fluxes_chunk = hydra.get_flux_from_ptsrc_amp(ptsrc_amps, freq_chunk, -2.7)
    beams = hydra.example.example_beams(args.beam_type, len(ants))

    # ---- operators (example.py:507-523, 597-624): factored source operator, SH response
    t0 = time.time()
    source_proj = hydra.ptsrc_sampler.calc_proj_operator(
        ra=ra, dec=dec, fluxes=fluxes_chunk, ant_pos=ant_pos, antpairs=antpairs,
        freqs=freq_chunk, times=time_chunk, beams=beams, latitude=lat, materialize=False)
    log("Source projection operator (factored) took %6.3f sec" % (time.time() - t0), "c")
    ones = torch.ones(args.Nptsrc, dtype=torch.float64, device=dev)
    model0_chunk = source_proj.apply(ones)               # the simulated sky = the catalogue
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
    if args.sample_cosmo_field:                          # example.py:539-594
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

    # ---- gains and simulated data (example.py:322-408)
    Lfreq, Ltime = freqs[-1] - freqs[0], times[-1] - times[0]
    Fb_all, k_freq, k_time = hydra.partial_fourier_basis_2d_from_nmax(
        freqs=freqs, times=times, nmaxfreq=args.gain_nmaxfreq, nmaxtime=args.gain_nmaxtime,
        Lfreq=Lfreq, Ltime=Ltime, freq0=freqs[0], time0=times[0],
        shape0=(freqs.size, times.size), positive_only=args.gain_only_positive_modes)
    Fbasis_h = np.ascontiguousarray(Fb_all[:, fidx, :])
    ngm = k_freq.size
    np.save(os.path.join(out, "Fbasis_w%04d" % myid), Fbasis_h)
    gain0_level = np.zeros(nants, dtype=np.complex128)
    delta_g_amps0 = np.zeros((nants, ngm), dtype=np.complex128)
    prior_std_delta_g = args.sim_gain_amp_std * np.ones(ngm)
    prior_std_delta_g[0] *= 0.0                           # example.py:362
    if myid == 0:
        gain0_level[:] = 0.05 * (np.random.randn(nants) + 1.0j * np.random.randn(nants))
        np.random.seed(1)                                 # example.py:368
        delta_g_amps0[:] = prior_std_delta_g * (np.random.randn(nants, ngm)
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
    gen = torch.Generator(device=dev).manual_seed(args.seed + 1000 + myid)
    data_chunk = data_chunk + args.sigma_noise * torch.randn(
        data_chunk.shape, dtype=torch.complex128, device=dev, generator=gen)
    inv_noise_var_chunk = torch.full((nbl, nf, nt), 1.0 / args.sigma_noise ** 2,
                                     dtype=torch.float64, device=dev)
    np.save(os.path.join(out, "sim_model0_chunk_w%04d" % myid), model0_chunk.cpu().numpy())
    np.save(os.path.join(out, "sim_data_chunk_w%04d" % myid), data_chunk.cpu().numpy())
    np.save(os.path.join(out, "sim_delta_g_amps0_w%04d" % myid), delta_g_amps0)
    log("Simulation step finished", "b")

    # ---- priors and starting state
    amp_prior_std = torch.full((args.Nptsrc,), args.ptsrc_amp_prior_level, dtype=torch.float64,
                               device=dev)
    gain_pspec_sqrt = args.gain_prior_amp * np.ones(ngm)
    gain_pspec_sqrt[0] = 1e-2 * args.gain_prior_amp       # example.py:562-570 (zero mode tight)
    gain_pspec_sqrt_d = tdev(gain_pspec_sqrt, torch.float64)
    A_real, A_imag = hydra.gain_sampler.proj_operator(ants, antpairs)
    current_delta_gain = torch.zeros_like(gains_chunk)
    zero_vis = torch.zeros((nbl, nf, nt), dtype=torch.complex128, device=dev)
    model_ptsrc = source_proj.apply(ones)
    model_sh = zero_vis.clone()
    x_soln = torch.zeros(args.Nptsrc, dtype=torch.float64, device=dev)
    if args.sample_sh:
        inv_prior_sh = torch.full((nmodes_sh,), 1.0 / args.sh_prior_std ** 2, dtype=torch.float64,
                                  device=dev)
        sh_current = torch.zeros(nmodes_sh, dtype=torch.float64, device=dev)
        a_0 = torch.zeros_like(sh_current)
        t0 = time.time()
        sh_lhs = hydra.sh_sampler.lhs_matrix(inv_noise_var_chunk, inv_prior_sh, sh_response,
                                             comm=comm)
        log("SH left-hand-side matrix took %6.3f sec" % (time.time() - t0), "c")

    # ---- Gibbs loop (example.py:663-1322)
    for n in range(args.Niters):
        log("Iteration %d" % n, "b")
        t_iter = time.time()

        if args.sample_gains:                            # (A) example.py:673-745
            t0 = time.time()
            ggv = hydra.apply_gains(model_ptsrc + model_sh, gains_chunk, ants, antpairs,
                                    inline=False)
            resid = data_chunk - ggv
            bgain = hydra.gain_sampler.construct_rhs_mpi(
                comm=comm, resid=resid, inv_noise_var=inv_noise_var_chunk,
                pspec_sqrt=gain_pspec_sqrt_d, A_real=A_real, A_imag=A_imag, model_vis=ggv,
                Fbasis=Fbasis, realisation=True, seed=100000 * myid + n)
            op = hydra.gain_sampler.GainOperator(inv_noise_var_chunk, gain_pspec_sqrt_d, A_real,
                                                 A_imag, ggv, Fbasis)
            if comm is None:
                lin = op.apply
            else:
                def lin(v, op=op):
                    t = op.local(v)
                    comm.allreduce_(t)
                    return v + t
            xgain = hydra.linear_solver.cg(Amat=None, bvec=bgain, linear_op=lin, use_norm_tol=True)
            half = xgain.numel() // 2
            xg = torch.complex(xgain[:half], xgain[half:]).reshape(nants, ngm) * gain_pspec_sqrt_d
            if myid == 0:
                np.save(os.path.join(out, "delta_g_amps_%05d" % n), xg.cpu().numpy())
            current_delta_gain = torch.tensordot(xg, Fbasis, dims=([1], [0]))
            timing(n, "(A) Gain sampler", time.time() - t0)

        if args.sample_cosmo_field:                      # (BBBB) example.py:751-851
            t0 = time.time()
            ggv = hydra.apply_gains(model_ptsrc + model_sh, gains_chunk, ants, antpairs,
                                    inline=False)
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

        if args.sample_ptsrc:                            # (B) example.py:856-984
            t0 = time.time()
            g_now = gains_chunk * (1.0 + current_delta_gain)
            resid = data_chunk - hydra.apply_gains(model_sh, gains_chunk, ants, antpairs,
                                                   inline=False)
            source_op, source_rhs = hydra.ptsrc_sampler.precompute_mpi(
                comm, ants=ants, antpairs=antpairs, freq_chunk=freq_chunk, time_chunk=time_chunk,
                proj_chunk=source_proj, data_chunk=resid, inv_noise_var_chunk=inv_noise_var_chunk,
                gain_chunk=g_now, amp_prior_std=amp_prior_std, realisation=True)
            if myid == 0:
                if args.solver_name == "cg_explicit":
                    x_soln = hydra.linear_solver.cg(source_op, source_rhs)
                else:                                     # the reference's default: dense solve
                    x_soln = hydra.linear_solver.solve_dense(source_op, source_rhs)
                x_soln = x_soln * amp_prior_std
                np.save(os.path.join(out, "src_amp_%05d" % n), x_soln.cpu().numpy())
            if comm is not None:
                comm.bcast_(x_soln, root=0)
            model_ptsrc = source_proj.apply(1.0 + x_soln)
            timing(n, "(B) Source sampler", time.time() - t0)

        if args.sample_sh:                               # (C) example.py:990-1069
            t0 = time.time()
            g_now = gains_chunk * (1.0 + current_delta_gain)
            # the SH operator carries no gains (sh_sampler.py:405-494): fit the gain-corrected
            # residual, as the reference does with its gain-free SH response
            gg = torch.ones((nbl, nf, nt), dtype=torch.complex128, device=dev)
            gg = hydra.apply_gains(gg, g_now, ants, antpairs, inline=False)
            resid = ((data_chunk / gg) - model_ptsrc).reshape(-1)
            omega_a = np.random.randn(nmodes_sh)
            omega_n = torch.randn(resid.shape, dtype=torch.complex128, device=dev, generator=gen)
            if comm is None:
                bsh = hydra.sh_sampler.construct_rhs_no_rot(
                    resid, inv_noise_var_chunk, inv_prior_sh, tdev(omega_a, torch.float64),
                    omega_n, a_0, sh_response)
            else:
                bsh = hydra.sh_sampler.construct_rhs_no_rot_mpi(
                    comm, resid, inv_noise_var_chunk, inv_prior_sh, omega_a, omega_n, a_0,
                    sh_response)
                bsh = _dev.f64(bsh)
            if myid == 0:
                sh_current = hydra.linear_solver.cg(sh_lhs, bsh, maxiters=1000, abs_tol=1e-8,
                                                    x0=sh_current)
                np.save(os.path.join(out, "sh_amp_%05d" % n), sh_current.cpu().numpy())
            if comm is not None:
                comm.bcast_(sh_current, root=0)
            _dev.call("hydra_apply_R", _dev.ptr(sh_response), nbl * nf * nt, nmodes_sh,
                      _dev.ptr(sh_current), _dev.ptr(None), _dev.ptr(model_sh), _dev.stream_ptr())
            timing(n, "(C) SH sampler", time.time() - t0)

        torch.cuda.synchronize()
        log("Iteration %d took %6.3f sec" % (n, time.time() - t_iter), "c")
    if comm is not None:
        comm.barrier()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
