"""
Pin the oracle (oracle/*.py) against golden vectors produced by the REFERENCE's
own pure-numpy functions (tests/golden/make_golden.py) and against the
reference's single numerical test (hydra/tests/test_linear_solver.py:9-54).
CPU only.
"""
import numpy as np

from conftest import load_golden, relerr
from oracle import cg as ocg
from oracle import gcr_gain, gcr_ptsrc, gcr_sh, helpers

TIGHT = 1e-13


def _pairs(arr):
    return [tuple(int(x) for x in p) for p in arr]


def test_helpers_match_reference():
    g = load_golden("helpers.npz")
    np.random.seed(0)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(
        25, ra_bounds=(0.0, 1.0), dec_bounds=(-0.6, 0.4), logflux_bounds=(-1.0, 2.0))
    assert np.array_equal(ra, g["ra"]) and np.array_equal(dec, g["dec"])
    assert np.array_equal(amps, g["amps"])
    fl = helpers.get_flux_from_ptsrc_amp(amps, g["freqs"], -2.7)
    assert np.array_equal(fl, g["fluxes"])
    flc = helpers.get_flux_from_ptsrc_amp(amps, g["freqs"], -2.7, curv_ptsrc=0.3)
    assert np.array_equal(flc, g["fluxes_curv"])
    for key, kw in [("hex34", dict(hex_spec=(3, 4))), ("hex47", dict(hex_spec=(4, 7))),
                    ("hexrows", dict(ants_per_row=[2, 3, 2], d=10.0))]:
        ants = helpers.build_hex_array(**kw)
        assert np.array_equal(np.array([ants[k] for k in ants]), g[key])
    assert len(helpers.build_hex_array((3, 4))) == 10
    assert len(helpers.build_hex_array((4, 7))) == 37
    assert len(helpers.build_hex_array((7, 13))) == 127
    freqs, times = g["freqs"], g["times"]
    Fb, kf, kt = helpers.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 2, Lfreq=freqs[-1] - freqs[0], Ltime=times[-1] - times[0],
        freq0=freqs[0], time0=times[0], shape0=(freqs.size, times.size),
        positive_only=True)
    assert np.array_equal(Fb, g["Fb"]) and np.array_equal(kf, g["kf"])
    assert np.array_equal(kt, g["kt"])
    Fb2, kf2, kt2 = helpers.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 1, Lfreq=25.0, Ltime=0.4, positive_only=False)
    assert np.array_equal(Fb2, g["Fb2"]) and np.array_equal(kf2, g["kf2"])


def test_apply_gains_and_extract_match_reference():
    g = load_golden("gains.npz")
    ap = _pairs(g["antpairs"])
    out = helpers.apply_gains(g["v"], g["g"], g["ants"], ap)
    assert np.array_equal(out, g["out_plain"])
    out = helpers.apply_gains(g["v"], g["g"], g["ants"], ap, perturbation=g["pert"])
    assert np.array_equal(out, g["out_pert"])
    assert np.array_equal(helpers.extract_vis_from_sim(g["ants"], ap, g["simvis"]),
                          g["extracted"])
    assert np.array_equal(helpers.flatten_vector(g["v"]), g["flat"])
    assert np.array_equal(helpers.reconstruct_vector(g["flat"], g["v"].shape), g["recon"])
    # inline=True mutates and returns the same buffer
    v = g["v"].copy()
    out = helpers.apply_gains(v, g["g"], g["ants"], ap, inline=True)
    assert out is v and np.array_equal(v, g["out_plain"])


def test_ptsrc_precompute_matches_reference():
    for name in ["ptsrc_small.npz", "ptsrc_c1like.npz"]:
        g = load_golden(name)
        ap = _pairs(g["antpairs"])
        # injected omegas
        A, b = gcr_ptsrc.precompute(g["ants"], ap, g["proj"], g["data"], g["inv_noise"],
                                    g["gains"], g["prior"], omega_n=g["omega_n"],
                                    omega_a=g["omega_a"])
        assert relerr(A, g["A"]) < TIGHT
        assert relerr(b, g["b"]) < TIGHT
        # drawing from the global RNG in the reference's order gives the same
        np.random.seed(int(g["seed"]))
        A2, b2 = gcr_ptsrc.precompute(g["ants"], ap, g["proj"], g["data"], g["inv_noise"],
                                      g["gains"], g["prior"])
        assert relerr(b2, g["b"]) < TIGHT
        # realisation off
        A0, b0 = gcr_ptsrc.precompute(g["ants"], ap, g["proj"], g["data"], g["inv_noise"],
                                      g["gains"], g["prior"], realisation=False)
        assert relerr(A0, g["A0"]) < TIGHT and relerr(b0, g["b0"]) < TIGHT
        # the input operator is not mutated
        assert np.array_equal(g["proj"], load_golden(name)["proj"])
        # A is symmetric positive definite; dense solve has a tiny residual
        assert relerr(A, A.T) < 1e-14
        x = gcr_ptsrc.dense_solve(A, b)
        assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-10


def test_sh_algebra_matches_reference():
    g = load_golden("sh_algebra.npz")
    lmax = int(g["lmax"])
    ems, ells, idx = gcr_sh.get_em_ell_idx(lmax)
    assert np.array_equal(ems, g["ems"]) and np.array_equal(ells, g["ells"])
    assert np.array_equal(idx, g["idx"])
    g10 = load_golden("sh_layout_lmax10.npz")
    ems10, ells10, idx10 = gcr_sh.get_em_ell_idx(10)
    assert np.array_equal(ems10, g10["ems"]) and np.array_equal(ells10, g10["ells"])
    # hydra/tests/test_sh_sampler.py:9-19
    assert np.sum(np.array(ems10) == 0) == 10 + 1
    rhs = gcr_sh.construct_rhs_no_rot(g["data"], g["inv_noise"], g["inv_prior"], g["omega_a"],
                                      g["omega_n"], g["a0"], g["R"])
    assert relerr(rhs, g["rhs"]) < TIGHT and relerr(rhs, g["rhs_mpi"]) < TIGHT
    lhs = gcr_sh.apply_lhs_no_rot(g["acr"], g["inv_noise"], g["inv_prior"], g["R"])
    assert relerr(lhs, g["lhs"]) < TIGHT and relerr(lhs, g["lhs_mpi"]) < TIGHT
    lhs2 = gcr_sh.apply_lhs_no_rot_matfree(g["acr"], g["inv_noise"], g["inv_prior"], g["R"])
    assert relerr(lhs2, g["lhs"]) < 1e-12
    hp_modes = gcr_sh.alms2healpy(g["packed"], lmax)
    assert np.array_equal(hp_modes, g["hp_modes"])
    assert np.array_equal(gcr_sh.healpy2alms(hp_modes), g["packed"])
    x = ocg.cg(None, g["rhs_mpi"], x0=g["a0"].copy(),
               linear_op=lambda v: gcr_sh.apply_lhs_no_rot(v, g["inv_noise"], g["inv_prior"],
                                                            g["R"]))
    assert relerr(x, g["x_cg"]) < 1e-12


def test_cg_matches_reference_and_its_own_test():
    g = load_golden("cg.npz")
    # the reference's test, verbatim numbers (test_linear_solver.py:12-54)
    np.random.seed(1)
    n = 401
    A1 = np.eye(n)
    xtrue = np.random.randn(n)
    assert np.array_equal(xtrue, g["xtrue"])
    x1 = ocg.cg(A1, A1 @ xtrue)
    assert np.allclose(xtrue, x1)
    x2 = ocg.cg(g["A2"], g["b2"])
    assert np.allclose(xtrue, x2)
    assert np.array_equal(x2, g["x2"])          # same operation order -> bit equal
    x2a = ocg.cg(None, g["b2"], linear_op=lambda v: g["A2"] @ v)
    assert np.allclose(x2, x2a)
    assert np.array_equal(ocg.cg(g["A2"], g["b2"], abs_tol=1e-6, use_norm_tol=True), g["x2n"])
    assert np.array_equal(ocg.cg(g["A2"], g["b2"], maxiters=5), g["x2_5"])
    assert np.array_equal(ocg.cg(g["A3"], g["b3"], abs_tol=1e-10, x0=g["x03"].copy()), g["x3"])
    # blocked CG (cg_mpi layout) converges to the same solution
    for split in (1, 2, 3):
        xb = ocg.cg_blocked(g["A3"], g["b3"], split, abs_tol=1e-10)
        assert relerr(xb, np.linalg.solve(g["A3"], g["b3"])) < 1e-9
    xb = ocg.cg_blocked(g["A2"], g["b2"], 2)
    assert np.allclose(xtrue, xb)


def test_gain_operator_matches_reference():
    g = load_golden("gain.npz")
    ap = _pairs(g["antpairs"])
    i1, i2 = gcr_gain.bl_indices(g["ants"], ap)
    # index form == sparse matrices of the reference
    A_real = np.zeros_like(g["A_real"])
    A_imag = np.zeros_like(g["A_imag"])
    for k in range(len(ap)):
        A_real[k, i1[k]] += 1
        A_real[k, i2[k]] += 1
        A_imag[k, i1[k]] += 1
        A_imag[k, i2[k]] -= 1
    assert np.array_equal(A_real, g["A_real"]) and np.array_equal(A_imag, g["A_imag"])
    assert relerr(gcr_gain.apply_proj(g["x"], i1, i2, g["mv"]), g["vproj"]) < TIGHT
    assert relerr(gcr_gain.apply_proj_conj(g["vv"], i1, i2, g["mv"], g["x"].shape),
                  g["gconj"]) < TIGHT
    opx = gcr_gain.apply_operator(g["xvec"], g["inv_noise"], g["ps_sqrt"], i1, i2, g["mv"],
                                  g["Fb"])
    assert relerr(opx, g["opx"]) < TIGHT
    nant, nmodes = g["x"].shape[0], g["Fb"].shape[0]
    om_y, om_r = gcr_gain.draw_gain_omegas(int(g["seed"]), nant, nmodes, g["resid"].shape)
    b = gcr_gain.construct_rhs(g["resid"], g["inv_noise"], g["ps_sqrt"], i1, i2, g["mv"],
                               g["Fb"], om_y, om_r)
    assert relerr(b, g["bvec"]) < TIGHT


# ---------------------------------------------------------------- cosmo / region heads

def test_cosmo_grid_modes_and_prior_match_reference():
    from oracle import gcr_cosmo as oc
    g = load_golden("cosmo.npz")
    ra, dec = oc.field_grid((10.0, 16.0), (-33.0, -28.0), 5, 4)
    assert np.array_equal(ra, g["g_ra"]) and np.array_equal(dec, g["g_dec"])
    hz_fn, r_fn = oc.background(h=0.69, omega_m=0.31)
    assert np.array_equal(hz_fn(g["zs"]), g["hz"]) and np.array_equal(r_fn(g["zs"]), g["r"])
    x, y = np.unique(ra), np.unique(dec)
    kx, ky, knu = oc.fourier_modes(x, y, g["freqs"], h=0.69, omega_m=0.31)
    assert np.array_equal(kx, g["kx"]) and np.array_equal(ky, g["ky"])
    assert np.array_equal(knu, g["knu"])
    p3 = oc.pspec_on_grid(g["kbins"], g["pk"], x, y, g["freqs"], h=0.69, omega_m=0.31)
    assert np.array_equal(p3, g["pspec3d"])
    assert p3.shape == (g["freqs"].size, x.size, y.size)
    for key, ex in [("s_plus", 1), ("s_inv", -1.0), ("s_mhalf", -0.5)]:
        assert np.array_equal(oc.apply_S(g["cube"], p3, exponent=ex), g[key])
    assert np.array_equal(oc.apply_lhs_operator(g["xvec"], g["op"], p3), g["lhs"])


def test_cosmo_precompute_matches_reference():
    from oracle import gcr_cosmo as oc
    g = load_golden("cosmo.npz")
    ap = _pairs(g["antpairs"])
    args = (g["ants"], ap, g["freqs"])
    op, rhs = oc.precompute(*args, g["freqs"], g["proj"], g["data"], g["inv_noise"], g["gains"],
                            g["pspec3d"], omega_n=g["omega_n"], omega_s=g["omega_s"])
    assert relerr(op, g["op"]) < TIGHT and relerr(rhs, g["rhs"]) < TIGHT
    np.random.seed(int(g["seed"]))          # draws in the reference's order
    op, rhs = oc.precompute(*args, g["freqs"], g["proj"], g["data"], g["inv_noise"], g["gains"],
                            g["pspec3d"])
    assert relerr(rhs, g["rhs"]) < TIGHT
    op0, rhs0 = oc.precompute(*args, g["freqs"], g["proj"], g["data"], g["inv_noise"], g["gains"],
                              g["pspec3d"], realisation=False)
    assert relerr(op0, g["op0"]) < TIGHT and relerr(rhs0, g["rhs0"]) < TIGHT
    sl = slice(2, 5)                        # a rank holding channels 2..4 only
    opp, rhsp = oc.precompute(*args, g["freqs"][sl], g["proj"][:, sl], g["data"][:, sl],
                              g["inv_noise"][:, sl], g["gains"][:, sl], g["pspec3d"],
                              realisation=False)
    assert relerr(opp, g["op_part"]) < TIGHT and relerr(rhsp, g["rhs_part"]) < TIGHT
    assert not opp[0].any() and not opp[5].any()


def test_region_segmentation_matches_reference():
    from oracle import gcr_region
    g = load_golden("region_segments.npz")
    for nreg in (3, 8):
        idx = gcr_region.segment_pixels(g["ra"], g["dec"], g["sky"], g["freqs"], nreg)
        assert len(idx) == int(g[f"n{nreg}_count"])
        for q, ix in enumerate(idx):
            assert np.array_equal(ix, g[f"n{nreg}_{q}"])
        assert sum(ix.size for ix in idx) == g["ra"].size


# ---------------------------------------------------------------- beam-head consumers

def test_beam_head_helpers_match_reference():
    from oracle import gcr_beam as ob
    g = load_golden("beam.npz")
    assert np.array_equal(ob.get_ant_inds(2, 5), g["keep"])
    bess = ob.get_bess_matr(g["nmodes"], g["mmodes"], g["rho"], g["phi"])
    assert np.array_equal(bess, g["bess"])
    assert np.array_equal(ob.get_bess_outer(bess), g["outer"])
    tr = ob.get_bess_to_vis_from_contraction(g["bsc"], g["coeffs"], 5, 2)
    assert tr.shape == g["trans"].shape and relerr(tr, g["trans"]) < TIGHT
    tr = ob.get_bess_to_vis_from_contraction(g["bsc_ref"], g["coeffs"], 5, 2, ref_contraction=True)
    assert tr.shape == g["trans_ref"].shape and relerr(tr, g["trans_ref"]) < TIGHT


def test_beam_head_two_routes_to_the_operator_agree():
    """get_bess_to_vis == get_bess_outer -> get_bess_sky_contraction -> ..._from_contraction
    (the reference documents the former as the slower alternative of the latter)."""
    from oracle import gcr_beam as ob, helpers
    rng = np.random.RandomState(5)
    ant_pos = helpers.build_hex_array(hex_spec=(2, 3), d=14.6)
    nant = len(ant_pos)
    np.random.seed(4)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(12, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.linspace(100.0, 110.0, 2)
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    lsts = np.linspace(0.2, 0.5, 3)
    nm, mm = np.array([1, 2, 1, 2]), np.array([0, 0, 1, 1])
    bess = ob.get_bess_matr(nm, mm, rng.uniform(0, 1, (3, 12)), rng.uniform(0, 6, (3, 12)))
    coeffs = rng.randn(4, 2, nant, 1, 1) + 1j * rng.randn(4, 2, nant, 1, 1)
    bsc = ob.get_bess_sky_contraction(ob.get_bess_outer(bess), ant_pos, fluxes, ra, dec,
                                      freqs * 1e6, lsts)
    a = ob.get_bess_to_vis_from_contraction(bsc, coeffs, nant, 3)
    b = ob.get_bess_to_vis(bess, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, coeffs, 3)
    assert a.shape == b.shape == (1, 1, 2, 3, nant - 1, 4)
    assert relerr(a, b) < 1e-12
