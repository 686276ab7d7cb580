"""
GPU parity tests of the cosmological-field and diffuse-region heads: the HIP
path through the hydra_amd API mirror against golden vectors produced by the reference's own
``cosmo_sampler`` / ``region_sampler`` functions (tests/golden/make_golden_cosmo.py) and against
the CPU oracle on seeded inputs.

Tolerances as in test_gpu_parity.py: operator entries <= 1e-10, samples <= 1e-8 (relative =
max|a-b| / max|b|).
"""
import argparse

import numpy as np
import pytest

from conftest import load_golden, operr, relerr

pytestmark = pytest.mark.gpu

OP_TOL = 1e-10
SAMPLE_TOL = 1e-8
LAT = -30.7215 * np.pi / 180.0


@pytest.fixture(scope="module")
def hy():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import hydra_amd
    return hydra_amd


def _pairs(arr):
    return [tuple(int(x) for x in p) for p in arr]


# ------------------------------------------------------------------ cosmological field

def test_cosmo_setup_functions_match_reference_golden(hy):
    cs = hy.cosmo_sampler
    g = load_golden("cosmo.npz")
    args = argparse.Namespace(cosmo_field_ra_bounds=(10.0, 16.0),
                              cosmo_field_dec_bounds=(-33.0, -28.0),
                              cosmo_field_ra_ngrid=5, cosmo_field_dec_ngrid=4)
    ra, dec = cs.make_cosmo_field_grid(args)
    assert np.array_equal(ra, g["g_ra"]) and np.array_equal(dec, g["g_dec"])
    Hz, dAc = cs.calculate_cosmo_fns(h=0.69, omega_m=0.31)
    assert np.array_equal(Hz(g["zs"]), g["hz"]) and np.array_equal(dAc(g["zs"]), g["r"])
    x, y = np.unique(ra), np.unique(dec)
    kx, ky, knu = cs.comoving_fourier_modes(x=x, y=y, freqs=g["freqs"], h=0.69, omega_m=0.31)
    assert np.array_equal(kx, g["kx"]) and np.array_equal(ky, g["ky"])
    assert np.array_equal(knu, g["knu"])
    p3 = cs.calculate_pspec_on_grid(kbins=g["kbins"], pspec=g["pk"], x=x, y=y, freqs=g["freqs"],
                                    h=0.69, omega_m=0.31)
    assert np.array_equal(p3, g["pspec3d"])


def test_cosmo_prior_and_operator_match_reference_golden(hy):
    cs = hy.cosmo_sampler
    g = load_golden("cosmo.npz")
    for key, ex in [("s_plus", 1), ("s_inv", -1.0), ("s_mhalf", -0.5)]:
        out = cs.apply_S(g["cube"], g["pspec3d"], exponent=ex)
        assert isinstance(out, np.ndarray) and out.shape == g["cube"].shape
        assert relerr(out, g[key]) < 1e-13
    lhs = cs.apply_lhs_operator(g["xvec"], g["op"], g["pspec3d"])
    assert lhs.shape == g["lhs"].shape
    assert relerr(lhs, g["lhs"]) < 1e-13


def test_cosmo_precompute_matches_reference_golden(hy):
    cs = hy.cosmo_sampler
    g = load_golden("cosmo.npz")
    ap = _pairs(g["antpairs"])
    op, rhs = cs.precompute_mpi(None, g["ants"], ap, g["freqs"], g["freqs"], g["times"],
                                g["proj"], g["data"], g["inv_noise"], g["gains"], g["pspec3d"],
                                omega_n=g["omega_n"], omega_s=g["omega_s"])
    assert op.shape == g["op"].shape and rhs.shape == g["rhs"].shape
    for f in range(op.shape[0]):
        assert operr(op[f], g["op"][f]) < OP_TOL
    assert operr(rhs, g["rhs"]) < OP_TOL
    # fluctuation vectors drawn inside, in the reference's order
    np.random.seed(int(g["seed"]))
    _, rhs2 = cs.precompute_mpi(None, g["ants"], ap, g["freqs"], g["freqs"], g["times"],
                                g["proj"], g["data"], g["inv_noise"], g["gains"], g["pspec3d"])
    assert operr(rhs2, g["rhs"]) < OP_TOL
    op0, rhs0 = cs.precompute_mpi(None, g["ants"], ap, g["freqs"], g["freqs"], g["times"],
                                  g["proj"], g["data"], g["inv_noise"], g["gains"], g["pspec3d"],
                                  realisation=False)
    assert operr(op0, g["op0"]) < OP_TOL and operr(rhs0, g["rhs0"]) < OP_TOL
    sl = slice(2, 5)      # a rank that holds channels 2..4 of the grid only
    opp, rhsp = cs.precompute_mpi(None, g["ants"], ap, g["freqs"], g["freqs"][sl], g["times"],
                                  g["proj"][:, sl], g["data"][:, sl], g["inv_noise"][:, sl],
                                  g["gains"][:, sl], g["pspec3d"], realisation=False)
    assert operr(opp, g["op_part"]) < OP_TOL and operr(rhsp, g["rhs_part"]) < OP_TOL
    assert not opp[0].any() and not opp[5].any()


@pytest.mark.parametrize("uniform_noise", [False, True])
def test_cosmo_head_factored_operator_and_cg_sample(hy, uniform_noise):
    """The whole head as the driver runs it (example.py:545-556,780-828): grid pixels as unit-flux
    sources, factored operator, per-channel Grams, CG on the FFT-prior operator -- against the
    oracle with the same injected fluctuation vectors."""
    import torch
    from oracle import beams as obeams, cg as ocg, gcr_cosmo as oc, gcr_ptsrc as op_, helpers
    cs, ps, ls = hy.cosmo_sampler, hy.ptsrc_sampler, hy.linear_solver
    rng = np.random.RandomState(7)
    ant_pos = helpers.build_hex_array(hex_spec=(3, 4), d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    freqs = np.linspace(100.0, 103.0, 4)
    times = np.linspace(0.2, 0.5, 5)
    g_ra, g_dec = oc.field_grid((20.0, 32.0), (-36.0, -26.0), 6, 5)
    npix = g_ra.size
    flux = np.ones((npix, freqs.size))
    x, y = np.unique(g_ra), np.unique(g_dec)
    kbins = np.linspace(0.0, 10.0, 25)
    pspec3d = oc.pspec_on_grid(kbins, 0.1 * kbins + 1.0, x, y, freqs, h=0.69, omega_m=0.31)
    assert pspec3d.shape == (freqs.size, x.size, y.size)

    proj_o = op_.calc_proj_operator(g_ra, g_dec, flux, ant_pos, antpairs, freqs, times,
                                    [obeams.GaussianBeam(14.0) for _ in range(nant)], LAT)
    proj_f = ps.calc_proj_operator(g_ra, g_dec, flux, ant_pos, antpairs, freqs, times,
                                   [hy.beams.GaussianBeam(14.0) for _ in range(nant)], LAT,
                                   materialize=False)
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, 1, 1) + 1j * rng.randn(nant, 1, 1))
    gains = gains * np.ones((nant, freqs.size, times.size))
    if uniform_noise:
        inv_noise = np.full((nbl, freqs.size, times.size), 400.0)
    else:
        inv_noise = rng.uniform(100.0, 400.0, size=(nbl, freqs.size, times.size))
    field = rng.randn(freqs.size, npix) * 0.5
    model = np.einsum("bftp,fp->bft", proj_o, field)
    data = helpers.apply_gains(model, gains, ants, antpairs)
    data = data + 0.05 * (rng.randn(*data.shape) + 1j * rng.randn(*data.shape))
    omega_n = (rng.randn(*data.shape) + 1j * rng.randn(*data.shape)) / np.sqrt(2.0)
    omega_s = rng.randn(*pspec3d.shape)

    op_ref, rhs_ref = oc.precompute(ants, antpairs, freqs, freqs, proj_o, data, inv_noise, gains,
                                    pspec3d, omega_n=omega_n, omega_s=omega_s)
    op, rhs = cs.precompute_mpi(None, ants, antpairs, freqs, freqs, times, proj_f, data,
                                inv_noise, gains, pspec3d, omega_n=omega_n, omega_s=omega_s)
    want = "gram_uniform_kernel" if uniform_noise else "gram_fused_kernel"
    assert proj_f.channel(0).complete
    for f in range(freqs.size):
        assert operr(op[f], op_ref[f]) < OP_TOL
    assert operr(rhs, rhs_ref) < OP_TOL
    sub = proj_f.channel(1)
    sub.gram(torch.as_tensor(gains[:, 1:2]).cuda().contiguous(),
             torch.as_tensor(inv_noise[:, 1:2]).cuda().contiguous())
    assert sub.last_gram_kernel == want

    shape3 = pspec3d.shape
    x_ref = ocg.cg(None, rhs_ref.flatten(), maxiters=2000, abs_tol=1e-10,
                   linear_op=lambda v: oc.apply_lhs_operator(v.reshape(shape3), op_ref,
                                                             pspec3d).flatten())
    opd, p3d = torch.as_tensor(op).cuda(), torch.as_tensor(pspec3d).cuda()
    x_gpu = ls.cg(Amat=None, bvec=torch.as_tensor(rhs).cuda().flatten(), maxiters=2000, abs_tol=1e-10,
                  linear_op=lambda v: cs.apply_lhs_operator(v.reshape(shape3), opd, p3d).flatten())
    assert x_gpu.is_cuda
    x_gpu = x_gpu.cpu().numpy()
    x_np = ls.cg(Amat=None, bvec=rhs.flatten(), maxiters=2000, abs_tol=1e-10,
                 linear_op=lambda v: cs.apply_lhs_operator(v.reshape(shape3), op, pspec3d).flatten())
    assert isinstance(x_np, np.ndarray) and relerr(x_np, x_ref) < SAMPLE_TOL
    assert relerr(x_gpu, x_ref) < SAMPLE_TOL
    res = oc.apply_lhs_operator(x_gpu.reshape(shape3), op_ref, pspec3d).flatten() - rhs_ref.flatten()
    assert np.abs(res).max() / np.abs(rhs_ref).max() < SAMPLE_TOL


# ------------------------------------------------------------------ diffuse regions

def test_region_segmentation_matches_reference_golden(hy):
    rs = hy.region_sampler
    g = load_golden("region_segments.npz")
    for nreg in (3, 8):
        idx = rs.segmented_diffuse_sky_model_pixels(g["ra"], g["dec"], g["sky"], g["freqs"], nreg)
        assert len(idx) == int(g[f"n{nreg}_count"])
        for q, ix in enumerate(idx):
            assert np.array_equal(ix, g[f"n{nreg}_{q}"])
    with pytest.raises(NotImplementedError):
        rs.segmented_diffuse_sky_model_pixels(g["ra"], g["dec"], g["sky"], g["freqs"], 3,
                                              smoothing_fwhm=2.0)
    with pytest.raises(NotImplementedError):
        rs.get_diffuse_sky_model_pixels(g["freqs"])


@pytest.mark.parametrize("hex_spec,npx,nreg,nf,nt", [((3, 4), 150, 4, 3, 4), ((2, 3), 70, 1, 2, 3),
                                                    ((4, 7), 333, 7, 2, 2)])
def test_region_proj_operator_matches_oracle(hy, hex_spec, npx, nreg, nf, nt):
    from oracle import beams as obeams, gcr_region, helpers
    rs = hy.region_sampler
    rng = np.random.RandomState(nreg)
    ant_pos = helpers.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    ra = rng.uniform(0.0, 2.0 * np.pi, npx)          # whole sky: most pixels below the horizon
    dec = np.arcsin(rng.uniform(-1.0, 1.0, npx))
    beta = rng.uniform(-3.0, -2.2, npx)
    sky = rng.uniform(1.0, 50.0, npx)[:, None] * (freqs[None, :] / 100.0) ** beta[:, None]
    if nreg > 1:
        idxs = gcr_region.segment_pixels(ra, dec, sky, freqs, nreg)
    else:
        idxs = [np.arange(npx)]                        # example.py:269: one region = whole map
    ref = gcr_region.calc_proj_operator(ra, dec, sky, idxs, ant_pos, antpairs, freqs, times,
                                        [obeams.GaussianBeam(14.0) for _ in range(nant)], LAT)
    out = rs.calc_proj_operator(region_pixel_ra=ra, region_pixel_dec=dec, region_fluxes=sky,
                                region_idxs=idxs, ant_pos=ant_pos, antpairs=antpairs,
                                freqs=freqs, times=times,
                                beams=[hy.beams.GaussianBeam(14.0) for _ in range(nant)],
                                latitude=LAT)
    assert isinstance(out, np.ndarray) and out.shape == ref.shape
    assert out.shape == (len(antpairs), nf, nt, len(idxs))
    assert operr(out, ref) < OP_TOL


def test_region_operator_edge_cases_and_joint_system(hy):
    """Empty region, overlapping regions, a reversed antenna pair; then the driver's joint
    point-source + region system (example.py:526-533,878-889) against the oracle."""
    from oracle import beams as obeams, gcr_ptsrc, gcr_region, helpers
    rs, ps = hy.region_sampler, hy.ptsrc_sampler
    rng = np.random.RandomState(3)
    ant_pos = helpers.build_hex_array(hex_spec=(3, 4), d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    freqs = np.linspace(100.0, 110.0, 3)
    times = np.linspace(0.2, 0.5, 4)
    npx = 90
    ra = rng.uniform(0.0, 1.2, npx)
    dec = rng.uniform(-0.9, -0.1, npx)
    sky = rng.uniform(0.5, 5.0, (npx, 1)) * (freqs[None, :] / 100.0) ** -2.5
    idxs = [np.arange(0, 40), np.array([], dtype=int), np.arange(30, 90), np.array([5])]
    ob = [obeams.GaussianBeam(14.0) for _ in range(nant)]
    hb = [hy.beams.GaussianBeam(14.0) for _ in range(nant)]
    odd_pairs = list(antpairs)
    odd_pairs[3] = (odd_pairs[3][1], odd_pairs[3][0])      # idx1 > idx2: zeros, as in the reference
    ref = gcr_region.calc_proj_operator(ra, dec, sky, idxs, ant_pos, odd_pairs, freqs, times, ob, LAT)
    out = rs.calc_proj_operator(ra, dec, sky, idxs, ant_pos, odd_pairs, freqs, times, hb, LAT)
    assert operr(out, ref) < OP_TOL
    assert not out[..., 1].any() and not out[3].any()

    # joint system: 12 point sources + the non-empty regions
    keep = [idxs[0], idxs[2]]
    r_ref = gcr_region.calc_proj_operator(ra, dec, sky, keep, ant_pos, antpairs, freqs, times, ob, LAT)
    r_out = rs.calc_proj_operator(ra, dec, sky, keep, ant_pos, antpairs, freqs, times, hb, LAT)
    np.random.seed(2)
    pra, pdec, amps = helpers.generate_random_ptsrc_catalogue(12, (0.0, 1.0), (-0.6, 0.4))
    pflux = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    p_ref = gcr_ptsrc.calc_proj_operator(pra, pdec, pflux, ant_pos, antpairs, freqs, times, ob, LAT)
    p_out = ps.calc_proj_operator(pra, pdec, pflux, ant_pos, antpairs, freqs, times, hb, LAT)
    proj_ref = np.concatenate((p_ref, r_ref), axis=-1)
    proj_out = np.concatenate((p_out, r_out), axis=-1)
    nsrc = proj_ref.shape[-1]
    prior = np.concatenate((0.1 * np.ones(12), 0.3 * np.ones(2)))
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, 3, 4) + 1j * rng.randn(nant, 3, 4))
    inv_noise = rng.uniform(50.0, 400.0, size=proj_ref.shape[:3])
    truth = 1.0 + 0.1 * rng.randn(nsrc)
    data = helpers.apply_gains((proj_ref.reshape(-1, nsrc) @ truth).reshape(proj_ref.shape[:3]),
                               gains, ants, antpairs)
    om_n = (rng.randn(*data.shape) + 1j * rng.randn(*data.shape)) / np.sqrt(2.0)
    om_a = rng.randn(nsrc)
    A_ref, b_ref = gcr_ptsrc.precompute(ants, antpairs, proj_ref, data, inv_noise, gains, prior,
                                        omega_n=om_n, omega_a=om_a)
    A, b = ps.precompute_mpi(None, ants, antpairs, freqs, times, proj_out, data, inv_noise, gains,
                             prior, omega_n=om_n, omega_a=om_a)
    assert operr(A, A_ref) < OP_TOL and operr(b, b_ref) < OP_TOL
    x = hy.linear_solver.solve_dense(A, b)
    assert relerr(np.asarray(x), gcr_ptsrc.dense_solve(A_ref, b_ref)) < SAMPLE_TOL


# ------------------------------------------------------------------ CG on long vectors

@pytest.mark.parametrize("use_norm_tol", [False, True])
def test_cg_long_vector_step_matches_oracle(hy, use_norm_tol):
    """n >= 8192 takes the multi-workgroup CG step (csrc/solver.hip): same iterates as the
    oracle's restatement of linear_solver.cg, for a callable operator and for a dense matrix."""
    import torch
    from oracle import cg as ocg
    rng = np.random.RandomState(17)
    n, k = 20011, 6
    d = rng.uniform(1.0, 3.0, n)
    U = rng.randn(n, k) / np.sqrt(n)
    b = rng.randn(n)
    x0 = rng.randn(n) * 0.1
    tol = 1e-9 if not use_norm_tol else 1e-7

    def op_np(v):
        return d * v + U @ (U.T @ v)
    x_ref = ocg.cg(None, b, maxiters=500, abs_tol=tol, use_norm_tol=use_norm_tol, x0=x0.copy(),
                   linear_op=op_np)
    dd, Ud = torch.as_tensor(d).cuda(), torch.as_tensor(U).cuda()
    x, iters = hy.linear_solver.cg(None, torch.as_tensor(b).cuda(), maxiters=500, abs_tol=tol,
                                   use_norm_tol=use_norm_tol, x0=torch.as_tensor(x0).cuda(),
                                   linear_op=lambda v: dd * v + Ud @ (Ud.T @ v), return_iters=True)
    assert 2 < iters < 500
    assert relerr(x.cpu().numpy(), x_ref) < SAMPLE_TOL
    # dense matrix path (hydra_cg_solve_dense) just above the switch-over length
    n2 = 8200
    B = rng.randn(n2, 40) / np.sqrt(n2)
    A = np.diag(rng.uniform(1.0, 2.0, n2)) + B @ B.T
    b2 = rng.randn(n2)
    x2_ref = ocg.cg(A, b2, maxiters=300, abs_tol=tol, use_norm_tol=use_norm_tol)
    x2 = hy.linear_solver.cg(A, b2, maxiters=300, abs_tol=tol, use_norm_tol=use_norm_tol)
    assert relerr(x2, x2_ref) < SAMPLE_TOL
    assert np.abs(A @ x2 - b2).max() < 10 * tol


@pytest.mark.parametrize("n,nb", [(64, 1), (1, 2), (130, 3), (257, 2), (1000, 1), (1100, 2), (2049, 1),
                                  (4160, 1)])
def test_symmetric_half_read_matvec_matches_numpy(hy, n, nb):
    """hydra_symv_batched (reads the lower blocks only) == A @ x for symmetric A; plain and
    accumulating; odd n takes the 8-byte-load variant."""
    import torch
    from hydra_amd import _dev
    rng = np.random.RandomState(n + nb)
    A = rng.randn(nb, n, n)
    A = A + A.transpose(0, 2, 1)
    x = rng.randn(nb, n)
    y0 = rng.randn(nb, n)
    ref = np.einsum("qij,qj->qi", A, x)
    Ad, xd = _dev.f64(A), _dev.f64(x)
    ws = _dev.workspace(_dev._lib.query("hydra_symv_workspace_bytes", n, nb))
    for acc in (0, 1):
        yd = _dev.f64(y0).clone()
        _dev.call("hydra_symv_batched", _dev.ptr(Ad), n, nb, _dev.ptr(xd), _dev.ptr(yd), acc,
                  _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        want = ref + (y0 if acc else 0.0)
        assert relerr(yd.cpu().numpy(), want) < 1e-13
    # the upper triangle above the diagonal blocks is never read
    if n > 64:
        A2 = A.copy()
        A2[:, :64, 64:] = np.nan
        yd = torch.empty_like(xd)
        _dev.call("hydra_symv_batched", _dev.ptr(_dev.f64(A2)), n, nb, _dev.ptr(xd), _dev.ptr(yd), 0,
                  _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        assert relerr(yd.cpu().numpy(), ref) < 1e-13


# ------------------------------------------------------------------ beam-head consumers

def test_beam_head_helpers_match_reference_golden(hy):
    bs = hy.per_ant_beam_sampler
    g = load_golden("beam.npz")
    assert np.array_equal(bs.get_ant_inds(2, 5), g["keep"])
    bess = bs.get_bess_matr(g["nmodes"], g["mmodes"], g["rho"], g["phi"])
    assert np.array_equal(bess, g["bess"])
    assert np.array_equal(bs.get_bess_outer(bess), g["outer"])
    tr = bs.get_bess_to_vis_from_contraction(g["bsc"], g["coeffs"], 5, 2)
    assert isinstance(tr, np.ndarray) and tr.shape == g["trans"].shape
    assert relerr(tr, g["trans"]) < 1e-13
    tr = bs.get_bess_to_vis_from_contraction(g["bsc_ref"], g["coeffs"], 5, 2, ref_contraction=True)
    assert tr.shape == g["trans_ref"].shape and relerr(tr, g["trans_ref"]) < 1e-13
    sub = bs.select_subarr(g["bsc"], 2, 5)
    assert sub.shape == g["bsc"][:, :, :, :, g["keep"], 2].shape


@pytest.mark.parametrize("hex_spec,ns,nf,nt,nmax", [((2, 3), 23, 2, 3, 2), ((3, 4), 150, 3, 2, 3),
                                                   ((4, 7), 70, 1, 2, 4)])
def test_bess_sky_contraction_and_operator_match_oracle(hy, hex_spec, ns, nf, nt, nmax):
    from oracle import gcr_beam as ob, helpers
    bs = hy.per_ant_beam_sampler
    rng = np.random.RandomState(ns)
    ant_pos = helpers.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    np.random.seed(ns)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 2 * np.pi), (-1.4, 0.6))
    freqs = np.linspace(100.0, 120.0, nf)
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    lsts = np.linspace(0.2, 0.5, nt)
    nm = np.repeat(np.arange(1, nmax + 1), 3)
    mm = np.tile(np.array([0, 1, -1]), nmax)
    nc = nm.size
    bess = ob.get_bess_matr(nm, mm, rng.uniform(0, 1, (nt, ns)), rng.uniform(0, 6.28, (nt, ns)))
    outer = ob.get_bess_outer(bess)
    ref = ob.get_bess_sky_contraction(outer, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, latitude=LAT)
    out = bs.get_bess_sky_contraction(outer, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts,
                                      latitude=LAT)
    assert isinstance(out, np.ndarray) and out.shape == ref.shape == (1, 1, nf, nt, nant, nant, nc, nc)
    assert operr(out, ref) < OP_TOL
    # Hermitian in (antenna pair, basis pair); autos doubled
    assert relerr(out, out.transpose(0, 1, 2, 3, 5, 4, 7, 6).conj()) < 1e-13
    # power-beam form with a reference response
    rbr = rng.uniform(0.2, 1.0, (nf, nt, ns)) + 1j * rng.uniform(-0.1, 0.1, (nf, nt, ns))
    ref2 = ob.get_bess_sky_contraction(bess, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts,
                                       latitude=LAT, outer=False, ref_beam_response=rbr)
    out2 = bs.get_bess_sky_contraction(bess, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts,
                                       latitude=LAT, outer=False, ref_beam_response=rbr)
    assert out2.shape == ref2.shape and operr(out2, ref2) < OP_TOL
    # operator for one antenna, both routes
    coeffs = rng.randn(nc, nf, nant, 1, 1) + 1j * rng.randn(nc, nf, nant, 1, 1)
    samp = nant // 2
    tr_ref = ob.get_bess_to_vis_from_contraction(ref, coeffs, nant, samp)
    tr = bs.get_bess_to_vis_from_contraction(out, coeffs, nant, samp)
    assert operr(tr, tr_ref) < OP_TOL
    tr_direct = bs.get_bess_to_vis(bess, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, coeffs, samp,
                                   latitude=LAT)
    tr_direct_ref = ob.get_bess_to_vis(bess, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, coeffs,
                                       samp, latitude=LAT)
    assert tr_direct.shape == tr_direct_ref.shape == (1, 1, nf, nt, nant - 1, nc)
    assert operr(tr_direct, tr_direct_ref) < OP_TOL
    assert relerr(tr_direct, tr) < 1e-12
    with pytest.raises(NotImplementedError):
        bs.get_bess_sky_contraction(outer, ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, polarized=True)


# ------------------------------------------------------------------ PolyBeam

def test_polybeam_generator_matches_oracle(hy):
    """The Chebyshev beam of hydra/example.py:143-168 evaluated in the generator kernel against
    the oracle's numpy.chebval restatement (formula parity unpinned: hera_sim is not installed),
    plus the host-side compute_response used when beams of several kinds are mixed."""
    from oracle import beams as obeams, helpers, response
    coeffs = hy.example.HERA_POLYBEAM_COEFFS
    ant_pos = helpers.build_hex_array(hex_spec=(2, 3), d=14.6)
    nant = len(ant_pos)
    np.random.seed(9)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(80, (0.0, 2 * np.pi), (-1.5, 0.9))
    freqs = np.linspace(100.0, 200.0, 5)
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    lsts = np.linspace(0.1, 2.0, 4)
    ob = [obeams.PolyBeam(coeffs, spectral_index=-0.6975, ref_freq=1.0e8) for _ in range(nant)]
    hb = hy.example.example_beams("polybeam", nant)
    assert type(hb[0]).__name__ == "PolyBeam"
    ref = response.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, beams=ob,
                                           latitude=LAT)
    out = hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, lsts,
                                                   beams=hb, latitude=LAT)
    assert operr(out, ref) < OP_TOL
    za = np.linspace(0.0, 1.5, 50)
    assert relerr(hb[0].compute_response(az_array=0 * za, za_array=za, freq_array=freqs * 1e6),
                  ob[0].compute_response(az_array=0 * za, za_array=za, freq_array=freqs * 1e6)) < 1e-15
    assert abs(hb[0].compute_response(az_array=[0.0], za_array=[0.0], freq_array=[1.5e8])[0, 0, 0, 0]
               - 1.0) < 1e-14
    # mixed kinds go through the host value table
    mix = [hb[0]] + [hy.beams.GaussianBeam(diameter=14.0) for _ in range(nant - 1)]
    omix = [ob[0]] + [obeams.GaussianBeam(14.0) for _ in range(nant - 1)]
    ref = response.simulate_vis(ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, beams=omix, latitude=LAT)
    out = hy.vis_simulator.simulate_vis(ant_pos, fluxes, ra, dec, freqs * 1e6, lsts, beams=mix,
                                        latitude=LAT)
    assert operr(out, ref) < OP_TOL
    with pytest.raises(ValueError):
        hy.beams.PolyBeam(np.ones(40))


# ------------------------------------------------------------------ antenna-space adjoint

@pytest.mark.parametrize("nant,ns,nf,nt,kind", [(5, 37, 2, 3, "all"), (17, 100, 3, 2, "all"),
                                               (37, 250, 2, 5, "all"), (50, 70, 1, 3, "autos"),
                                               (64, 130, 2, 2, "dups"), (33, 64, 40, 7, "sparse"),
                                               (41, 90, 3, 3, "flipped"), (60, 50, 1, 5, "flipped")])
def test_antenna_space_adjoint_matches_pair_loop_and_numpy(hy, nant, ns, nf, nt, kind):
    """hydra_adjoint_V on the matrix cores (W = Zc V per (f,t)) == the pair-loop kernel == numpy,
    for complete arrays, lists with autos, repeated pairs, and sparse lists."""
    import torch
    from hydra_amd import _dev
    rng = np.random.RandomState(nant)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    if kind == "autos":
        pairs += [(i, i) for i in range(0, nant, 3)]
    elif kind == "dups":
        pairs += [pairs[k] for k in range(0, len(pairs), 17)] + [pairs[5], pairs[5]]
    elif kind == "sparse":
        pairs = [p for k, p in enumerate(pairs) if k % 5 == 1]
    elif kind == "flipped":   # pairs below the block diagonal: the kernel's all-blocks loop
        pairs = [(j, i) if k % 3 == 0 else (i, j) for k, (i, j) in enumerate(pairs)]
    bl_i = np.array([p[0] for p in pairs], dtype=np.int32)
    bl_j = np.array([p[1] for p in pairs], dtype=np.int32)
    nbl = len(pairs)
    V = rng.randn(nf, nt, nant, ns) + 1j * rng.randn(nf, nt, nant, ns)
    z = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    scale, add = rng.uniform(0.5, 2.0, ns), rng.randn(ns)
    R = np.conj(V[:, :, bl_i, :]) * V[:, :, bl_j, :]                 # (nf, nt, nbl, ns)
    ref = scale * np.einsum("ftbs,bft->s", np.conj(R), z).real + add
    Vd, zd = _dev.c128(V), _dev.c128(z)
    bid, bjd, sd, ad = _dev.i32(bl_i), _dev.i32(bl_j), _dev.f64(scale), _dev.f64(add)
    ws = _dev.workspace(_dev._lib.query("hydra_adjoint_V_workspace_bytes", nf, nt, nbl, ns))
    outs = []
    for enable in (1, 0):
        prev = _dev._lib.query("hydra_debug_adjoint_mfma", enable)
        try:
            out = _dev.empty((ns,), torch.float64)
            _dev.call("hydra_adjoint_V", _dev.ptr(Vd), nf, nt, nant, ns, _dev.ptr(bid), _dev.ptr(bjd),
                      nbl, _dev.ptr(zd), _dev.ptr(sd), _dev.ptr(ad), _dev.ptr(out), _dev.ptr(ws),
                      ws.numel(), _dev.stream_ptr())
            outs.append(out.cpu().numpy())
        finally:
            _dev._lib.query("hydra_debug_adjoint_mfma", prev)
    assert relerr(outs[0], ref) < 1e-12 and relerr(outs[1], ref) < 1e-12
    assert relerr(outs[0], outs[1]) < 1e-12


@pytest.mark.parametrize("nant,ns,nf,nt,kind", [(5, 37, 2, 3, "all"), (17, 100, 3, 2, "all"),
                                               (37, 250, 2, 5, "all"), (50, 70, 1, 3, "autos"),
                                               (64, 130, 2, 2, "dups"), (33, 64, 40, 7, "flipped"),
                                               (20, 1000, 1, 1, "all")])
def test_antenna_space_apply_matches_pair_loop_and_numpy(hy, nant, ns, nf, nt, kind):
    """hydra_apply_V on the matrix cores (C = conj(V) diag(x) V^T per (f,t) + gather) == the
    pair-loop kernel == numpy; x given or NULL (= ones), with and without the row factors; few
    (f,t) samples exercise the split over source ranges."""
    import torch
    from hydra_amd import _dev
    rng = np.random.RandomState(nant + 1)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    if kind == "autos":
        pairs += [(i, i) for i in range(0, nant, 3)]
    elif kind == "dups":
        pairs += [pairs[k] for k in range(0, len(pairs), 17)] + [pairs[5], pairs[5]]
    elif kind == "flipped":
        pairs = [(j, i) if k % 3 == 0 else (i, j) for k, (i, j) in enumerate(pairs) if k % 2 == 0]
    bl_i = np.array([p[0] for p in pairs], dtype=np.int32)
    bl_j = np.array([p[1] for p in pairs], dtype=np.int32)
    nbl = len(pairs)
    V = rng.randn(nf, nt, nant, ns) + 1j * rng.randn(nf, nt, nant, ns)
    x = rng.randn(ns)
    rowmul = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    R = np.conj(V[:, :, bl_i, :]) * V[:, :, bl_j, :]                 # (nf, nt, nbl, ns)
    Vd, bid, bjd = _dev.c128(V), _dev.i32(bl_i), _dev.i32(bl_j)
    ws = _dev.workspace(_dev._lib.query("hydra_apply_V_workspace_bytes", nf, nt, nant, ns))
    assert ws.numel() >= nf * nt * 16 * 16 * 16
    for xv, rm in [(x, None), (None, rowmul), (x, rowmul)]:
        ref = np.einsum("ftbs,s->bft", R, np.ones(ns) if xv is None else xv)
        if rm is not None:
            ref = ref * rm
        xd = None if xv is None else _dev.f64(xv)
        rd = None if rm is None else _dev.c128(rm)
        outs = []
        for enable in (1, 0):
            prev = _dev._lib.query("hydra_debug_adjoint_mfma", enable)
            try:
                out = _dev.empty((nbl, nf, nt), torch.complex128)
                _dev.call("hydra_apply_V", _dev.ptr(Vd), nf, nt, nant, ns, _dev.ptr(bid), _dev.ptr(bjd),
                          nbl, _dev.ptr(xd), _dev.ptr(rd), _dev.ptr(out), _dev.ptr(ws), ws.numel(),
                          _dev.stream_ptr())
                outs.append(out.cpu().numpy())
            finally:
                _dev._lib.query("hydra_debug_adjoint_mfma", prev)
        assert relerr(outs[0], ref) < 1e-12 and relerr(outs[1], ref) < 1e-12
    # no workspace: the pair loop serves the call
    out = _dev.empty((nbl, nf, nt), torch.complex128)
    _dev.call("hydra_apply_V", _dev.ptr(Vd), nf, nt, nant, ns, _dev.ptr(bid), _dev.ptr(bjd), nbl,
              _dev.ptr(_dev.f64(x)), _dev.ptr(None), _dev.ptr(out), _dev.ptr(None), 0,
              _dev.stream_ptr())
    assert relerr(out.cpu().numpy(), np.einsum("ftbs,s->bft", R, x)) < 1e-12


@pytest.mark.parametrize("hex_spec,ns,nf,nt", [((5, 9), 77, 3, 2), ((6, 11), 100, 2, 3),
                                               ((7, 13), 45, 1, 5), ((7, 14), 33, 2, 1),
                                               ((8, 15), 50, 2, 2), ((11, 21), 70, 1, 3)])
def test_antenna_space_adjoint_for_more_than_64_antennas(hex_spec, ns, nf, nt):
    """hydra_adjoint_V_antspace (W = Zc V on the matrix cores, any array size) against the pair
    loop and the oracle's Re(R^H z): 61, 91, 127 and 154 antennas (row blocks 4..10, odd and even
    counts, antennas not a multiple of 16), sources not a multiple of the 32-wide tile, scale / add;
    169 and 331 antennas: the super-row / super-block kernels beyond one LDS tile of all antennas
    (3 and 6 super-rows, the last one partial)."""
    import torch
    import hydra_amd as hy
    from oracle import beams as obeams, gcr_ptsrc, helpers
    np.random.seed(41)
    ant_pos = helpers.build_hex_array(hex_spec)
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs, times = np.linspace(100.0, 120.0, nf), np.linspace(0.2, 0.5, nt)
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh,
                                              materialize=False)
    assert fac.unique_upper
    rng = np.random.RandomState(5)
    nbl = len(antpairs)
    z = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    scale, add = rng.rand(ns) + 0.5, rng.randn(ns)
    zd = torch.as_tensor(z, device=fac.V.device)
    sd, ad = torch.as_tensor(scale, device=fac.V.device), torch.as_tensor(add, device=fac.V.device)
    fac.antspace_adjoint = False
    want = fac.adjoint(zd, scale=sd, add=ad).cpu().numpy()
    fac.antspace_adjoint = True
    got = fac.adjoint(zd, scale=sd, add=ad).cpu().numpy()
    if nant > 64:
        assert relerr(got, want) < 1e-12
    else:
        assert np.array_equal(got, want)          # 61 antennas: the same (<= 64) kernel either way
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    R = gcr_ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo).reshape(-1, ns)
    ref = scale * (R.real.T @ z.reshape(-1).real + R.imag.T @ z.reshape(-1).imag) + add
    assert operr(got, ref) < OP_TOL
    # a list with autos and missing pairs (still i <= j, unique)
    pairs2 = [(i, j) for i in range(nant) for j in range(i, nant) if (i + 2 * j) % 3 != 1]
    fac2 = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, pairs2, freqs, times, bh,
                                               materialize=False)
    z2 = rng.randn(len(pairs2), nf, nt) + 1j * rng.randn(len(pairs2), nf, nt)
    z2d = torch.as_tensor(z2, device=fac.V.device)
    fac2.antspace_adjoint = False
    want2 = fac2.adjoint(z2d).cpu().numpy()
    fac2.antspace_adjoint = True
    got2 = fac2.adjoint(z2d).cpu().numpy()
    assert relerr(got2, want2) < 1e-12

    # R x in antenna space (apply_V_big_kernel from 65 antennas on) against the pair loop and the
    # oracle's R @ x; x = None (ones), a row factor, and the list with autos / missing pairs
    from hydra_amd import _dev
    x = rng.randn(ns)
    xd = torch.as_tensor(x, device=fac.V.device)
    rm = torch.as_tensor(rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt), device=fac.V.device)

    def both(f, xx, rowmul=None):
        res = []
        for enable in (0, 1):
            prev = _dev._lib.query("hydra_debug_adjoint_mfma", enable)
            try:
                res.append(f.apply(xx, rowmul=rowmul).cpu().numpy())
            finally:
                _dev._lib.query("hydra_debug_adjoint_mfma", prev)
        return res
    pair, ant = both(fac, xd)
    assert relerr(ant, pair) < 1e-12
    assert operr(ant, (R @ x).reshape(nbl, nf, nt)) < OP_TOL
    pair, ant = both(fac, None, rowmul=rm)
    assert relerr(ant, pair) < 1e-12
    assert operr(ant, rm.cpu().numpy() * R.sum(axis=1).reshape(nbl, nf, nt)) < OP_TOL
    pair, ant = both(fac2, xd)
    assert relerr(ant, pair) < 1e-12


@pytest.mark.parametrize("nant,nft", [(380, 21), (430, 10), (9, 3)])
def test_streamed_gain_pairs_layouts(hy, nant, nft):
    """hydra_gain_pairs_stream in its other layouts -- 8 samples per workgroup with three LDS
    buffers (345-408 antennas), 4 samples per workgroup (beyond), a single sweep -- against the
    adjacency kernel on a pair list with holes (gain_sampler.py:278-284)."""
    import torch
    from hydra_amd import _dev, gain_sampler as gs
    rng = np.random.RandomState(nant)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    keep = rng.rand(len(pairs)) > 0.2
    pairs = [p for p, k in zip(pairs, keep) if k]
    nbl = len(pairs)
    i1 = np.array([p[0] for p in pairs], dtype=np.int32)
    i2 = np.array([p[1] for p in pairs], dtype=np.int32)
    idx = gs._Index(i1, i2, nant)
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(nant)
    mv = torch.randn((nbl, nft), dtype=torch.complex128, device=dev, generator=gen)
    w = 1.0 + torch.rand((nbl, nft), dtype=torch.float64, device=dev, generator=gen)
    g = torch.randn((1, nant, nft), dtype=torch.complex128, device=dev, generator=gen)
    st = _dev.stream_ptr()
    W = _dev.empty((nbl, nft), torch.float64)
    _dev.call("hydra_gain_weights", _dev.ptr(mv), _dev.ptr(w), nbl * nft, _dev.ptr(W), st)
    _, _, adj_ptr, adj_b, adj_o, _ = idx.dev()
    y_c = _dev.empty((1, nant, nft), torch.complex128)
    _dev.call("hydra_gain_pairs", _dev.ptr(g), 1, nant, nft, _dev.ptr(W), nbl,
              _dev.host_ptr(idx.adj_ptr), _dev.ptr(adj_ptr), _dev.ptr(adj_b), _dev.ptr(adj_o),
              int(idx.adj_b.size), _dev.ptr(y_c), st)
    count = _dev._lib.query("hydra_gain_stream_weight_count", nant, nft)
    assert count > 0
    Wt = _dev.empty((count,), torch.float64)
    _dev.call("hydra_gain_weights_stream", _dev.ptr(mv), _dev.ptr(w), nbl, nant, nft,
              _dev.ptr(idx.pair_dev()), _dev.ptr(Wt), st)
    ones = torch.ones((nant, nft), dtype=torch.complex128, device=dev)
    y1 = _dev.empty((nant, nft), torch.complex128)
    _dev.call("hydra_gain_pairs_stream", _dev.ptr(ones), nant, nft, _dev.ptr(Wt), _dev.ptr(None),
              _dev.ptr(y1), st)
    sw = y1.real.contiguous()
    y_s = _dev.empty((nant, nft), torch.complex128)
    _dev.call("hydra_gain_pairs_stream", _dev.ptr(g), nant, nft, _dev.ptr(Wt), _dev.ptr(sw),
              _dev.ptr(y_s), st)
    scale = float(y_c.abs().max())
    assert float((y_s - y_c[0]).abs().max()) <= 1e-12 * scale
    # every weight sits in exactly one slot of the stream layout
    assert abs(float(Wt.sum()) - float(W.sum())) <= 1e-12 * float(W.sum())
    assert _dev._lib.query("hydra_gain_stream_weight_count", 1800, nft) == 0


@pytest.mark.parametrize("nant,nft,nvec", [(9, 3, 1), (40, 10, 8), (177, 7, 5), (331, 9, 8),
                                           (360, 4, 11), (520, 2, 3)])
def test_matrix_core_batch_pair_kernel_matches_adjacency_kernel(hy, nant, nft, nvec):
    """hydra_gain_pairs_mfma (batch of fields: symmetric weight matrix of a sample x 2 nvec real
    columns on the FP64 matrix cores, weights in 16 x 16 antenna blocks) against the adjacency
    kernel and numpy on pair lists with holes (gain_sampler.py:278-284): 1, 2 and 3 block rows per
    wave, odd sample counts (a half-filled chunk), fewer and more than 8 vectors; the weights from
    a gain-applied model and from (model, gains)."""
    import torch
    from hydra_amd import _dev, gain_sampler as gs
    rng = np.random.RandomState(nant)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    keep = rng.rand(len(pairs)) > 0.2
    pairs = [p for p, k in zip(pairs, keep) if k]
    nbl = len(pairs)
    i1 = np.array([p[0] for p in pairs], dtype=np.int32)
    i2 = np.array([p[1] for p in pairs], dtype=np.int32)
    idx = gs._Index(i1, i2, nant)
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(nant)
    mv = torch.randn((nbl, nft), dtype=torch.complex128, device=dev, generator=gen)
    gains = torch.randn((nant, nft), dtype=torch.complex128, device=dev, generator=gen)
    w = 1.0 + torch.rand((nbl, nft), dtype=torch.float64, device=dev, generator=gen)
    g = torch.randn((nvec, nant, nft), dtype=torch.complex128, device=dev, generator=gen)
    st = _dev.stream_ptr()
    i1d, i2d = torch.as_tensor(i1, device=dev).long(), torch.as_tensor(i2, device=dev).long()
    ggv = (gains[i1d] * gains[i2d].conj() * mv).contiguous()
    W = _dev.empty((nbl, nft), torch.float64)
    _dev.call("hydra_gain_weights", _dev.ptr(ggv), _dev.ptr(w), nbl * nft, _dev.ptr(W), st)
    _, _, adj_ptr, adj_b, adj_o, _ = idx.dev()
    y_c = _dev.empty((nvec, nant, nft), torch.complex128)
    _dev.call("hydra_gain_pairs", _dev.ptr(g), nvec, nant, nft, _dev.ptr(W), nbl,
              _dev.host_ptr(idx.adj_ptr), _dev.ptr(adj_ptr), _dev.ptr(adj_b), _dev.ptr(adj_o),
              int(idx.adj_b.size), _dev.ptr(y_c), st)
    count = _dev._lib.query("hydra_gain_blocks_weight_count", nant, nft)
    assert count > 0
    scale = float(y_c.abs().max())
    for mvin, gin in ((ggv, None), (mv, gains)):
        Wb = torch.full((count,), float("nan"), dtype=torch.float64, device=dev)
        _dev.call("hydra_gain_weights_blocks", _dev.ptr(mvin), _dev.ptr(gin), _dev.ptr(w), nbl, nant,
                  nft, _dev.ptr(idx.pair_dev()), _dev.ptr(Wb), st)
        assert bool(torch.isfinite(Wb).all())             # every slot written
        # every pair sits in exactly two slots of an off-diagonal... one slot, or two inside a
        # diagonal block: the total is between once and twice the sum of the weights
        tot, once = float(Wb.sum()), float(W.sum())
        assert once * (1 - 1e-12) <= tot <= 2 * once * (1 + 1e-12)
        ones = torch.ones((1, nant, nft), dtype=torch.complex128, device=dev)
        y1 = _dev.empty((1, nant, nft), torch.complex128)
        _dev.call("hydra_gain_pairs_mfma", _dev.ptr(ones), 1, nant, nft, _dev.ptr(Wb), _dev.ptr(None),
                  _dev.ptr(y1), st)
        sw = y1[0].real.contiguous()
        sw_ref = torch.zeros((nant, nft), dtype=torch.float64, device=dev)
        sw_ref.index_add_(0, i1d, W)
        sw_ref.index_add_(0, i2d, W)
        assert float((sw - sw_ref).abs().max()) <= 1e-12 * float(sw_ref.abs().max())
        assert float(y1[0].imag.abs().max()) == 0.0
        y_m = torch.full((nvec, nant, nft), float("nan"), dtype=torch.complex128, device=dev)
        _dev.call("hydra_gain_pairs_mfma", _dev.ptr(g), nvec, nant, nft, _dev.ptr(Wb), _dev.ptr(sw),
                  _dev.ptr(y_m), st)
        assert float((y_m - y_c).abs().max()) <= 1e-12 * scale
    # numpy on one sample: y_a = g_a sum_b W_b + conj(sum_b W_b g_other)
    Wn, gn = W[:, 0].cpu().numpy(), g[0, :, 0].cpu().numpy()
    T = np.zeros(nant, dtype=complex)
    np.add.at(T, i1, Wn * gn[i2])
    np.add.at(T, i2, Wn * gn[i1])
    ref = gn * sw_ref[:, 0].cpu().numpy() + np.conj(T)
    assert np.abs(y_m[0, :, 0].cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    assert _dev._lib.query("hydra_gain_blocks_weight_count", 529, nft) == 0
    # the whole data term in one call (fields sample-major between the three kernels) against
    # basis transform -> adjacency pair kernel -> adjoint basis transform
    for nmodes in (5, 16, 21, 40):
        F = torch.randn((nmodes, nft), dtype=torch.complex128, device=dev, generator=gen)
        ps = 0.5 + torch.rand(nmodes, dtype=torch.float64, device=dev, generator=gen)
        n = 2 * nant * nmodes
        X = torch.randn((nvec, n), dtype=torch.float64, device=dev, generator=gen)
        add = torch.randn((nvec, n), dtype=torch.float64, device=dev, generator=gen)
        gf = _dev.empty((nvec, nant, nft), torch.complex128)
        _dev.call("hydra_gain_basis_fwd", _dev.ptr(X), nvec, nant, nmodes, nft, _dev.ptr(F), _dev.ptr(ps),
                  _dev.ptr(gf), st)
        yf = _dev.empty((nvec, nant, nft), torch.complex128)
        _dev.call("hydra_gain_pairs", _dev.ptr(gf), nvec, nant, nft, _dev.ptr(W), nbl,
                  _dev.host_ptr(idx.adj_ptr), _dev.ptr(adj_ptr), _dev.ptr(adj_b), _dev.ptr(adj_o),
                  int(idx.adj_b.size), _dev.ptr(yf), st)
        want = _dev.empty((nvec, n), torch.float64)
        _dev.call("hydra_gain_basis_adj", _dev.ptr(yf), nvec, nant, nmodes, nft, _dev.ptr(F), _dev.ptr(ps),
                  _dev.ptr(add), _dev.ptr(want), st)
        ws = _dev.workspace(_dev._lib.query("hydra_gain_operator_blocks_workspace_bytes", nant, nft))
        got = torch.full((nvec, n), float("nan"), dtype=torch.float64, device=dev)
        _dev.call("hydra_gain_operator_blocks", _dev.ptr(X), nvec, nant, nmodes, nft, _dev.ptr(F),
                  _dev.ptr(ps), _dev.ptr(Wb), _dev.ptr(sw), _dev.ptr(add), _dev.ptr(got), _dev.ptr(ws),
                  ws.numel(), st)
        assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max()), nmodes


@pytest.mark.parametrize("nant,nf,nt,holes,flip", [(12, 3, 5, False, False), (40, 4, 8, True, False),
                                                   (9, 1, 3, False, False), (380, 3, 7, True, False),
                                                   (14, 2, 5, False, True)])
def test_gain_prepare_sample_matches_the_unfused_sequence(hy, nant, nf, nt, holes, flip):
    """gain_sampler.prepare_sample (one pass over the visibilities: hydra_gain_prepare_stream)
    against the driver's sequence apply_gains -> residual -> construct_rhs_mpi -> GainOperator
    (example.py:690-745, gain_sampler.py:145-231, 278-284): same right-hand side, same stream
    weights, same diagonal term, same operator.  `flip` lists some baselines as (higher, lower):
    prepare_sample must then run the sequence itself."""
    import torch
    from hydra_amd import gain_sampler as gs
    rng = np.random.RandomState(100 + nant)
    pairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    if holes:
        pairs = [p for p in pairs if rng.rand() > 0.15]
    if flip:
        pairs = [(j, i) if k % 3 == 0 else (i, j) for k, (i, j) in enumerate(pairs)]
    nbl = len(pairs)
    ants = np.arange(nant)
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(nant)
    shape = (nbl, nf, nt)
    model = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    data = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    w = 0.5 + torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
    gains = 1.0 + 0.3 * torch.randn((nant, nf, nt), dtype=torch.complex128, device=dev, generator=gen)
    om_r = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    freqs, times = np.linspace(100., 120., nf), np.linspace(0.1, 0.4, nt)
    Fb, _, _ = hy.utils.partial_fourier_basis_2d_from_nmax(freqs, times, 1, 1, 20., 0.3)
    nmodes = Fb.shape[0]
    Fb = torch.as_tensor(np.ascontiguousarray(Fb), device=dev)
    ps = torch.as_tensor(0.1 + rng.rand(nmodes), device=dev)
    om_y = rng.randn(nant, nmodes) + 1j * rng.randn(nant, nmodes)
    A_real, A_imag = gs.proj_operator(ants, pairs)

    ggv = hy.utils.apply_gains(model, gains, ants, pairs, inline=False)
    b0 = gs.construct_rhs_mpi(None, data - ggv, w, ps, A_real, A_imag, ggv, Fb,
                              omega_y=om_y, omega_r=om_r)
    op0 = gs.GainOperator(w, ps, A_real, A_imag, ggv, Fb)
    b1, op1 = gs.prepare_sample(None, data, model, gains, w, ps, A_real, A_imag, Fb,
                                omega_y=om_y, omega_r=om_r)
    assert (op1._mv is None) == (not flip)            # the one-pass form ran unless told otherwise
    assert float((b1 - b0).abs().max()) <= 1e-12 * float(b0.abs().max())
    x = torch.as_tensor(rng.randn(op0.n), device=dev)
    y0, y1 = op0.apply(x), op1.apply(x)
    assert float((y1 - y0).abs().max()) <= 1e-12 * float(y0.abs().max())
    if op0.stream:
        assert float((op1.Wt - op0.Wt).abs().max()) <= 1e-13 * float(op0.Wt.abs().max())
        assert float((op1.sw - op0.sw).abs().max()) <= 1e-12 * float(op0.sw.abs().max())
    X = torch.as_tensor(rng.randn(3, op0.n), device=dev)     # batched form: by-baseline weights
    Y0, Y1 = op0.apply(X), op1.apply(X)
    assert float((Y1 - Y0).abs().max()) <= 1e-12 * float(Y0.abs().max())
    # numpy in, numpy out, and the reference's own draws when nothing is injected
    bn, _ = gs.prepare_sample(None, data.cpu().numpy(), model.cpu().numpy(), gains.cpu().numpy(),
                              w.cpu().numpy(), ps.cpu().numpy(), A_real, A_imag, Fb.cpu().numpy(),
                              seed=11)
    bs = gs.construct_rhs_mpi(None, (data - ggv).cpu().numpy(), w.cpu().numpy(), ps.cpu().numpy(),
                              A_real, A_imag, ggv.cpu().numpy(), Fb.cpu().numpy(), seed=11)
    assert isinstance(bn, np.ndarray)
    assert np.abs(bn - bs).max() <= 1e-12 * np.abs(bs).max()
