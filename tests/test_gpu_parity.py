"""
GPU parity tests: the HIP path (through the C ABI, via the hydra_amd API mirror) against the
CPU oracle on identical seeded inputs and injected fluctuation vectors, and against the golden
vectors produced by the reference itself (tests/golden/).

Tolerances (BASELINE.json north_star): operator entries <= 1e-10 relative (FP64); GCR samples
<= 1e-8 relative.  The operator checks use conftest.operr: max-norm relative error AND every entry within 1e-10 |b| + 1e-13 max|b|.
"""
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


@pytest.fixture(scope="module")
def orc():
    from oracle import beams, cg, gcr_ptsrc, gcr_sh, geometry, helpers, response, sphere

    class O:
        pass
    o = O()
    o.beams, o.cg, o.ptsrc, o.sh, o.geometry = beams, cg, gcr_ptsrc, gcr_sh, geometry
    o.helpers, o.response, o.sphere = helpers, response, sphere
    return o


def _pairs(arr):
    return [tuple(int(x) for x in p) for p in arr]


def _setup(hex_spec=(3, 4), ns=40, nf=3, nt=4, seed=0, wide=False):
    from oracle import helpers
    np.random.seed(seed)
    ant_pos = helpers.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    if wide:
        ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 2 * np.pi),
                                                                (-0.5 * np.pi, 0.5 * np.pi))
    else:
        ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    return ant_pos, antpairs, ra, dec, freqs, times, fluxes


# ------------------------------------------------------------------ response generator

def test_geometry_kernels_match_oracle(hy, orc):
    import torch
    from hydra_amd import _dev, vis_simulator as vs
    rng = np.random.RandomState(1)
    ra, dec = rng.uniform(0, 2 * np.pi, 300), rng.uniform(-1.5, 1.5, 300)
    lsts = np.linspace(0.0, 6.0, 7)
    crd_eq, eq2tops = vs._geometry(ra, dec, lsts, LAT)
    assert relerr(crd_eq.cpu().numpy(), orc.geometry.point_source_crd_eq(ra, dec)) < 1e-15
    ref = np.array([orc.geometry.eci_to_enu_matrix(t, LAT) for t in lsts])
    assert relerr(eq2tops.cpu().numpy(), ref) < 1e-15
    az = _dev.empty((7, 300), torch.float64)
    za = _dev.empty((7, 300), torch.float64)
    up = _dev.empty((7, 300), torch.uint8)
    _dev.call("hydra_az_za", _dev.ptr(eq2tops), 7, _dev.ptr(crd_eq), 300, _dev.ptr(az),
              _dev.ptr(za), _dev.ptr(up), _dev.stream_ptr())
    for t in range(7):
        top = ref[t] @ orc.geometry.point_source_crd_eq(ra, dec)
        a, z = orc.geometry.enu_to_az_za(top[0], top[1], "uvbeam")
        d = np.abs(az[t].cpu().numpy() - a)
        assert np.max(np.minimum(d, 2 * np.pi - d)) < 1e-9     # az is ill-conditioned at zenith
        assert np.max(np.abs(za[t].cpu().numpy() - z)) < 1e-10   # sqrt(1 - l^2) cancels at the horizon
        assert np.array_equal(up[t].cpu().numpy().astype(bool), top[2] > 0)


def test_per_source_cube_matches_oracle(hy, orc):
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(ns=70, nf=3, nt=4, wide=True)
    times = np.linspace(0.0, 3.0, 4)
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    ref = orc.response.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times, bo,
                                               latitude=LAT)
    got = hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times,
                                                   bh, latitude=LAT)
    assert got.shape == ref.shape and got.dtype == np.complex128
    assert np.count_nonzero(ref) > 0 and np.any(ref == 0)       # sources above and below horizon
    assert operr(got, ref) < OP_TOL
    # lower triangle left at zero, autos populated (vis_simulator.py:281-289)
    assert np.all(got[:, :, 3, 1] == 0) and np.any(got[:, :, 2, 2] != 0)
    # sum over sources (simulate_vis) and the subarr_ant variant
    tot = hy.vis_simulator.simulate_vis(ant_pos, fluxes, ra, dec, freqs * 1e6, times, bh,
                                        latitude=LAT)
    assert operr(tot, ref.sum(axis=-1)) < OP_TOL
    sub = hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times,
                                                   bh, latitude=LAT, subarr_ant=2)
    subref = orc.response.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times,
                                                  bo, latitude=LAT, subarr_ant=2)
    assert operr(sub, subref) < OP_TOL


def test_vis_sim_per_source_single_channel_and_beam_variants(hy, orc):
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(ns=33, nf=2, nt=3)
    antpos = np.array([ant_pos[k] for k in ant_pos])
    crd_eq = orc.geometry.point_source_crd_eq(ra, dec)
    eq2tops = np.array([orc.geometry.eci_to_enu_matrix(t, LAT) for t in times])
    # one shared beam / fixed-sigma beams with per-antenna widths / uniform beam
    cases = [
        ([orc.beams.GaussianBeam(diameter=12.0)], [hy.beams.GaussianBeam(diameter=12.0)]),
        ([orc.beams.GaussianBeam(sigma=0.2 + 0.01 * k, reference_frequency=1e8,
                                 spectral_index=-1.0) for k in range(len(ant_pos))],
         [hy.beams.GaussianBeam(sigma=0.2 + 0.01 * k, reference_frequency=1e8,
                                spectral_index=-1.0) for k in range(len(ant_pos))]),
        ([orc.beams.UniformBeam()], [hy.beams.UniformBeam()]),
    ]
    for bo, bh in cases:
        ref = orc.response.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0], bo)
        got = hy.vis_simulator.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0], bh)
        assert got.shape == (times.size, len(ant_pos), len(ant_pos), ra.size)
        assert operr(got, ref) < OP_TOL
    # a beam object without an in-kernel form goes through compute_response on the host
    class HostBeam(orc.beams.GaussianBeam):
        pass
    hb = [HostBeam(diameter=14.0)]
    ref = orc.response.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0], hb)
    got = hy.vis_simulator.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0], hb)
    assert operr(got, ref) < OP_TOL
    # force_no_beam_sqrt
    ref = orc.response.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0],
                                          cases[0][0], force_no_beam_sqrt=True)
    got = hy.vis_simulator.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0],
                                              cases[0][1], force_no_beam_sqrt=True)
    assert operr(got, ref) < OP_TOL
    # error behaviour of the reference: wrong beam count, NaN beam, shape asserts
    with pytest.raises(ValueError):
        hy.vis_simulator.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0],
                                            [hy.beams.UniformBeam()] * 2)

    class NanBeam:
        def compute_response(self, *, az_array, za_array, freq_array, **kw):
            return np.full((1, 1, np.size(freq_array), np.size(za_array)), np.nan)
    with pytest.raises(ValueError, match="invalid value"):
        hy.vis_simulator.vis_sim_per_source(antpos, 1.1e8, eq2tops, crd_eq, fluxes[:, 0],
                                            [NanBeam()])
    with pytest.raises(AssertionError):
        hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes[:, :1], ra, dec, freqs * 1e6,
                                                 times, cases[0][1] * len(ant_pos))
    with pytest.warns(RuntimeWarning):
        hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra - 1.0, dec, freqs * 1e6,
                                                 times, cases[0][1] * len(ant_pos))


def test_calc_proj_operator_matches_oracle(hy, orc):
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(ns=50, nf=4, nt=5)
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    ref = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
    got = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh)
    assert got.shape == (45, 4, 5, 50)
    assert operr(got, ref) < OP_TOL
    # a pair listed against dict order picks up the zero lower triangle, as in the reference
    ap2 = [(1, 0)] + antpairs[1:]
    got2 = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, ap2, freqs, times, bh)
    ref2 = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, ap2, freqs, times, bo)
    assert np.all(got2[0] == 0) and operr(got2, ref2) < OP_TOL
    # reference smoke test (hydra/tests/test_ptsrc_sampler.py): 3 antennas, 17 sources
    np.random.seed(3)
    ant3 = {0: (0.0, 0.0, 0.0), 1: (0.0, 14.0, 0.0), 2: (0.0, 0.0, 14.0)}
    ap3 = [(0, 1), (0, 2), (1, 2)]
    lsts = np.linspace(0.0, 3.0, 5)
    fr = np.linspace(100.0, 200.0, 10)
    ra3, dec3, amps3 = hy.example.generate_random_ptsrc_catalogue(
        17, (0.0, 2 * np.pi), (-0.5 * np.pi, 0.5 * np.pi), (-1.0, 2.0))
    fl3 = hy.utils.get_flux_from_ptsrc_amp(amps3, fr, -2.7)
    proj = hy.ptsrc_sampler.calc_proj_operator(ra3, dec3, fl3, ant3, ap3, fr, lsts,
                                               [hy.beams.GaussianBeam(diameter=14.0)] * 3)
    assert np.all(~np.isnan(proj))
    ref3 = orc.ptsrc.calc_proj_operator(ra3, dec3, fl3, ant3, ap3, fr, lsts,
                                        [orc.beams.GaussianBeam(diameter=14.0)] * 3)
    assert operr(proj, ref3) < OP_TOL
    data = (proj.reshape((-1, 17)) @ np.ones(17)).reshape(3, 10, 5)
    A, b = hy.ptsrc_sampler.precompute_mpi(
        comm=None, ants=list(ant3.keys()), antpairs=ap3, freq_chunk=fr, time_chunk=lsts,
        proj_chunk=proj, data_chunk=data, inv_noise_var_chunk=np.ones(data.shape),
        gain_chunk=(1.1 + 1.0j) * np.ones((3, 10, 5)), amp_prior_std=0.1 * np.ones(17),
        realisation=True)
    assert np.all(~np.isnan(A)) and np.all(~np.isnan(b))


# ------------------------------------------------------------------ GCR operator and RHS

@pytest.mark.parametrize("name", ["ptsrc_small.npz", "ptsrc_c1like.npz"])
def test_precompute_matches_reference_golden(hy, name):
    g = load_golden(name)
    ap = _pairs(g["antpairs"])
    nf, nt = g["data"].shape[1:]
    A, b = hy.ptsrc_sampler.precompute_mpi(
        None, g["ants"], ap, np.arange(nf), np.arange(nt), g["proj"], g["data"], g["inv_noise"],
        g["gains"], g["prior"], realisation=True, omega_n=g["omega_n"], omega_a=g["omega_a"])
    assert operr(A, g["A"]) < OP_TOL
    assert operr(b, g["b"]) < OP_TOL
    assert np.array_equal(A, A.T)
    # drawing omega from the global RNG in the reference's order reproduces the reference
    np.random.seed(int(g["seed"]))
    A2, b2 = hy.ptsrc_sampler.precompute_mpi(
        None, g["ants"], ap, np.arange(nf), np.arange(nt), g["proj"], g["data"], g["inv_noise"],
        g["gains"], g["prior"], realisation=True)
    assert operr(b2, g["b"]) < OP_TOL
    A0, b0 = hy.ptsrc_sampler.precompute_mpi(
        None, g["ants"], ap, np.arange(nf), np.arange(nt), g["proj"], g["data"], g["inv_noise"],
        g["gains"], g["prior"], realisation=False)
    assert operr(A0, g["A0"]) < OP_TOL and operr(b0, g["b0"]) < OP_TOL
    # samples: CG and the dense fallback against the reference's dense solve of its own system
    import scipy.linalg
    xref = scipy.linalg.solve(g["A"], g["b"], assume_a="her")
    x = hy.linear_solver.cg(A, b, maxiters=1000, abs_tol=1e-10)
    assert relerr(x, xref) < SAMPLE_TOL
    assert np.linalg.norm(g["A"] @ x - g["b"]) / np.linalg.norm(g["b"]) < SAMPLE_TOL
    xd = hy.linear_solver.solve_dense(A, b)
    assert relerr(xd, xref) < SAMPLE_TOL


def _gcr_inputs(nbl, nant, nf, nt, ns, seed, uniform_noise=False):
    rng = np.random.RandomState(seed)
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, 1, 1) + 1j * rng.randn(nant, 1, 1)) \
        + 0.01 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    inv_noise = np.full((nbl, nf, nt), 400.0) if uniform_noise else \
        rng.uniform(100.0, 900.0, (nbl, nf, nt))
    if not uniform_noise:
        inv_noise[rng.randint(nbl), :, 0] = 0.0
    prior = rng.uniform(0.05, 0.2, ns)
    om_n = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    om_a = rng.randn(ns)
    return gains, inv_noise, prior, om_n, om_a


@pytest.mark.parametrize("hex_spec,ns,nf,nt", [((3, 4), 100, 4, 5), ((4, 7), 150, 2, 3),
                                               ((5, 9), 200, 2, 2), ((3, 4), 100, 16, 20),
                                               ((3, 4), 60, 2, 3), ((3, 4), 208, 3, 2),
                                               ((3, 4), 209, 2, 2), ((2, 3), 150, 2, 2),
                                               ((2, 3), 300, 1, 3)])
def test_factored_system_matches_oracle_and_materialised(hy, orc, hex_spec, ns, nf, nt):
    """End to end on the simulator's own operator: fused (factored) path == materialised HIP
    path == oracle, for one-segment (10, 37 antennas) and multi-segment (61 antennas) tables;
    Ns not a multiple of the 128 tile, odd row counts.  The last case is BASELINE config 1 at
    its exact size: hex (3,4) = 10 antennas / 45 baselines, 100 sources, 16 freqs, 20 times.
    Last row band of the tile grid: 60 = one 80-row tile on the diagonal, 150 / 200 / 208 = an
    80-row band under full tiles (22, 72, exactly 80 rows; one and two wave groups), 209 = 81
    rows, a full tile again.  hex (2,3) = 7 antennas: the staged voltages alone (28 KiB) are smaller
    than the 32 KiB the diagonal tiles' hand-over needs; 150 sources there = no full tile at all,
    300 = full, diagonal and short-band tiles in one launch."""
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(hex_spec, ns, nf, nt, seed=5)
    nant, nbl = len(ant_pos), len(antpairs)
    ants = np.array(list(ant_pos.keys()))
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    proj_ref = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
    gains, inv_noise, prior, om_n, om_a = _gcr_inputs(nbl, nant, nf, nt, ns, seed=7)
    truth = 1.0 + 0.1 * np.random.RandomState(8).randn(ns)
    data = orc.helpers.apply_gains((proj_ref.reshape(-1, ns) @ truth).reshape(nbl, nf, nt),
                                   gains, ants, antpairs)
    data = data + 0.05 * om_n[::-1]
    A_ref, b_ref = orc.ptsrc.precompute(ants, antpairs, proj_ref, data, inv_noise, gains, prior,
                                        omega_n=om_n, omega_a=om_a)
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh,
                                              materialize=False)
    assert fac.shape == proj_ref.shape
    if nant > 37:
        assert fac.seg_host.shape[0] > 1
    R = fac.materialize().cpu().numpy()
    assert operr(R, proj_ref) < OP_TOL
    args = (None, ants, antpairs, freqs, times)
    kw = dict(data_chunk=data, inv_noise_var_chunk=inv_noise, gain_chunk=gains,
              amp_prior_std=prior, omega_n=om_n, omega_a=om_a)
    A_f, b_f = hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=fac, **kw)
    assert fac.last_gram_kernel == "gram_fused_kernel"      # per-baseline weights: general kernel
    A_m, b_m = hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=R, **kw)
    for A, b in ((A_f, b_f), (A_m, b_m)):
        assert operr(A, A_ref) < OP_TOL
        assert operr(b, b_ref) < OP_TOL
    # model update (example.py:974-976) in both forms
    mod_ref = (proj_ref.reshape(-1, ns) @ truth).reshape(nbl, nf, nt)
    assert operr(hy.ptsrc_sampler.apply_proj(R, truth), mod_ref) < OP_TOL
    assert operr(hy.ptsrc_sampler.apply_proj(fac, truth).cpu().numpy(), mod_ref) < OP_TOL
    # the sample
    x_ref = orc.ptsrc.dense_solve(A_ref, b_ref)
    x = hy.linear_solver.cg(A_f, b_f, maxiters=2000, abs_tol=1e-9)
    assert relerr(x, x_ref) < SAMPLE_TOL
    assert np.linalg.norm(A_ref @ x - b_ref) / np.linalg.norm(b_ref) < SAMPLE_TOL


def test_gram_kernel_shapes_and_symmetry(hy, orc):
    """hydra_gram alone on random (asymmetric) operators: tile edges (Ns = 1, 127, 128, 129,
    300), row counts that do not divide the split, with and without row factors."""
    import torch
    from hydra_amd import _dev
    rng = np.random.RandomState(11)
    for m, ns, with_rows in [(5, 1, True), (1000, 127, True), (777, 128, False),
                             (4099, 129, True), (2500, 300, True)]:
        R = rng.randn(m, ns) + 1j * rng.randn(m, ns)
        rows = rng.randn(m) + 1j * rng.randn(m) if with_rows else None
        Rp = R * rows[:, None] if with_rows else R
        ref = Rp.real.T @ Rp.real + Rp.imag.T @ Rp.imag
        Rd = _dev.c128(R)
        rd = _dev.c128(rows) if with_rows else None
        ws = _dev.workspace(_dev._lib.query("hydra_gram_workspace_bytes", m, ns))
        G = _dev.empty((ns, ns), torch.float64)
        _dev.call("hydra_gram", _dev.ptr(Rd), _dev.ptr(rd), m, ns, _dev.ptr(None), _dev.ptr(None),
                  _dev.ptr(G), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        Gh = G.cpu().numpy()
        assert relerr(Gh, ref) < 1e-12, (m, ns)
        assert np.array_equal(Gh, Gh.T)
        # bitwise reproducible (fixed-order slab reduction, no atomics)
        G2 = _dev.empty((ns, ns), torch.float64)
        _dev.call("hydra_gram", _dev.ptr(Rd), _dev.ptr(rd), m, ns, _dev.ptr(None), _dev.ptr(None),
                  _dev.ptr(G2), _dev.ptr(ws), ws.numel(), _dev.stream_ptr())
        assert torch.equal(G, G2)


# ------------------------------------------------------------------ solver

def test_cg_matches_reference_golden(hy, orc):
    g = load_golden("cg.npz")
    # hydra/tests/test_linear_solver.py:9-54 on the GPU path
    n = 401
    A1 = np.eye(n)
    x1 = hy.linear_solver.cg(A1, A1 @ g["xtrue"], maxiters=1000, abs_tol=1e-8)
    assert np.allclose(g["xtrue"], x1)
    x2, it2 = hy.linear_solver.cg(g["A2"], g["b2"], maxiters=1000, abs_tol=1e-8,
                                  return_iters=True)
    assert np.allclose(g["xtrue"], x2)
    # both runs stop at |r_i| < 1e-8: they agree to a few times that, and each meets the
    # 1e-8 relative-residual gate
    assert relerr(x2, g["x2"]) < 5 * SAMPLE_TOL
    assert np.linalg.norm(g["A2"] @ x2 - g["b2"]) / np.linalg.norm(g["b2"]) < SAMPLE_TOL
    _, it_ref = orc.cg.cg(g["A2"], g["b2"], return_iters=True)
    # same algorithm; this matrix needs > n iterations, i.e. the trajectory is set by rounding, and
    # the products / dot products are summed in another order than numpy's
    assert abs(it2 - it_ref) <= 0.03 * it_ref
    x2a = hy.linear_solver.cg(None, g["b2"], maxiters=1000, abs_tol=1e-8,
                              linear_op=lambda v: g["A2"] @ v)
    assert np.allclose(g["xtrue"], x2a) and np.allclose(x2, x2a)
    assert relerr(hy.linear_solver.cg(g["A2"], g["b2"], abs_tol=1e-6, use_norm_tol=True),
                  g["x2n"]) < 1e-6
    # maxiters is honoured exactly, no convergence flag is raised
    x5 = hy.linear_solver.cg(g["A2"], g["b2"], maxiters=5)
    assert relerr(x5, g["x2_5"]) < 1e-10
    x3 = hy.linear_solver.cg(g["A3"], g["b3"], abs_tol=1e-10, x0=g["x03"].copy())
    assert relerr(x3, g["x3"]) < SAMPLE_TOL
    with pytest.raises(AssertionError):
        hy.linear_solver.cg(g["A3"], g["b3"], x0=np.zeros(5))
    # already-converged start: zero iterations, x0 returned
    xs = np.linalg.solve(g["A3"], g["b3"])
    xc, itc = hy.linear_solver.cg(g["A3"], g["b3"], abs_tol=1e-6, x0=xs.copy(), return_iters=True)
    assert itc == 0 and np.array_equal(xc, xs)
    # torch (device-resident) operator path
    import torch
    Ad = torch.as_tensor(g["A3"]).cuda()
    xt = hy.linear_solver.cg(None, torch.as_tensor(g["b3"]).cuda(), abs_tol=1e-10,
                             linear_op=lambda v: Ad @ v)
    assert xt.is_cuda and relerr(xt.cpu().numpy(), xs) < SAMPLE_TOL


def test_dense_fallback_matches_scipy(hy):
    import scipy.linalg
    rng = np.random.RandomState(2)
    # (from 1024 unknowns on the sweeps run one launch per 256-wide panel -- even n only -- and
    # the factorisation one launch per 32 columns: sizes on both sides of those edges, a last
    # panel of 2 and of 254 entries, an odd size that keeps the one-workgroup sweeps)
    for n in (1, 31, 32, 33, 100, 257, 1022, 1024, 1026, 1279, 1280, 1534, 2000, 2050):
        B = rng.randn(2 * n + 3, n)
        A = np.eye(n) + 0.1 * (B.T @ B)
        b = rng.randn(n)
        x = hy.linear_solver.solve_dense(A, b)
        xref = scipy.linalg.solve(A, b, assume_a="her")
        assert relerr(x, xref) < 1e-10, n
    with pytest.raises(np.linalg.LinAlgError):
        hy.linear_solver.solve_dense(-np.eye(4), np.ones(4))


@pytest.mark.parametrize("ns,factored_proj", [(40, True), (40, False), (6146, True)])
def test_system_cache_reuses_operator_and_factor_bit_for_bit(hy, orc, ns, factored_proj):
    """precompute_mpi(cache=) + cache.solve: while gains / noise weights / prior are unchanged the
    operator and its Cholesky factor are reused (the reference rebuilds them every iteration,
    example.py:878,944) and every sample is BIT-identical to the rebuilding path with the same
    fluctuation vectors; a change of the gains rebuilds.  6146 sources: the matrix-core
    factorisation (n >= 6144), whose in-place factor leaves the operator readable above the
    diagonal (cache.matvec)."""
    import torch
    from hydra_amd import _dev, linear_solver, ptsrc_sampler
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(ns=ns, nf=2, nt=3)
    nant, nbl = len(ant_pos), len(antpairs)
    ants = np.arange(nant)
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    proj = hy.ptsrc_sampler.calc_proj_operator(ra, dec, _dev.f64(fluxes), ant_pos, antpairs, freqs,
                                               times, bh, materialize=not factored_proj)
    rng = np.random.RandomState(3)
    shape = (nbl, 2, 3)
    g = _dev.c128((1.0 + 0.5j) + 0.1 * (rng.randn(nant, 2, 3) + 1j * rng.randn(nant, 2, 3)))
    w = _dev.f64(np.full(shape, 4.0))
    data = _dev.c128(rng.randn(*shape) + 1j * rng.randn(*shape))
    prior = _dev.f64(0.1 + 0.05 * rng.rand(ns))

    def omegas(k):
        r = np.random.RandomState(100 + k)
        return (_dev.c128((r.randn(*shape) + 1j * r.randn(*shape)) / np.sqrt(2)), _dev.f64(r.randn(ns)))

    def rebuild(gains, k):
        om_n, om_a = omegas(k)
        A, b = ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, proj, data, w, gains,
                                            prior, omega_n=om_n, omega_a=om_a)
        return A, b, linear_solver.solve_dense(A, b)

    cache = ptsrc_sampler.SystemCache()

    def cached(gains, k):
        om_n, om_a = omegas(k)
        A, b = ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, proj, data, w, gains,
                                            prior, omega_n=om_n, omega_a=om_a, cache=cache)
        return A, b, cache.solve(b)

    for k in range(3):                                   # one build, then two reuses
        A0, b0, x0 = rebuild(g, k)
        A1, b1, x1 = cached(g.clone(), k)                # a NEW tensor with the same values
        assert torch.equal(b0, b1) and torch.equal(x0, x1), k
        assert relerr(cache.matvec(x1).cpu().numpy(), (A0 @ x0).cpu().numpy()) < 1e-13
    assert (cache.misses, cache.hits) == (1, 2) and cache.factored
    # the upper triangle of the cached operator is still the operator
    assert torch.equal(torch.triu(A1, 1), torch.triu(A0, 1))
    assert not getattr(A1, "hydra_symmetric", False)
    g2 = g * 1.01                                        # the gain head moved: rebuild
    A0, b0, x0 = rebuild(g2, 7)
    A1, b1, x1 = cached(g2, 7)
    assert (cache.misses, cache.hits) == (2, 2)
    assert torch.equal(torch.triu(A0, 1), torch.triu(A1, 1)) and torch.equal(cache.diag, A0.diagonal())
    assert torch.equal(b0, b1) and torch.equal(x0, x1)
    w2 = w.clone()
    w2[1, 0, 2] = 0.0                                    # a flag appeared: rebuild
    om_n, om_a = omegas(9)
    ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, proj, data, w2, g2, prior,
                                 omega_n=om_n, omega_a=om_a, cache=cache)
    assert (cache.misses, cache.hits) == (3, 2)
    # against the oracle once (the cache must not have changed what is computed)
    if ns == 40:
        bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
        R = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
        A_ref, b_ref = orc.ptsrc.precompute(ants, antpairs, R, data.cpu().numpy(), w.cpu().numpy(),
                                            g2.cpu().numpy(), prior.cpu().numpy(),
                                            omega_n=omegas(7)[0].cpu().numpy(),
                                            omega_a=omegas(7)[1].cpu().numpy())
        assert operr(b0.cpu().numpy(), b_ref) < OP_TOL
        assert relerr(x0.cpu().numpy(), orc.ptsrc.dense_solve(A_ref, b_ref)) < SAMPLE_TOL


# ------------------------------------------------------------------ gains helper

def test_operator_cg_reads_the_flag_on_a_learned_schedule(hy):
    """cg(linear_op=) enqueues iterations in blocks between reads of the device-side converged
    flag; steps past convergence are no-ops but their operator applications are real work.  The
    second solve of a system of the same size starts with a block of the first one's iteration
    count: same iterates bit for bit, at most a couple of surplus applications."""
    import torch
    from hydra_amd import _dev, linear_solver
    rng = np.random.RandomState(12)
    n = 300
    M = rng.randn(n, n)
    A = _dev.f64(M @ M.T / n + np.diag(1.0 + rng.rand(n)))
    b = _dev.f64(rng.randn(n))
    calls = [0]

    def op(v):
        calls[0] += 1
        return A @ v

    linear_solver._ITER_HISTORY.clear()
    x1, it1 = linear_solver.cg(None, b, linear_op=op, abs_tol=1e-10, return_iters=True)
    c1 = calls[0]
    calls[0] = 0
    x2, it2 = linear_solver.cg(None, b, linear_op=op, abs_tol=1e-10, return_iters=True)
    c2 = calls[0]
    assert it1 == it2 and 0 < it1 < 1000 and torch.equal(x1, x2)
    assert it1 <= c1 <= it1 + linear_solver.CHECK_EVERY       # blocks of CHECK_EVERY, no history
    assert c2 == it2                                          # the learned first block is exact here
    # a harder right-hand side of the same size: the first block falls short, short blocks follow
    b3 = _dev.f64(rng.randn(n) * 1e3)
    calls[0] = 0
    x3, it3 = linear_solver.cg(None, b3, linear_op=op, abs_tol=1e-10, return_iters=True)
    assert it3 >= it2 and it3 <= calls[0] <= it3 + max(2, it2 // 16)
    assert relerr(x3.cpu().numpy(), np.linalg.solve(A.cpu().numpy(), b3.cpu().numpy())) < 1e-9


@pytest.mark.parametrize("n", [1, 33, 257, 1681, 6145])
def test_cho_factor_then_solves_match_scipy(hy, n):
    """Factor once, solve many (the SH head's constant left-hand side): every right-hand side
    against scipy's dense solve, and the factor-only call leaves what hydra_chol_apply reads
    (6145: the matrix-core path from 6144 on, padded to even size)."""
    import scipy.linalg
    rng = np.random.RandomState(n)
    M = rng.randn(n, n)
    A = M @ M.T + n * np.eye(n)
    fac = hy.linear_solver.cho_factor(A)
    assert (fac.linv is not None) == (n <= hy.linear_solver.INVERSE_FACTOR_MAX_N)
    sub = hy.linear_solver.cho_factor(A, inverse_factor=False)      # the substitution path
    for k in range(3):
        b = rng.randn(n)
        want = scipy.linalg.solve(A, b, assume_a="pos")
        assert relerr(hy.linear_solver.cho_solve(fac, b), want) < 1e-11
        assert relerr(hy.linear_solver.cho_solve(sub, b), want) < 1e-11
    with pytest.raises(np.linalg.LinAlgError):
        hy.linear_solver.cho_factor(A - 2 * np.diag(np.diag(A)))


@pytest.mark.parametrize("n", [1, 5, 32, 33, 100, 1000, 1681, 4096])
def test_triangular_inverse_kernel(hy, n):
    """hydra_tri_inverse (the inverse of the Cholesky factor that cho_factor keeps for small
    systems): X L = 1 for the lower triangle of a well-conditioned factor, zeros above the
    diagonal, nothing above the diagonal of L read (it holds NaN here); a padded row stride."""
    import torch
    from hydra_amd import _dev
    rng = np.random.RandomState(n)
    L = np.tril(rng.randn(n, n)) / np.sqrt(n) + np.diag(1.0 + rng.rand(n))
    ld = n + 3
    Ld = np.full((n, ld), np.nan)
    Ld[:, :n] = np.where(np.tril(np.ones((n, n), dtype=bool)), L, np.nan)
    Lg = torch.as_tensor(Ld, device="cuda")
    X = torch.full((n, n), float("nan"), dtype=torch.float64, device="cuda")
    _dev.call("hydra_tri_inverse", _dev.ptr(Lg), ld, n, _dev.ptr(X), n, _dev.stream_ptr())
    Xh = X.cpu().numpy()
    assert np.all(np.isfinite(Xh))
    assert np.all(np.triu(Xh, 1) == 0.0)
    assert np.max(np.abs(Xh @ L - np.eye(n))) <= 1e-12 * np.linalg.cond(L)
    assert relerr(Xh, np.linalg.inv(L)) <= 1e-12 * np.linalg.cond(L)


@pytest.mark.parametrize("n,size", [(1, 1), (7, 1), (300, 1), (1000, 3), (2306, 2)])
def test_symmetric_product_from_the_upper_triangle(hy, n, size):
    """hydra_symv_upper (SystemCache.matvec after an in-place factorisation): A x from the strict
    upper triangle and the saved diagonal, for all rows and for the block-cyclic rows of each of
    ``size`` ranks (summed here); what lies on and below the diagonal is NaN and must not be read."""
    import torch
    from hydra_amd import _dev, _dist
    rng = np.random.RandomState(n)
    M = rng.randn(n, n)
    A = M + M.T
    x = rng.randn(n)
    want = A @ x
    upper = np.where(np.triu(np.ones((n, n), dtype=bool), 1), A, np.nan)
    total = np.zeros(n)
    for r in range(size):
        rows = _dist.RowLayout(n, size, r, block=256).rows()
        if rows.size == 0:
            continue
        # (named tensors: a temporary handed to _dev.ptr() is freed, and its block reused by the
        # next allocation, before the kernel runs)
        Ad = torch.as_tensor(upper[rows], device="cuda").contiguous()
        rows_d = torch.as_tensor(rows, dtype=torch.int64, device="cuda")
        diag_d = torch.as_tensor(np.diag(A)[rows].copy(), device="cuda")
        x_d = torch.as_tensor(x, device="cuda")
        y = torch.zeros(n, dtype=torch.float64, device="cuda")
        _dev.call("hydra_symv_upper", _dev.ptr(Ad), n, rows.size, n, _dev.ptr(rows_d),
                  _dev.ptr(diag_d), _dev.ptr(x_d), _dev.ptr(y), _dev.stream_ptr())
        torch.cuda.synchronize()
        total += y.cpu().numpy()
    assert np.all(np.isfinite(total))
    assert relerr(total, want) < 1e-13


def test_apply_gains_matches_reference_golden(hy):
    g = load_golden("gains.npz")
    ap = _pairs(g["antpairs"])
    out = hy.utils.apply_gains(g["v"], g["g"], g["ants"], ap)
    assert relerr(out, g["out_plain"]) < 1e-14
    out = hy.utils.apply_gains(g["v"], g["g"], g["ants"], ap, perturbation=g["pert"])
    assert relerr(out, g["out_pert"]) < 1e-14
    v = g["v"].copy()
    r = hy.utils.apply_gains(v, g["g"], g["ants"], ap, inline=True)
    assert r is v and relerr(v, g["out_plain"]) < 1e-14
    assert np.array_equal(hy.utils.extract_vis_from_sim(g["ants"], ap, g["simvis"]),
                          g["extracted"])


# ------------------------------------------------------------------ spherical harmonics

def test_healpix_and_mode_maps_match_oracle(hy, orc):
    for nside in (1, 2, 8, 16):
        th, ph = hy.vis_simulator.healpix_pixel_coords(nside)
        tr, pr = orc.sphere.pix2ang_ring(nside, np.arange(12 * nside * nside))
        assert np.max(np.abs(th.cpu().numpy() - tr)) < 1e-14
        assert np.max(np.abs(ph.cpu().numpy() - pr)) < 1e-14
    for lmax, nside in [(0, 1), (3, 2), (10, 8), (40, 8)]:
        npix = 12 * nside * nside
        ref = orc.sphere.unit_mode_maps(lmax, nside) * (4 * np.pi / npix)
        got = hy.vis_simulator.sh_mode_maps(lmax, nside, 1.0, packed=False).cpu().numpy()
        assert got.shape == (npix, ref.shape[0])
        assert relerr(got.T, ref) < 1e-11
        nalm = ref.shape[0] // 2
        keep = np.concatenate((np.arange(nalm), np.arange(nalm + lmax + 1, 2 * nalm)))
        gp = hy.vis_simulator.sh_mode_maps(lmax, nside, 2.0, packed=True).cpu().numpy()
        assert relerr(gp.T, 2.0 * ref[keep]) < 1e-11
    ell, m = hy.vis_simulator.getlm(7)
    er, mr = orc.sphere.getlm(7)
    assert np.array_equal(ell, er) and np.array_equal(m, mr)


def test_sh_response_operator_matches_oracle(hy, orc):
    # hydra/tests/test_sh_sampler.py:21-67 geometry; lmax/nside reduced for the CPU oracle
    ant_pos = {0: (0.0, 0.0, 0.0), 1: (0.0, 14.0, 0.0), 2: (0.0, 0.0, 14.0), 3: (20.0, 5.0, 1.0)}
    lsts = np.linspace(0.0, 3.0, 3)
    freqs = np.linspace(100.0, 200.0, 2)
    lmax, nside = 5, 4
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    pr, ar, ell_r, m_r = orc.sh.vis_proj_operator_no_rot(freqs, lsts, bo, ant_pos, lmax, nside,
                                                         spectral_idx=-2.5)
    pg, ag, ell, m = hy.sh_sampler.vis_proj_operator_no_rot(freqs, lsts, bh, ant_pos, lmax, nside,
                                                            spectral_idx=-2.5)
    ems, ells, idx = hy.sh_sampler.get_em_ell_idx(lmax)
    assert pg.shape == (6 * freqs.size * lsts.size, len(ems)) and ag.shape[0] == 4 * 6
    assert np.array_equal(ell, ell_r) and np.array_equal(m, m_r)
    assert operr(pg, pr) < OP_TOL and operr(ag, ar) < OP_TOL
    assert not np.allclose(pg.imag, 0)
    # include_autos / autos_only variants return 3-tuples
    p2, _, _ = hy.sh_sampler.vis_proj_operator_no_rot(freqs, lsts, bh, ant_pos, lmax, nside,
                                                      include_autos=True)
    r2, _, _ = orc.sh.vis_proj_operator_no_rot(freqs, lsts, bo, ant_pos, lmax, nside,
                                               include_autos=True)
    assert p2.shape == r2.shape and operr(p2, r2) < OP_TOL
    # the raw simulate_vis_per_alm cube (2*nalm modes, zero lower triangle)
    e3, m3, cube = hy.vis_simulator.simulate_vis_per_alm(lmax, nside, ant_pos, freqs * 1e6, lsts, bh,
                                                         latitude=-0.5361913261514378)
    e4, m4, cube_r = orc.response.simulate_vis_per_alm(lmax, nside, ant_pos, freqs * 1e6, lsts, bo,
                                                       latitude=-0.5361913261514378)
    e5, m5, cube5 = hy.vis_simulator.simulate_vis_per_region(lmax, nside, ant_pos, freqs * 1e6, lsts,
                                                             bh, latitude=LAT)
    assert np.array_equal(e5, e3) and np.array_equal(m5, m3) and np.array_equal(cube5, cube)
    assert cube.shape == cube_r.shape and operr(cube, cube_r) < OP_TOL


def test_sh_algebra_matches_reference_golden(hy, orc):
    g = load_golden("sh_algebra.npz")
    lmax = int(g["lmax"])
    ems, ells, idx = hy.sh_sampler.get_em_ell_idx(lmax)
    assert np.array_equal(ems, g["ems"]) and np.array_equal(ells, g["ells"])
    rhs = hy.sh_sampler.construct_rhs_no_rot(g["data"].flatten(), g["inv_noise"].flatten(),
                                             g["inv_prior"], g["omega_a"], g["omega_n"].flatten(),
                                             g["a0"], g["R"])
    assert operr(rhs, g["rhs"]) < OP_TOL
    rhs2 = hy.sh_sampler.construct_rhs_no_rot_mpi(None, g["data"], g["inv_noise"], g["inv_prior"],
                                                  g["omega_a"].copy(), g["omega_n"], g["a0"], g["R"])
    assert operr(rhs2, g["rhs_mpi"]) < OP_TOL
    lhs = hy.sh_sampler.apply_lhs_no_rot(g["acr"], g["inv_noise"].flatten(), g["inv_prior"], g["R"])
    assert operr(lhs, g["lhs"]) < OP_TOL
    lhs2 = hy.sh_sampler.apply_lhs_no_rot_mpi(None, g["acr"].copy(), g["inv_noise"],
                                              g["inv_prior"], g["R"])
    assert operr(lhs2, g["lhs_mpi"]) < OP_TOL
    assert np.array_equal(hy.sh_sampler.alms2healpy(g["packed"], lmax), g["hp_modes"])
    assert np.array_equal(hy.sh_sampler.healpy2alms(g["hp_modes"]), g["packed"])
    # explicit operator (Gram once) reproduces the matrix-free apply
    A = hy.sh_sampler.lhs_matrix(g["inv_noise"], g["inv_prior"], g["R"])
    assert operr(A @ g["acr"], g["lhs"]) < OP_TOL
    # the GCR sample: CG with the matrix-free operator (the driver's call, example.py:1045-1052)
    # and with the explicit matrix, against the reference's own CG solution
    x = hy.linear_solver.cg(
        Amat=None, bvec=rhs2, maxiters=1000, abs_tol=1e-8, use_norm_tol=False, x0=g["a0"].copy(),
        linear_op=lambda v: hy.sh_sampler.apply_lhs_no_rot_mpi(None, v, g["inv_noise"],
                                                               g["inv_prior"], g["R"]))
    assert relerr(x, g["x_cg"]) < SAMPLE_TOL
    xA = hy.linear_solver.cg(A, rhs2, maxiters=1000, abs_tol=1e-8, x0=g["a0"].copy())
    assert relerr(xA, g["x_cg"]) < SAMPLE_TOL


# ------------------------------------------------------------------ gain head (config 4)

def test_gain_head_matches_reference_golden(hy):
    """proj_operator / apply_proj / apply_proj_conj / apply_operator_mpi / construct_rhs_mpi
    against the outputs of the reference's own functions (tests/golden/gain.npz)."""
    from oracle import gcr_gain
    gs = hy.gain_sampler
    g = load_golden("gain.npz")
    ap = _pairs(g["antpairs"])
    A_real, A_imag = gs.proj_operator(g["ants"], ap)
    assert np.array_equal(np.asarray(A_real.todense()), g["A_real"])
    assert np.array_equal(np.asarray(A_imag.todense()), g["A_imag"])
    assert operr(gs.apply_proj(g["x"], A_real, A_imag, g["mv"]), g["vproj"]) < OP_TOL
    assert operr(gs.apply_proj_conj(g["vv"], A_real, A_imag, g["mv"], g["x"].shape),
                  g["gconj"]) < OP_TOL
    # projectors handed over as plain dense arrays (not made by proj_operator) work too
    assert operr(gs.apply_proj(g["x"], g["A_real"], g["A_imag"], g["mv"]), g["vproj"]) < OP_TOL
    opx = gs.apply_operator_mpi(None, g["xvec"], g["inv_noise"], g["ps_sqrt"], A_real, A_imag,
                                g["mv"], g["Fb"])
    assert operr(opx, g["opx"]) < OP_TOL
    # the reference's own draws (seeded inside construct_rhs_mpi) ...
    b = gs.construct_rhs_mpi(None, g["resid"], g["inv_noise"], g["ps_sqrt"], A_real, A_imag,
                             g["mv"], g["Fb"], realisation=True, seed=int(g["seed"]))
    assert operr(b, g["bvec"]) < OP_TOL
    # ... and the same vectors injected
    nant, nmodes = g["x"].shape[0], g["Fb"].shape[0]
    om_y, om_r = gcr_gain.draw_gain_omegas(int(g["seed"]), nant, nmodes, g["resid"].shape)
    b2 = gs.construct_rhs_mpi(None, g["resid"], g["inv_noise"], g["ps_sqrt"], A_real, A_imag,
                              g["mv"], g["Fb"], omega_y=om_y, omega_r=om_r)
    assert operr(b2, g["bvec"]) < OP_TOL


@pytest.mark.parametrize("hex_spec,nf,nt,nmax,nvec", [((3, 4), 7, 5, 2, 3), ((4, 7), 16, 12, 3, 9)])
def test_gain_operator_batched_matches_oracle_and_solves(hy, orc, hex_spec, nf, nt, nmax, nvec):
    """GainOperator (device-resident weights, batched vectors) against the oracle's restatement
    of apply_operator_mpi, with subsets of baselines and non-uniform noise; then a CG solve of
    the gain system through linear_solver.cg(linear_op=...) as the driver does
    (example.py:719-728), checked against a dense solve of the oracle operator."""
    from oracle import gcr_gain
    gs = hy.gain_sampler
    rng = np.random.RandomState(42)
    ant_pos = orc.helpers.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    antpairs = [p for k, p in enumerate(antpairs) if k % 7 != 3]          # some baselines flagged out
    nbl = len(antpairs)
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    Fb, kf, kt = orc.helpers.partial_fourier_basis_2d_from_nmax(
        freqs, times, nmax, nmax, freqs[-1] - freqs[0], times[-1] - times[0])
    nmodes = Fb.shape[0]
    mv = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    w = 50.0 + 400.0 * rng.rand(nbl, nf, nt)
    ps = 0.05 + 0.1 * rng.rand(nmodes)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    i1, i2 = gcr_gain.bl_indices(ants, antpairs)
    n = 2 * nant * nmodes
    X = rng.randn(nvec, n)
    op = gs.GainOperator(w, ps, A_real, A_imag, mv, Fb)
    Y = op.apply(X)
    for v in range(nvec):
        ref = gcr_gain.apply_operator(X[v], w, ps, i1, i2, mv, Fb)
        assert operr(Y[v], ref) < OP_TOL
    assert op.stream                                       # single vectors: streamed pair kernel
    assert relerr(op.apply(X[0]), Y[0]) < 1e-13
    ref0 = gcr_gain.apply_operator(X[0], w, ps, i1, i2, mv, Fb)
    assert operr(op.apply(X[0]), ref0) < OP_TOL
    op.stream = False
    assert relerr(op.apply(X[0]), Y[0]) < 1e-14                             # single-vector path
    op.stream = True
    op.tiled = False                                                        # CSR adjacency kernel
    assert relerr(op.apply(X), Y) < 1e-13
    op.tiled = True
    assert relerr(gs.apply_operator_mpi(None, X[1], w, ps, A_real, A_imag, mv, Fb), Y[1]) < 1e-14

    resid = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    om_y, om_r = gcr_gain.draw_gain_omegas(5, nant, nmodes, resid.shape)
    b_ref = gcr_gain.construct_rhs(resid, w, ps, i1, i2, mv, Fb, om_y, om_r)
    b = gs.construct_rhs_mpi(None, resid, w, ps, A_real, A_imag, mv, Fb, seed=5)
    assert operr(b, b_ref) < OP_TOL
    x = hy.linear_solver.cg(None, b, maxiters=2000, abs_tol=1e-11, use_norm_tol=True,
                            linear_op=op.apply)
    Aref = np.column_stack([gcr_gain.apply_operator(e, w, ps, i1, i2, mv, Fb) for e in np.eye(n)])
    x_ref = np.linalg.solve(Aref, b_ref)
    assert relerr(x, x_ref) < SAMPLE_TOL


# ------------------------------------------------------------------ closed-form (antenna-space) Gram

def test_closed_form_gram_beyond_65535_time_frequency_samples(hy):
    """More (f,t) samples than a grid dimension holds (700 x 100): the closed-form kernel still
    runs (the pre-scale kernel strides over (f,t)) and agrees with the general kernel."""
    nant, ns, nf, nt = 4, 40, 700, 100
    rng = np.random.RandomState(5)
    ant_pos = {a: np.array([rng.uniform(-80, 80), rng.uniform(-80, 80), 0.0]) for a in range(nant)}
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    np.random.seed(3)
    ra, dec, amps = hy.example.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    fluxes = hy.utils.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    gains = (1.0 + 0.5j) + 0.2 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    inv_noise = np.broadcast_to(100.0 + 400.0 * rng.rand(nf, nt), (nbl, nf, nt)).copy()
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    om_n = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    om_a = rng.randn(ns)
    prior = 0.05 + 0.1 * rng.rand(ns)
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in range(nant)]
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, beams,
                                              materialize=False)
    out = {}
    for uniform in (True, False):
        fac.allow_uniform = uniform
        out[uniform] = hy.ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, fac, data,
                                                       inv_noise, gains, prior, omega_n=om_n,
                                                       omega_a=om_a)
        assert fac.last_gram_kernel == ("gram_uniform_kernel" if uniform else "gram_fused_kernel")
    assert relerr(out[True][0], out[False][0]) < 1e-10
    assert relerr(out[True][1], out[False][1]) < 1e-12


@pytest.mark.parametrize("nant,ns,nf,nt,mixed", [(10, 100, 3, 4, False), (37, 300, 2, 3, False),
                                                 (37, 300, 2, 3, True), (39, 140, 1, 2, False),
                                                 (40, 140, 1, 2, False), (45, 130, 2, 2, False),
                                                 (45, 130, 2, 2, True), (64, 257, 1, 3, False)])
def test_uniform_weight_gram_matches_oracle_and_general_kernel(hy, orc, nant, ns, nf, nt, mixed):
    """Complete array + noise weights independent of the baseline: precompute_mpi takes the
    closed-form kernel (gram_uniform_kernel).  It must agree with the oracle's
    ptsrc_sampler.precompute restatement and with the general per-baseline kernel, with gains
    varying over (antenna, f, t) and weights varying over (f, t).  nant = 45 / 64 exercise an
    odd / even number of 20-antenna LDS phases, nant = 39 / 40 the position of the same-beam
    row (last slot of a k-step / a k-step and LDS phase of its own), ns = 130 / 257 the edge
    tiles; ``mixed`` = two beam models, i.e. the general per-antenna q term."""
    rng = np.random.RandomState(nant)
    ant_pos = {a: np.array([rng.uniform(-150, 150), rng.uniform(-150, 150), 0.0])
               for a in range(nant)}
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    rng.shuffle(antpairs)                       # completeness does not depend on the order
    antpairs = [tuple(p) for p in antpairs]
    nbl = len(antpairs)
    np.random.seed(3)
    ra, dec, amps = orc.helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.linspace(100.0, 120.0, max(nf, 2))[:nf]
    times = np.linspace(0.2, 0.5, max(nt, 2))[:nt]
    fluxes = orc.helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    gains = (1.0 + 0.5j) + 0.2 * (rng.randn(nant, nf, nt) + 1j * rng.randn(nant, nf, nt))
    w_ft = 100.0 + 400.0 * rng.rand(nf, nt)
    inv_noise = np.broadcast_to(w_ft, (nbl, nf, nt)).copy()
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    om_n = (rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)) / np.sqrt(2)
    om_a = rng.randn(ns)
    prior = 0.05 + 0.1 * rng.rand(ns)

    # one beam model for the whole array (rank-one q term) or two different ones (general q term)
    diam = [14.0 if (not mixed or a % 3) else 11.0 for a in range(nant)]
    bo = [orc.beams.GaussianBeam(diameter=d) for d in diam]
    proj_ref = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
    A_ref, b_ref = orc.ptsrc.precompute(ants, antpairs, proj_ref, data, inv_noise, gains, prior,
                                        omega_n=om_n, omega_a=om_a)
    bh = [hy.beams.GaussianBeam(diameter=d) for d in diam]
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh,
                                              materialize=False)
    assert fac.complete and fac.same_beam == (not mixed)
    A_u, b_u = hy.ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, fac, data,
                                               inv_noise, gains, prior, omega_n=om_n, omega_a=om_a)
    assert fac.last_gram_kernel == "gram_uniform_kernel"
    assert operr(A_u, A_ref) < OP_TOL and operr(b_u, b_ref) < OP_TOL
    assert np.array_equal(A_u, A_u.T)
    fac.allow_uniform = False
    A_g, _ = hy.ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, fac, data,
                                             inv_noise, gains, prior, omega_n=om_n, omega_a=om_a)
    assert fac.last_gram_kernel == "gram_fused_kernel"
    assert relerr(A_u, A_g) < 1e-12
    # one baseline with a different weight: the closed form takes its one-row correction
    fac.allow_uniform = True
    inv_noise[3, 0, 0] *= 0.5
    A_ref2, _ = orc.ptsrc.precompute(ants, antpairs, proj_ref, data, inv_noise, gains, prior,
                                     omega_n=om_n, omega_a=om_a)
    A_2, _ = hy.ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, fac, data,
                                             inv_noise, gains, prior, omega_n=om_n, omega_a=om_a)
    assert fac.last_gram_kernel == "gram_uniform_kernel" and fac.last_gram_flagged_rows == 1
    assert operr(A_2, A_ref2) < OP_TOL
    # an incomplete array (one pair missing) never takes it
    fac2 = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs[:-1], freqs,
                                               times, bh, materialize=False)
    assert not fac2.complete


# ------------------------------------------------------------------ gridded (spline) beams

def _gridded_beam_pair(hy, orc, nfreq=5):
    za = np.linspace(0.0, np.pi / 2, 46)
    az = np.arange(0.0, 360.0, 4.0) * np.pi / 180.0
    freqs = np.linspace(95e6, 125e6, nfreq)
    rng = np.random.RandomState(2)
    nu = (freqs - 100e6) / 20e6
    grid = (np.exp(-(za[None, :, None] / (0.25 + 0.05 * nu[:, None, None])) ** 2)
            * (1.0 + 0.2 * np.cos(2 * az[None, None, :]) * np.sin(za[None, :, None]) ** 2)
            + 0.02 + 1e-4 * rng.rand(nfreq, za.size, az.size))
    return (hy.beams.GriddedBeam(freqs, za, az, grid), orc.beams.SplineBeam(freqs, za, az, grid))


def test_gridded_beam_device_spline_matches_scipy(hy, orc):
    """hydra_beam_bspline (de Boor on the GPU) against scipy's RectBivariateSpline.ev -- the
    routine pyuvdata's az_za_simple interpolation calls -- incl. points on knots, on the domain
    edges and outside the domain (clamped), and cubic interpolation in frequency."""
    bh, bo = _gridded_beam_pair(hy, orc)
    rng = np.random.RandomState(4)
    zq = np.concatenate([rng.uniform(0.0, np.pi / 2, 3000), bo.za_grid[::5], [0.0, np.pi / 2],
                         [-0.01, np.pi / 2 + 0.2]])
    aq = np.concatenate([rng.uniform(0.0, 2 * np.pi, 3000), bo.az_grid[:zq.size - 3000]])
    aq = aq[:zq.size]
    fq = np.array([95e6, 101.7e6, 118.2e6, 125e6])
    got = bh.compute_response(az_array=aq, za_array=zq, freq_array=fq)
    want = bo.compute_response(az_array=aq, za_array=zq, freq_array=fq)
    assert got.shape == want.shape == (1, 1, fq.size, zq.size)
    assert relerr(got, want) < 1e-13


def test_simulator_with_gridded_beams_matches_oracle(hy, orc):
    """Per-source response and point-source GCR system with one shared gridded beam (evaluated
    once on the GPU) and with a mix of gridded and analytic beams (value-table path)."""
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup((3, 4), 60, 3, 4, seed=9, wide=True)
    nant = len(ant_pos)
    bh, bo = _gridded_beam_pair(hy, orc)
    for mixed in (False, True):
        beams_h = [bh] * nant
        beams_o = [bo] * nant
        if mixed:
            beams_h = [bh if a % 2 else hy.beams.GaussianBeam(diameter=14.0) for a in range(nant)]
            beams_o = [bo if a % 2 else orc.beams.GaussianBeam(diameter=14.0) for a in range(nant)]
        ref = orc.response.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times,
                                                   beams_o, latitude=LAT)
        got = hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6,
                                                       times, beams_h, latitude=LAT)
        assert operr(got, ref) < OP_TOL


def _healpix_beam_pair(hy, orc, nside=16, nfreq=3):
    th, ph = orc.sphere.pix2ang_ring(nside, np.arange(12 * nside * nside))
    freqs = np.linspace(95e6, 125e6, nfreq)
    nu = (freqs - 100e6) / 20e6
    rng = np.random.RandomState(8)
    maps = (np.exp(-(th[None, :] / (0.3 + 0.05 * nu[:, None])) ** 2)
            * (1.0 + 0.2 * np.cos(2 * ph[None, :]) * np.sin(th[None, :]) ** 2)
            + 0.02 + 1e-4 * rng.rand(nfreq, th.size))
    return hy.beams.HealpixBeam(freqs, maps), orc.beams.HealpixBeam(freqs, maps)


def test_healpix_beam_device_interpolation_matches_oracle(hy, orc):
    """hydra_beam_healpix (pyuvdata "healpix_simple": four-pixel bilinear weights, linear in
    frequency) against the oracle's restatement at random directions, pixel centres, both poles,
    ring boundaries and phi = 0 / 2 pi; several resolutions."""
    rng = np.random.RandomState(12)
    for nside in (1, 2, 16, 64):
        bh, bo = _healpix_beam_pair(hy, orc, nside=nside)
        tc, pc = orc.sphere.pix2ang_ring(nside, np.arange(min(12 * nside * nside, 500)))
        zq = np.concatenate([np.arccos(rng.uniform(-1, 1, 4000)), tc, tc + 1e-12, tc - 1e-12,
                             [0.0, np.pi, 1e-9, np.pi - 1e-9, 0.7, 0.7, 0.7]])
        aq = np.concatenate([rng.uniform(-np.pi, 3 * np.pi, 4000), pc, pc, pc,
                             [0.3, 1.0, 2.0, 4.0, 0.0, 2 * np.pi, 2 * np.pi - 1e-14]])
        fq = np.array([95e6, 101.7e6, 110e6, 125e6])
        got = bh.compute_response(az_array=aq, za_array=zq, freq_array=fq)
        want = bo.compute_response(az_array=aq, za_array=zq, freq_array=fq)
        assert got.shape == want.shape == (1, 1, fq.size, zq.size)
        assert operr(got, want) < 1e-12
    with pytest.raises(ValueError):
        bh.compute_response(az_array=aq[:3], za_array=zq[:3], freq_array=[90e6])


def test_simulator_with_healpix_beams_matches_oracle(hy, orc):
    """Per-source response with HEALPix-tabulated beams (shared, and mixed with analytic ones)."""
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup((3, 4), 60, 3, 4, seed=9, wide=True)
    nant = len(ant_pos)
    bh, bo = _healpix_beam_pair(hy, orc)
    for mixed in (False, True):
        beams_h = [bh] * nant
        beams_o = [bo] * nant
        if mixed:
            beams_h = [bh if a % 2 else hy.beams.GaussianBeam(diameter=14.0) for a in range(nant)]
            beams_o = [bo if a % 2 else orc.beams.GaussianBeam(diameter=14.0) for a in range(nant)]
        ref = orc.response.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6, times,
                                                   beams_o, latitude=LAT)
        got = hy.vis_simulator.simulate_vis_per_source(ant_pos, fluxes, ra, dec, freqs * 1e6,
                                                       times, beams_h, latitude=LAT)
        assert operr(got, ref) < OP_TOL


def test_large_array_factored_operator_matches_oracle(hy, orc):
    """170 antennas: all antennas x 64 sources no longer fit in LDS, so R x and R^H z use the
    16-source staging; the Gram walks 9 x 9 antenna-block segments.  Factored system vs oracle."""
    nant, ns, nf, nt = 170, 48, 1, 2
    rng = np.random.RandomState(170)
    ant_pos = {a: np.array([rng.uniform(-300, 300), rng.uniform(-300, 300), 0.0])
               for a in range(nant)}
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    np.random.seed(5)
    ra, dec, amps = orc.helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4))
    freqs = np.array([110.0])
    times = np.array([0.25, 0.4])
    fluxes = orc.helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    gains, inv_noise, prior, om_n, om_a = _gcr_inputs(nbl, nant, nf, nt, ns, seed=3)
    data = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    proj_ref = orc.ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
    A_ref, b_ref = orc.ptsrc.precompute(ants, antpairs, proj_ref, data, inv_noise, gains, prior,
                                        omega_n=om_n, omega_a=om_a)
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh,
                                              materialize=False)
    assert fac.seg_host.shape[0] == 45
    A, b = hy.ptsrc_sampler.precompute_mpi(None, ants, antpairs, freqs, times, fac, data,
                                           inv_noise, gains, prior, omega_n=om_n, omega_a=om_a)
    assert operr(A, A_ref) < OP_TOL and operr(b, b_ref) < OP_TOL
    truth = 1.0 + 0.1 * rng.randn(ns)
    mod_ref = (proj_ref.reshape(-1, ns) @ truth).reshape(nbl, nf, nt)
    assert operr(hy.ptsrc_sampler.apply_proj(fac, truth).cpu().numpy(), mod_ref) < OP_TOL


def test_tiled_sh_projection_matches_untiled_kernel(hy):
    """hydra_sh_project_tiled (antenna-block tiles, horizon-compacted pixel lists, LDS-DMA) against
    the plain kernel: arbitrary pair list (autos, both orientations, gaps), 37 antennas (partial
    8 x 16 blocks), Nm not a multiple of 128, and the frequency axis filled block by block."""
    import torch
    vs = hy.vis_simulator
    ant_pos = hy.utils.build_hex_array(hex_spec=(4, 7), d=14.6)
    nant = len(ant_pos)
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    freqs = np.array([100e6, 110e6, 120e6])
    lsts = np.array([0.2, 1.7, 4.0])
    lmax, nside = 12, 8
    V = vs.sh_voltages(nside, ant_pos, freqs, lsts, beams)
    counts = V.hydra_npix_t.cpu().numpy()
    assert np.all(counts > 0) and np.all(counts < 12 * nside ** 2)
    up = (V[0, :, 0, :].abs() > 0).sum(dim=1).cpu().numpy()
    assert np.array_equal(counts, up)
    ymap = vs.sh_mode_maps(lmax, nside, 1.0, packed=True)
    rng = np.random.RandomState(1)
    pairs = [(i, j) for i in range(nant) for j in range(nant)]
    pairs = [pairs[k] for k in rng.permutation(len(pairs))[:500]]
    bi = np.array([p[0] for p in pairs], dtype=np.int32)
    bj = np.array([p[1] for p in pairs], dtype=np.int32)
    scale = np.array([1.0, 0.8, 1.3])
    tiled = vs.sh_project(V, bi, bj, ymap, scale_f=scale)
    Vp = V.clone()                                   # no horizon lists attached -> plain kernel
    plain = vs.sh_project(Vp, bi, bj, ymap, scale_f=scale)
    scale_ref = torch.max(torch.abs(plain)).item()
    assert torch.max(torch.abs(tiled - plain)).item() < 1e-12 * scale_ref
    # block by block along frequency into one output array
    out = torch.zeros_like(tiled)
    for a, b in ((0, 2), (2, 3)):
        Vb = V[a:b].contiguous()
        Vb.hydra_pixidx, Vb.hydra_npix_t = V.hydra_pixidx, V.hydra_npix_t
        vs.sh_project(Vb, bi, bj, ymap, scale_f=scale[a:b], out=out, f0=a, nf_total=3)
    assert torch.equal(out, tiled)


def test_tiled_sh_projection_matches_oracle_on_full_antenna_blocks(hy, orc):
    """The tiled projection kernel (16-antenna blocks) against the ORACLE on whole blocks: all 666
    cross pairs of a 37-antenna array = block pairs (0,0), (0,1), (0,2), (1,1), (1,2), (2,2) with
    two full 16-antenna blocks and a partial one, two channels x three LSTs (one with most of
    the sky set), lmax 12 (169 real modes: a full and a partial 128-mode tile), spectral factor.
    (test_tiled_sh_projection_matches_untiled_kernel compares two kernels with each other; the
    full-size check of config 3 reaches the oracle on a 5-antenna sub-array only.)"""
    ant_pos = hy.utils.build_hex_array(hex_spec=(4, 7), d=14.6)
    nant = len(ant_pos)
    freqs, lsts = np.array([100.0, 117.0]), np.array([0.2, 1.7, 4.0])
    lmax, nside = 12, 8
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bo = [orc.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    R_ref, _, ell_ref, m_ref = orc.sh.vis_proj_operator_no_rot(freqs, lsts, bo, ant_pos, lmax, nside,
                                                               latitude=LAT, ref_freq=100.0,
                                                               spectral_idx=-2.7)
    from hydra_amd import vis_simulator as vs
    V = vs.sh_voltages(nside, ant_pos, freqs * 1e6, lsts, bh, LAT)
    assert getattr(V, "hydra_pixidx", None) is not None          # the tiled kernel's pixel lists
    R, ell, m = hy.sh_sampler.vis_proj_operator_no_rot(freqs, lsts, bh, ant_pos, lmax, nside,
                                                       latitude=LAT, ref_freq=100.0,
                                                       spectral_idx=-2.7, cross_only=True)
    assert R.shape == R_ref.shape == (nant * (nant - 1) // 2 * 2 * 3, (lmax + 1) ** 2)
    assert np.array_equal(ell, ell_ref) and np.array_equal(m, m_ref)
    assert operr(R, R_ref) < OP_TOL


def test_gain_operator_with_autos_and_repeated_pairs(hy):
    """Pair lists with auto-correlations (count twice in A^H A) go through the LDS-tiled kernel,
    lists that repeat a pair fall back to the adjacency kernel; both against the oracle."""
    from oracle import gcr_gain
    gs = hy.gain_sampler
    rng = np.random.RandomState(8)
    nant, nf, nt, nmodes = 21, 5, 13, 3
    ants = np.arange(nant)
    base = [(i, j) for i in range(nant) for j in range(i, nant) if (i + 2 * j) % 5]   # incl. autos
    for pairs, tiled in ((base, True), (base + base[3:9], False)):
        nbl = len(pairs)
        Fb = (rng.randn(nmodes, nf, nt) + 1j * rng.randn(nmodes, nf, nt)) / np.sqrt(nf * nt)
        mv = rng.randn(nbl, nf, nt) + 1j * rng.randn(nbl, nf, nt)
        w = 1.0 + rng.rand(nbl, nf, nt)
        ps = 0.1 + rng.rand(nmodes)
        A_real, A_imag = gs.proj_operator(ants, pairs)
        i1, i2 = gcr_gain.bl_indices(ants, pairs)
        op = gs.GainOperator(w, ps, A_real, A_imag, mv, Fb)
        assert (op.idx.pair_table is not None) == tiled
        X = rng.randn(5, 2 * nant * nmodes)
        Y = op.apply(X)
        for v in range(5):
            assert operr(Y[v], gcr_gain.apply_operator(X[v], w, ps, i1, i2, mv, Fb)) < OP_TOL
