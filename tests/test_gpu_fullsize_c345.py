"""
BASELINE configs 3, 4 and 5 at their FULL array / mode / source sizes (the oracle cannot run
them whole), checked the way tests/test_gpu_fullsize.py checks config 2:

  c3  SH head, HERA-127 (8001 baselines), lmax 40 (1681 real modes), nside 32, 128 channels:
      rows of the response operator for a 5-antenna sub-array x 3 channels against the oracle
      (element-wise), adjoint identity of R, and the left-hand-side matrix against R a.
  c4  gain head, HERA-331 (54 615 baselines), 64 x 64 (f,t), 16 modes: the tiled pair kernel
      against the CSR kernel on every sample, both against the oracle on (f,t) slices, the basis
      transforms against the oracle, and symmetry of the whole operator.
  c5  point-source head, 350 antennas (61 075 baselines), gridded beam: closed-form Gram = general
      Gram = oracle on scattered source pairs; the tile maps of BOTH Gram kernels in the
      no-split regime (>= 16 384 tiles: Ns = 23 200) and at the full 100 000 sources (408 000
      closed-form tiles, 80 GB Gram) against the oracle on scattered source pairs incl. the last
      tile rows; and the dense solve of a config-5-shaped system to the 1e-8 gate.
"""
import numpy as np
import pytest

from conftest import elem_err, relerr

pytestmark = pytest.mark.gpu

LAT = -30.7215 * np.pi / 180.0


# ------------------------------------------------------------------------------------- c3

@pytest.fixture(scope="module")
def c3():
    import torch
    import bench
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    hex_spec, lmax, nside, nf, _ = bench.SH_CONFIGS["c3"]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.array([0.2, 0.5])
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    R, ell, m = sh_sampler.vis_proj_operator_no_rot(freqs, times, beams, ant_pos, lmax, nside,
                                                    as_torch=True, cross_only=True)
    torch.cuda.synchronize()
    return dict(R=R, ant_pos=ant_pos, freqs=freqs, times=times, lmax=lmax, nside=nside, ell=ell, m=m)


def test_c3_response_rows_match_oracle_on_a_subarray(c3):
    import torch
    from oracle import beams as obeams, gcr_sh
    ant_pos, freqs, times = c3["ant_pos"], c3["freqs"], c3["times"]
    nant, nf, nt = len(ant_pos), freqs.size, times.size
    assert nant == 127 and c3["R"].shape == (8001 * nf * nt, 1681)
    sel = [0, 5, 63, 100, 126]                 # incl. the longest baseline of the array
    fsel = [0, 64, 127]
    sub = {k: ant_pos[a] for k, a in enumerate(sel)}
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in sel]
    R_ref, _, ell, m = gcr_sh.vis_proj_operator_no_rot(freqs[fsel], times, bo, sub, c3["lmax"],
                                                       c3["nside"], latitude=LAT)
    assert np.array_equal(ell, c3["ell"]) and np.array_equal(m, c3["m"])
    rows = []
    for a in range(len(sel)):                  # the oracle's pair order: i, then j > i
        for b in range(a + 1, len(sel)):
            i, j = sel[a], sel[b]
            bl = i * nant - i * (i + 1) // 2 + (j - i - 1)
            for f in fsel:
                for t in range(nt):
                    rows.append((bl * nf + f) * nt + t)
    got = c3["R"][torch.as_tensor(rows, device=c3["R"].device)].cpu().numpy()
    assert got.shape == R_ref.shape
    assert relerr(got, R_ref) < 1e-10
    assert elem_err(got, R_ref) <= 1.0, elem_err(got, R_ref)


def test_c3_adjoint_identity_and_lhs_matrix_full_size(c3):
    import torch
    from hydra_amd import _dev, sh_sampler
    R = c3["R"]
    mrows, nm = R.shape
    gen = torch.Generator(device=R.device).manual_seed(11)
    a = torch.randn(nm, dtype=torch.float64, device=R.device, generator=gen)
    z = torch.randn(mrows, dtype=torch.complex128, device=R.device, generator=gen)
    w = 300.0 + 200.0 * torch.rand(mrows, dtype=torch.float64, device=R.device, generator=gen)
    Ra = _dev.empty((mrows,), torch.complex128)
    _dev.call("hydra_apply_R", _dev.ptr(R), mrows, nm, _dev.ptr(a), _dev.ptr(None), _dev.ptr(Ra),
              _dev.stream_ptr())
    RHz = sh_sampler._adjoint(R, z)
    lhs = torch.sum(Ra.real * z.real + Ra.imag * z.imag).item()
    rhs = torch.dot(a, RHz).item()
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    # a^T (R^T N^-1 R + S^-1) a = sum_r w_r |(R a)_r|^2 + a^T S^-1 a, matrix and matrix-free forms
    ip = torch.full((nm,), 100.0, dtype=torch.float64, device=R.device)
    A = sh_sampler.lhs_matrix(w, ip, R)
    assert torch.equal(A, A.T)
    quad = (torch.sum(w * (Ra.real ** 2 + Ra.imag ** 2)) + torch.sum(ip * a * a)).item()
    assert abs(torch.dot(a, A @ a).item() - quad) <= 1e-11 * abs(quad)
    Aa = sh_sampler.apply_lhs_no_rot(a, w, ip, R)
    ref = A @ a
    assert float((Aa - ref).abs().max() / ref.abs().max()) < 1e-11


# ------------------------------------------------------------------------------------- c4

@pytest.fixture(scope="module")
def c4():
    import torch
    import bench
    import hydra_amd as hy
    from hydra_amd import gain_sampler as gs
    hex_spec, nf, nt, nmaxf, nmaxt, _ = bench.GAIN_CONFIGS["c4"]
    ant_pos = hy.utils.build_hex_array(hex_spec=hex_spec, d=14.6)
    nant = len(ant_pos)
    ants = np.arange(nant)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    freqs = np.linspace(100.0, 120.0, nf)
    times = np.linspace(0.2, 0.5, nt)
    Fb, _, _ = hy.utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, nmaxf, nmaxt, freqs[-1] - freqs[0], times[-1] - times[0])
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(3)
    shape = (len(antpairs), nf, nt)
    mv = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    w = 300.0 + 200.0 * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
    ps = torch.as_tensor(0.05 * (1.0 + np.arange(Fb.shape[0]) / Fb.shape[0]), device=dev)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    op = gs.GainOperator(w, ps, A_real, A_imag, mv, Fb)
    return dict(op=op, Fb=Fb, mv=mv, w=w, ps=ps, ants=ants, antpairs=antpairs, nant=nant, nf=nf,
                nt=nt, gen=gen, dev=dev)


def test_c4_pair_kernels_agree_and_match_oracle_on_slices(c4):
    import torch
    from hydra_amd import _dev
    from oracle import gcr_gain
    op, nant, nf, nt = c4["op"], c4["nant"], c4["nf"], c4["nt"]
    assert nant == 331 and op.nbl == 54615 and op.nmodes == 16
    nft = nf * nt
    x = torch.randn((2, op.n), dtype=torch.float64, device=c4["dev"], generator=c4["gen"])
    g = _dev.empty((2, nant, nft), torch.complex128)
    _dev.call("hydra_gain_basis_fwd", _dev.ptr(x), 2, nant, op.nmodes, nft, _dev.ptr(op.F),
              _dev.ptr(op.s), _dev.ptr(g), _dev.stream_ptr())
    idx = op.idx
    _, _, adj_ptr, adj_b, adj_o, _ = idx.dev()
    y_t = _dev.empty((2, nant, nft), torch.complex128)
    y_c = _dev.empty((2, nant, nft), torch.complex128)
    _dev.call("hydra_gain_pairs_tiled", _dev.ptr(g), 2, nant, nft, _dev.ptr(op.W), op.nbl,
              _dev.ptr(idx.pair_dev()), _dev.ptr(y_t), _dev.stream_ptr())
    _dev.call("hydra_gain_pairs", _dev.ptr(g), 2, nant, nft, _dev.ptr(op.W), op.nbl,
              _dev.host_ptr(idx.adj_ptr), _dev.ptr(adj_ptr), _dev.ptr(adj_b), _dev.ptr(adj_o),
              int(idx.adj_b.size), _dev.ptr(y_c), _dev.stream_ptr())
    scale = float(y_c.abs().max())
    assert float((y_t - y_c).abs().max()) <= 1e-12 * scale      # summation order differs
    # streamed form (every weight read once): one field per call
    assert op.stream
    y_s = _dev.empty((2, nant, nft), torch.complex128)
    for v in range(2):
        _dev.call("hydra_gain_pairs_stream", _dev.ptr(g[v]), nant, nft, _dev.ptr(op.Wt),
                  _dev.ptr(op.sw), _dev.ptr(y_s[v]), _dev.stream_ptr())
    assert float((y_s - y_c).abs().max()) <= 1e-12 * scale
    # oracle on (f,t) slices: the pair stage is point-wise in (f,t)
    i1, i2 = gcr_gain.bl_indices(c4["ants"], c4["antpairs"])
    fsel, tsel = np.array([0, 31, 63]), np.array([0, 17, 63])
    ft = (fsel[:, None] * nt + tsel[None, :])
    ftd = torch.as_tensor(ft.reshape(-1), device=c4["dev"])
    Fb = c4["Fb"]
    xs = x[0].cpu().numpy()
    ps = c4["ps"].cpu().numpy()
    xc = ps[None, :] * (xs[: op.n // 2].reshape(nant, op.nmodes)
                        + 1j * xs[op.n // 2:].reshape(nant, op.nmodes))
    g_ref = np.tensordot(xc, Fb[:, fsel][:, :, tsel], axes=((1,), (0,)))      # (nant, 3, 3)
    g_got = g[0][:, ftd].cpu().numpy().reshape(nant, 3, 3)
    assert elem_err(g_got, g_ref) <= 1.0
    mv_s = c4["mv"].reshape(op.nbl, nft)[:, ftd].cpu().numpy().reshape(op.nbl, 3, 3)
    w_s = c4["w"].reshape(op.nbl, nft)[:, ftd].cpu().numpy().reshape(op.nbl, 3, 3)
    y_ref = gcr_gain.apply_proj_conj(gcr_gain.apply_proj(g_ref, i1, i2, mv_s) * w_s, i1, i2, mv_s,
                                     (nant, 3, 3))
    for y in (y_t, y_c, y_s):
        y_got = y[0][:, ftd].cpu().numpy().reshape(nant, 3, 3)
        assert relerr(y_got, y_ref) < 1e-10
        assert elem_err(y_got, y_ref) <= 1.0
    # the weights themselves
    W_ref = (np.abs(mv_s) ** 2 * w_s)
    assert elem_err(op.W[:, ftd].cpu().numpy().reshape(op.nbl, 3, 3), W_ref, floor=0.0) <= 1.0


def test_c4_operator_is_symmetric_and_batch_equals_single(c4):
    import torch
    op = c4["op"]
    X = torch.randn((8, op.n), dtype=torch.float64, device=c4["dev"], generator=c4["gen"])
    Y = op.local(X)
    for k in (0, 3, 7):
        yk = op.local(X[k].contiguous())
        assert float((yk - Y[k]).abs().max()) <= 1e-12 * float(Y[k].abs().max())
    lhs = torch.dot(X[1], Y[2]).item()
    rhs = torch.dot(Y[1], X[2]).item()
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))
    # positive semi-definite data term: x^T L x >= 0, and apply() adds the identity
    assert torch.dot(X[0], Y[0]).item() > 0.0
    Z = op.apply(X[0].contiguous())
    assert float((Z - (X[0] + Y[0])).abs().max()) <= 1e-13 * float(Z.abs().max())


def test_c4_one_pass_preparation_matches_separate_passes_and_oracle_on_slices(c4):
    """gain_sampler.prepare_sample at HERA-331 x 64 x 64: right-hand side and stream weights equal
    to the separate passes (apply_gains, residual, construct_rhs_mpi, GainOperator), and the
    antenna-space term A^H (N^-1 r + N^-1/2 omega) equal to the oracle's on (f,t) slices
    (gain_sampler.py:145-231; the projection is point-wise in (f,t))."""
    import torch
    import hydra_amd as hy
    from hydra_amd import _dev, gain_sampler as gs
    from oracle import gcr_gain
    op0, nant, nf, nt, dev, gen = c4["op"], c4["nant"], c4["nf"], c4["nt"], c4["dev"], c4["gen"]
    ants, antpairs = c4["ants"], c4["antpairs"]
    nbl, nft = op0.nbl, nf * nt
    shape = (nbl, nf, nt)
    model, w, ps, Fb = c4["mv"], c4["w"], c4["ps"], c4["Fb"]
    gains = 1.0 + 0.1 * torch.randn((nant, nf, nt), dtype=torch.complex128, device=dev, generator=gen)
    data = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    om_r = torch.randn(shape, dtype=torch.complex128, device=dev, generator=gen)
    rng = np.random.RandomState(5)
    om_y = rng.randn(nant, op0.nmodes) + 1j * rng.randn(nant, op0.nmodes)
    A_real, A_imag = gs.proj_operator(ants, antpairs)
    b1, op1 = gs.prepare_sample(None, data, model, gains, w, ps, A_real, A_imag, Fb,
                                omega_y=om_y, omega_r=om_r)
    assert op1._mv is None                                   # the one-pass form ran
    ggv = hy.utils.apply_gains(model, gains, ants, antpairs, inline=False)
    resid = data - ggv
    b0 = gs.construct_rhs_mpi(None, resid, w, ps, A_real, A_imag, ggv, Fb, omega_y=om_y,
                              omega_r=om_r)
    assert float((b1 - b0).abs().max()) <= 1e-12 * float(b0.abs().max())
    op2 = gs.GainOperator(w, ps, A_real, A_imag, ggv, Fb)
    assert float((op1.Wt - op2.Wt).abs().max()) <= 1e-13 * float(op2.Wt.abs().max())
    assert float((op1.sw - op2.sw).abs().max()) <= 1e-12 * float(op2.sw.abs().max())
    x = torch.randn(op2.n, dtype=torch.float64, device=dev, generator=gen)
    y1, y2 = op1.apply(x), op2.apply(x)
    assert float((y1 - y2).abs().max()) <= 1e-12 * float(y2.abs().max())
    # oracle on slices: yy = A^H z straight from the kernel
    count = _dev._lib.query("hydra_gain_stream_weight_count", nant, nft)
    Wt, sw = _dev.empty((count,), torch.float64), _dev.empty((nant, nft), torch.float64)
    yy = _dev.empty((nant, nft), torch.complex128)
    _dev.call("hydra_gain_prepare_stream", _dev.ptr(data), _dev.ptr(model), _dev.ptr(gains),
              _dev.ptr(w), _dev.ptr(om_r), nbl, nant, nft, _dev.ptr(op1.idx.pair_dev()),
              _dev.ptr(Wt), _dev.ptr(sw), _dev.ptr(yy), _dev.stream_ptr())
    fsel, tsel = np.array([0, 31, 63]), np.array([0, 17, 63])
    ftd = torch.as_tensor((fsel[:, None] * nt + tsel[None, :]).reshape(-1), device=dev)
    i1, i2 = gcr_gain.bl_indices(ants, antpairs)
    sl = lambda a, n: a.reshape(n, nft)[:, ftd].cpu().numpy().reshape(n, 3, 3)
    g_s, m_s, d_s, w_s, o_s = sl(gains, nant), sl(model, nbl), sl(data, nbl), sl(w, nbl), sl(om_r, nbl)
    ggv_s = m_s * g_s[i1] * np.conj(g_s[i2])
    z_s = (d_s - ggv_s) * w_s + o_s * np.sqrt(w_s)
    yy_ref = gcr_gain.apply_proj_conj(z_s, i1, i2, ggv_s, (nant, 3, 3))
    yy_got = sl(yy, nant)
    assert relerr(yy_got, yy_ref) < 1e-11
    assert elem_err(yy_got, yy_ref) <= 1.0
    sw_ref = np.zeros((nant, 3, 3))
    Wb = np.abs(ggv_s) ** 2 * w_s
    np.add.at(sw_ref, i1, Wb)
    np.add.at(sw_ref, i2, Wb)
    assert relerr(sl(sw, nant), sw_ref) < 1e-12


# ------------------------------------------------------------------------------------- c5

def _c5_layout():
    import bench
    import hydra_amd as hy
    ant_pos = hy.utils.build_hex_array(hex_spec=(11, 21), d=14.6)
    n0 = len(ant_pos)
    for k in range(19):
        ang = 2.0 * np.pi * k / 19
        ant_pos[n0 + k] = np.array([300.0 * np.cos(ang), 300.0 * np.sin(ang), 0.0])
    return ant_pos


def _grid_beam_tables():
    za_g = np.linspace(0.0, np.pi, 181)
    az_g = np.arange(360) * np.pi / 180.0
    f_g = np.array([90e6, 105e6, 120e6, 135e6])
    from oracle import beams as obeams
    sig = obeams.GaussianBeam(diameter=14.0).get_sigmas(f_g)
    tab = np.exp(-(za_g[None, :, None] / sig[:, None, None]) ** 2) * np.ones((1, 1, az_g.size)) + 1e-3
    return f_g, za_g, az_g, tab


def _oracle_gram_block(ant_pos, antpairs, ra, dec, fluxes, freqs, times, beams_o, gains, w, ssel):
    """The oracle's Gram restricted to the sources ``ssel`` (a Gram entry depends on its two
    sources only, so the restriction is exact)."""
    from oracle import gcr_ptsrc
    ants = np.arange(len(ant_pos))
    proj = gcr_ptsrc.calc_proj_operator(ra[ssel], dec[ssel], fluxes[ssel], ant_pos, antpairs, freqs,
                                        times, beams_o)
    return gcr_ptsrc.gram(gcr_ptsrc.apply_gains_to_proj(ants, antpairs, proj, gains), w)


def test_c5_array_closed_form_general_and_oracle_grams_agree():
    """350 antennas / 61 075 baselines, gridded (bicubic-spline) beam, 3000 sources, 1 x 2 (f,t)."""
    import torch
    import hydra_amd as hy
    from hydra_amd import ptsrc_sampler
    from oracle import beams as obeams, helpers
    ant_pos = _c5_layout()
    nant = len(ant_pos)
    assert nant == 350
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    ns, freqs, times = 3000, np.array([113.0]), np.array([0.2, 0.45])
    np.random.seed(2)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4), (-1.0, 2.0))
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    f_g, za_g, az_g, tab = _grid_beam_tables()
    gb = hy.beams.GriddedBeam(f_g, za_g, az_g, tab)
    fac = ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times,
                                           [gb for _ in ant_pos], LAT, materialize=False)
    dev = fac.V.device
    rng = np.random.RandomState(4)
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, 1, 2) + 1j * rng.randn(nant, 1, 2))
    w_ft = 300.0 + 200.0 * rng.rand(1, 2)
    w = np.broadcast_to(w_ft, (len(antpairs), 1, 2)).copy()
    gd, wd = torch.as_tensor(gains, device=dev), torch.as_tensor(w, device=dev)
    fac.allow_uniform = True
    G_u = fac.gram(gd, wd)
    assert fac.last_gram_kernel == "gram_uniform_kernel"
    fac.allow_uniform = False
    G_g = fac.gram(gd, wd)
    assert fac.last_gram_kernel == "gram_fused_kernel"
    scale = float(G_g.abs().max())
    assert float((G_u - G_g).abs().max()) <= 1e-11 * scale
    ssel = np.array([0, 1, 127, 128, 1500, 2047, 2048, 2990, 2999])
    bo = [obeams.SplineBeam(f_g, za_g, az_g, tab) for _ in ant_pos]
    G_ref = _oracle_gram_block(ant_pos, antpairs, ra, dec, fluxes, freqs, times, bo, gains, w, ssel)
    sd = torch.as_tensor(ssel, device=dev)
    for G in (G_u, G_g):
        got = G[sd][:, sd].cpu().numpy()
        assert relerr(got, G_ref) < 1e-10
        assert elem_err(got, G_ref) <= 1.0, elem_err(got, G_ref)


@pytest.mark.parametrize("ns", [23200, 100000])
def test_c5_source_counts_tile_maps_of_both_gram_kernels(ns):
    """>= 16 384 tiles of 128 x 128 (Ns >= 23 041): no K split, slabs are the tiles themselves
    (gram_plan, gram_finish with nsplit = 1), and the closed-form kernel's compact tile map
    (k3_tile_of / k3_prefix) at 22 000 and 408 000 tiles.  Small array (10 antennas) so that the
    Gram itself is cheap; scattered source pairs incl. the first and last tile rows against the
    oracle, and G x of the two kernels against each other (touches every tile)."""
    import torch
    import hydra_amd as hy
    from hydra_amd import ptsrc_sampler
    from oracle import beams as obeams, helpers
    ant_pos = helpers.build_hex_array((3, 4))
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    freqs, times = np.array([104.0, 118.0]), np.array([0.25, 0.4])
    np.random.seed(6)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4), (-1.0, 2.0))
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    fac = ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh, LAT,
                                           materialize=False)
    dev = fac.V.device
    rng = np.random.RandomState(8)
    gains = (1.0 + 1.0j) + 0.05 * (rng.randn(nant, 2, 2) + 1j * rng.randn(nant, 2, 2))
    w = np.broadcast_to(300.0 + 200.0 * rng.rand(1, 2, 2), (len(antpairs), 2, 2)).copy()
    gd, wd = torch.as_tensor(gains, device=dev), torch.as_tensor(w, device=dev)
    ssel = np.unique(np.concatenate([np.array([0, 1, 95, 96, 127, 128, ns // 2, ns - 129, ns - 128,
                                               ns - 97, ns - 96, ns - 2, ns - 1]),
                                     rng.randint(0, ns, 40)]))
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    G_ref = _oracle_gram_block(ant_pos, antpairs, ra, dec, fluxes, freqs, times, bo, gains, w, ssel)
    sd = torch.as_tensor(ssel, device=dev)
    x = torch.as_tensor(rng.randn(ns), device=dev)
    prods = []
    for allow, name in ((True, "gram_uniform_kernel"), (False, "gram_fused_kernel")):
        fac.allow_uniform = allow
        G = fac.gram(gd, wd)
        assert fac.last_gram_kernel == name
        got = G[sd][:, sd].cpu().numpy()
        assert relerr(got, G_ref) < 1e-10, name
        assert elem_err(got, G_ref) <= 1.0, (name, elem_err(got, G_ref))
        # symmetric mirror: a few rows against the matching columns
        rows = G[sd[:6]]
        cols = G[:, sd[:6]].T
        assert torch.equal(rows, cols)
        prods.append(G @ x)
        del G, rows, cols
        torch.cuda.empty_cache()
    assert float((prods[0] - prods[1]).abs().max()) <= 1e-11 * float(prods[1].abs().max())


def test_c5_shaped_sample_meets_the_residual_gate_with_the_dense_solve():
    """A config-5-shaped GCR sample (350 antennas, gridded beam, one channel; 12 288 sources so
    that the 256-wide-panel Cholesky path is the one that runs): A x = b solved by
    linear_solver.solve_dense to <= 1e-8 relative residual (north_star gate), where the
    reference's unpreconditioned CG stops at maxiters with 1e-4."""
    import torch
    import hydra_amd as hy
    from hydra_amd import linear_solver, ptsrc_sampler
    from oracle import helpers
    ant_pos = _c5_layout()
    nant = len(ant_pos)
    antpairs = [(i, j) for i in range(nant) for j in range(i + 1, nant)]
    nbl = len(antpairs)
    ns, freqs, times = 12288, np.array([110.0]), np.array([0.3])
    np.random.seed(3)
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(ns, (0.0, 1.0), (-0.6, 0.4), (-1.0, 2.0))
    fluxes = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    f_g, za_g, az_g, tab = _grid_beam_tables()
    gb = hy.beams.GriddedBeam(f_g, za_g, az_g, tab)
    fac = ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times,
                                           [gb for _ in ant_pos], LAT, materialize=False)
    dev = fac.V.device
    gen = torch.Generator(device=dev).manual_seed(9)
    gains = torch.full((nant, 1, 1), 1.0 + 1.0j, dtype=torch.complex128, device=dev)
    data = fac.apply(torch.ones(ns, dtype=torch.float64, device=dev))
    data = data + 0.05 * torch.randn(data.shape, dtype=torch.complex128, device=dev, generator=gen)
    w = torch.full((nbl, 1, 1), 400.0, dtype=torch.float64, device=dev)
    prior = torch.full((ns,), 0.1, dtype=torch.float64, device=dev)
    om_n = torch.randn((nbl, 1, 1), dtype=torch.complex128, device=dev, generator=gen)
    om_a = torch.randn(ns, dtype=torch.float64, device=dev, generator=gen)
    A, b = ptsrc_sampler.precompute_mpi(None, np.arange(nant), antpairs, freqs, times, fac, data, w,
                                        gains, prior, omega_n=om_n, omega_a=om_a)
    x = linear_solver.solve_dense(A, b)
    res = float(torch.linalg.norm(A @ x - b) / torch.linalg.norm(b))
    assert res <= 1e-8, res
    x_cg, iters = linear_solver.cg(A, b, maxiters=1000, abs_tol=1e-8, return_iters=True)
    res_cg = float(torch.linalg.norm(A @ x_cg - b) / torch.linalg.norm(b))
    assert res < res_cg or iters < 1000


# ---------------------------------------------------------------------------- c3, streamed

def test_streamed_sh_operator_matches_materialised_operator_and_oracle():
    """sh_sampler.StreamedSHOperator (no (Nvis, Nm) matrix in memory) against the materialised
    operator and the oracle on a small array: right-hand side, operator application, model
    update and the block-accumulated left-hand-side matrix; spectral index != 0; LST count not a
    multiple of the block."""
    import torch
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    from oracle import beams as obeams, gcr_sh, helpers
    ant_pos = helpers.build_hex_array((2, 3))
    nant = len(ant_pos)
    freqs, lsts = np.linspace(100.0, 120.0, 4), np.linspace(0.2, 0.5, 5)
    lmax, nside = 6, 8
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    R_ref, _, ell, m = gcr_sh.vis_proj_operator_no_rot(freqs, lsts, bo, ant_pos, lmax, nside,
                                                       latitude=LAT, ref_freq=110.0, spectral_idx=-2.5)
    op = sh_sampler.StreamedSHOperator(freqs, lsts, bh, ant_pos, lmax, nside, latitude=LAT,
                                       ref_freq=110.0, spectral_idx=-2.5, time_block=2)
    mrows, nm = R_ref.shape
    assert op.shape == (mrows, nm) and mrows == nant * (nant - 1) // 2 * 4 * 5
    rng = np.random.RandomState(4)
    w = 300.0 + 200.0 * rng.rand(mrows)
    ip = 50.0 + 50.0 * rng.rand(nm)
    d = rng.randn(mrows) + 1j * rng.randn(mrows)
    om1 = (rng.randn(mrows) + 1j * rng.randn(mrows)) / np.sqrt(2)
    om0, a0, a = rng.randn(nm), 0.1 * rng.randn(nm), rng.randn(nm)
    b_ref = gcr_sh.construct_rhs_no_rot(d, w, ip, om0, om1, a0, R_ref)
    b = op.construct_rhs(d, w, ip, om0, om1, a0).cpu().numpy()
    assert relerr(b, b_ref) < 1e-10 and elem_err(b, b_ref) <= 1.0
    y_ref = (R_ref @ a).reshape(op.nbl, op.nf, op.nt)
    y = op.apply(a).cpu().numpy()
    assert relerr(y, y_ref) < 1e-10
    lhs_ref = gcr_sh.apply_lhs_no_rot(a, w, ip, R_ref)
    lhs = op.apply_lhs(a, w, ip).cpu().numpy()
    assert relerr(lhs, lhs_ref) < 1e-10
    A = op.lhs_matrix(w, ip).cpu().numpy()
    A_ref = R_ref.real.T @ (w[:, None] * R_ref.real) + R_ref.imag.T @ (w[:, None] * R_ref.imag) \
        + np.diag(ip)
    assert relerr(A, A_ref) < 1e-10 and np.array_equal(A, A.T)
    # weights that do not depend on the baseline: the closed form through pixel space, against
    # the block-accumulated form and the oracle; in several frequency chunks too
    wu = np.broadcast_to(300.0 + 200.0 * rng.rand(1, 4, 5), (op.nbl, 4, 5)).reshape(-1).copy()
    A_ref = R_ref.real.T @ (wu[:, None] * R_ref.real) + R_ref.imag.T @ (wu[:, None] * R_ref.imag) \
        + np.diag(ip)
    assert op.fac._noise_plan(hy._dev.f64(wu).reshape(op.nbl, 4, 5))[0] == "uniform"
    A_cf = op.lhs_matrix(wu, ip).cpu().numpy()
    A_bl = op.lhs_matrix(wu, ip, closed_form=False).cpu().numpy()
    assert relerr(A_cf, A_ref) < 1e-10 and relerr(A_bl, A_ref) < 1e-10
    assert np.array_equal(A_cf, A_cf.T)
    op.LHS_CHUNK_BYTES = 1          # one channel at a time
    assert relerr(op.lhs_matrix(wu, ip).cpu().numpy(), A_ref) < 1e-10
    # and the materialised HIP operator agrees with both
    R, _, _ = sh_sampler.vis_proj_operator_no_rot(freqs, lsts, bh, ant_pos, lmax, nside, latitude=LAT,
                                                  ref_freq=110.0, spectral_idx=-2.5, cross_only=True)
    assert relerr(R, R_ref) < 1e-10


def test_c3_streamed_operator_full_array_equals_resident_operator(c3):
    """HERA-127, lmax 40, nside 32, 128 channels x 2 LSTs: the streamed operator's right-hand
    side, application and left-hand-side matrix equal those of the resident (materialised)
    operator of the fixture."""
    import torch
    import hydra_amd as hy
    from hydra_amd import _dev, sh_sampler
    R = c3["R"]
    mrows, nm = R.shape
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in c3["ant_pos"]]
    op = sh_sampler.StreamedSHOperator(c3["freqs"], c3["times"], beams, c3["ant_pos"], c3["lmax"],
                                       c3["nside"], latitude=sh_sampler.vis_proj_operator_no_rot.__defaults__[0],
                                       time_block=1)
    assert op.shape == (mrows, nm)
    gen = torch.Generator(device=R.device).manual_seed(21)
    a = torch.randn(nm, dtype=torch.float64, device=R.device, generator=gen)
    z = torch.randn(mrows, dtype=torch.complex128, device=R.device, generator=gen)
    w = 300.0 + 200.0 * torch.rand(mrows, dtype=torch.float64, device=R.device, generator=gen)
    ip = torch.full((nm,), 100.0, dtype=torch.float64, device=R.device)
    want = sh_sampler._adjoint(R, z)
    got = op.adjoint(z)
    assert float((got - want).abs().max() / want.abs().max()) < 1e-11
    Ra = _dev.empty((mrows,), torch.complex128)
    _dev.call("hydra_apply_R", _dev.ptr(R), mrows, nm, _dev.ptr(a), _dev.ptr(None), _dev.ptr(Ra),
              _dev.stream_ptr())
    got = op.apply(a).reshape(-1)
    assert float((got - Ra).abs().max() / Ra.abs().max()) < 1e-11
    A_res = sh_sampler.lhs_matrix(w, ip, R)
    A_str = op.lhs_matrix(w, ip)
    assert float((A_str - A_res).abs().max() / A_res.abs().max()) < 1e-11
    # baseline-independent weights: the closed form through pixel space
    nbl, nf, nt = op.nbl, op.nf, op.nt
    wu = (300.0 + 200.0 * torch.rand((1, nf, nt), dtype=torch.float64, device=R.device,
                                     generator=gen)).expand(nbl, nf, nt).contiguous().reshape(-1)
    A_res = sh_sampler.lhs_matrix(wu, ip, R)
    A_cf = op.lhs_matrix(wu, ip)
    assert float((A_cf - A_res).abs().max() / A_res.abs().max()) < 1e-11
    assert torch.equal(A_cf, A_cf.T)


def test_streamed_sh_operator_at_hera_350():
    """The SH head on config 5's array (350 antennas, 61 075 baselines; the north star asks for
    every head there): StreamedSHOperator -- antenna-space R a and R^H z on the matrix cores through
    the super-block / super-row kernels (more than 160 antennas), closed-form left-hand side --
    against the materialised operator built by the projection kernel: application, right-hand
    side, both left-hand-side paths, and the pair-loop kernels on the same voltages."""
    import torch
    import hydra_amd as hy
    from hydra_amd import _dev, sh_sampler
    ant_pos = _c5_layout()
    nant = len(ant_pos)
    freqs, lsts = np.linspace(100.0, 120.0, 2), np.linspace(0.2, 0.5, 2)
    lmax, nside = 5, 8
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    op = sh_sampler.StreamedSHOperator(freqs, lsts, bh, ant_pos, lmax, nside, latitude=LAT)
    R, _, _ = sh_sampler.vis_proj_operator_no_rot(freqs, lsts, bh, ant_pos, lmax, nside, latitude=LAT,
                                                  as_torch=True, cross_only=True)
    mrows, nm = R.shape
    assert op.shape == (mrows, nm) and mrows == nant * (nant - 1) // 2 * 4 and nant == 350
    g = torch.Generator(device=R.device).manual_seed(8)
    a = torch.randn(nm, dtype=torch.float64, device=R.device, generator=g)
    d = torch.randn(mrows, dtype=torch.complex128, device=R.device, generator=g)
    om1 = torch.randn(mrows, dtype=torch.complex128, device=R.device, generator=g)
    om0 = torch.randn(nm, dtype=torch.float64, device=R.device, generator=g)
    a0 = 0.1 * torch.randn(nm, dtype=torch.float64, device=R.device, generator=g)
    ip = 50.0 + 50.0 * torch.rand(nm, dtype=torch.float64, device=R.device, generator=g)
    wu = (300.0 + 200.0 * torch.rand((1, 2, 2), dtype=torch.float64, device=R.device, generator=g)
          ).expand(op.nbl, 2, 2).contiguous()

    def rel(x, y):
        return float(((x - y).abs().max() / y.abs().max()).item())

    y_ref = (R @ a.to(torch.complex128)).reshape(op.nbl, 2, 2)
    assert rel(op.apply(a), y_ref) < 1e-11
    b_ref = sh_sampler.construct_rhs_no_rot(d, wu.reshape(-1), ip, om0, om1, a0, R)
    assert rel(op.construct_rhs(d, wu, ip, om0, om1, a0), b_ref) < 1e-11
    # the same two products through the pair loops on the vector units
    prev = _dev._lib.query("hydra_debug_adjoint_mfma", 0)
    op.fac.antspace_adjoint = False
    try:
        y_pair = op.apply(a)
        b_pair = op.construct_rhs(d, wu, ip, om0, om1, a0)
    finally:
        _dev._lib.query("hydra_debug_adjoint_mfma", prev)
        op.fac.antspace_adjoint = True
    assert rel(op.apply(a), y_pair) < 1e-12 and rel(op.construct_rhs(d, wu, ip, om0, om1, a0), b_pair) < 1e-12
    A_ref = sh_sampler.lhs_matrix(wu.reshape(-1), ip, R)
    A_cf = op.lhs_matrix(wu, ip)
    assert op.fac.last_gram_kernel == "gram_uniform_kernel"
    assert rel(A_cf, A_ref) < 1e-10 and torch.equal(A_cf, A_cf.T)
    assert rel(op.lhs_matrix(wu, ip, closed_form=False), A_ref) < 1e-10
