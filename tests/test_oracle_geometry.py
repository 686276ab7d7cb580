"""
Analytic known-answer tests for the oracle pieces whose arithmetic the reference
delegates to third-party packages that are not installed (matvis, pyuvdata,
healpy): PARITY UNPINNED at that boundary, so these defend the restated
formulas.  CPU only.
"""
import numpy as np
import scipy.special

from oracle import beams, geometry, gcr_ptsrc, gcr_sh, helpers, response, sphere

LAT = -30.7215 * np.pi / 180.0


def test_zenith_and_east_conventions():
    lst = 0.7
    m = geometry.eci_to_enu_matrix(lst, LAT)
    assert np.allclose(m @ m.T, np.eye(3), atol=1e-15)
    zen = m @ geometry.point_source_crd_eq(np.array([lst]), np.array([LAT]))
    assert np.allclose(zen[:, 0], [0, 0, 1], atol=1e-15)
    east = m @ geometry.point_source_crd_eq(np.array([lst + 0.1]), np.array([LAT]))
    assert east[0, 0] > 0            # RA > LST: not yet transited, east of meridian
    north = m @ geometry.point_source_crd_eq(np.array([lst]), np.array([LAT + 0.1]))
    assert north[1, 0] > 0 and abs(north[0, 0]) < 1e-15
    pole = m @ np.array([0.0, 0.0, 1.0])      # celestial pole: due north at altitude |lat|
    assert np.allclose(pole, [0, np.cos(LAT), np.sin(LAT)], atol=1e-15)


def test_az_za_uvbeam_convention():
    # east -> az 0, north -> az pi/2 in the uvbeam convention; za from (e, n) only
    az, za = geometry.enu_to_az_za(np.array([0.5, 0.0, -0.5, 0.0]),
                                   np.array([0.0, 0.5, 0.0, -0.5]), orientation="uvbeam")
    assert np.allclose(az, [0.0, np.pi / 2, np.pi, 3 * np.pi / 2])
    assert np.allclose(za, np.arcsin(0.5))
    az2, _ = geometry.enu_to_az_za(np.array([0.5]), np.array([0.0]), orientation="astropy")
    assert np.allclose(az2, np.pi / 2)
    _, za0 = geometry.enu_to_az_za(np.array([0.0]), np.array([0.0]))
    assert za0[0] == 0.0


def test_gaussian_beam_formula():
    bm = beams.GaussianBeam(diameter=14.0)
    f = np.array([1.0e8, 1.5e8])
    sig = bm.get_sigmas(f)
    lam = geometry.C_LIGHT / f
    assert np.allclose(sig, np.arcsin(2.2150894 * lam / (np.pi * 14.0)) * 2 / 2.355)
    za = np.array([0.0, 0.1, 0.3])
    p = bm.compute_response(az_array=np.zeros(3), za_array=za, freq_array=f)
    assert p.shape == (1, 1, 2, 3)
    assert np.allclose(p[0, 0, 0], np.exp(-za**2 / sig[0] ** 2))
    assert p[0, 0, 0, 0] == 1.0
    # FWHM of the power beam ~ 2.355 sigma / sqrt2
    half = np.sqrt(np.log(2.0)) * sig[0]
    ph = bm.compute_response(az_array=np.zeros(1), za_array=np.array([half]), freq_array=f[:1])
    assert np.allclose(ph, 0.5)


def _tiny_setup(ns=9, seed=4):
    rng = np.random.RandomState(seed)
    ants = {0: (0.0, 0.0, 0.0), 1: (14.6, 0.0, 0.0), 2: (3.0, 20.0, 0.5)}
    ra = rng.uniform(0.0, 1.0, ns)
    dec = rng.uniform(-0.9, -0.2, ns)
    freqs = np.array([100e6, 130e6])
    lsts = np.array([0.3, 0.6, 0.9])
    flux = rng.uniform(0.5, 2.0, (ns, freqs.size))
    bms = [beams.GaussianBeam(diameter=14.0) for _ in ants]
    return ants, flux, ra, dec, freqs, lsts, bms


def test_single_baseline_phase_closed_form():
    ants, flux, ra, dec, freqs, lsts, bms = _tiny_setup()
    vis = response.simulate_vis_per_source(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT)
    assert vis.shape == (2, 3, 3, 3, ra.size)
    # closed form for baseline (0,1): E-W separation b = 14.6 m
    for fi, f in enumerate(freqs):
        for ti, lst in enumerate(lsts):
            ha = lst - ra
            e = -np.cos(dec) * np.sin(ha)
            n = np.cos(LAT) * np.sin(dec) - np.sin(LAT) * np.cos(dec) * np.cos(ha)
            u = np.sin(LAT) * np.sin(dec) + np.cos(LAT) * np.cos(dec) * np.cos(ha)
            za = np.arcsin(np.sqrt(e * e + n * n))
            sig = bms[0].get_sigmas(np.array([f]))[0]
            amp = 0.5 * flux[:, fi] * np.exp(-za**2 / sig**2) * (u > 0)
            phase = 2 * np.pi * f * 14.6 * e / geometry.C_LIGHT
            expect = amp * np.exp(1j * phase)
            assert np.allclose(vis[fi, ti, 0, 1], expect, rtol=1e-10, atol=1e-14)
            # autos: |v|^2, real
            assert np.allclose(vis[fi, ti, 0, 0], amp, rtol=1e-12, atol=1e-15)
    # lower triangle is left at zero (vis_simulator.py:281-289)
    assert np.all(vis[:, :, 1, 0] == 0) and np.all(vis[:, :, 2, 1] == 0)


def test_zenith_source_has_zero_phase_and_unit_beam():
    ants = {0: (0.0, 0.0, 0.0), 1: (100.0, -40.0, 3.0)}
    lst = 1.1
    vis = response.simulate_vis_per_source(
        ants, np.array([[2.0]]), np.array([lst]), np.array([LAT]), np.array([150e6]),
        np.array([lst]), [beams.GaussianBeam(diameter=14.0)] * 2, latitude=LAT)
    # up-component of antenna 1 gives phase 2 pi nu * 3 / c only
    ph = 2 * np.pi * 150e6 * 3.0 / geometry.C_LIGHT
    assert np.allclose(vis[0, 0, 0, 1, 0], 1.0 * np.exp(1j * ph), rtol=1e-9)


def test_below_horizon_sources_are_zeroed():
    ants, flux, ra, dec, freqs, lsts, bms = _tiny_setup()
    ra = ra + np.pi          # other side of the sky
    dec = -dec
    vis = response.simulate_vis_per_source(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT)
    assert np.all(vis == 0)


def test_sum_over_sources_equals_simulate_vis_and_subarr():
    ants, flux, ra, dec, freqs, lsts, bms = _tiny_setup()
    per = response.simulate_vis_per_source(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT)
    tot = response.simulate_vis(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT)
    assert np.allclose(per.sum(axis=-1), tot)
    sub = response.simulate_vis_per_source(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT,
                                           subarr_ant=1)
    assert sub.shape == (2, 3, 3, ra.size)
    assert np.allclose(sub[:, :, 0], per[:, :, 0, 1])       # conj(v_0) v_1
    assert np.allclose(sub[:, :, 2], per[:, :, 1, 2].conj())  # conj(v_2) v_1


def test_proj_operator_gathers_baselines():
    ants, flux, ra, dec, freqs, lsts, bms = _tiny_setup()
    ap = [(0, 1), (0, 2), (1, 2)]
    proj = gcr_ptsrc.calc_proj_operator(ra, dec, flux, ants, ap, freqs / 1e6, lsts, bms,
                                        latitude=LAT)
    per = response.simulate_vis_per_source(ants, flux, ra, dec, freqs, lsts, bms, latitude=LAT)
    assert proj.shape == (3, 2, 3, ra.size)
    assert np.array_equal(proj[1], per[:, :, 0, 2])
    assert not np.any(np.isnan(proj))


def test_reference_smoke_case_ptsrc():
    # hydra/tests/test_ptsrc_sampler.py:8-75 with the oracle's Gaussian beam
    np.random.seed(10)
    ns = 17
    ant_pos = {0: (0.0, 0.0, 0.0), 1: (0.0, 14.0, 0.0), 2: (0.0, 0.0, 14.0)}
    ap = [(0, 1), (0, 2), (1, 2)]
    lsts = np.linspace(0.0, 3.0, 5)
    freqs = np.linspace(100.0, 200.0, 10)
    bms = [beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    ra, dec, amps = helpers.generate_random_ptsrc_catalogue(
        ns, (0.0, 2 * np.pi), (-0.5 * np.pi, 0.5 * np.pi), (-1.0, 2.0))
    fl = helpers.get_flux_from_ptsrc_amp(amps, freqs, -2.7)
    proj = gcr_ptsrc.calc_proj_operator(ra, dec, fl, ant_pos, ap, freqs, lsts, bms)
    assert np.all(~np.isnan(proj))
    data = (proj.reshape((-1, ns)) @ np.ones(ns)).reshape(3, 10, 5)
    gains = (1.1 + 1.0j) * np.ones((3, 10, 5))
    A, b = gcr_ptsrc.precompute(list(ant_pos.keys()), ap, proj, data, np.ones(data.shape),
                                gains, 0.1 * np.ones(ns))
    assert np.all(~np.isnan(A)) and np.all(~np.isnan(b))


# ----------------------------------------------------------------- sphere

def test_pix2ang_ring_properties():
    for nside in (1, 2, 8, 32):
        npix = sphere.nside2npix(nside)
        th, ph = sphere.pix2ang_ring(nside, np.arange(npix))
        assert np.all(np.diff(th) >= -1e-15)             # rings ordered north -> south
        assert np.all((ph >= 0) & (ph < 2 * np.pi))
        z = np.cos(th)
        assert abs(z.sum()) < 1e-9                       # symmetric about the equator
        assert np.allclose(th + th[::-1], np.pi, atol=1e-14)
        # equal-area rings: number of distinct z values = 4 nside - 1
        assert np.unique(np.round(z, 12)).size == 4 * nside - 1
    th, ph = sphere.pix2ang_ring(1, np.arange(12))
    assert np.allclose(np.cos(th[:4]), 2.0 / 3.0) and np.allclose(np.cos(th[4:8]), 0.0)
    assert np.allclose(ph[:4], np.pi / 4 + np.arange(4) * np.pi / 2)
    assert np.allclose(ph[4:8], np.arange(4) * np.pi / 2)
    th, ph = sphere.pix2ang_ring(2, np.arange(48))
    assert np.allclose(np.cos(th[0]), 1 - 1.0 / 12.0) and np.allclose(ph[0], np.pi / 4)
    assert np.allclose(np.cos(th[4:12]), 1 - 4.0 / 12.0)
    assert np.allclose(ph[4], np.pi / 8)


def test_lambda_lm_against_scipy_and_closed_forms():
    lmax = 12
    th = np.linspace(0.05, np.pi - 0.05, 41)
    lam = sphere.lambda_lm(lmax, th)
    ell, em = sphere.getlm(lmax)
    assert ell.size == (lmax + 1) * (lmax + 2) // 2
    for n in range(ell.size):
        assert sphere.getidx(lmax, ell[n], em[n]) == n
        y = scipy.special.sph_harm_y(int(ell[n]), int(em[n]), th, 0.0)
        assert np.allclose(lam[n], y.real, rtol=1e-11, atol=1e-13)
    assert np.allclose(lam[sphere.getidx(lmax, 0, 0)], np.sqrt(1 / (4 * np.pi)))
    assert np.allclose(lam[sphere.getidx(lmax, 1, 0)], np.sqrt(3 / (4 * np.pi)) * np.cos(th))
    assert np.allclose(lam[sphere.getidx(lmax, 1, 1)], -np.sqrt(3 / (8 * np.pi)) * np.sin(th))
    assert np.allclose(lam[sphere.getidx(lmax, 2, 0)],
                       np.sqrt(5 / (16 * np.pi)) * (3 * np.cos(th) ** 2 - 1))
    assert sphere.getlmax(ell.size) == lmax and sphere.getlmax(ell.size + 1) == -1


def test_unit_mode_maps_and_alm2map():
    lmax, nside = 6, 16
    maps = sphere.unit_mode_maps(lmax, nside)
    ell, em = sphere.getlm(lmax)
    nalm = ell.size
    npix = sphere.nside2npix(nside)
    assert maps.shape == (2 * nalm, npix)
    th, ph = sphere.pix2ang_ring(nside, np.arange(npix))
    # closed forms: a_00 = 1 -> 1/sqrt(4 pi); a_11 = 1 -> -2 sqrt(3/8pi) sin th cos ph
    assert np.allclose(maps[sphere.getidx(lmax, 0, 0)], 1 / np.sqrt(4 * np.pi))
    k = sphere.getidx(lmax, 1, 1)
    assert np.allclose(maps[k], -2 * np.sqrt(3 / (8 * np.pi)) * np.sin(th) * np.cos(ph))
    assert np.allclose(maps[k + nalm], 2 * np.sqrt(3 / (8 * np.pi)) * np.sin(th) * np.sin(ph))
    # imaginary part of m=0 is ignored by the synthesis
    assert np.all(maps[nalm: nalm + lmax + 1] == 0)
    # alm2map is linear and agrees with the unit maps
    rng = np.random.RandomState(0)
    alm = rng.randn(nalm) + 1j * rng.randn(nalm)
    m1 = sphere.alm2map(alm, nside, lmax)
    m2 = alm.real @ maps[:nalm] + alm.imag @ maps[nalm:]
    assert np.allclose(m1, m2, atol=1e-13)
    # quadrature orthogonality of the real-mode maps on the pixelisation
    # (m = 0: norm 1;  m > 0: map = 2 Re Y -> norm 2), approximately at this nside
    gram = (maps[:nalm] @ maps[:nalm].T) * (4 * np.pi / npix)
    expect = np.diag(np.where(em == 0, 1.0, 2.0))
    assert np.max(np.abs(gram - expect)) < 2e-2


def test_sh_response_operator_shapes_and_consistency():
    # hydra/tests/test_sh_sampler.py:21-67 (smaller lmax/nside to keep it quick)
    ant_pos = {0: (0.0, 0.0, 0.0), 1: (0.0, 14.0, 0.0), 2: (0.0, 0.0, 14.0)}
    lsts = np.linspace(0.0, 3.0, 3)
    freqs = np.linspace(100.0, 200.0, 2)
    bms = [beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    lmax, nside = 4, 4
    ems, ells, idx = gcr_sh.get_em_ell_idx(lmax)
    proj, autos, ell, m = gcr_sh.vis_proj_operator_no_rot(freqs, lsts, bms, ant_pos, lmax, nside)
    assert proj.shape == (3 * freqs.size * lsts.size, len(ems))
    assert autos.shape == (3 * freqs.size * lsts.size, len(ems))
    assert np.all(~np.isnan(proj))
    assert not np.allclose(proj.imag, 0)
    # response to a_00 = sum over pixels of the pixel visibilities * Y_00 * pixel area
    npix = sphere.nside2npix(nside)
    dec, ra = sphere.pix2ang_ring(nside, np.arange(npix))
    vis_pix = response.simulate_vis_per_source(
        ant_pos, np.ones((npix, freqs.size)), ra, dec - 0.5 * np.pi, freqs * 1e6, lsts, bms,
        latitude=-0.5361913261514378)
    a00 = vis_pix[:, :, 0, 1, :].sum(axis=-1) * (4 * np.pi / npix) / np.sqrt(4 * np.pi)
    assert np.allclose(proj.reshape(3, freqs.size, lsts.size, -1)[0, :, :, 0], a00)
    # spectral index scales rows by (f/ref)^beta
    proj2, _, _, _ = gcr_sh.vis_proj_operator_no_rot(freqs, lsts, bms, ant_pos, lmax, nside,
                                                     spectral_idx=-2.5)
    sc = ((freqs / 100.0) ** -2.5)[None, :, None, None]
    assert np.allclose(proj2.reshape(3, freqs.size, lsts.size, -1),
                       proj.reshape(3, freqs.size, lsts.size, -1) * sc)


def test_spline_beam_reproduces_a_smooth_beam_and_wraps_in_azimuth():
    """oracle.beams.SplineBeam (scipy interp1d + RectBivariateSpline, the pyuvdata az_za_simple
    recipe): a finely gridded smooth beam is reproduced, azimuth is periodic across 2 pi, and
    the cubic frequency interpolation is exact for a cubic frequency dependence."""
    from oracle import beams as obeams
    za = np.linspace(0.0, np.pi / 2, 91)
    az = np.arange(0.0, 360.0, 2.0) * np.pi / 180.0
    freqs = np.linspace(100e6, 120e6, 6)

    def truth(f, z, a):
        nu = (f - 100e6) / 20e6
        return np.exp(-(z / 0.3) ** 2) * (1.0 + 0.2 * np.cos(2 * a) * np.sin(z) ** 2) \
            * (1.0 + 0.3 * nu - 0.2 * nu ** 2 + 0.1 * nu ** 3)
    grid = truth(freqs[:, None, None], za[None, :, None], az[None, None, :])
    bm = obeams.SplineBeam(freqs, za, az, grid)
    rng = np.random.RandomState(0)
    zq = rng.uniform(0.0, np.pi / 2, 500)
    aq = rng.uniform(0.0, 2 * np.pi, 500)
    fq = np.array([103.3e6, 117.9e6])
    got = bm.compute_response(az_array=aq, za_array=zq, freq_array=fq)[0, 0]
    want = truth(fq[:, None], zq[None, :], aq[None, :])
    assert np.max(np.abs(got - want)) < 1e-5      # interpolation error of a 1 x 2 degree grid
    lo = bm.compute_response(az_array=np.array([1e-9]), za_array=np.array([0.4]), freq_array=fq)
    hi = bm.compute_response(az_array=np.array([2 * np.pi - 1e-9]), za_array=np.array([0.4]),
                             freq_array=fq)
    assert np.max(np.abs(lo - hi)) < 1e-8


def test_healpix_bilinear_weights_known_answers():
    """oracle.sphere.interp_weights_ring (the HEALPix bilinear scheme behind pyuvdata's
    "healpix_simple"): weights are a partition of unity, reproduce a map exactly at pixel centres
    (all resolutions, caps and equatorial belt), are continuous across ring boundaries, treat the
    poles symmetrically, and converge to a smooth field as the pixels shrink."""
    from oracle import sphere
    rng = np.random.RandomState(1)
    for nside in (1, 2, 4, 8):
        npix = 12 * nside * nside
        th, ph = sphere.pix2ang_ring(nside, np.arange(npix))
        pix, w = sphere.interp_weights_ring(nside, th, ph)
        assert np.allclose(w.sum(0), 1.0, atol=1e-14)
        m = rng.randn(npix)
        assert np.max(np.abs((m[pix] * w).sum(0) - m)) < 1e-13
    th = np.arccos(rng.uniform(-1, 1, 20000))
    ph = rng.uniform(0, 2 * np.pi, 20000)
    errs = []
    for nside in (8, 16, 32):
        pix, w = sphere.interp_weights_ring(nside, th, ph)
        assert np.allclose(w.sum(0), 1.0, atol=1e-13) and w.min() > -1e-12
        assert pix.min() >= 0 and pix.max() < 12 * nside * nside
        tp, pp = sphere.pix2ang_ring(nside, np.arange(12 * nside * nside))
        f = lambda t, p: np.cos(t) + 0.3 * np.sin(t) * np.cos(p - 0.4)        # noqa: E731
        errs.append(np.max(np.abs((f(tp, pp)[pix] * w).sum(0) - f(th, ph))))
    assert errs[1] < 0.6 * errs[0] and errs[2] < 0.6 * errs[1] and errs[2] < 3e-3   # converges
    # exactly at a pole: the mean of the four cap pixels; phi wraps at 2 pi
    m = rng.randn(12 * 16)
    pix, w = sphere.interp_weights_ring(4, [0.0, np.pi], [0.3, 1.0])
    assert np.allclose((m[pix] * w).sum(0), [m[:4].mean(), m[-4:].mean()])
    pa, wa = sphere.interp_weights_ring(4, [1.0, 1.0], [1e-13, 2 * np.pi - 1e-13])
    assert abs((m[pa] * wa).sum(0)[0] - (m[pa] * wa).sum(0)[1]) < 1e-10
    # continuity across a ring: just above / below a pixel centre's latitude
    t0, p0 = sphere.pix2ang_ring(8, np.array([200]))
    va = [(m2 := np.arange(768.0) ** 0.5)[px] @ wx for px, wx in
          [tuple(a[:, 0] for a in sphere.interp_weights_ring(8, t0 + d, p0)) for d in (-1e-9, 1e-9)]]
    assert abs(va[0] - va[1]) < 1e-6


def test_healpix_beam_oracle_interpolates_linearly_in_frequency():
    from oracle import beams, sphere
    nside = 4
    th, ph = sphere.pix2ang_ring(nside, np.arange(12 * nside * nside))
    maps = np.stack([np.cos(th) ** 2, 2.0 * np.cos(th) ** 2 + 1.0])
    hb = beams.HealpixBeam([100e6, 120e6], maps)
    out = hb.compute_response(az_array=ph[:7], za_array=th[:7], freq_array=[100e6, 105e6, 120e6])
    assert out.shape == (1, 1, 3, 7)
    assert np.allclose(out[0, 0, 0], maps[0, :7]) and np.allclose(out[0, 0, 2], maps[1, :7])
    assert np.allclose(out[0, 0, 1], 0.75 * maps[0, :7] + 0.25 * maps[1, :7])


def test_polybeam_known_answers():
    """hera_sim PolyBeam as restated in oracle/beams.py (hydra/example.py:143-168): unit response
    at zenith whatever the coefficients and the channel; a pure T_1 beam is (1 - 2 sin za)^2; the
    spectral index only rescales the zenith angle, za -> za / (nu / nu_ref)^index."""
    rng = np.random.RandomState(3)
    coeffs = rng.randn(9)
    pb = beams.PolyBeam(coeffs, spectral_index=-0.7, ref_freq=1.0e8)
    za = np.linspace(0.0, 1.2, 7)
    freqs = np.array([0.8e8, 1.0e8, 1.3e8])
    r = pb.compute_response(az_array=np.zeros_like(za), za_array=za, freq_array=freqs)[0, 0]
    assert r.shape == (3, 7)
    assert np.allclose(r[:, 0], 1.0, atol=1e-14)
    t1 = beams.PolyBeam([0.0, 1.0], spectral_index=0.0, ref_freq=1.0e8)
    r1 = t1.compute_response(az_array=np.zeros_like(za), za_array=za, freq_array=freqs[:1])[0, 0, 0]
    assert np.allclose(r1, (1.0 - 2.0 * np.sin(za)) ** 2, atol=1e-14)
    scale = (freqs / 1.0e8) ** -0.7
    ref = beams.PolyBeam(coeffs, spectral_index=0.0, ref_freq=1.0e8)
    for k in range(3):
        rk = ref.compute_response(az_array=np.zeros_like(za), za_array=za / scale[k],
                                  freq_array=np.array([1.0e8]))[0, 0, 0]
        assert np.allclose(r[k], rk, rtol=1e-13, atol=1e-15)


def test_sidereal_invariance_and_transit_symmetry():
    """Topocentric coordinates depend on (ra - lst) only, and a source is mirrored east-west
    about its transit: e(-h) = -e(h), n and up even in the hour angle h (matvis
    eci_to_enu_matrix / point_source_crd_eq as restated in oracle/geometry.py)."""
    rng = np.random.RandomState(4)
    ra, dec = rng.uniform(0, 2 * np.pi, 20), rng.uniform(-1.2, 0.6, 20)
    lst, shift = 1.1, 0.37
    a = geometry.eci_to_enu_matrix(lst, LAT) @ geometry.point_source_crd_eq(ra, dec)
    b = geometry.eci_to_enu_matrix(lst + shift, LAT) @ geometry.point_source_crd_eq(ra + shift, dec)
    assert np.allclose(a, b, atol=1e-14)
    h = rng.uniform(0, 1.0, 20)
    m = geometry.eci_to_enu_matrix(lst, LAT)
    before = m @ geometry.point_source_crd_eq(lst + h, dec)     # ra > lst: east
    after = m @ geometry.point_source_crd_eq(lst - h, dec)
    assert np.allclose(before[0], -after[0], atol=1e-14) and np.all(before[0] >= 0)
    assert np.allclose(before[1:], after[1:], atol=1e-14)
