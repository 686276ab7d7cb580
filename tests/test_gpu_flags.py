"""
Closed-form Gram with flagged data (ptsrc_sampler.py:62-75 for an inv_noise_var with flags):
G = G_closed(all pairs, w_ref) - Gram over the materialised rows of the flagged samples.
Against the oracle with 1 % random flags, one fully flagged baseline, one fully flagged (f,t)
sample and a few down-weighted (not zero) samples; the plan cache; the fall-back to the general
kernel when too many samples deviate; and at BASELINE config 2's full size against the general
kernel.
"""
import numpy as np
import pytest

from conftest import elem_err, relerr

pytestmark = pytest.mark.gpu


def _flagged_noise(nbl, nf, nt, rng, frac=0.01):
    w_ft = 300.0 + 200.0 * rng.rand(1, nf, nt)
    w = np.broadcast_to(w_ft, (nbl, nf, nt)).copy()
    flags = rng.rand(nbl, nf, nt) < frac
    w[flags] = 0.0
    w[rng.randint(nbl)] = 0.0                      # a dead baseline
    w[:, nf // 2, nt // 2] = 0.0                   # a fully flagged (f,t) sample: w_ref = 0 there
    k = rng.randint(0, nbl, 5)
    w[k, 0, 0] *= 0.5                              # down-weighted, not flagged
    return w


@pytest.mark.parametrize("hex_spec,ns,nf,nt", [((3, 4), 100, 6, 8), ((4, 7), 300, 4, 6),
                                               ((5, 9), 130, 2, 3)])
def test_flagged_closed_form_system_matches_oracle(hex_spec, ns, nf, nt):
    import torch
    import hydra_amd as hy
    from oracle import beams as obeams, gcr_ptsrc, helpers
    from test_gpu_parity import _gcr_inputs, _setup
    ant_pos, antpairs, ra, dec, freqs, times, fluxes = _setup(hex_spec, ns, nf, nt, seed=15)
    nant, nbl = len(ant_pos), len(antpairs)
    ants = np.array(list(ant_pos.keys()))
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    proj_ref = gcr_ptsrc.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bo)
    gains, _, prior, om_n, om_a = _gcr_inputs(nbl, nant, nf, nt, ns, seed=17)
    rng = np.random.RandomState(19)
    w = _flagged_noise(nbl, nf, nt, rng)
    data = helpers.apply_gains((proj_ref.reshape(-1, ns) @ (1.0 + 0.1 * rng.randn(ns))).reshape(
        nbl, nf, nt), gains, ants, antpairs) + 0.05 * om_n[::-1]
    A_ref, b_ref = gcr_ptsrc.precompute(ants, antpairs, proj_ref, data, w, gains, prior,
                                        omega_n=om_n, omega_a=om_a)
    fac = hy.ptsrc_sampler.calc_proj_operator(ra, dec, fluxes, ant_pos, antpairs, freqs, times, bh,
                                              materialize=False)
    args = (None, ants, antpairs, freqs, times)
    wd = torch.as_tensor(w, device=fac.V.device)
    kw = dict(data_chunk=torch.as_tensor(data, device=fac.V.device), inv_noise_var_chunk=wd,
              gain_chunk=gains, amp_prior_std=prior, omega_n=om_n, omega_a=om_a)
    A, b = hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=fac, **kw)
    assert fac.last_gram_kernel == "gram_uniform_kernel"
    n_dev = int((w != w.max(axis=0, keepdims=True)).sum())
    assert fac.last_gram_flagged_rows == n_dev > 0
    A, b = A.cpu().numpy(), b.cpu().numpy()
    assert relerr(A, A_ref) < 1e-10 and relerr(b, b_ref) < 1e-10
    assert elem_err(A, A_ref, floor=1e-12) <= 1.0
    assert np.array_equal(A, A.T)
    # the plan is cached with the weight tensor: a second sample does not re-analyse ...
    plan = fac._plan_cache[2]
    hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=fac, **kw)
    assert fac._plan_cache[2] is plan
    # ... until the weights change in place
    wd[0, 0, 0] = 123.0
    A2, _ = hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=fac, **kw)
    assert fac._plan_cache[2] is not plan
    w2 = w.copy()
    w2[0, 0, 0] = 123.0
    A2_ref, _ = gcr_ptsrc.precompute(ants, antpairs, proj_ref, data, w2, gains, prior,
                                     omega_n=om_n, omega_a=om_a)
    assert relerr(A2.cpu().numpy(), A2_ref) < 1e-10
    # too many deviating samples: the general kernel takes over, same numbers
    w3 = 100.0 + 800.0 * rng.rand(nbl, nf, nt)
    kw["inv_noise_var_chunk"] = torch.as_tensor(w3, device=fac.V.device)
    A3, b3 = hy.ptsrc_sampler.precompute_mpi(*args, proj_chunk=fac, **kw)
    assert fac.last_gram_kernel == "gram_fused_kernel"
    A3_ref, b3_ref = gcr_ptsrc.precompute(ants, antpairs, proj_ref, data, w3, gains, prior,
                                          omega_n=om_n, omega_a=om_a)
    assert relerr(A3.cpu().numpy(), A3_ref) < 1e-10 and relerr(b3.cpu().numpy(), b3_ref) < 1e-10


def test_flagged_closed_form_equals_general_kernel_at_config2_full_size():
    """37 antennas, 2000 sources, 64 x 200 (f,t), 1 % flags + a dead baseline (85 000+ rows
    corrected): closed form + correction == general per-baseline-weight kernel, and the quadratic
    form x^T G x = sum_r w_r |(R' x)_r|^2 with the flagged weights."""
    import torch
    import bench
    import hydra_amd as hy
    from hydra_amd import _dev, ptsrc_sampler
    hex_spec, ns, nf, nt = bench.CONFIGS["c2"]
    inp = bench.synth_inputs(hex_spec, ns, nf, nt)
    beams = [hy.beams.GaussianBeam(diameter=14.0) for _ in inp["ant_pos"]]
    fac = ptsrc_sampler.calc_proj_operator(inp["ra"], inp["dec"], inp["fluxes"], inp["ant_pos"],
                                           inp["antpairs"], inp["freqs"], inp["times"], beams,
                                           materialize=False)
    dev = fac.V.device
    nant, nbl = len(inp["ant_pos"]), len(inp["antpairs"])
    gen = torch.Generator(device=dev).manual_seed(5)
    g = (1.0 + 1.0j) + 0.05 * torch.randn((nant, nf, nt), dtype=torch.complex128, device=dev,
                                          generator=gen)
    w = (300.0 + 200.0 * torch.rand((1, nf, nt), dtype=torch.float64, device=dev, generator=gen)
         ).expand(nbl, nf, nt).contiguous()
    w[torch.rand((nbl, nf, nt), device=dev, generator=gen) < 0.01] = 0.0
    w[123] = 0.0
    fac.allow_uniform = True
    G_f = fac.gram(g, w)
    assert fac.last_gram_kernel == "gram_uniform_kernel" and fac.last_gram_flagged_rows > 85000
    fac.allow_uniform = False
    G_g = fac.gram(g, w)
    assert fac.last_gram_kernel == "gram_fused_kernel"
    scale = float(G_g.abs().max())
    assert float((G_f - G_g).abs().max()) <= 1e-11 * scale
    assert torch.equal(G_f, G_f.T)
    x = torch.randn(ns, dtype=torch.float64, device=dev, generator=gen)
    gg = _dev.empty((nbl, nf, nt), torch.complex128)
    _dev.call("hydra_row_factors", _dev.ptr(g), _dev.ptr(None), _dev.ptr(fac.bl_i),
              _dev.ptr(fac.bl_j), nbl, nf, nt, _dev.ptr(gg), _dev.ptr(None), _dev.stream_ptr())
    y = fac.apply(x, rowmul=gg)
    quad = torch.sum(w * (y.real ** 2 + y.imag ** 2)).item()
    assert abs(torch.dot(x, G_f @ x).item() - quad) <= 1e-11 * abs(quad)


def test_streamed_sh_left_hand_side_with_flags_matches_oracle():
    """StreamedSHOperator.lhs_matrix with flagged data (sh_sampler.py:389-402 takes any
    inv_noise_var): the closed form through pixel space at the common weight of each (f,t) sample
    minus the Gram of the flagged rows, frequency chunk by frequency chunk -- against the
    oracle's Re(R^H N^-1 R) + S^-1 and against the block-accumulated path; weights that depend on
    the baseline everywhere still take the block path."""
    import torch
    import hydra_amd as hy
    from hydra_amd import sh_sampler
    from oracle import beams as obeams, gcr_sh, helpers
    LAT = -30.7215 * np.pi / 180.0
    ant_pos = helpers.build_hex_array((3, 4))
    freqs, lsts = np.linspace(100.0, 120.0, 5), np.linspace(0.2, 0.5, 4)
    lmax, nside = 6, 8
    bh = [hy.beams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in ant_pos]
    R_ref, _, _, _ = gcr_sh.vis_proj_operator_no_rot(freqs, lsts, bo, ant_pos, lmax, nside,
                                                     latitude=LAT)
    op = sh_sampler.StreamedSHOperator(freqs, lsts, bh, ant_pos, lmax, nside, latitude=LAT)
    mrows, nm = R_ref.shape
    rng = np.random.RandomState(23)
    w = _flagged_noise(op.nbl, op.nf, op.nt, rng, frac=0.03)
    ip = 50.0 + 50.0 * rng.rand(nm)
    wf = w.reshape(-1)
    A_ref = R_ref.real.T @ (wf[:, None] * R_ref.real) + R_ref.imag.T @ (wf[:, None] * R_ref.imag) \
        + np.diag(ip)
    n_dev = int((w != w.max(axis=0, keepdims=True)).sum())
    for chunk_bytes in (16 << 30, 1):              # all channels at once / one channel at a time
        op.LHS_CHUNK_BYTES = chunk_bytes
        A = op.lhs_matrix(w, ip).cpu().numpy()
        assert op.fac.last_gram_kernel == "gram_uniform_kernel"
        assert op.last_lhs_flagged_rows == n_dev > 0
        assert relerr(A, A_ref) < 1e-10 and np.array_equal(A, A.T)
    A_bl = op.lhs_matrix(w, ip, closed_form=False).cpu().numpy()
    assert op.fac.last_gram_kernel == "gram_kernel" and relerr(A_bl, A_ref) < 1e-10
    # per-baseline weights everywhere: no closed form
    w3 = 100.0 + 800.0 * rng.rand(op.nbl, op.nf, op.nt)
    A3 = op.lhs_matrix(w3, ip).cpu().numpy()
    assert op.fac.last_gram_kernel == "gram_kernel"
    w3f = w3.reshape(-1)
    A3_ref = R_ref.real.T @ (w3f[:, None] * R_ref.real) + R_ref.imag.T @ (w3f[:, None] * R_ref.imag) \
        + np.diag(ip)
    assert relerr(A3, A3_ref) < 1e-10
