"""
Size-independent properties of the hot path at BASELINE config 2's FULL size (37 antennas,
2000 sources, 64 x 200 (f,t), complex128: M = 8.5e6 rows; the materialised operator would be
273 GB, so the oracle cannot be run here):

  * adjoint identity  <R x, z> = <x, R^H z>  ties R x and R^H z to each other;
  * quadratic form    x^T G x = sum_r w_r |(R' x)_r|^2  ties the Gram (both kernels) to R x;
  * the closed-form antenna-space Gram equals the general per-baseline Gram;
  * a slice of the full-size Gram equals the oracle's Gram computed from a materialised
    sub-block of rows (additivity over (f,t): G = sum over channel blocks).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c2():
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
    dev = _dev.device()
    nant, nbl = len(inp["ant_pos"]), len(inp["antpairs"])
    gen = torch.Generator(device=dev).manual_seed(5)
    g = (1.0 + 1.0j) + 0.05 * torch.randn((nant, nf, nt), dtype=torch.complex128, device=dev,
                                          generator=gen)
    w_ft = 300.0 + 200.0 * torch.rand((nf, nt), dtype=torch.float64, device=dev, generator=gen)
    w = w_ft.expand(nbl, nf, nt).contiguous()
    return dict(inp=inp, fac=fac, gains=g, w=w, dev=dev, gen=gen, shape=(nbl, nf, nt), ns=ns)


def test_adjoint_identity_full_size(c2):
    import torch
    fac, gen, dev = c2["fac"], c2["gen"], c2["dev"]
    x = torch.randn(c2["ns"], dtype=torch.float64, device=dev, generator=gen)
    z = torch.randn(c2["shape"], dtype=torch.complex128, device=dev, generator=gen)
    Rx = fac.apply(x)
    RHz = fac.adjoint(z)
    lhs = torch.sum(Rx.real * z.real + Rx.imag * z.imag).item()      # Re <R x, z>
    rhs = torch.dot(x, RHz).item()
    assert abs(lhs - rhs) <= 1e-11 * max(abs(lhs), abs(rhs))


def test_gram_kernels_agree_and_match_quadratic_form_full_size(c2):
    import torch
    from hydra_amd import _dev
    fac, gen, dev = c2["fac"], c2["gen"], c2["dev"]
    nbl, nf, nt = c2["shape"]
    fac.allow_uniform = True
    G_u = fac.gram(c2["gains"], c2["w"])
    assert fac.last_gram_kernel == "gram_uniform_kernel"
    fac.allow_uniform = False
    G_g = fac.gram(c2["gains"], c2["w"])
    assert fac.last_gram_kernel == "gram_fused_kernel"
    fac.allow_uniform = True
    scale = torch.max(torch.abs(G_g)).item()
    assert torch.max(torch.abs(G_u - G_g)).item() <= 1e-11 * scale
    assert torch.equal(G_u, G_u.T) and torch.equal(G_g, G_g.T)
    # x^T G x = sum_r w_r |g_i conj(g_j) (R x)_r|^2
    x = torch.randn(c2["ns"], dtype=torch.float64, device=dev, generator=gen)
    gg = _dev.empty((nbl, nf, nt), torch.complex128)
    _dev.call("hydra_row_factors", _dev.ptr(c2["gains"]), _dev.ptr(None), _dev.ptr(fac.bl_i),
              _dev.ptr(fac.bl_j), nbl, nf, nt, _dev.ptr(gg), _dev.ptr(None), _dev.stream_ptr())
    y = fac.apply(x, rowmul=gg)
    quad = torch.sum(c2["w"] * (y.real ** 2 + y.imag ** 2)).item()
    for G in (G_u, G_g):
        xGx = torch.dot(x, G @ x).item()
        assert abs(xGx - quad) <= 1e-11 * abs(quad)


def test_full_size_gram_is_the_sum_of_channel_blocks_checked_by_the_oracle(c2):
    """Additivity over rows: the Gram restricted to a (2 channel x 3 time) block of the full
    operator, computed by the HIP kernels on sliced device buffers, equals the oracle's Gram of
    the materialised block for a 128-source sub-catalogue."""
    import torch
    import hydra_amd as hy
    from oracle import beams as obeams, gcr_ptsrc
    inp = c2["inp"]
    fsel, tsel, ssel = np.array([5, 40]), np.array([0, 99, 199]), np.arange(700, 828)
    ants = np.arange(len(inp["ant_pos"]))
    bo = [obeams.GaussianBeam(diameter=14.0) for _ in inp["ant_pos"]]
    proj = gcr_ptsrc.calc_proj_operator(inp["ra"][ssel], inp["dec"][ssel],
                                        inp["fluxes"][ssel][:, fsel], inp["ant_pos"],
                                        inp["antpairs"], inp["freqs"][fsel], inp["times"][tsel], bo)
    g = c2["gains"].cpu().numpy()[:, fsel][:, :, tsel]
    w = c2["w"].cpu().numpy()[:, fsel][:, :, tsel]
    G_ref = gcr_ptsrc.gram(gcr_ptsrc.apply_gains_to_proj(ants, inp["antpairs"], proj, g), w)
    # the same block cut out of the full-size device operator
    from hydra_amd.ptsrc_sampler import FactoredProj
    V = c2["fac"].V[torch.as_tensor(fsel, device=c2["dev"])][:, torch.as_tensor(tsel, device=c2["dev"])]
    V = V[:, :, :, ssel[0]:ssel[-1] + 1].contiguous()
    sub = FactoredProj(V, c2["fac"].bl_i_host, c2["fac"].bl_j_host)
    for allow in (True, False):
        sub.allow_uniform = allow
        G = sub.gram(torch.as_tensor(g, device=c2["dev"]).contiguous(),
                     torch.as_tensor(w, device=c2["dev"]).contiguous()).cpu().numpy()
        assert np.max(np.abs(G - G_ref)) <= 1e-10 * np.max(np.abs(G_ref))
