"""
The Gibbs driver (example.py, counterpart of the reference's example.py for the ptsrc / gains /
SH heads) end to end on small problems: output files of the reference's names and shapes, and
samples that are statistically sensible (the source head recovers the simulated catalogue, whose
fractional amplitude deviations are zero, to within a few posterior sigmas).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, extra, niters=4, on_head=None):
    sys.path.insert(0, ROOT)
    import example
    out = str(tmp_path / "out")
    argv = ["--hex-array", "3", "4", "--Nptsrc", "20", "--Ntimes", "6", "--Nfreqs", "8",
            "--Niters", str(niters), "--output-dir", out, "--seed", "3"] + extra
    assert example.main(argv, on_head=on_head) == 0
    return out


def _np(x):
    return x.cpu().numpy() if hasattr(x, "cpu") else np.asarray(x)


def test_one_seeded_iteration_of_each_head_matches_the_oracle(tmp_path):
    """The driver calls on_head(...) right before a head draws its random numbers.  From the
    arrays it hands over and numpy's generator state at that moment the CPU oracle replays the
    head -- same draws in the reference's order (ptsrc_sampler.py:109-116,138; example.py:1022-1025;
    gain_sampler.py:163-199) -- and must land on the sample the driver saved."""
    from oracle import beams as obeams, cg as ocg, gcr_gain, gcr_ptsrc, gcr_sh
    want = {}

    def on_head(name, n, st):
        if n != 0:
            return
        rng_state = np.random.get_state()
        if name == "ptsrc":
            bo = [obeams.GaussianBeam(diameter=14.0) for _ in st["ant_pos"]]
            proj = gcr_ptsrc.calc_proj_operator(st["ra"], st["dec"], st["fluxes"], st["ant_pos"],
                                                st["antpairs"], st["freqs"], st["times"], bo,
                                                st["latitude"])
            A, b = gcr_ptsrc.precompute(st["ants"], st["antpairs"], proj, _np(st["resid"]),
                                        _np(st["inv_noise_var"]), _np(st["gains"]),
                                        st["amp_prior_std"])         # draws from the global RNG
            want["src"] = gcr_ptsrc.dense_solve(A, b) * st["amp_prior_std"]
        elif name == "gains":
            i1, i2 = gcr_gain.bl_indices(st["ants"], st["antpairs"])
            Fb, ps = st["Fbasis"], st["pspec_sqrt"]
            nmodes, nant = Fb.shape[0], len(st["ants"])
            mv, w = _np(st["model_vis"]), _np(st["inv_noise_var"])
            om_y, om_r = gcr_gain.draw_gain_omegas(st["seed"], nant, nmodes, mv.shape)
            b = gcr_gain.construct_rhs(_np(st["resid"]), w, ps, i1, i2, mv, Fb, om_y, om_r)
            x = ocg.cg(None, b, use_norm_tol=True,
                       linear_op=lambda v: gcr_gain.apply_operator(v, w, ps, i1, i2, mv, Fb))
            want["gain"] = (x[: x.size // 2] + 1j * x[x.size // 2:]).reshape(nant, nmodes) * ps
        elif name == "sh":
            R, w = _np(st["vis_response"]), _np(st["inv_noise_var"]).reshape(-1)
            ip, resid = _np(st["inv_prior_var"]), _np(st["resid"])
            om_a = np.random.randn(ip.size)
            om_n = (1.0 * np.random.randn(*resid.shape) + 1.0j * np.random.randn(*resid.shape)) \
                / np.sqrt(2.0)
            b = gcr_sh.construct_rhs_no_rot(resid, w, ip, om_a, om_n, _np(st["a_0"]), R)
            want["sh"] = ocg.cg(None, b, maxiters=1000, abs_tol=1e-8, x0=_np(st["x0"]),
                                linear_op=lambda a: gcr_sh.apply_lhs_no_rot(a, w, ip, R))
        np.random.set_state(rng_state)        # the driver's head draws the same numbers next

    out = _run(tmp_path, ["--ptsrc", "--gains", "--sh", "--sh-lmax", "3", "--sh-nside", "8"],
               niters=1, on_head=on_head)
    x = np.load(os.path.join(out, "src_amp_00000.npy"))
    g = np.load(os.path.join(out, "delta_g_amps_00000.npy"))
    a = np.load(os.path.join(out, "sh_amp_00000.npy"))
    assert set(want) == {"src", "gain", "sh"}
    assert np.max(np.abs(x - want["src"])) <= 1e-8 * np.max(np.abs(want["src"]))
    assert np.max(np.abs(g - want["gain"])) <= 1e-6 * np.max(np.abs(want["gain"]))
    assert np.max(np.abs(a - want["sh"])) <= 1e-6 * np.max(np.abs(want["sh"]))


def test_driver_rejects_what_is_outside_the_hot_path(tmp_path):
    sys.path.insert(0, ROOT)
    import example
    for flag in ("--regions", "--beam", "--cl", "--pspec", "--stats"):
        with pytest.raises(NotImplementedError):
            example.main(["--output-dir", str(tmp_path), flag])
    with pytest.raises(ValueError, match="not recognised"):
        example.main(["--output-dir", str(tmp_path), "--solver", "lu"])


def test_ptsrc_and_gain_heads_write_reference_outputs(tmp_path):
    out = _run(tmp_path, ["--ptsrc", "--gains", "--timing"])
    for name in ["ptsrc_amps0.npy", "ptsrc_coords0.npy", "sim_gain0_level.npy", "k_freq.npy",
                 "k_time.npy", "Fbasis_w0000.npy", "sim_model0_chunk_w0000.npy",
                 "sim_data_chunk_w0000.npy", "sim_delta_g_amps0_w0000.npy", "timing.dat"]:
        assert os.path.exists(os.path.join(out, name)), name
    assert np.load(os.path.join(out, "sim_data_chunk_w0000.npy")).shape == (45, 8, 6)
    for n in range(4):
        x = np.load(os.path.join(out, "src_amp_%05d.npy" % n))
        g = np.load(os.path.join(out, "delta_g_amps_%05d.npy" % n))
        assert x.shape == (20,) and np.all(np.isfinite(x))
        assert g.shape == (10, 4) and np.all(np.isfinite(g))
    # fractional amplitude deviations of the simulated catalogue are zero; prior std is 0.1
    assert np.max(np.abs(x)) < 0.5
    # gain perturbations stay at the level of the prior / simulated amplitudes
    assert np.max(np.abs(g)) < 1.0


def test_sh_head_samples_the_simulated_modes(tmp_path):
    out = _run(tmp_path, ["--sh", "--sh-lmax", "4", "--sh-nside", "8", "--sim-gain-amp-std", "0",
                          "--sigma-noise", "0.0005"])
    truth = np.load(os.path.join(out, "sim_sh_amps0.npy"))
    a = np.load(os.path.join(out, "sh_amp_00003.npy"))
    assert a.shape == truth.shape == (25,) and np.all(np.isfinite(a))
    # a posterior sample: within the prior range, and correlated with the simulated sky
    assert np.max(np.abs(a)) < 1.0
    assert np.corrcoef(a, truth)[0, 1] > 0.3


def test_cosmo_head_writes_field_samples(tmp_path):
    out = _run(tmp_path, ["--cosmo", "--ptsrc", "--cosmo-ra-bounds", "5", "40", "--cosmo-dec-bounds",
                          "-40", "-20", "--cosmo-ra-ngrid", "6", "--cosmo-dec-ngrid", "5"])
    for n in range(4):
        f = np.load(os.path.join(out, "cosmo_field_%05d.npy" % n))
        assert f.shape == (8, 6, 5) and np.all(np.isfinite(f))
        x = np.load(os.path.join(out, "src_amp_%05d.npy" % n))
        assert x.shape == (20,) and np.all(np.isfinite(x))
    # the simulated sky holds no cosmological field: samples scatter about zero at the prior level
    assert np.abs(f).max() < 20.0 and abs(f.mean()) < 1.0
