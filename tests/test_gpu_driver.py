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


def _run(tmp_path, extra):
    sys.path.insert(0, ROOT)
    import example
    out = str(tmp_path / "out")
    argv = ["--hex-array", "3", "4", "--Nptsrc", "20", "--Ntimes", "6", "--Nfreqs", "8",
            "--Niters", "4", "--output-dir", out, "--seed", "3"] + extra
    assert example.main(argv) == 0
    return out


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
