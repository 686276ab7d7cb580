#!/usr/bin/env python
"""
Golden vectors for three more support helpers, from the REFERENCE's own hydra/utils.py
(load_gain_model :113-157, extend_coords_with_padding :261-288, gain_prior_pspec_sqrt :565-622),
loaded with the technique of make_golden.py.  Run in the build container only; writes utils3.npz.

Usage:  python tests/golden/make_golden_utils3.py
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, load_reference  # noqa: E402


def main():
    utils = load_reference()["utils"]
    rng = np.random.RandomState(23)
    out = {}
    freqs, lsts = np.linspace(100.0, 120.0, 12), np.linspace(0.2, 0.5, 9)
    out["freqs"], out["lsts"] = freqs, lsts
    out["pad_f"] = utils.extend_coords_with_padding(freqs, pad=[2, 3])
    out["pad_t"] = utils.extend_coords_with_padding(lsts, pad=[0, 4])
    out["pad_none"] = utils.extend_coords_with_padding(lsts)
    out["ps_flat"] = utils.gain_prior_pspec_sqrt(lsts, freqs, 0.1)
    out["ps_taper"] = utils.gain_prior_pspec_sqrt(lsts, freqs, 0.1, gain_prior_sigma_frate=0.5,
                                                  gain_prior_sigma_delay=80.0,
                                                  gain_prior_zeropoint_std=0.3, frate0=0.1,
                                                  delay0=-20.0)
    out["ps_frate"] = utils.gain_prior_pspec_sqrt(lsts, freqs, 2.0, gain_prior_sigma_frate=1.5)
    model = rng.randn(3, 5, 4) + 1j * rng.randn(3, 5, 4)
    out["gain_model"] = model
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "g.npy")
        np.save(path, model)
        out["gain_padded"] = utils.load_gain_model(path, lst_pad=[1, 2], freq_pad=[3, 0],
                                                   pad_value=1.0)
        out["gain_plain"] = utils.load_gain_model(path)
    np.savez(os.path.join(OUT, "utils3.npz"), **out)
    print("wrote utils3.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
