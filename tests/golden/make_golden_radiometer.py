#!/usr/bin/env python
"""
Golden vectors for the radiometer equation of the SH head, from the REFERENCE's own
hydra/sh_sampler.py (radiometer_eq :497-522), loaded with the technique of make_golden.py
(placeholder modules for the missing third-party packages, which this function never calls).
Run in the build container only; writes radiometer.npz.

Usage:  python tests/golden/make_golden_radiometer.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, load_reference  # noqa: E402


def main():
    sh = load_reference()["sh_sampler"]
    rng = np.random.RandomState(17)
    out = {}
    for tag, nant, indx in (("a", 4, 6), ("b", 7, 5), ("c", 1, 3)):
        autos = 50.0 + 10.0 * rng.rand(nant * indx)
        ants = np.arange(nant)
        out[tag + "_autos"], out[tag + "_nant"] = autos, np.array(nant)
        out[tag + "_cross"] = sh.radiometer_eq(autos, ants, 10.7, 97.6e3)
        out[tag + "_with_autos"] = sh.radiometer_eq(autos, ants, 10.7, 97.6e3, Nnights=3,
                                                    include_autos=True)
    np.savez(os.path.join(OUT, "radiometer.npz"), **out)
    print("wrote radiometer.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
