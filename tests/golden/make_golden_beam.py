#!/usr/bin/env python
"""
Golden vectors for the simulator-free functions of the per-antenna beam head, from the
REFERENCE's own ``hydra/per_ant_beam_sampler.py`` (imported with the placeholder technique of
make_golden.py; the functions that call the visibility simulator are NOT run).  Build container
only; beam.npz is committed.

  get_ant_inds, get_bess_matr, get_bess_outer, get_bess_to_vis_from_contraction (both forms)

Usage:  python tests/golden/make_golden_beam.py
"""
import importlib
import os

import numpy as np

from make_golden import OUT, load_reference


def main():
    load_reference()
    bs = importlib.import_module("hydra.per_ant_beam_sampler")
    rng = np.random.RandomState(51)
    nt, ns, nant, nf = 3, 17, 5, 4
    nmodes = np.array([1, 2, 3, 1, 2, 3, 1, 2])
    mmodes = np.array([0, 0, 0, 1, 1, 1, -1, -1])
    nc = nmodes.size
    rho = rng.uniform(0.0, 1.0, size=(nt, ns))
    phi = rng.uniform(0.0, 2.0 * np.pi, size=(nt, ns))
    bess = bs.get_bess_matr(nmodes, mmodes, rho, phi)
    outer = bs.get_bess_outer(bess)
    keep = bs.get_ant_inds(2, nant)
    bsc = rng.randn(1, 1, nf, nt, nant, nant, nc, nc) + 1j * rng.randn(1, 1, nf, nt, nant, nant, nc, nc)
    bsc_ref = rng.randn(1, 1, nf, nt, nant, nant, nc) + 1j * rng.randn(1, 1, nf, nt, nant, nant, nc)
    coeffs = rng.randn(nc, nf, nant, 1, 1) + 1j * rng.randn(nc, nf, nant, 1, 1)
    trans = bs.get_bess_to_vis_from_contraction(bsc, coeffs, nant, 2)
    trans_ref = bs.get_bess_to_vis_from_contraction(bsc_ref, coeffs, nant, 2, ref_contraction=True)
    np.savez(os.path.join(OUT, "beam.npz"), nmodes=nmodes, mmodes=mmodes, rho=rho, phi=phi,
             bess=bess, outer=outer, keep=keep, bsc=bsc, bsc_ref=bsc_ref, coeffs=coeffs,
             trans=trans, trans_ref=trans_ref)
    print("beam golden vectors written to", OUT)


if __name__ == "__main__":
    main()
