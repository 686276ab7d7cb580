#!/usr/bin/env python
"""
Golden vectors for the chunk map and the beam-shape helper, from the REFERENCE's own
hydra/utils.py (freqs_times_for_worker :313-371, get_beam_interp_shape :542-562), loaded with the
technique of make_golden.py (placeholder modules for the missing third-party packages, which
these two functions never call).  Run in the build container only; writes worker_map.npz.

Usage:  python tests/golden/make_golden_utils2.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import OUT, load_reference  # noqa: E402


class FakeComm:
    """The two calls freqs_times_for_worker makes on its communicator."""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size


def main():
    utils = load_reference()["utils"]
    freqs, times = np.linspace(100.0, 120.0, 11), np.linspace(0.2, 0.5, 7)
    cases = [(4, 2, 2), (6, 3, 2), (3, 2, 2), (1, 1, 1), (5, 5, 1), (9, 3, 3)]
    out = {"freqs": freqs, "times": times, "cases": np.array(cases)}
    for nworkers, fchunks, tchunks in cases:
        for rank in range(nworkers):
            fi, ti, wmap = utils.freqs_times_for_worker(FakeComm(rank, nworkers), freqs, times,
                                                        fchunks, tchunks)
            key = f"n{nworkers}_f{fchunks}_t{tchunks}_r{rank}"
            out[key + "_fi"], out[key + "_ti"] = fi, ti
            out[key + "_grid"] = np.array([wmap[w][:2] for w in range(nworkers)])
            for w in range(nworkers):
                out[key + f"_w{w}_f"], out[key + f"_w{w}_t"] = wmap[w][2], wmap[w][3]
    # get_beam_interp_shape: analytic beam / beam interface -> False, UVBeam -> not
    # future_array_shapes, anything else -> False
    import pyuvdata
    from pyuvdata.analytic_beam import AnalyticBeam
    uvb_old, uvb_new = pyuvdata.UVBeam(), pyuvdata.UVBeam()
    uvb_old.future_array_shapes, uvb_new.future_array_shapes = False, True
    out["interp_shape"] = np.array([utils.get_beam_interp_shape(AnalyticBeam()),
                                    utils.get_beam_interp_shape(pyuvdata.BeamInterface()),
                                    utils.get_beam_interp_shape(uvb_old),
                                    utils.get_beam_interp_shape(uvb_new),
                                    utils.get_beam_interp_shape(object())])
    np.savez(os.path.join(OUT, "worker_map.npz"), **out)
    print("wrote worker_map.npz:", len(out), "arrays; interp_shape =", out["interp_shape"])


if __name__ == "__main__":
    main()
