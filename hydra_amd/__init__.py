"""
hydra_amd -- MI355X (gfx950) implementation of the Gaussian-constrained-realisation hot path
of HydraRadio/Hydra, behind the Python API of the reference's modules:

    hydra_amd.vis_simulator   <->  hydra/vis_simulator.py
    hydra_amd.ptsrc_sampler   <->  hydra/ptsrc_sampler.py
    hydra_amd.sh_sampler      <->  hydra/sh_sampler.py     (GCR functions)
    hydra_amd.linear_solver   <->  hydra/linear_solver.py
    hydra_amd.gain_sampler    <->  hydra/gain_sampler.py
    hydra_amd.region_sampler  <->  hydra/region_sampler.py (projection operator, segmentation)
    hydra_amd.cosmo_sampler   <->  hydra/cosmo_sampler.py
    hydra_amd.per_ant_beam_sampler <-> hydra/per_ant_beam_sampler.py (simulator consumers)
    hydra_amd.utils, .example <->  hydra/utils.py, hydra/example.py (hot-path helpers)

Every arithmetic step runs in libhydra_hip.so (hand-written HIP kernels, C ABI in
include/hydra_hip.h).  There is no CPU fallback: importing the package needs the built
library, and calling into it needs a GPU.
"""
from . import _lib

_lib.load()          # fail loudly at import if the HIP library has not been built

from . import (beams, cosmo_sampler, example, gain_sampler, linear_solver, per_ant_beam_sampler,  # noqa: E402,F401
               ptsrc_sampler, region_sampler, sh_sampler, utils, vis_simulator)
from .utils import *  # noqa: E402,F401,F403  (the reference re-exports utils: hydra/__init__.py:12)

__all__ = ["beams", "cosmo_sampler", "example", "gain_sampler", "linear_solver", "per_ant_beam_sampler",
           "ptsrc_sampler",
           "region_sampler", "sh_sampler", "utils", "vis_simulator"]
