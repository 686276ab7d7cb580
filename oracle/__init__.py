"""
CPU oracle for the Hydra GCR hot path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy/scipy restatement of the reference algorithm
(HydraRadio/Hydra @ 2025-10-31) for the path named by BASELINE.json:north_star:

    visibility response  ->  GCR operator / RHS  ->  CG or dense solve

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker or the reported CPU
baseline.  Nothing under ``hydra_amd/`` imports it; the product path runs on the
HIP library alone and raises if that library is missing.

Pinning status (see DESIGN.md, section "Oracle"):

* ``gcr_ptsrc``, ``gcr_sh``, ``cg``, ``gcr_gain``, ``helpers``: PINNED.  Checked
  against golden vectors produced by the reference's own pure-numpy functions,
  imported in the build container by ``tests/golden/make_golden.py`` (the
  fixtures are committed under ``tests/golden/``), plus the reference's one
  numerical test (``hydra/tests/test_linear_solver.py:9-54``).
* ``geometry``, ``beams``, ``sphere`` and hence ``response``: PARITY UNPINNED at
  the third-party boundary.  The arithmetic lives in matvis / pyuvdata / healpy /
  astropy (un-vendored, unpinned in ``requirements.txt``; not installed, no
  network) and the reference's tests hold no numerical value for it.  These
  modules restate the published formulas and are defended by analytic
  known-answer tests (``tests/test_oracle_geometry.py``).  The code *inside*
  ``vis_simulator.py`` that consumes those third-party pieces (operation order
  of delay, phase, voltage, outer product) is followed line by line.
"""
