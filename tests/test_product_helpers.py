"""
The PRODUCT-side helpers (hydra_amd.utils / hydra_amd.example / hydra_amd.sh_sampler packing),
bit for bit against the golden vectors made by the reference's own functions
(tests/golden/make_golden.py): the oracle's copies are pinned by tests/test_oracle_golden.py,
these are the functions a user of the package actually calls.  Pure host code: no GPU needed.
"""
import numpy as np
import pytest

from conftest import load_golden

from hydra_amd import example, sh_sampler, utils


def _pairs(arr):
    return [tuple(int(x) for x in p) for p in arr]


def test_catalogue_seds_hex_array_and_fourier_basis_match_reference():
    g = load_golden("helpers.npz")
    np.random.seed(0)
    ra, dec, amps = example.generate_random_ptsrc_catalogue(
        25, ra_bounds=(0.0, 1.0), dec_bounds=(-0.6, 0.4), logflux_bounds=(-1.0, 2.0))
    assert np.array_equal(ra, g["ra"]) and np.array_equal(dec, g["dec"])
    assert np.array_equal(amps, g["amps"])
    assert np.array_equal(utils.get_flux_from_ptsrc_amp(amps, g["freqs"], -2.7), g["fluxes"])
    assert np.array_equal(utils.get_flux_from_ptsrc_amp(amps, g["freqs"], -2.7, curv_ptsrc=0.3),
                          g["fluxes_curv"])
    for key, kw in [("hex34", dict(hex_spec=(3, 4))), ("hex47", dict(hex_spec=(4, 7))),
                    ("hexrows", dict(ants_per_row=[2, 3, 2], d=10.0))]:
        ants = utils.build_hex_array(**kw)
        assert np.array_equal(np.array([ants[k] for k in ants]), g[key])
        assert list(ants.keys()) == list(range(len(ants)))
    assert [len(utils.build_hex_array(h)) for h in ((3, 4), (4, 7), (7, 13), (11, 21))] == \
        [10, 37, 127, 331]
    freqs, times = g["freqs"], g["times"]
    Fb, kf, kt = utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 2, Lfreq=freqs[-1] - freqs[0], Ltime=times[-1] - times[0],
        freq0=freqs[0], time0=times[0], shape0=(freqs.size, times.size), positive_only=True)
    assert np.array_equal(Fb, g["Fb"]) and np.array_equal(kf, g["kf"]) and np.array_equal(kt, g["kt"])
    Fb2, kf2, kt2 = utils.partial_fourier_basis_2d_from_nmax(
        freqs, times, 2, 1, Lfreq=25.0, Ltime=0.4, positive_only=False)
    assert np.array_equal(Fb2, g["Fb2"]) and np.array_equal(kf2, g["kf2"])
    assert np.array_equal(kt2, g["kt2"])


def test_flatten_reconstruct_extract_match_reference():
    g = load_golden("gains.npz")
    ap = _pairs(g["antpairs"])
    assert np.array_equal(utils.flatten_vector(g["v"]), g["flat"])
    assert np.array_equal(utils.reconstruct_vector(g["flat"], g["v"].shape), g["recon"])
    assert np.array_equal(utils.extract_vis_from_sim(g["ants"], ap, g["simvis"]), g["extracted"])
    # reduced index sets round-trip (utils.py:21-60)
    idx = np.arange(0, 2 * g["v"].size, 3)
    red = utils.flatten_vector(g["v"], reduced_idxs=idx)
    back = utils.reconstruct_vector(red, g["v"].shape, reduced_idxs=idx)
    full = np.zeros(2 * g["v"].size)
    full[idx] = g["flat"][idx]
    assert np.array_equal(back, (full[: full.size // 2] + 1j * full[full.size // 2:]).reshape(
        g["v"].shape))


def test_sh_parameter_packing_matches_reference():
    g = load_golden("sh_algebra.npz")
    lmax = int(g["lmax"])
    ems, ells, idx = sh_sampler.get_em_ell_idx(lmax)
    assert np.array_equal(ems, g["ems"]) and np.array_equal(ells, g["ells"])
    assert np.array_equal(idx, g["idx"])
    assert np.array_equal(sh_sampler.alms2healpy(g["packed"], lmax), g["hp_modes"])
    assert np.array_equal(sh_sampler.healpy2alms(g["hp_modes"]), g["packed"])
    g10 = load_golden("sh_layout_lmax10.npz")
    ems, ells, idx = sh_sampler.get_em_ell_idx(10)
    assert np.array_equal(ems, g10["ems"]) and np.array_equal(ells, g10["ells"])
    assert sum(1 for m in ems if m == 0) == 11          # hydra/tests/test_sh_sampler.py:9-19


class _Comm:
    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def Get_rank(self):
        return self.rank

    def Get_size(self):
        return self.size


def test_freqs_times_for_worker_grid():
    """utils.py:313-371 against the reference's own output (tests/golden/worker_map.npz) for the
    product helper and the oracle's chunk_grid: workers on an fchunks x tchunks grid, row-major;
    every (freq, time) sample belongs to exactly one worker when the grid is full."""
    from oracle import helpers
    g = load_golden("worker_map.npz")
    freqs, times = g["freqs"], g["times"]
    for nworkers, fchunks, tchunks in g["cases"].tolist():
        seen = np.zeros((freqs.size, times.size), dtype=int)
        for rank in range(nworkers):
            fi, ti, wmap = utils.freqs_times_for_worker(_Comm(rank, nworkers), freqs, times,
                                                        fchunks, tchunks)
            ref = helpers.chunk_grid(nworkers, freqs.size, times.size, fchunks, tchunks)
            key = f"n{nworkers}_f{fchunks}_t{tchunks}_r{rank}"
            assert np.array_equal(fi, g[key + "_fi"]) and np.array_equal(ti, g[key + "_ti"])
            assert sorted(wmap) == list(range(nworkers)) == sorted(ref)
            for w in range(nworkers):
                assert tuple(g[key + "_grid"][w]) == wmap[w][:2] == ref[w][:2]
                for m in (wmap, ref):
                    assert np.array_equal(m[w][2], g[key + f"_w{w}_f"])
                    assert np.array_equal(m[w][3], g[key + f"_w{w}_t"])
            assert wmap[rank][:2] == (rank // tchunks, rank % tchunks)
            assert np.array_equal(fi, wmap[rank][2]) and np.array_equal(ti, wmap[rank][3])
            seen[np.ix_(fi, ti)] += 1
        if nworkers == fchunks * tchunks:
            assert np.all(seen == 1)
        else:
            assert seen.max() == 1
    with pytest.raises(AssertionError, match="more workers"):
        utils.freqs_times_for_worker(_Comm(0, 5), freqs, times, 2, 2)


def test_get_beam_interp_shape():
    from hydra_amd import beams

    class OldUVBeam:
        future_array_shapes = False

    class NewUVBeam:
        future_array_shapes = True
    got = [utils.get_beam_interp_shape(b) for b in (beams.GaussianBeam(diameter=14.0),
                                                    beams.UniformBeam(), OldUVBeam(), NewUVBeam(),
                                                    object())]
    # the reference's answers for (AnalyticBeam, BeamInterface, old UVBeam, new UVBeam, other)
    assert got == load_golden("worker_map.npz")["interp_shape"].tolist()


@pytest.mark.gpu
def test_convert_to_tops_matches_oracle_geometry():
    from oracle import geometry
    rng = np.random.RandomState(2)
    ra, dec = rng.uniform(0, 2 * np.pi, 50), rng.uniform(-1.5, 1.5, 50)
    lsts = np.linspace(0.0, 6.0, 9)
    lat = -30.7215 * np.pi / 180.0
    crd = geometry.point_source_crd_eq(ra, dec)
    for precision, tol in ((2, 1e-15), (1, 1e-15)):
        tx, ty, tz = utils.convert_to_tops(ra, dec, lsts, lat, precision=precision)
        assert tx.shape == (lsts.size, ra.size)
        for t, lst in enumerate(lsts):
            m = geometry.eci_to_enu_matrix(lst, lat)
            if precision == 1:
                m = m.astype(np.float32)
            want = np.dot(m, crd)
            assert np.max(np.abs(np.stack([tx[t], ty[t], tz[t]]) - want)) <= 1e-7 * (precision == 1) + tol
    # zenith: a source at ra = lst, dec = latitude is straight up
    tx, ty, tz = utils.convert_to_tops([1.2], [lat], [1.2], lat, precision=2)
    assert abs(tx[0, 0]) < 1e-15 and abs(ty[0, 0]) < 1e-15 and abs(tz[0, 0] - 1.0) < 1e-15


def test_config_flags_match_the_reference_parser():
    """hydra_amd.config.get_config: every flag of hydra/config.py with the same dest, type,
    nargs, action and default (tests/golden/config_flags.json, parsed from the reference)."""
    import json
    import os

    from conftest import GOLDEN
    from hydra_amd import config
    ref = json.load(open(os.path.join(GOLDEN, "config_flags.json")))
    assert len(ref) == 59
    parser = config.build_parser()
    mine = {a.option_strings[0]: a for a in parser._actions if a.option_strings
            and a.option_strings[0] != "-h"}
    assert set(mine) == {r["flag"] for r in ref}
    args = config.get_config([])
    for r in ref:
        a = mine[r["flag"]]
        assert a.dest == r["dest"], r["flag"]
        if r["action"] == "store_true":
            assert a.const is True and a.default is False and getattr(args, a.dest) is False
            continue
        assert a.type.__name__ == r["type"], r["flag"]
        assert a.nargs == r["nargs"], r["flag"]
        want, got = r["default"], getattr(args, a.dest)
        if isinstance(want, list):
            assert list(got) == want, r["flag"]
        elif isinstance(want, float):
            assert got == pytest.approx(want, rel=0, abs=0), r["flag"]
        else:
            assert got == want, r["flag"]
    # a reference-style command line
    a = config.get_config("--ptsrc --gains --hex-array 4 7 --Nptsrc 50 --solver mpicg "
                          "--ra-bounds 0.1 0.9 --gain-prior-zero-mode-std 0.01".split())
    assert a.sample_ptsrc and a.sample_gains and not a.sample_sh
    assert a.hex_array == [4, 7] and a.Nptsrc == 50 and a.solver_name == "mpicg"
    assert a.ra_bounds == [0.1, 0.9] and a.gain_prior_zero_mode_std == 0.01


def test_radiometer_eq_matches_the_reference():
    """sh_sampler.radiometer_eq against the reference's outputs (tests/golden/radiometer.npz,
    made by make_golden_radiometer.py from hydra/sh_sampler.py:497-522), bit for bit."""
    from hydra_amd import sh_sampler
    g = load_golden("radiometer.npz")
    for tag in ("a", "b", "c"):
        ants = np.arange(int(g[tag + "_nant"]))
        got = sh_sampler.radiometer_eq(g[tag + "_autos"], ants, 10.7, 97.6e3)
        assert got.shape == g[tag + "_cross"].shape and np.array_equal(got, g[tag + "_cross"])
        got = sh_sampler.radiometer_eq(g[tag + "_autos"], ants, 10.7, 97.6e3, Nnights=3,
                                       include_autos=True)
        assert np.array_equal(got, g[tag + "_with_autos"])


def test_padding_and_gain_prior_helpers_match_the_reference(tmp_path):
    """utils.extend_coords_with_padding, gain_prior_pspec_sqrt and load_gain_model against the
    reference's outputs (tests/golden/utils3.npz, made by make_golden_utils3.py), bit for bit."""
    g = load_golden("utils3.npz")
    freqs, lsts = g["freqs"], g["lsts"]
    assert np.array_equal(utils.extend_coords_with_padding(freqs, pad=[2, 3]), g["pad_f"])
    assert np.array_equal(utils.extend_coords_with_padding(lsts, pad=[0, 4]), g["pad_t"])
    assert np.array_equal(utils.extend_coords_with_padding(lsts), g["pad_none"])
    assert np.array_equal(utils.gain_prior_pspec_sqrt(lsts, freqs, 0.1), g["ps_flat"])
    got = utils.gain_prior_pspec_sqrt(lsts, freqs, 0.1, gain_prior_sigma_frate=0.5,
                                      gain_prior_sigma_delay=80.0, gain_prior_zeropoint_std=0.3,
                                      frate0=0.1, delay0=-20.0)
    assert np.array_equal(got, g["ps_taper"])
    assert np.array_equal(utils.gain_prior_pspec_sqrt(lsts, freqs, 2.0, gain_prior_sigma_frate=1.5),
                          g["ps_frate"])
    path = str(tmp_path / "g.npy")
    np.save(path, g["gain_model"])
    assert np.array_equal(utils.load_gain_model(path, lst_pad=[1, 2], freq_pad=[3, 0]),
                          g["gain_padded"])
    assert np.array_equal(utils.load_gain_model(path), g["gain_plain"])
