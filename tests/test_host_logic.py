"""CPU-only: host-side packing / validation mirrors forward_solver.solve's
argument checks (forward_solver.py:100-117,:140,:203,:225; sim_utils.py:269-270)."""
import numpy as np
import pytest

from metrotrpl_amd import batch, workloads as W
from metrotrpl_amd.sim_utils import Grid


def test_grid_matches_reference_fields():
    t = np.linspace(0, 100, 1001)
    g = Grid(1000, 100, t, 4)
    assert g.dx == 10.0 and g.nt == 1000 and g.final_time == 100 and g.start_time == 0
    np.testing.assert_array_equal(g.xSteps, np.linspace(5.0, 995.0, 100))
    import sys
    assert g.min_y == sys.float_info.min
    with pytest.raises(ValueError, match="times must start at t=0"):
        Grid(1000, 100, t + 1, 4)


def test_workloads_are_seeded_and_approved():
    a, fa = W.draw_states(32)
    b, fb = W.draw_states(32)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(fa, fb)
    ix = W.indexes()
    assert np.all(a[:, ix["p0"]] > a[:, ix["n0"]])
    assert np.all(np.abs(np.log10(a[:, ix["tauN"]]) - np.log10(a[:, ix["tauP"]])) <= 2)
    for name, (lo, hi) in W.PRIOR_BOX.items():
        assert np.all((a[:, ix[name]] > lo) & (a[:, ix[name]] < hi)), name
    assert np.all((fa >= 2.12e11) & (fa <= 2.75e13))
    # prefix property: config2(16) is the head of config2(1024) (golden subset = first 16)
    np.testing.assert_array_equal(W.config2(16)["states"], W.config2(64)["states"][:16])
    lad = W.ladder(64)
    assert lad[0] == 1 and abs(lad[-1] - 1000) < 1e-9 and np.all(np.diff(lad) > 0)


def test_pack_and_descs():
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    d, ini = batch.build_batch(np.stack([s0, s0]), ix, un, np.array(W.STAUB_FLUENCES)[:2], [311.0, 2000.0], 128, 1001)
    assert d.shape == (2,) and d["nx"][0] == 128 and d["nt"][1] == 1001
    np.testing.assert_array_equal(d["thickness"], [311.0, 2000.0])
    np.testing.assert_array_equal(d["out_off"], [0, 1001])
    assert d["par"][0][1] == s0[ix["p0"]] * un[ix["p0"]]
    assert d["par"][0][11] == 10 and d["par"][0][12] == 300
    np.testing.assert_array_equal(d["ini"][0], [2.12e11, 6e4, 1])
    assert d["rtol"][0] == 1e-7 and d["atol"][0] == 1e-10
    # density mode packs nx values per solve
    d2, ini2 = batch.build_batch(s0[None, :], ix, un, np.full((1, 64), 1e15), 311.0, 64, 11, ini_mode="density")
    assert ini2.shape == (64,) and d2["ini_mode"][0] == 0 and d2["ini_off"][0] == 0


def test_argument_errors_match_reference():
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()[None, :]
    with pytest.raises(ValueError, match="Expected 64 initial densities"):
        batch.build_batch(s0, ix, un, np.ones((1, 63)), 311.0, 64, 11, ini_mode="density")
    with pytest.raises(ValueError, match="Expected only fluence"):
        batch.build_batch(s0, ix, un, np.ones((1, 4)), 311.0, 64, 11, ini_mode="fluence")
    with pytest.raises(ValueError, match="Invalid ini_mode"):
        batch.build_batch(s0, ix, un, np.ones((1, 3)), 311.0, 64, 11, ini_mode="other")
    with pytest.raises(ValueError, match="Invalid model"):
        batch.build_batch(s0, ix, un, np.ones((1, 3)), 311.0, 64, 11, model="nope")
    with pytest.raises(NotImplementedError):
        batch.build_batch(s0, ix, un, np.ones((1, 3)), 311.0, 64, 11, meas="something else")
    with pytest.raises(KeyError):
        batch.pack_params(s0, {"n0": 0})
