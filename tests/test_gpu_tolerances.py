"""What the caller's RTOL / ATOL do (include/metrotrpl_amd.h, mtrpl_solve_desc.rtol / .atol; reference
semantics: forward_solver.py:163-165 hands both to LSODA).  The engine's contract is the converged trace,
so a loose RTOL changes nothing, a tight one tightens the integration below the policy's own floor, a
request below the noise floor is raised to it and reported, and ATOL bounds the absolute term from above."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _solve(z, sel, rtol=None, atol=None):
    from metrotrpl_amd import batch
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    return batch.solve_batch(z["states"][sel], idx, z["units"], z["iniPar"][sel], z["thickness"][sel], 128, z["tSteps"],
                             RTOL=rtol, ATOL=atol, return_stats=True)


def _err(sig, ref, lo_dec, hi_dec):
    with np.errstate(all="ignore"):
        dec = np.floor(np.log10(ref[:, :1] / ref))
        rel = np.abs(sig - ref) / ref
    m = (dec >= lo_dec) & (dec < hi_dec)
    return float(rel[m].max())


def test_tight_rtol_is_honoured_loose_rtol_is_not_a_licence():
    z = np.load(os.path.join(G, "tail_decades.npz"))
    sel = np.arange(0, 48)
    ref = z["sig"][sel]
    base, st0, stats0 = _solve(z, sel)                 # defaults 1e-7 / 1e-10
    loose, st1, stats1 = _solve(z, sel, rtol=1e-3)     # looser than the default: same integration
    tight, st2, stats2 = _solve(z, sel, rtol=1e-9)     # 0.1 * 1e-9 = 1e-10 < the policy floor of 3e-10: honoured
    assert np.all(st0 == 0) and np.all(st1 == 0) and np.all(st2 == 0)
    np.testing.assert_array_equal(loose, base)
    np.testing.assert_array_equal(stats1[:, 0], stats0[:, 0])
    e_base, e_tight = _err(base, ref, 0, 4), _err(tight, ref, 0, 4)
    print(f"first four decades: default {e_base:.2e} ({stats0[:, 0].mean():.0f} steps), RTOL=1e-9 {e_tight:.2e} "
          f"({stats2[:, 0].mean():.0f} steps)")
    assert e_base <= 1e-6
    assert e_tight < 0.5 * e_base and stats2[:, 0].mean() > 1.15 * stats0[:, 0].mean()
    # below the noise floor of this system: raised to it, and the solve says so
    floor_, st3, stats3 = _solve(z, sel[:8], rtol=1e-13)
    assert np.all(st3 == 2), st3  # MTRPL_SOLVE_OK_TOL_CLAMPED
    assert np.all(np.isfinite(floor_)) and _err(floor_, ref[:8], 0, 4) <= 1e-6


def test_atol_bounds_the_absolute_term():
    """ATOL is an upper bound of the densities' absolute tolerance.  Measured with the far-tail relaxations off
    (far_factor = deep_factor = 1: by default the local tolerance is loosened again nine and eleven decades down, where
    nothing is resolved, and would mask what the absolute term does there)."""
    from metrotrpl_amd import _lib as L
    ctx = L.default_context()
    ctx.set_option("far_factor", 1.0)
    ctx.set_option("deep_factor", 1.0)
    try:
        _atol_case()
    finally:
        ctx.set_option("far_factor", 3.0)
        ctx.set_option("deep_factor", 1000.0)
    assert ctx.get_option("far_factor") == 3.0 and ctx.get_option("deep_factor") == 1000.0


def _atol_case():
    z = np.load(os.path.join(G, "tail_decades.npz"))
    sel = np.arange(256, 256 + 48)  # the config-3 geometry: thin films decay through many decades
    ref = z["sig"][sel]
    base, st0, stats0 = _solve(z, sel)
    same, st1, stats1 = _solve(z, sel, atol=1e-3)      # looser than the rule (about 3e-22 nm^-3): nothing changes
    deep, st2, stats2 = _solve(z, sel, atol=1e-26)     # a smaller ATOL resolves a still deeper tail
    assert np.all(st0 == 0) and np.all(st2 == 0)
    np.testing.assert_array_equal(same, base)
    e_base, e_deep = _err(base, ref, 9, 12), _err(deep, ref, 9, 12)
    print(f"decades 9-11: default {e_base:.2e} ({stats0[:, 0].mean():.0f} steps), ATOL=1e-26 {e_deep:.2e} "
          f"({stats2[:, 0].mean():.0f} steps)")
    assert stats2[:, 0].mean() > 1.005 * stats0[:, 0].mean(), (stats2[:, 0].mean(), stats0[:, 0].mean())
    assert e_deep < 0.9 * e_base  # (two-fold with round 2's ten times larger default floor)
    # the field keeps its own absolute floor: a tiny ATOL does not send the integrator crawling
    assert stats2[:, 0].max() < 3 * stats0[:, 0].max()


def test_deep_zone_relaxation_touches_no_sample_above_it():
    """Eleven decades below its first sample a solve's local tolerance is relaxed by deep_factor (default 1000) and the
    reference's max_step is lifted (hmax_deep).  The zone is entered at an output sample, so every sample up to and
    including the first one below 1e-11 x the first is the SAME BITS with and without the relaxation; what comes after
    stays a decaying tail of that size, and solves that never get there do not change at all."""
    from metrotrpl_amd import _lib as L
    z = np.load(os.path.join(G, "tail_decades.npz"))
    sel = np.arange(256, 256 + 96)  # the config-3 geometry: thin films decay through many decades
    ctx = L.default_context()
    shipped, st0, stats0 = _solve(z, sel)
    ctx.set_option("deep_factor", 1.0)
    try:
        plain, st1, stats1 = _solve(z, sel)
    finally:
        ctx.set_option("deep_factor", 1000.0)
    assert np.all(st0 == 0) and np.all(st1 == 0)
    n_deep = 0
    for a, b in zip(shipped, plain):
        below = np.flatnonzero(np.abs(b) < 1e-11 * abs(b[0]))
        if below.size == 0:
            np.testing.assert_array_equal(a, b)
            continue
        n_deep += 1
        k = below[0]
        np.testing.assert_array_equal(a[:k + 1], b[:k + 1])
        assert np.all(np.abs(a[k + 1:]) < 1e-10 * abs(a[0]))
    assert n_deep >= 10, n_deep  # the fixture does reach down there
    assert stats0[:, 0].sum() < stats1[:, 0].sum()
    print(f"{n_deep} of {len(sel)} traces reach the zone; steps {stats1[:, 0].mean():.0f} -> {stats0[:, 0].mean():.0f}")


def test_atol_zero_is_integrated_like_the_reference_does():
    """ATOL = 0 (or a denormal) is legal for the reference's LSODA.  The fp32 error weights 1/(atol + rtol |y|) of
    the controller must stay finite next to a component that is (or has gone to) zero -- ADVICE r3: they flushed to
    1/0, every attempt was booked as a Newton failure and the solve stopped early with a zero-filled tail."""
    from metrotrpl_amd import batch, workloads as W
    z = np.load(os.path.join(G, "tail_decades.npz"))
    sel = np.arange(0, 16)
    ref = z["sig"][sel]
    base, st0, stats0 = _solve(z, sel)
    for atol in (0.0, 1e-45):
        sig, st, stats = _solve(z, sel, atol=atol)
        assert np.all(st == 0), (atol, st)
        assert np.all(np.isfinite(sig)) and np.all(stats[:, 5] == 0)  # no Newton failure
        assert _err(sig, ref, 0, 9) <= 1e-6
        assert stats[:, 0].max() < 3 * stats0[:, 0].max()
    # the traps model starts from a trapped density of exactly zero
    team = np.load(os.path.join(G, "team_cases.npz"))
    idx = {n: i for i, n in enumerate(str(team["traps_names"]).split(","))}
    t0, s0, _ = batch.solve_batch(team["traps_states"][:2], idx, team["traps_units"], team["traps_ini"][:2],
                                  team["traps_thickness"][:2], 128, team["traps_t"], model="traps", return_stats=True)
    t1, s1, x1 = batch.solve_batch(team["traps_states"][:2], idx, team["traps_units"], team["traps_ini"][:2],
                                   team["traps_thickness"][:2], 128, team["traps_t"], model="traps", ATOL=0.0,
                                   return_stats=True)
    assert np.all(s0 == 0) and np.all(s1 == 0) and np.all(x1[:, 5] == 0)
    m = np.abs(t0) >= 1e-4 * np.abs(t0[:, :1])
    assert np.max(np.abs(t1 - t0)[m] / np.abs(t0)[m]) < 1e-6
