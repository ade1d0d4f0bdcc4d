"""GPU edge cases through the C ABI: empty and ragged batches, tiny grids, a single
sample time, mixed measurement types and grid sizes in one launch, integrator give-up."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx():
    from metrotrpl_amd import _lib as L
    return L, L.default_context()


def test_empty_batch_is_a_noop():
    L, ctx = _ctx()
    assert L.lib().mtrpl_solve_batch(ctx.handle, 0, None, None, 0, None, 0, None, 0, None, None) == 0
    assert L.lib().mtrpl_rhs_batch(ctx.handle, 0, 16, 0, None, None, None, None) == 0


def test_ragged_batch_matches_individual_calls():
    """Solves with different nx, thickness, nt, time grids, meas types and ini modes in ONE launch."""
    from metrotrpl_amd import batch, workloads as W
    L, ctx = _ctx()
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    grids = [np.linspace(0, 50, 26), np.linspace(0, 300, 301), np.array([0.0]), np.linspace(0, 10, 5) ** 2 / 10,
             np.linspace(0, 120, 61)]
    nxs = [16, 128, 64, 3, 100]
    meas = ["TRPL", "TRTS", "TRPL", "TRPL", "TRTS"]
    thick = [311.0, 2000.0, 500.0, 100.0, 311.0]
    singles = []
    for k in range(5):
        ini = [[1.92e12, 6e4, 1]] if k != 2 else [np.full(nxs[k], 1e15)]
        singles.append(batch.solve_batch(s0[None, :], ix, un, ini, thick[k], nxs[k], grids[k], meas=meas[k],
                                         ini_mode="fluence" if k != 2 else "density")[0])
    # pack all five into one descriptor array by hand
    par = batch.pack_params((s0 * un)[None, :], ix)
    d = np.zeros(5, dtype=L.SOLVE_DESC_DTYPE)
    t_all = np.concatenate(grids)
    t_off = np.concatenate([[0], np.cumsum([len(g) for g in grids])[:-1]])
    ini = np.full(64, 1e15)
    for k in range(5):
        d[k]["par"] = par[0]
        d[k]["thickness"], d[k]["hmax"], d[k]["rtol"], d[k]["atol"] = thick[k], 4.0, 1e-7, 1e-10
        d[k]["min_y"] = np.finfo(float).tiny
        d[k]["ini"] = [1.92e12, 6e4, 1]
        d[k]["nx"], d[k]["nt"], d[k]["meas"] = nxs[k], len(grids[k]), L.MEAS[meas[k]]
        d[k]["model"], d[k]["ini_mode"] = 0, 1 if k != 2 else 0
        d[k]["ini_off"], d[k]["t_off"], d[k]["out_off"] = 0, t_off[k], t_off[k]
    out = np.empty(len(t_all))
    status = np.zeros(5, dtype=np.int32)
    ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, 5, d.ctypes.data_as(C.c_void_p), L.dptr(ini), 64, L.dptr(t_all),
                                        len(t_all), L.dptr(out), len(out), L.iptr(status), None))
    assert np.all(status == 0)
    # every nx here is <= 128: alone or packed next to the others a solve runs on the same wave kernel (the one- and
    # two-per-SIMD builds give the same bits), so the batch reproduces the individual calls exactly
    for k in range(5):
        np.testing.assert_array_equal(out[t_off[k]:t_off[k] + len(grids[k])], singles[k])
    assert len(singles[2]) == 1 and singles[2][0] > 0  # nt = 1: just the t=0 sample
    # the same with the kernel pinned to the fp64-block wave kernel and to the team kernel
    for variant in (2, 1):
        ctx.set_option("kernel_variant", variant)
        try:
            out2 = np.empty(len(t_all))
            ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, 5, d.ctypes.data_as(C.c_void_p), L.dptr(ini), 64, L.dptr(t_all),
                                                len(t_all), L.dptr(out2), len(out2), L.iptr(status), None))
            k = 1  # the nx = 128 solve runs on the same kernel alone or in the batch
            one = batch.solve_batch(s0[None, :], ix, un, [[1.92e12, 6e4, 1]], thick[k], nxs[k], grids[k], meas=meas[k])[0]
        finally:
            ctx.set_option("kernel_variant", 0)
        np.testing.assert_array_equal(out2[t_off[k]:t_off[k] + len(grids[k])], one)


def test_descriptor_validation_errors():
    from metrotrpl_amd import batch, workloads as W
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    with pytest.raises(ValueError, match="times must start at t=0"):
        batch.solve_batch(s0[None, :], ix, un, [[1e12, 6e4, 1]], 311.0, 16, np.linspace(1, 10, 5))
    with pytest.raises(NotImplementedError):
        batch.solve_batch(s0[None, :], ix, un, [[1e12, 6e4, 1]], 311.0, 1024, np.linspace(0, 10, 5))
    with pytest.raises(ValueError):
        batch.solve_batch(s0[None, :], ix, un, [[1e12, 6e4, 1]], -5.0, 16, np.linspace(0, 10, 5))


def test_integrator_give_up_zero_fills_like_the_reference():
    """forward_solver.py:168 + :190-192: samples after the last solved time come from a zeroed
    state; with max_steps exhausted the status says so."""
    from metrotrpl_amd import batch, workloads as W
    L, ctx = _ctx()
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    t = np.linspace(0, 2000, 101)
    try:
        ctx.set_option("max_steps", 40)
        sig, status, stats = batch.solve_batch(s0[None, :], ix, un, [[1.92e12, 6e4, 1]], 311.0, 64, t, return_stats=True)
    finally:
        ctx.set_option("max_steps", 2000000)
    assert status[0] == 1 and stats[0, 0] == 40
    full = batch.solve_batch(s0[None, :], ix, un, [[1.92e12, 6e4, 1]], 311.0, 64, t)[0]
    k = int(np.argmax(sig[0] != full))  # first sample that differs = first unsolved time
    assert k > 0
    np.testing.assert_array_equal(sig[0, :k], full[:k])
    # zeroed state -> PL = -ks n0 p0 L 1e23 < min_y -> clamped to min_y (float_info.min) from there on
    assert np.all(sig[0, k:] == np.finfo(float).tiny)


def test_units_none_and_list_state():
    """solve() with units=None and a python list for `state` (Tests/test_eval_trial_move.py:65,77)."""
    from metrotrpl_amd.forward_solver import solve
    from metrotrpl_amd.sim_utils import Grid
    from metrotrpl_amd import workloads as W
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    g = Grid(311, 32, np.linspace(0, 20, 11), 4)
    a = solve([1.92e12, 6e4, 1], g, list(s0 * un), ix, ini_mode="fluence")
    b = solve([1.92e12, 6e4, 1], g, s0, ix, units=un, ini_mode="fluence")
    np.testing.assert_array_equal(a, b)


def test_odd_and_tiny_grids_against_the_oracle():
    """nx that is odd, tiny or not a multiple of the lanes' two nodes (2, 3, 5, 33, 101, 127) and the
    wave kernel's maximum (128): padded lanes must stay inert.  Oracle: SciPy LSODA at 1e-10/1e-14
    (converged to ~1e-8 over the first decades)."""
    from metrotrpl_amd import batch, workloads as W
    from oracle import oracle as O
    O.build()
    ix, un, s0 = W.indexes(), W.units(), W.initial_state()
    t = np.linspace(0, 100, 51)
    for nx in (2, 3, 5, 33, 101, 127, 128):
        for thick, meas in ((311.0, "TRPL"), (2000.0, "TRTS")):
            ini = np.array([1.92e12, 6e4, 1.0])
            sig = batch.solve_batch(s0[None, :], ix, un, ini[None, :], thick, nx, t, meas=meas)[0]
            g = O.Grid(thick, nx, t, 4.0)
            ref = O.solve(ini, g, s0.copy(), ix, meas=meas, units=un, ini_mode="fluence", RTOL=1e-10, ATOL=1e-14)
            m = ref >= 1e-3 * ref[0]
            err = np.max(np.abs(sig[m] - ref[m]) / ref[m])
            assert err < 1e-6, (nx, thick, meas, err)


def test_the_library_issues_a_launch_as_given():
    """Round 3's device-side issue ordering (a cost model of the descriptors, two kernels in front of every launch) is
    gone -- it cost more than it bought.  A launch runs its solves in the order of the descriptor array; a
    caller who knows better (LikelihoodEngine: the cost measured one Metropolis iteration earlier) orders the array."""
    L, ctx = _ctx()
    with pytest.raises(Exception):
        ctx.get_option("order_min_batch")
    with pytest.raises(Exception):
        ctx.set_option("order_min_batch", 0)
    assert not hasattr(L.lib(), "mtrpl_launch_order")


def test_every_option_the_header_documents_exists():
    """include/metrotrpl_amd.h lists the integrator options in the comment above mtrpl_set_option: each of them is
    a key the library knows (get works, set of the same value works), and an unknown key is an error."""
    import re
    from metrotrpl_amd import _lib as L
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "metrotrpl_amd.h")).read()
    block = hdr[hdr.index("/* Integrator options"):hdr.index("int mtrpl_set_option")]
    keys = []
    for line in block.splitlines():
        m = re.match(r"^ \*   ([a-z_0-9]+(?:, [a-z_0-9/]+)*)\s{2,}", line)
        if m:
            for k in m.group(1).split(", "):
                keys.extend(k.split("/"))
    keys = [k if not k.startswith("_") else "nq_bias" + k for k in keys]
    assert len(keys) >= 30, keys
    ctx = L.default_context()
    for k in keys:
        v = ctx.get_option(k)
        ctx.set_option(k, v)
        assert ctx.get_option(k) == v, k
    with pytest.raises(Exception):
        ctx.get_option("no_such_option")
