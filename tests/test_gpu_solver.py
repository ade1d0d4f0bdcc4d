"""GPU parity tests for the forward solver: HIP path (through the C ABI) vs
golden vectors captured from the reference and vs the CPU oracle.

Accuracy gate (SURVEY.md 8d / DESIGN.md): max |sig_gpu - sig_ref| / |sig_ref|
<= 1e-6 against the reference at tight tolerance (RTOL=1e-10, ATOL=1e-14), over
the samples within 4 decades of the trace's first sample -- the range over
which the tight reference is itself converged to ~1e-8 (below that LSODA's own
atol floor dominates: measured 2e-5 at 8 decades between 1e-10/1e-14 and
1e-12/1e-18 runs).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
REL_TOL = 1e-6
DECADES = 1e-4


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "solve.npz"))


@pytest.fixture(scope="module")
def conv():
    """Oracle at RTOL=1e-12/ATOL=1e-17 (tests/golden/make_converged.py)."""
    return np.load(os.path.join(G, "solve_converged.npz"))


@pytest.fixture(scope="module")
def rhsg():
    return np.load(os.path.join(G, "rhs.npz"))


def _case(gold, key):
    names, meas, solver, model, ini_mode = [str(x) for x in gold[key + "__meta"]]
    thickness, nx, hmax, rtol, atol, min_y = gold[key + "__geom"]
    return dict(names=names.split(","), meas=meas, solver=solver, model=model, ini_mode=ini_mode,
                thickness=float(thickness), nx=int(nx), hmax=float(hmax), rtol=float(rtol), atol=float(atol),
                min_y=None if min_y < 0 else float(min_y), tSteps=gold[key + "__tSteps"],
                state=gold[key + "__state"], units=gold[key + "__units"], iniPar=gold[key + "__iniPar"],
                sig=gold[key + "__sig"])


def _relerr(sig, ref, floor=DECADES):
    m = np.abs(ref) >= floor * np.abs(ref[0])
    return np.max(np.abs(sig[m] - ref[m]) / np.abs(ref[m])), int(m.sum())


def test_rhs_matches_reference(rhsg):
    """forward_solver.py:332-372 on random states (a4).  The kernel keeps the
    reference's operation order; hipcc may contract a*b+c, so agreement is to
    rounding of the individual flux terms, not bitwise."""
    from metrotrpl_amd import _lib as L_, batch
    from metrotrpl_amd.forward_solver import q_C, eps0
    ctx = L_.default_context()
    n = 0
    for k in range(int(rhsg["n_cases"])):
        a = rhsg[f"std{k}_args"]
        L = int(a[0])
        if L > 128:
            continue
        par = np.array([[a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12],
                         q_C / (a[13] * eps0), a[14], 0, 0, 1]])
        y = rhsg[f"std{k}_y"]
        ref = rhsg[f"std{k}_dy"]
        # dN, dP are differences of fluxes ~1e3x larger than the result: scale by the flux magnitude
        scale = np.max(np.abs(ref[:2 * L])) + 1e-300
        for variant in (2, 1):  # the single-wave layout and the team layout (boundary node via LDS)
            ctx.set_option("kernel_variant", variant)
            try:
                dy = batch.rhs_batch(par, [a[1]], y[None, :], L)[0]
            finally:
                ctx.set_option("kernel_variant", 0)
            np.testing.assert_allclose(dy[:2 * L], ref[:2 * L], rtol=1e-9, atol=1e-12 * scale)
            np.testing.assert_allclose(dy[2 * L:], ref[2 * L:], rtol=1e-12, atol=1e-14 * np.max(np.abs(ref[2 * L:])))
        n += 1
    assert n >= 4


def test_newton_linear_solver():
    """E-elimination + cyclic reduction == dense solve of
    (I - cJ) x = b."""
    from metrotrpl_amd import _lib as L, batch, workloads as W
    ctx = L.default_context()
    from tests.dense_model import Problem
    ix, un = W.indexes(), W.units()
    c2 = W.config2(6)
    rng = np.random.default_rng(0)
    worst_f32 = 0.0
    for nx in (128, 100, 37):
        for k in range(6):
            th = 2000.0 if k % 2 else 311.0
            prob = Problem(c2["states"][k], un, ix, c2["iniPar"][k], th, nx)
            y = prob.y0.copy()
            y[:2 * nx] *= 10 ** rng.uniform(-0.3, 0.3, 2 * nx)
            y[2 * nx:] += rng.normal(0, 1e-5, nx + 1)
            par = batch.pack_params((c2["states"][k] * un)[None, :], ix)
            for c in (1e-6, 1e-2, 1.0):
                M = np.eye(y.size) - c * prob.jac(y)
                b = rng.normal(0, 1, y.size) * np.abs(y)
                b[2 * nx] = 0.0
                b[-1] = 0.0
                x_ref = np.linalg.solve(M, b)
                # one wave, fp64 blocks | one wave, fp32-stored blocks + fp32 re-solve (the default)
                for variant in (2, 3):
                    ctx.set_option("kernel_variant", variant)
                    try:
                        x = batch.newton_probe(par, [prob.dx], [c], y[None, :], b[None, :], nx)[0]
                    finally:
                        ctx.set_option("kernel_variant", 0)
                    res = np.linalg.norm(M @ x - b) / np.linalg.norm(b)
                    if variant == 3:
                        # blocks rounded to fp32 on their way into LDS, exchanged vectors and block products in fp32:
                        # an approximate inverse, good for a contraction of the simplified Newton iteration by three
                        # orders of magnitude per sweep
                        assert res < 1e-3, (variant, nx, k, c, res)
                        worst_f32 = max(worst_f32, res)
                        continue
                    assert res < 1e-6, (variant, nx, k, c, res)
                    np.testing.assert_allclose(x, x_ref, rtol=1e-5, atol=1e-7 * np.max(np.abs(x_ref)))
    print(f"fp32-stored blocks: worst relative residual {worst_f32:.1e}")


def _run(c, rtol=None, atol=None):
    from metrotrpl_amd import batch
    idx = {n: i for i, n in enumerate(c["names"])}
    sig, status, stats = batch.solve_batch(c["state"][None, :], idx, c["units"], c["iniPar"][None, :], c["thickness"],
                                           c["nx"], c["tSteps"], hmax=c["hmax"], meas=c["meas"], model=c["model"],
                                           ini_mode=c["ini_mode"], RTOL=c["rtol"] if rtol is None else rtol,
                                           ATOL=c["atol"] if atol is None else atol, min_y=c["min_y"],
                                           return_stats=True)
    return sig[0], int(status[0]), stats[0]


def test_golden_solves_tight(gold, conv):
    """Every std-model golden case with nx <= 128, called with the DEFAULT
    tolerances (1e-7 / 1e-10), against (a) the converged trace and (b) the
    reference's own tight-tolerance output wherever that output is itself
    converged (it is not for cfg2_11: 2.2e-6 from the converged trace)."""
    worst = 0.0
    n = 0
    for key in [str(k) for k in gold["order"]]:
        if not key.endswith("__tight"):
            continue
        c = _case(gold, key)
        if c["model"] != "std" or c["nx"] > 128 or c["min_y"] is not None:
            continue
        sig, status, stats = _run(c, rtol=1e-7, atol=1e-10)
        assert status == 0, key
        err, npts = _relerr(sig, c["sig"])
        cv = conv[key.replace("__tight", "__conv")]
        err_c, _ = _relerr(sig, cv)
        ref_gap, _ = _relerr(c["sig"], cv)
        print(f"{key}: vs tight ref={err:.2e} vs converged={err_c:.2e} (ref gap {ref_gap:.1e}) over {npts} pts; "
              f"steps={stats[0]} rhs={stats[1]} fac={stats[2]} jac={stats[3]} rej={stats[4]} nfail={stats[5]}")
        worst = max(worst, err_c)
        assert err_c <= REL_TOL, (key, err_c)
        if ref_gap <= 3e-7:
            assert err <= REL_TOL, (key, err)
        n += 1
    assert n >= 30
    print("worst rel err:", worst)


def test_golden_solves_default_reference(gold, conv):
    """Against the reference at ITS default tolerances (RTOL=1e-7 / ATOL=1e-10) the floor is the reference's
    own error: its default-tolerance traces are up to 1e-3 from the converged ones inside the 4-decade
    window (cfg2_11).  So every `__default` golden of configs 1, 2 and 3 is held to two bounds that cannot be
    loosened by picking cases: the GPU trace is within the gate of the CONVERGED trace, and it is at least
    as close to the reference's default output as that output is to the converged trace."""
    n = 0
    worst_gap = 0.0
    for key in [str(k) for k in gold["order"]]:
        if not key.endswith("__default") or not key.startswith(("cfg1", "cfg2", "cfg3")):
            continue
        c = _case(gold, key)
        ck = key.replace("__default", "__conv")
        if c["model"] != "std" or c["nx"] > 128 or c["min_y"] is not None or ck not in conv.files:
            continue
        sig, status, _ = _run(c)
        assert status == 0, key
        err_conv, _ = _relerr(sig, conv[ck])
        err_ref, _ = _relerr(sig, c["sig"])
        ref_gap, _ = _relerr(c["sig"], conv[ck])
        worst_gap = max(worst_gap, ref_gap)
        print(f"{key}: gpu vs converged {err_conv:.2e}, gpu vs ref default {err_ref:.2e}, ref default vs converged {ref_gap:.2e}")
        assert err_conv <= REL_TOL, (key, err_conv)
        assert err_ref <= 1.05 * ref_gap + 1e-6, (key, err_ref, ref_gap)
        n += 1
    assert n >= 12, n
    assert worst_gap > 1e-5  # the reference's default tolerance really is the floor here


def test_min_y_clamp(gold):
    """forward_solver.py:190-192, pin Tests/test_metropolis.py:199-266."""
    c = _case(gold, "tm_depl__tight")
    sig, status, _ = _run(c)
    ref = c["sig"]
    assert status in (0, 2)  # the case asks for RTOL=1e-10: below the noise floor, raised to it (status 2)
    assert np.min(sig) >= c["min_y"]
    np.testing.assert_array_equal(sig[-10:], c["min_y"])
    first_ref = int(np.argmax(ref == c["min_y"]))
    first = int(np.argmax(sig == c["min_y"]))
    assert abs(first - first_ref) <= 1
    k = min(first, first_ref)
    np.testing.assert_allclose(sig[:k], ref[:k], rtol=REL_TOL)


def test_batch_equals_single(gold):
    """A batch of different solves gives exactly what one-by-one calls give."""
    from metrotrpl_amd import batch, workloads as W
    ix, un = W.indexes(), W.units()
    c2 = W.config2(8)
    t = np.linspace(0, 300, 151)
    sig = batch.solve_batch(c2["states"], ix, un, c2["iniPar"], 2000.0, 128, t)
    for k in range(8):
        one = batch.solve_batch(c2["states"][k:k + 1], ix, un, c2["iniPar"][k:k + 1], 2000.0, 128, t)[0]
        np.testing.assert_array_equal(one, sig[k])


def test_oracle_cross_check_random_draws():
    """Config-2 draws beyond the committed golden subset, against the oracle
    (SciPy LSODA + C RHS) at RTOL=1e-12 / ATOL=1e-17, computed on this machine."""
    from metrotrpl_amd import batch, workloads as W
    from oracle import oracle as O
    ix, un = W.indexes(), W.units()
    c2 = W.config2(40)
    sel = np.arange(16, 40)
    t = np.linspace(0, 2000, 1001)
    sig, status, stats = batch.solve_batch(c2["states"][sel], ix, un, c2["iniPar"][sel], 2000.0, 128, t,
                                           return_stats=True)
    assert np.all(status == 0)
    g = O.Grid(2000.0, 128, t, 4.0)
    worst = 0.0
    for j, k in enumerate(sel):
        # converged trace: the 1e-10/1e-14 run is itself off by up to 2e-6 on fast decays
        ref = O.solve(c2["iniPar"][k], g, c2["states"][k], ix, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)
        err, _ = _relerr(sig[j], ref)
        worst = max(worst, err)
        assert err <= REL_TOL, (k, err, stats[j])
    print("worst:", worst, "mean steps:", stats[:, 0].mean())


def test_hard_cases_tauP_much_larger_than_tauN():
    """The worst-conditioned census draws (tests/golden/make_hard_cases.py): (tauN+tauP)/tauN up to 91
    amplifies every relative error made before the high-/low-injection transition.  TRPL and TRTS,
    thick and thin films, all within the 1e-6 gate over the 4-decade window."""
    from metrotrpl_amd import batch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hard_cases.npz"))
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    sig, status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"],
                                           meas=[str(m) for m in z["meas"]], return_stats=True)
    assert np.all(status == 0)
    worst = 0.0
    for k in range(len(sig)):
        ref = z["sig"][k]
        m = np.abs(ref) >= 1e-4 * abs(ref[0])
        err = np.max(np.abs(sig[k][m] - ref[m]) / np.abs(ref[m]))
        worst = max(worst, err)
        assert err <= 1e-6, (k, str(z["meas"][k]), err)
    print(f"hard cases: worst {worst:.2e}, steps {stats[:, 0].min()}..{stats[:, 0].max()}")


def test_every_backend_lands_on_the_converged_traces():
    """kernel_variant 6 / 3 (one wave per solve, fp32-stored elimination blocks, one / two solves per SIMD: the
    default pair), 2 (fp64 blocks) and 1 (team kernel) are different distributions of the same integrator: all within
    the gate on the hard cases, and not one kernel behind four switches.  The two builds of the default kernel give
    the same bits."""
    from metrotrpl_amd import _lib as L, batch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "hard_cases.npz"))
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    ctx = L.default_context()
    out = {}
    variants = (6, 3, 2, 1)
    try:
        for variant in variants:
            ctx.set_option("kernel_variant", variant)
            out[variant], status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128,
                                                            z["tSteps"], meas=[str(m) for m in z["meas"]], return_stats=True)
            assert np.all(status == 0), variant
            print(f"variant {variant}: steps {stats[:, 0].min()}..{stats[:, 0].max()}")
    finally:
        ctx.set_option("kernel_variant", 0)
    for k in range(len(z["sig"])):
        ref = z["sig"][k]
        m = np.abs(ref) >= 1e-4 * abs(ref[0])
        for variant in variants:
            assert np.max(np.abs(out[variant][k][m] - ref[m]) / np.abs(ref[m])) <= 1e-6, (variant, k)
    assert not np.array_equal(out[2], out[3])
    np.testing.assert_array_equal(out[6], out[3])  # one or two solves per SIMD: the same arithmetic


@pytest.mark.parametrize("variant", [0, 3, 2])
def test_census_subset_within_gate(variant):
    """160 committed census traces (config-2 draws and PT-ensemble solves, thin and thick films, TRPL and
    TRTS; tests/golden/make_census_subset.py): every one within 1e-6 of the converged oracle over the
    4-decade window -- the regression guard of the tolerance policy.  Through the default
    kernel choice (fp32-stored elimination blocks, one solve per SIMD at this size), its two-per-SIMD build, and the
    fp64-block kernel."""
    from metrotrpl_amd import _lib as L, batch
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "census_subset.npz"))
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    ctx = L.default_context()
    ctx.set_option("kernel_variant", variant)
    try:
        sig, status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"],
                                               meas=[str(m) for m in z["meas"]], return_stats=True)
    finally:
        ctx.set_option("kernel_variant", 0)
    assert np.all(status == 0)
    errs = np.zeros(len(sig))
    for k in range(len(sig)):
        ref = z["sig"][k]
        m = np.abs(ref) >= 1e-4 * abs(ref[0])
        errs[k] = np.max(np.abs(sig[k][m] - ref[m]) / np.abs(ref[m]))
    print(f"census subset: max {errs.max():.2e} p90 {np.percentile(errs, 90):.2e} median {np.median(errs):.2e}")
    assert errs.max() <= 1e-6, (int(errs.argmax()), errs.max())
    assert np.median(errs) <= 1e-7


def test_controller_wave_reductions():
    """device_math.hpp wave_sums_f<1|2|4>: the merged fp32 DPP / permlane-swap ladders the error norms go through,
    on data whose rows of 16 lanes differ wildly (a norm concentrated in one quarter of the grid must not be lost)."""
    import ctypes as C
    from metrotrpl_amd import _lib as L
    ctx = L.default_context()
    rng = np.random.default_rng(5)
    B = 64
    x = rng.uniform(0, 1, (B, 4, 64)).astype(np.float32)
    x[1, :, :48] = 0                      # everything in the last row of 16 lanes
    x[2, :, 16:] = 0                      # ... in the first
    x[3] = 0
    x[3, :, 37] = [1, 2, 3, 4]            # one lane
    x[4] *= (10.0 ** rng.uniform(-20, 20, (4, 64))).astype(np.float32)
    out = np.zeros((B, 7), dtype=np.float32)
    ctx.check(L.lib().mtrpl_reduce_probe(ctx.handle, B, x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
    ref = x.astype(np.float64).sum(axis=2)
    want = np.stack([ref[:, 0], ref[:, 0], ref[:, 1], ref[:, 0], ref[:, 1], ref[:, 2], ref[:, 3]], axis=1)
    np.testing.assert_allclose(out, want, rtol=2e-6)
