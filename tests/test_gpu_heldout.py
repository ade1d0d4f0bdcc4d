"""The HELD-OUT accuracy gate (missing 6).

tests/golden/heldout.npz (tests/golden/make_heldout.py) holds converged-oracle traces drawn from a box 10x wider than
the prior box every other converged fixture comes from, at thicknesses 100-5000 nm, over 50 ns / 2000 ns / 20 us
windows, on uniform and geometric sample grids, from Beer-Lambert and from tabulated initial profiles, in an
Auger-dominated and a surface-dominated corner, TRPL and TRTS, std and traps, nx 128 / 100 / 64 -- generated before the
round-4 error-control policy was tuned.  (What it changed: round 3's policy FAILS it -- 1.1e-6 / 1.2e-6 / 1.2e-5 in
decades 5 / 7 / 8 --, and it exposed the cancellation in the photoconductivity formula that model_physics.hpp now avoids.
Policy constants were afterwards chosen on the other fixtures and only checked here.)  Same gate as
tests/test_gpu_accuracy.py: per decade of signal decay, error <= 1e-6 + the oracle's own uncertainty for that trace and
decade, decades 0-8 -- where the oracle itself is converged to 1e-7 or better (TRTS below its seventh decade is not:
the reference's own formula subtracts p0 from P there)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def solve_all(z, ctx=None):
    """Every valid case of the fixture through batch.solve_batch, grouped by what one launch must share."""
    from metrotrpl_amd import batch
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    n = len(z["nx"])
    sig = np.full(z["sig"].shape, np.nan)
    steps = np.zeros(n, dtype=int)
    status = np.full(n, -1)
    keys = {}
    for k in range(n):
        if not z["valid"][k]:
            continue
        key = (int(z["nx"][k]), int(z["nt"][k]), float(z["t"][k, 1]), float(z["t"][k, int(z["nt"][k]) - 1]), str(z["model"][k]),
               str(z["ini_mode"][k]), float(z["hmax"][k]))
        keys.setdefault(key, []).append(k)
    for (nx, nt, _, _, model, ini_mode, hmax), ks in keys.items():
        ks = np.array(ks)
        ini = z["ini"][ks, :nx] if ini_mode == "density" else z["ini"][ks, :3]
        s, st, stats = batch.solve_batch(z["states"][ks], idx, z["units"], ini, z["thickness"][ks], nx, z["t"][ks[0], :nt],
                                         hmax=hmax, meas=[str(m) for m in z["meas"][ks]], model=model, ini_mode=ini_mode,
                                         return_stats=True, ctx=ctx)
        sig[ks, :nt] = s
        steps[ks] = stats[:, 0]
        status[ks] = st
    return sig, status, steps


U_MAX = 1e-7


def decade_table(z, sig, n_dec=11):
    """per decade: (judged samples, worst error of all samples, worst error - u_oracle of the judged ones, index of
    that trace)"""
    valid = z["valid"]
    ref = z["sig"]
    with np.errstate(all="ignore"):
        dec = np.floor(np.log10(ref[:, :1]) - np.log10(ref))
        rel = np.abs(sig - ref) / ref
    clamped = ~(ref > 1e-300)  # samples the reference clamped to min_y (or its padding) do not count
    out = []
    for d in range(n_dec):
        m = (dec == d) & valid[:, None] & ~clamped & np.isfinite(ref)
        if not m.any():
            out.append((0, 0.0, 0.0, -1))
            continue
        # a reference adjudicates a 1e-6 gate only where it is itself converged ten times better: (trace, decade)
        # cells whose oracle uncertainty exceeds U_MAX are reported, not gated
        judged = m & (z["u_oracle"][:, d:d + 1] <= U_MAX)
        e = np.where(m, rel, -np.inf)
        ex = np.where(judged, rel - z["u_oracle"][:, d:d + 1], -np.inf)
        who = int(np.unravel_index(ex.argmax(), ex.shape)[0]) if judged.any() else -1
        out.append((int(judged.sum()), float(e.max()), float(ex.max()) if judged.any() else 0.0, who))
    return out


# heldout.npz: frozen since round 4.  heldout2.npz: a second draw of the same recipe (other seed), generated in round 5 AFTER that
# round's changes to the error control (the field out of the norms, no growth cap): it passed at 8.5e-7 where round 4's build had
# 5.7e-7 -- the margins of the gating fixtures had not told -- and the internal tolerance went from 2.4e-9 to 2e-9 (and its policy
# floor with it) with that in view.  heldout3.npz: a third draw, generated after that and used to choose nothing.  The set of
# unseen traces grows; no set is re-drawn.
@pytest.mark.parametrize("fixture", ["heldout.npz", "heldout2.npz", "heldout3.npz"])
@pytest.mark.parametrize("variant", [0, 3])  # the default build | two solves per SIMD
def test_heldout_traces_decade_by_decade(variant, fixture):
    from metrotrpl_amd import _lib as L
    fn = os.path.join(G, fixture)
    z = np.load(fn)
    assert int(z["valid"].sum()) >= 128
    ctx = L.default_context()
    ctx.set_option("kernel_variant", variant)
    try:
        sig, status, steps = solve_all(z)
    finally:
        ctx.set_option("kernel_variant", 0)
    v = z["valid"]
    assert np.all(status[v] == 0), (np.flatnonzero(v & (status != 0)), status[v & (status != 0)])
    # where the oracle itself is not converged (TRTS below its sixth decade: mu_n (N - n0) + mu_p (P - p0) is a
    # difference of nearly equal numbers there) the gate widens by exactly that much; it stays sharp nearly everywhere
    u9 = z["u_oracle"][v][:, :9]
    assert u9[:, :6].max() < U_MAX and np.mean(u9 < U_MAX) > 0.95
    table = decade_table(z, sig)
    for g_ in sorted(set(str(x) for x in z["group"])):
        m = v & (z["group"] == g_)
        print(f"  {g_:8s} {int(m.sum()):3d} traces, steps mean {steps[m].mean():6.0f} max {steps[m].max():6d}")
    for d, (n, mx, ex, who) in enumerate(table):
        print(f"  decade {d:2d}: n={n:6d} max={mx:.2e} (- u_oracle: {ex:.2e}) worst trace {who} "
              f"[{z['group'][who] if who >= 0 else ''}]")
    for d, (n, mx, ex, who) in enumerate(table[:9]):
        assert ex <= 1e-6, (d, mx, ex, who, str(z["group"][who]))
