"""The accuracy gate, decade by decade (SURVEY.md 8d "Accuracy gate": max |sig_gpu - sig_oracle| / |sig_oracle|
over the samples not clamped to min_y, bar 1e-6).

tests/golden/tail_decades.npz holds converged oracle traces for 256 draws of config 2 (nx=128, 2000 nm) and
16 chains x 6 measurements of the config-3 geometry -- none of their samples is clamped (min_y = the smallest
double), so every sample counts.  Samples are binned by how many decades the trace has fallen below its
first sample.  Decades 0-8 hold the 1e-6 bar plus the oracle's OWN uncertainty there -- stored in the fixture per
trace and decade (`u_oracle`: the same solves at a 100x smaller density ATOL and at a 3x smaller RTOL; at most
5e-9 through decade 8, 2.6e-8 / 1.9e-7 / 1.3e-6 in decades 9 / 10 / 11).  From decade 9 on the signal is the
difference N P - n0 p0 of nearly equal numbers, nothing measured reaches down there and the integrator's local
tolerance is relaxed again (options far_frac / far_factor): the bounds widen and are regression guards, not
accuracy claims.  DESIGN.md section 2 prints the measured table.

The MEDIAN bounds are regression guards too, ten to a hundred times inside the contract.  Round 5 took the field components
out of the controller's norms (they were setting the step size of most steps without being part of any signal: 9 % fewer
steps) and tightened the internal rtol from 3e-9 to 2e-9 in exchange: the maxima stayed where they were (config 2 decade 8:
6.1e-7 against 6.5e-7; config-3 geometry 6.9e-7 against 6.3e-7), the medians of decades 1 and 2 moved from 7e-9 / 1.7e-8 to
1.3e-8 / 2.6e-8 -- the error budget is spent more evenly -- and their guards from 1e-8 / 3e-8 to 2e-8 / 4e-8."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")

#          decade: 0     1     2     3     4     5     6     7     8     9     10
MAX_BOUND = [1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 2e-5, 2e-4]
MEDIAN_BOUND = [1e-8, 2e-8, 4e-8, 1e-7, 1e-7, 2e-7, 3e-7, 5e-7, 8e-7, 1.5e-6, 3e-6]


def per_decade(sig, ref, u_oracle):
    """-> per decade (n, max error, median error, max of error - oracle uncertainty of that trace and decade)"""
    dec = np.floor(np.log10(ref[:, :1]) - np.log10(ref)).astype(int)
    rel = np.abs(sig - ref) / ref
    out = []
    for d in range(len(MAX_BOUND)):
        m = dec == d
        excess = np.where(m, rel - u_oracle[:, d:d + 1], -np.inf)
        out.append((int(m.sum()), float(rel[m].max()), float(np.median(rel[m])), float(excess.max())))
    return out


@pytest.mark.parametrize("variant", [0, 3, 2])  # default (fp32-stored blocks, one per SIMD here) | two per SIMD | fp64 blocks
@pytest.mark.parametrize("part", ["config2", "config3"])
def test_every_decade_of_every_trace(part, variant):
    from metrotrpl_amd import _lib as L, batch
    ctx = L.default_context()
    ctx.set_option("kernel_variant", variant)
    try:
        _every_decade(part, batch)
    finally:
        ctx.set_option("kernel_variant", 0)


def _every_decade(part, batch):
    z = np.load(os.path.join(G, "tail_decades.npz"))
    n2 = int(z["n_cfg2"])
    sel = slice(0, n2) if part == "config2" else slice(n2, None)
    idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
    ref = z["sig"][sel]
    assert np.all(ref > 0)
    sig, status, stats = batch.solve_batch(z["states"][sel], idx, z["units"], z["iniPar"][sel], z["thickness"][sel], 128,
                                           z["tSteps"], return_stats=True)
    assert np.all(status == 0)
    u = z["u_oracle"][sel]
    assert u[:, :9].max() < 1e-8  # the oracle is converged far inside the gate through decade 8
    table = per_decade(sig, ref, u)
    print(f"{part}: {len(ref)} traces, mean steps {stats[:, 0].mean():.0f}")
    for d, (n, mx, med, ex) in enumerate(table):
        print(f"  decade {d:2d}: n={n:6d} max={mx:.2e} median={med:.2e}   (bounds {MAX_BOUND[d]:.0e} + u_oracle / {MEDIAN_BOUND[d]:.0e})")
    for d, (n, mx, med, ex) in enumerate(table):
        assert n > 500, (part, d, n)
        assert ex <= MAX_BOUND[d], (part, d, mx, ex)  # error <= bound + u_oracle(trace, decade)
        assert med <= MEDIAN_BOUND[d], (part, d, med)
