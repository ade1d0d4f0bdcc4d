"""Accuracy census on the GPU box against the cached converged references (tests/census/census_refs.py),
sweeping integrator options given as key=value[,value...] arguments (dev tool).
    python tools/census.py e_weight=1,100 honor_hmax=1,0 rtol_cap=3e-9"""
import itertools
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
MEAS = os.environ.get("CENSUS_MEAS", "TRPL")
TAG = os.environ.get("CENSUS_TAG", "")
z = np.load(os.path.join(ROOT, "tools", "census_cache", ("census_refs" if MEAS == "TRPL" else f"census_refs_{MEAS.lower()}") + TAG + ".npz"))
N, NCH = int(z["n"]), int(z["nch"])
c = W.config2(N); e = W.pt_ensemble(NCH); c2 = W.config2(1024)
ix, un = W.indexes(), W.units()
ctx = L.default_context()
keys, vals = [], []
for a in sys.argv[1:]:
    k, v = a.split("=")
    keys.append(k); vals.append([float(x) for x in v.split(",")])


def errs_of(sig, refs):
    out = np.zeros(len(refs))
    for k in range(len(refs)):
        m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
        out[k] = np.max(np.abs(sig[k][m] - refs[k][m]) / np.abs(refs[k][m]))
    return out


for combo in itertools.product(*vals):
    for k, v in zip(keys, combo):
        if k == "rtol_cap":
            ctx.set_option("rtol_floor", min(v, ctx.get_option("rtol_floor")))
        ctx.set_option(k, v)
    s1, st1, stat1 = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], meas=MEAS, return_stats=True)
    e1 = errs_of(s1, z["cfg2"])
    s2, st2, stat2 = batch.solve_batch(e["states"], ix, un, e["iniPar"], e["thickness"], 128, e["tSteps"], meas=MEAS, return_stats=True)
    e2 = errs_of(s2, z["pt"])
    thin = e["thickness"] < 1000
    batch.solve_batch(c2["states"], ix, un, c2["iniPar"], c2["thickness"], 128, c2["tSteps"])
    t0 = time.time()
    s3, st3, stat3 = batch.solve_batch(c2["states"], ix, un, c2["iniPar"], c2["thickness"], 128, c2["tSteps"], return_stats=True)
    dt = time.time() - t0
    print(" ".join(f"{k}={v:g}" for k, v in zip(keys, combo)),
          f"| cfg2 max={e1.max():.2e} p99={np.percentile(e1, 99):.2e} med={np.median(e1):.2e} (#{e1.argmax()})"
          f" | thin max={e2[thin].max():.2e} p99={np.percentile(e2[thin], 99):.2e} | thick max={e2[~thin].max():.2e}"
          f" | bad={(st1 != 0).sum() + (st2 != 0).sum() + (st3 != 0).sum()}"
          f" | cfg2x1024 {dt * 1e3:.1f} ms steps {stat3[:, 0].mean():.0f}/{stat3[:, 0].max()} rhs {stat3[:, 1].mean():.0f} fac {stat3[:, 2].mean():.0f} rej {stat3[:, 4].mean():.1f} nfail {stat3[:, 5].mean():.1f}",
          flush=True)
