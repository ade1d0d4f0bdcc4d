"""Per-solve 4-decade error of the fp64 path on the config-5 batch at the shipped settings (dev tool)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
z = np.load(os.path.join(ROOT, "tools", "_cache", "config5_refs.npz"))
refs, NS = z["refs"], int(z["n_states"])
c = W.config5(NS); ix, un = W.indexes(), W.units()
ctx = L.default_context()
for opts in ({}, {"rtol_cap": 1.5e-9}, {"atol_rel": 1e-6}, {"atol_rel": 1e-6, "rtol_cap": 1.5e-9}):
    for k, v in {"rtol_cap": 3e-9, "atol_rel": 1e-4, **opts}.items():
        ctx.set_option(k, v)
    sig, status, stats = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], meas=c["meas"], return_stats=True)
    errs = np.zeros(len(refs)); where = np.zeros(len(refs), int)
    for k in range(len(refs)):
        m = np.abs(refs[k]) >= 1e-4 * abs(refs[k][0])
        r = np.abs(sig[k] - refs[k]) / np.abs(refs[k]); r[~m] = 0
        errs[k] = r.max(); where[k] = r.argmax()
    print(opts, "steps", stats[:, 0].mean())
    for combo in range(4):
        sel = np.arange(len(refs)) % 4 == combo
        print(f"  {c['thickness'][combo]:.0f} nm {c['meas'][combo]}: max {errs[sel].max():.2e} median {np.median(errs[sel]):.2e} worst solve {np.arange(len(refs))[sel][errs[sel].argmax()]} at sample {where[sel][errs[sel].argmax()]}")
    w = errs.argmax(); s = c["states"][w]
    print("  worst state:", {n: f"{s[i]:.3g}" for n, i in ix.items()}, "fluence", c["iniPar"][w][0], "sig/sig0 at worst", refs[w][where[w]] / refs[w][0])
