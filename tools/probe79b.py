import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
zt = np.load(os.path.join(ROOT, "tools", "_cache", "census_refs_trts.npz"))
c = W.config2(384); ix, un = W.indexes(), W.units()
ctx = L.default_context()
ctx.set_option("trts_rtol_factor", 1.0)
for k in (79,):
    s = c["states"][k]; ref = zt["cfg2"][k]
    for rt in (3e-9, 1.5e-9):
        for ar in (1e-4, 1e-5, 1e-6, 1e-7):
            ctx.set_option("rtol_cap", rt); ctx.set_option("rtol_floor", rt); ctx.set_option("atol_rel", ar)
            sig, st, stats = batch.solve_batch(s[None], ix, un, c["iniPar"][k][None], 2000.0, 128, c["tSteps"], meas="TRTS", return_stats=True)
            m = np.abs(ref) >= 1e-4 * abs(ref[0])
            with np.errstate(all="ignore"):
                r = np.abs(sig[0] - ref) / np.abs(ref)
            r[~m] = 0
            print(k, f"rtol={rt:g} atol_rel={ar:g}: max {r.max():.2e} at {r.argmax()}; r[60]={r[60]:.1e} r[70]={r[70]:.1e} r[78]={r[78]:.1e} steps {stats[0][0]} rhs {stats[0][1]}")
