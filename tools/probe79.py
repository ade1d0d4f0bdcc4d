import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import workloads as W, batch, _lib as L
zt = np.load(os.path.join(ROOT, "tools", "_cache", "census_refs_trts.npz")); zp = np.load(os.path.join(ROOT, "tools", "_cache", "census_refs.npz"))
c = W.config2(384); ix, un = W.indexes(), W.units()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 79
s = c["states"][k]; print({n: f"{s[i]:.3g}" for n, i in ix.items()}, "fluence %.3g" % c["iniPar"][k][0])
ctx = L.default_context()
for meas, z in (("TRPL", zp), ("TRTS", zt)):
    ref = z["cfg2"][k]
    for floor, fac in ((1.5e-9, 1.0), (1.5e-9, 0.5), (5e-10, 0.33), (5e-10, 0.2), (2e-10, 0.1)):
        ctx.set_option("rtol_floor", floor); ctx.set_option("trts_rtol_factor", fac)
        ctx.set_option("rtol_cap", 3e-9 if meas == "TRTS" else 3e-9 * fac)
        if meas == "TRPL": ctx.set_option("rtol_floor", min(floor, 3e-9 * fac))
        sig, st, stats = batch.solve_batch(s[None], ix, un, c["iniPar"][k][None], 2000.0, 128, c["tSteps"], meas=meas, return_stats=True)
        m = np.abs(ref) >= 1e-4 * abs(ref[0])
        r = np.abs(sig[0] - ref) / np.abs(ref); r[~m] = 0
        idx = [10, 50, 100, 200, 400, 700, 1000]
        print(meas, f"rtol_eff={3e-9*fac:.2g}: max {r.max():.2e} at sample {r.argmax()} (sig/sig0 {ref[r.argmax()]/ref[0]:.2e}); err at {idx}: " + " ".join(f"{r[i]:.1e}" for i in idx), "stats", stats[0][:6].tolist())
