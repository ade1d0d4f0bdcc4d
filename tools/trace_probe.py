import os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from metrotrpl_amd import _lib as L, batch
z = np.load("tests/golden/heldout2.npz")
names = str(z["names"]).split(","); idx = {n: i for i, n in enumerate(names)}
ctx = L.default_context()
for k in (76, 20, 0):
    nx, nt = int(z["nx"][k]), int(z["nt"][k])
    print("trace", k, z["group"][k], "nx", nx, "nt", nt, "meas", z["meas"][k], "thick %.0f" % z["thickness"][k], "ini", z["ini"][k, :3], {n: "%.3g" % z["states"][k, idx[n]] for n in ("p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Sf", "Sb", "tauN", "tauP")})
    ref = z["sig"][k, :nt]
    dec = np.floor(np.log10(ref[0]) - np.log10(ref)).astype(int)
    for setting in ("shipped", "fun_weight=4", "rtol_force=6e-10", "rtol_force=3e-10", "newton_tol=0.03", "early_factor=1", "amp_ref=0", "shape_factor=10", "kernel_variant=2"):
        saved = {}
        if setting != "shipped":
            kk, v = setting.split("="); saved[kk] = ctx.get_option(kk); ctx.set_option(kk, float(v))
        ini = z["ini"][k:k+1, :nx] if str(z["ini_mode"][k]) == "density" else z["ini"][k:k+1, :3]
        s, st, stats = batch.solve_batch(z["states"][k:k+1], idx, z["units"], ini, z["thickness"][k:k+1], nx, z["t"][k, :nt], hmax=float(z["hmax"][k]), meas=[str(z["meas"][k])], model=str(z["model"][k]), ini_mode=str(z["ini_mode"][k]), return_stats=True)
        for kk, v in saved.items(): ctx.set_option(kk, v)
        rel = (s[0] - ref) / ref
        print("  %-18s steps %5d | signed max-err per decade 0..9: " % (setting, stats[0, 0]) + " ".join("%+.1e" % (rel[dec == d][np.abs(rel[dec == d]).argmax()] if (dec == d).any() else 0) for d in range(10)))
