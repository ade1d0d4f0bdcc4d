"""Where along the time axis do the integrator's steps go?  Re-runs selected config-2 solves on truncated
windows of the same 2 ns sample grid and prints the cumulative step count (dev tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        ctx.set_option(k, float(v))
c = W.config2(1024)
sel = [217, 634, 79, 35, 359, 0, 1, 2, 3, 500]
ends = [2, 10, 20, 50, 100, 200, 400, 800, 1200, 2000]
tab = np.zeros((len(sel), len(ends)), dtype=int)
dec = np.zeros((len(sel), len(ends)))
for j, te in enumerate(ends):
    t = np.linspace(0, te, int(te / 2) + 1)
    sig, status, stats = batch.solve_batch(c["states"][sel], W.indexes(), W.units(), c["iniPar"][sel], 2000.0, 128, t, return_stats=True, ctx=ctx)
    tab[:, j] = stats[:, 0]
    with np.errstate(all="ignore"):
        dec[:, j] = np.log10(sig[:, 0] / sig[:, -1])
print("t_end:      " + " ".join(f"{e:6d}" for e in ends))
for i, k in enumerate(sel):
    print(f"#{k:4d} steps " + " ".join(f"{v:6d}" for v in tab[i]))
    print(f"      decay " + " ".join(f"{v:6.1f}" for v in dec[i]))
