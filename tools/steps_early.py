"""Steps the integrator needs to reach t_end for tiny t_end: the start-up phase (dev tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    ctx.set_option(k, float(v))
ctx.set_option("early_factor", 1.0)  # early_frac refers to the first sample: keep the policy of the full grid out of it
c = W.config2(1024)
sel = [0, 1, 2, 3, 217, 634, 79, 500]
ends = [1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.5, 1, 2, 4]
print("t_end      " + " ".join(f"{e:7g}" for e in ends))
rows = []
for te in ends:
    t = np.array([0.0, te])
    sig, status, stats = batch.solve_batch(c["states"][sel], W.indexes(), W.units(), c["iniPar"][sel], 2000.0, 128, t, return_stats=True, ctx=ctx)
    rows.append((stats[:, 0], stats[:, 6], stats[:, 4]))
for i, k in enumerate(sel):
    print(f"#{k:4d} steps " + " ".join(f"{r[0][i]:7d}" for r in rows))
    print(f"      order " + " ".join(f"{r[1][i]:7d}" for r in rows))
    print(f"      rej   " + " ".join(f"{r[2][i]:7d}" for r in rows))
