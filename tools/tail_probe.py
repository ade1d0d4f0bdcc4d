"""Who is right in the deep tail?  For the traces with the largest disagreement between the GPU solver and the
oracle in decades 5-8: GPU at the shipped policy, GPU at forced rtol 3e-10 / 1e-10 (two more tolerance levels of
an independent integrator) and the oracle, at the worst sample."""
import os
import sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from metrotrpl_amd import _lib as L, batch
z = np.load(os.path.join(REPO, "tests", "golden", "tail_decades.npz"))
idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
ctx = L.default_context()
ref = z["sig"]
runs = {}
for name, opts in (("shipped", {}), ("3e-10", {"rtol_force": 3e-10}), ("1e-10", {"rtol_force": 1e-10})):
    for k, v in opts.items():
        ctx.set_option(k, v)
    runs[name], status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"], return_stats=True)
    ctx.set_option("rtol_force", 0.0)
    print(name, "ok", (status == 0).sum(), "steps", stats[:, 0].mean(), stats[:, 0].max())
pos = ref > 0
with np.errstate(all="ignore"):
    dec = np.where(pos, np.floor(np.log10(ref[:, :1] / np.where(pos, ref, 1.0))), 99).astype(int)
    rel = {k: np.abs(v - ref) / np.abs(ref) for k, v in runs.items()}
    relg = np.abs(runs["shipped"] - runs["1e-10"]) / np.abs(runs["1e-10"])
print("non-positive oracle samples:", (~pos).sum(), "of", pos.size)
for d in range(1, 12):
    m = dec == d
    if not m.any():
        continue
    print(f"decade {d:2d} n={m.sum():6d}: vs oracle max shipped {rel['shipped'][m].max():.2e} 3e-10 {rel['3e-10'][m].max():.2e} 1e-10 {rel['1e-10'][m].max():.2e} | "
          f"shipped vs gpu-1e-10: max {relg[m].max():.2e} p99 {np.percentile(relg[m], 99):.2e} median {np.median(relg[m]):.2e}")
for d in (5, 6, 7, 8):
    m = dec == d
    r = np.where(m, rel["shipped"], 0)
    i, j = np.unravel_index(np.argmax(r), r.shape)
    print(f"decade {d}: worst trace {i} (thickness {z['thickness'][i]}) sample {j} t={z['tSteps'][j]}: oracle {ref[i, j]:.10e} shipped {runs['shipped'][i, j]:.10e} "
          f"3e-10 {runs['3e-10'][i, j]:.10e} 1e-10 {runs['1e-10'][i, j]:.10e}; neighbours oracle {ref[i, j-1:j+2]}")
