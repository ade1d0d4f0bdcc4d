"""dev: which traces of tests/golden/tail_decades.npz carry the worst errors in decades 6..8, and what they look like"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from metrotrpl_amd import _lib as L, batch
z = np.load(os.path.join(REPO, "tests", "golden", "tail_decades.npz"))
idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
ctx = L.default_context()
for kv in sys.argv[1:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
sig, status, st = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"], return_stats=True)
ref = z["sig"]
with np.errstate(all="ignore"):
    dec = np.clip(np.floor(np.log10(ref[:, :1]) - np.log10(ref)).astype(int), 0, 12)
    rel = np.abs(sig - ref) / ref
w = np.array([np.where((dec[k] >= 6) & (dec[k] <= 8), rel[k], 0).max() for k in range(len(ref))])
for k in np.argsort(-w)[:12]:
    s = z["states"][k]
    per = [np.where(dec[k] == d, rel[k], 0).max() for d in range(4, 10)]
    t8 = z["tSteps"][np.argmax(dec[k] >= 8)] if (dec[k] >= 8).any() else -1
    print(f"#{k:3d} th {z['thickness'][k]:.0f} worst(6-8) {w[k]:.2e} per decade 4..9 " + " ".join(f"{v:.1e}" for v in per) +
          f" | steps {st[k,0]} t(dec8) {t8:.0f} | tauN {s[idx['tauN']]:.0f} tauP {s[idx['tauP']]:.0f} Sf {s[idx['Sf']]:.1f} Sb {s[idx['Sb']]:.1f} "
          f"mu {s[idx['mu_n']]:.0f}/{s[idx['mu_p']]:.0f} p0 {s[idx['p0']]:.1e} ks {s[idx['ks']]:.1e} flu {z['iniPar'][k,0]:.1e}")
