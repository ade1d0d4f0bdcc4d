"""dev: who are the slowest solves of the config-2 batch and where along the time axis do their steps go"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, float(v))
c = W.config2(1024)
ix, un = W.indexes(), W.units()
sig, status, st = batch.solve_batch(c["states"], ix, un, c["iniPar"], 2000.0, 128, c["tSteps"], return_stats=True, ctx=ctx)
steps = st[:, 0]
print("steps: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %d | rhs/step %.3f fac %.0f rej %.1f nfail %.2f" % (
    steps.mean(), np.percentile(steps, 50), np.percentile(steps, 90), np.percentile(steps, 99), steps.max(),
    st[:, 1].sum() / steps.sum(), st[:, 2].mean(), st[:, 4].mean(), st[:, 5].mean()))
order = np.argsort(-steps)
names = W.PARAM_NAMES
top = order[:12]
ends = [2, 10, 50, 200, 500, 1000, 2000]
tab = np.zeros((len(top), len(ends)), dtype=int)
for j, te in enumerate(ends):
    t = np.linspace(0, te, int(te / 2) + 1)
    _, _, s2 = batch.solve_batch(c["states"][top], ix, un, c["iniPar"][top], 2000.0, 128, t, return_stats=True, ctx=ctx)
    tab[:, j] = s2[:, 0]
print("cumulative steps up to t =", ends)
for i, k in enumerate(top):
    s = c["states"][k]
    with np.errstate(all="ignore"):
        dec = np.log10(sig[k, 0] / np.abs(sig[k, [5, 50, 250, 1000]]))
    print(f"#{k:4d} steps {steps[k]} rhs {st[k,1]} fac {st[k,2]} rej {st[k,4]} | " + " ".join(f"{v:5d}" for v in tab[i]) +
          f" | tauN {s[ix['tauN']]:.0f} tauP {s[ix['tauP']]:.0f} Sf {s[ix['Sf']]:.1f} Sb {s[ix['Sb']]:.1f} mu {s[ix['mu_n']]:.0f}/{s[ix['mu_p']]:.0f} p0 {s[ix['p0']]:.1e} ks {s[ix['ks']]:.1e} flu {c['iniPar'][k,0]:.1e} | decades at 10/100/500/2000 ns " +
          " ".join(f"{v:.1f}" for v in dec))
# the same table for a typical solve
med = order[len(order) // 2: len(order) // 2 + 3]
for j, te in enumerate(ends):
    t = np.linspace(0, te, int(te / 2) + 1)
    _, _, s2 = batch.solve_batch(c["states"][med], ix, un, c["iniPar"][med], 2000.0, 128, t, return_stats=True, ctx=ctx)
    tab[:3, j] = s2[:, 0]
for i, k in enumerate(med):
    print(f"median #{k:4d} steps {steps[k]} | " + " ".join(f"{v:5d}" for v in tab[i]))
