import sys, os
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import workloads as W, batch, _lib as L
from oracle import oracle as O
e = W.pt_ensemble(64); ix, un = W.indexes(), W.units()
k = 74
g = O.Grid(e["thickness"][k], 128, e["tSteps"], 4.0)
refs = {}
for rt, at in ((1e-10, 1e-14), (1e-11, 1e-16), (1e-12, 1e-17), (1e-13, 1e-18)):
    refs[rt] = O.solve(e["iniPar"][k], g, e["states"][k], ix, units=un, ini_mode="fluence", RTOL=rt, ATOL=at)
r = refs[1e-12]; m = np.abs(r) >= 1e-4 * abs(r[0])
for rt in refs: print("oracle", rt, "vs 1e-12: %.2e" % np.max(np.abs(refs[rt] - r)[m] / np.abs(r[m])))
ctx = L.default_context()
for cap in (2e-8, 1e-8, 5e-9, 3e-9):
    ctx.set_option("rtol_force", cap)
    sig, st, stats = batch.solve_batch(e["states"][k:k+1], ix, un, e["iniPar"][k:k+1], e["thickness"][k], 128, e["tSteps"], return_stats=True)
    err = np.abs(sig[0] - r)[m] / np.abs(r[m])
    print("gpu rtol", cap, "err max %.2e at sample %d of %d, steps %d" % (err.max(), err.argmax(), m.sum(), stats[0, 0]), "err[::25]", np.array2string(err[::25], precision=1))
print("state", dict(zip(W.PARAM_NAMES, e["states"][k])), e["iniPar"][k], e["thickness"][k])
