"""How far down a trace is the 'converged' oracle (SciPy LSODA + gcc RHS at RTOL=1e-12 / ATOL=1e-17) itself
converged?  Re-integrates a few config-2 draws and PT-ensemble solves at RTOL=1e-13 / ATOL=1e-21 and prints
the disagreement per decade of signal decay (dev tool; needs tools/_cache/census_refs_big.npz)."""
import os
import sys
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from metrotrpl_amd import workloads as W
c = W.config2(64)
e = W.pt_ensemble(8)
ix, un = W.indexes(), W.units()
SEL = [("cfg2", k) for k in (0, 2, 3, 7, 11, 14, 21, 30)] + [("pt", k) for k in range(0, 12)]


def one(item):
    from oracle import oracle as O
    kind, k = item
    src, th = (c, 2000.0) if kind == "cfg2" else (e, float(e["thickness"][k]))
    g = O.Grid(th, 128, src["tSteps"], 4.0)
    return O.solve(src["iniPar"][k], g, src["states"][k], ix, units=un, ini_mode="fluence", RTOL=1e-13, ATOL=1e-21)


if __name__ == "__main__":
    z = np.load(os.path.join(REPO, "tools", "_cache", "census_refs_big.npz"))
    with Pool(7) as p:
        tight = p.map(one, SEL)
    worst = {}
    for (kind, k), s in zip(SEL, tight):
        ref = z[kind][k]
        dec = np.floor(np.log10(ref[0] / np.maximum(ref, 1e-300))).astype(int)
        rel = np.abs(s - ref) / np.abs(ref)
        for d in range(0, 14):
            m = dec == d
            if m.any():
                worst[d] = max(worst.get(d, 0.0), float(rel[m].max()))
    for d in sorted(worst):
        print(f"decade {d:2d}: max |1e-13/1e-21 - 1e-12/1e-17| / ref = {worst[d]:.2e}")
