"""Converged-oracle (1e-12/1e-17) reference traces for the accuracy census (dev tool):
N config-2 draws + NCH PT-ensemble chains x 6 measurements -> tools/_cache/census_refs.npz"""
import os
import sys
import time
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrotrpl_amd import workloads as W
N = int(sys.argv[1]) if len(sys.argv) > 1 else 384
NCH = int(sys.argv[2]) if len(sys.argv) > 2 else 64
NP = int(sys.argv[3]) if len(sys.argv) > 3 else 7
MEAS = sys.argv[4] if len(sys.argv) > 4 else "TRPL"
TAG = sys.argv[5] if len(sys.argv) > 5 else ""
c = W.config2(N)
e = W.pt_ensemble(NCH)
ix, un = W.indexes(), W.units()


def one(k):
    from oracle import oracle as O
    if k < N:
        g = O.Grid(c["thickness"], c["nx"], c["tSteps"], c["hmax"])
        return O.solve(c["iniPar"][k], g, c["states"][k], ix, meas=MEAS, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)
    k -= N
    g = O.Grid(e["thickness"][k], 128, e["tSteps"], 4.0)
    return O.solve(e["iniPar"][k], g, e["states"][k], ix, meas=MEAS, units=un, ini_mode="fluence", RTOL=1e-12, ATOL=1e-17)


if __name__ == "__main__":
    t0 = time.time()
    out = []
    with Pool(NP) as p:
        for i, r in enumerate(p.imap(one, range(N + NCH * 6), chunksize=2)):
            out.append(r)
            if i % 16 == 0:
                print(i, time.time() - t0, flush=True)
    out = np.array(out)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "_cache", ("census_refs" if MEAS == "TRPL" else f"census_refs_{MEAS.lower()}") + TAG + ".npz")
    np.savez_compressed(dst, cfg2=out[:N], pt=out[N:], n=N, nch=NCH)
    print("done", time.time() - t0)
