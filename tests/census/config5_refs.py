"""Converged-oracle (1e-12/1e-17) traces for the config-5 workload (nx=512, thick/thin, TRPL/TRTS)
-> tools/_cache/config5_refs.npz (dev tool; ~30 s per solve)."""
import os
import sys
import time
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from metrotrpl_amd import workloads as W
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 16
NP = int(sys.argv[2]) if len(sys.argv) > 2 else 7
c = W.config5(NS)
ix, un = W.indexes(), W.units()


def one(k):
    from oracle import oracle as O
    g = O.Grid(c["thickness"][k], c["nx"], c["tSteps"], c["hmax"])
    return O.solve(c["iniPar"][k], g, c["states"][k], ix, meas=c["meas"][k], units=un, ini_mode="fluence",
                   RTOL=1e-12, ATOL=1e-17)


if __name__ == "__main__":
    t0 = time.time()
    out = []
    with Pool(NP) as p:
        for i, r in enumerate(p.imap(one, range(4 * NS))):
            out.append(r)
            if i % 8 == 0:
                print(i, time.time() - t0, flush=True)
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "_cache", "config5_refs.npz")
    np.savez_compressed(dst, refs=np.array(out), n_states=NS)
    print("done", time.time() - t0)
