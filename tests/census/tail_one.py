"""One trace of tests/golden/tail_decades.npz re-integrated by the oracle at a given RTOL / ATOL (dev tool).
usage: python tests/census/tail_one.py <trace> <rtol> <atol N,P> <atol E> [sample ...]"""
import os, sys, time
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import oracle as O
z = np.load(os.path.join(REPO, "tests", "golden", "tail_decades.npz"))
k, rtol, atol, atol_e = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
g = O.Grid(float(z["thickness"][k]), 128, z["tSteps"], 4.0)
t0 = time.time()
s = O.solve(z["iniPar"][k], g, z["states"][k], idx, units=z["units"], ini_mode="fluence", RTOL=rtol, ATOL=np.r_[np.full(256, atol), np.full(129, atol_e)])
print(f"trace {k} rtol {rtol:g} atol {atol:g} / E {atol_e:g}: {time.time() - t0:.1f} s")
for j in [int(a) for a in sys.argv[5:]]:
    print(f"  sample {j}: {s[j]:.10e}   (fixture {z['sig'][k, j]:.10e})")
