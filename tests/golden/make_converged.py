"""make_converged.py -- companion to make_golden.py.

For every golden solve case, run the CPU oracle (SciPy LSODA + the C RHS; pinned
bit-exact to the reference's own tight output by tests/test_oracle_golden.py) at
RTOL=1e-12, ATOL=1e-17 and store the trace as `<case>__conv`.  Where the
reference's RTOL=1e-10/ATOL=1e-14 output is not itself converged to 1e-6 over 4
decades (fast decays: LSODA's own global error), the parity gate uses this trace.
Runs without the reference (oracle only):  python tests/golden/make_converged.py
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))


def one(job):
    from oracle import oracle as O
    key, names, meas, solver, model, ini_mode, geom, t, state, units, ini = job
    th, nx, hmax, _, _, min_y = geom
    g = O.Grid(float(th), int(nx), t, float(hmax))
    if min_y >= 0:
        g.min_y = float(min_y)
    idx = {n: i for i, n in enumerate(names)}
    t0 = time.time()
    sig = O.solve(ini, g, state, idx, meas=meas, units=units, solver=("solveivp",), model=model,
                  ini_mode=ini_mode, RTOL=1e-12, ATOL=1e-17)
    return key, sig, time.time() - t0


if __name__ == "__main__":
    gold = np.load(os.path.join(HERE, "solve.npz"))
    jobs = []
    for key in [str(k) for k in gold["order"]]:
        if not key.endswith("__tight"):
            continue
        names, meas, solver, model, ini_mode = [str(x) for x in gold[key + "__meta"]]
        if int(gold[key + "__geom"][1]) > 128:
            continue  # dense 1537^2 Jacobians at this tolerance take hours
        jobs.append((key, names.split(","), meas, solver, model, ini_mode, gold[key + "__geom"],
                     gold[key + "__tSteps"], gold[key + "__state"], gold[key + "__units"], gold[key + "__iniPar"]))
    out = {}
    with Pool(7) as pool:
        for key, sig, dt in pool.imap_unordered(one, jobs):
            ref = gold[key + "__sig"]
            m = np.abs(ref) >= 1e-4 * abs(ref[0])
            d = np.max(np.abs(sig - ref)[m] / np.abs(ref[m]))
            print(f"{key}: {dt:.1f}s  reference-tight vs converged over 4 decades: {d:.2e}", flush=True)
            out[key.replace("__tight", "__conv")] = sig
    np.savez_compressed(os.path.join(HERE, "solve_converged.npz"), **out)
