"""
make_hard_cases.py -- the worst-conditioned draws of the accuracy census as a committed
fixture (tests/golden/hard_cases.npz): inputs + converged oracle traces (SciPy LSODA +
gcc RHS at RTOL=1e-12 / ATOL=1e-17, the same recipe as make_converged.py).

They are the draws where tauP >> tauN: the SRH decay rate rises from 1/(tauN+tauP) at
high injection to 1/tauN at low injection inside the 4-decade window, which amplifies
every earlier relative error by (tauN+tauP)/tauN (up to 91x here) -- the cases that set
the integrator's tolerance policy (DESIGN.md "error control").

    python tests/golden/make_hard_cases.py        (about 1 minute)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
from metrotrpl_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

CFG2 = [79, 34, 11, 112, 142, 214]   # indices into the seeded config-2 stream (W.config2)
PT = [72, 74, 76, 77, 244, 302, 303]  # indices into the seeded PT ensemble (W.pt_ensemble(64)): thin + thick films

if __name__ == "__main__":
    O.build()
    ix, un = W.indexes(), W.units()
    c = W.config2(384)
    e = W.pt_ensemble(64)
    t = c["tSteps"]
    states, ini, thick, meas, sig = [], [], [], [], []
    for src, idxs, th_of in ((c, CFG2, lambda k: 2000.0), (e, PT, lambda k: float(e["thickness"][k]))):
        for k in idxs:
            for m in ("TRPL", "TRTS"):
                g = O.Grid(th_of(k), 128, t, 4.0)
                s = O.solve(src["iniPar"][k], g, src["states"][k], ix, meas=m, units=un, ini_mode="fluence",
                            RTOL=1e-12, ATOL=1e-17)
                states.append(src["states"][k]); ini.append(src["iniPar"][k]); thick.append(th_of(k))
                meas.append(m); sig.append(s)
                print(k, m, th_of(k), flush=True)
    np.savez_compressed(os.path.join(HERE, "hard_cases.npz"), states=np.array(states), iniPar=np.array(ini),
                        thickness=np.array(thick), meas=np.array(meas), sig=np.array(sig), tSteps=t,
                        names=np.array(",".join(W.PARAM_NAMES)), units=un)
