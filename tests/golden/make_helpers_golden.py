"""
make_helpers_golden.py -- golden vectors for the small helper functions of the reference's
forward_solver / laplace modules (forward_solver.py:250-329, laplace.py:129-167), produced by IMPORTING the
reference in the build container exactly as make_golden.py does.  Only data is committed
(tests/golden/helpers.npz).

    python tests/golden/make_helpers_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _import_reference, REF  # noqa: E402

if __name__ == "__main__":
    fs, lp, _, _ = _import_reference()
    rng = np.random.default_rng(7)
    out = {}
    N = 10 ** rng.uniform(-9, -3, (5, 40))
    P = 10 ** rng.uniform(-9, -3, (5, 40))
    out["N"], out["P"] = N, P
    out["args_rr"] = np.array([3.7e1, 1e-13, 3e-6])            # ks, n0, p0
    out["args_pc"] = np.array([2e6, 1.1e6, 1e-13, 3e-6])       # mu_n, mu_p, n0, p0
    out["rr"] = fs.calculate_RR(N, P, *out["args_rr"])
    out["photoc"] = fs.calculate_photoc(N, P, *out["args_pc"])
    out["dx"] = np.array(7.3)
    out["int2d"] = fs.integrate_2D(7.3, out["rr"])
    out["int1d"] = np.array(fs.integrate_1D(7.3, out["rr"][2]))
    irf = np.loadtxt(os.path.join(REF, "IRFs", "irf_745nm.csv"), delimiter=",")
    out["irf"] = irf
    cases = []
    vals = []
    for (m, n, lo, hi, nu) in ((0, 0, 0, 1, 100), (5, 1, 0, 1, 100), (17, 2, 0, 1, 100), (3, 1, 0, 1, 101), (40, 2, 0, 1, 7),
                               (40, 0, 0, 1, 1000), (12, 2, 0, 1, 2), (9, 1, 0, 1, 3), (100, 2, 0, 1, 64)):
        cases.append([m, n, lo, hi, nu])
        vals.append(lp.I_moment(irf[:, 0], irf[:, 1], m, n, u_lower=lo, u_upper=hi, u_spacing=nu))
    out["moment_cases"] = np.array(cases)
    out["moment_vals"] = np.array(vals)
    # bounds that take y[m+1-u] to a NEGATIVE index: numpy wraps around (laplace.py:164), no IndexError
    wcases, wvals = [], []
    for (m, n, lo, hi, nu) in ((0, 1, 0, 3, 100), (1, 0, 5, 1, 50), (2, 2, 0, 150, 33)):
        wcases.append([m, n, lo, hi, nu])
        wvals.append(lp.I_moment(irf[:, 0], irf[:, 1], m, n, u_lower=lo, u_upper=hi, u_spacing=nu))
    out["moment_wrap_cases"] = np.array(wcases)
    out["moment_wrap_vals"] = np.array(wvals)
    np.savez_compressed(os.path.join(HERE, "helpers.npz"), **out)
    print("wrote helpers.npz:", {k: np.shape(v) for k, v in out.items()})
