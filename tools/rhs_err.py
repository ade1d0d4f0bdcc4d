"""dev: worst error of the kernel's right-hand side against the reference's vectors (tests/golden/rhs.npz), scaled
by the flux magnitude -- to judge what a cheaper reciprocal costs"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import batch
from metrotrpl_amd.forward_solver import q_C, eps0
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "rhs.npz"))
worst = worst_rel = 0
for k in range(int(g["n_cases"])):
    a = g[f"std{k}_args"]
    L = int(a[0])
    if L > 128:
        continue
    par = np.array([[a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], q_C / (a[13] * eps0), a[14], 0, 0, 1]])
    y, ref = g[f"std{k}_y"], g[f"std{k}_dy"]
    dy = batch.rhs_batch(par, [a[1]], y[None, :], L)[0]
    scale = np.max(np.abs(ref[:2 * L]))
    worst = max(worst, np.max(np.abs(dy[:2 * L] - ref[:2 * L])) / scale)
    worst_rel = max(worst_rel, np.max(np.abs(dy[:2 * L] - ref[:2 * L]) / np.abs(ref[:2 * L])))
print(f"dN,dP: worst |err| / max|ref| = {worst:.2e}, worst relative = {worst_rel:.2e}")
