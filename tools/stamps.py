"""Phase breakdown from the -DMTRPL_STAMPS diagnostic build (dev tool)."""
import sys, os
os.environ["MTRPL_LIB_PATH"] = "/root/repo/metrotrpl_amd/lib/libmetrotrpl_amd_stamps.so"
import numpy as np
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import batch, workloads as W
c = W.config2(1024); ix, un = W.indexes(), W.units()
for rep in range(2):
    sig, status, st = batch.solve_batch(c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"], return_stats=True)
names = ["steps", "sel+chgD", "pred", "build_fac", "newton", "err+reject", "D update", "dense out"]
tot = st[:, 1:].sum(axis=1).astype(float)
print("mean steps", st[:, 0].mean(), "mean total kcycles", tot.mean(), "max", tot.max())
for k in range(1, 8):
    print(f"{names[k]:12s}: {st[:,k].mean():9.0f} kcyc  {100*st[:,k].sum()/tot.sum():5.1f}%   per step {1024*st[:,k].mean()/st[:,0].mean():7.0f} cyc")
