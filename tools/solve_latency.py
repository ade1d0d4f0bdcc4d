"""Latency of the drop-in forward_solver.solve() -- the B = 1 case of the host-pointer entry point: H2D of one
descriptor and the time grid, one workgroup on an otherwise empty GPU, D2H of 1001 samples (dev tool)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import forward_solver as fs, workloads as W
from metrotrpl_amd.sim_utils import Grid
ix, un, s0 = W.indexes(), W.units(), W.initial_state()
t = W.default_times()
for label, th, flu in (("thick film, high fluence", 2000.0, W.STAUB_FLUENCES[5]), ("thin film, low fluence", 311.0, W.STAUB_FLUENCES[0])):
    g = Grid(th, 128, t, 4.0)
    fs.solve(np.array(flu), g, s0.copy(), ix, units=un, ini_mode="fluence")
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        fs.solve(np.array(flu), g, s0.copy(), ix, units=un, ini_mode="fluence")
    print(f"solve(), nx=128, 1001 samples, {label}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per call")
# the host-pointer batch entry point on the config-2 batch: descriptors up, 1024 x 1001 samples down (PCIe-inclusive)
from metrotrpl_amd import batch
c = W.config2(1024)
args = (c["states"], ix, un, c["iniPar"], c["thickness"], c["nx"], c["tSteps"])
batch.solve_batch(*args)
n = 10
t0 = time.perf_counter()
for _ in range(n):
    batch.solve_batch(*args)
dt = (time.perf_counter() - t0) / n
print(f"solve_batch(), config 2, 1024 solves, host arrays in and out: {dt * 1e3:.2f} ms per call = {1024 / dt:.0f} solves/s")
