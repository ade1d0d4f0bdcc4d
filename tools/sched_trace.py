"""Where and when every solve of one prefetched PT launch ran (dev tool; -DMTRPL_SCHED diagnostic build).
usage: python tools/sched_trace.py [chains]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "metrotrpl_amd", "lib", "libmetrotrpl_amd_sched.so")
CSRC = os.path.join(REPO, "metrotrpl_amd", "csrc")
_newest = max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC))
if not os.path.exists(LIB) or os.path.getmtime(LIB) < _newest:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DMTRPL_SCHED", "-o", LIB, os.path.join(CSRC, "mtrpl_capi.hip")])
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["MTRPL_LIB_PATH"] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: E402
sys.path.insert(0, REPO)
from metrotrpl_amd import pt, pt_bench  # noqa: E402

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
S = pt_bench.setup(n_chains, dev, 0)
eng = S["eng"]
ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluator"], rng_mode="counter")
ens.initialise()
ens.advance(S["proposer"], 6, 10, True)
torch.cuda.synchronize()
eng.timing = []
ens.advance(S["proposer"], 2, 10, True)
torch.cuda.synchronize()
e0, e1, e2, n = eng.timing[-1]
kern_ms = e0.elapsed_time(e1)
st = eng.d_stats[:8 * n].view(-1, 8).cpu().numpy().astype(np.int64)
xcc = st[:, 4] & 7
start = (st[:, 5] - st[:, 5].min()) & 0x7fffffff   # 100 MHz real-time counter, one origin for the whole device
end = (st[:, 3] - st[:, 5].min()) & 0x7fffffff
dur = end - start
tick = 1e-5  # ms per count
print(f"{n} solves, kernel {kern_ms:.3f} ms (HIP events); last end {end.max() * tick:.3f} ms; positions in launch order")
print("solves per XCD:", np.bincount(xcc, minlength=8))
print(f"slowest solve {dur.max() * tick:.3f} ms ({st[dur.argmax(), 0]} steps, started {start[dur.argmax()] * tick:.3f} ms, "
      f"position {dur.argmax()})")
late = np.where(start * tick > 0.2)[0]
if late.size:
    print(f"{late.size} solves started later than 0.2 ms: start min {start[late].min() * tick:.3f} median "
          f"{np.median(start[late]) * tick:.3f} max {start[late].max() * tick:.3f} ms; positions {late.min()}..{late.max()}")
for i in np.argsort(-end)[:10]:
    print(f"  pos {i:5d} xcc {xcc[i]} hw {st[i, 6] & 0xffffffff:08x} start {start[i] * tick:6.3f} dur {dur[i] * tick:6.3f} "
          f"end {end[i] * tick:6.3f} ms steps {st[i, 0]}")
print("duration by position (mean of blocks of 128):", np.round(dur.reshape(-1, 128).mean(axis=1) * tick, 2))
print("start by position (max of blocks of 128):   ", np.round(start.reshape(-1, 128).max(axis=1) * tick, 2))
print("end per XCD:", np.round([end[xcc == x].max() * tick for x in range(8)], 3))
per_step = dur * tick * 1e3 / np.maximum(st[:, 0], 1)
print(f"us per step: mean {per_step.mean():.3f}, of the 64 slowest solves {per_step[np.argsort(-dur)[:64]].mean():.3f}")
