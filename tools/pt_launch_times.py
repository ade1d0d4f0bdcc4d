"""Per-launch solve / likelihood kernel times of the PT loop (dev tool): python tools/pt_launch_times.py [chains] [iters] [prefetch]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import pt, pt_bench
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prefetch = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
groups = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda", 0)
S = pt_bench.setup(n_chains, dev, 0, sort_by_cost=os.environ.get("MTRPL_NO_SORT") != "1",
                   sort_min=int(os.environ.get("MTRPL_SORT_MIN", "1024")), groups=groups,
                   parts=int(os.environ.get("MTRPL_PARTS", "2" if groups == 1 else "1")))
eng = S["eng"]
ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluators"] if groups > 1 else S["evaluator"],
                    rng_mode="counter", streams=S["streams"])
ens.initialise()
ens.advance(S["proposer"], 4, 10, prefetch)
torch.cuda.synchronize()
for e_ in S["engines"]:
    e_.timing = eng.timing = [] if e_ is eng else eng.timing
kinds = []
inner = eng.descriptors
def spy(states):
    out = inner(states)
    kinds.append("measured" if getattr(eng, "_cost_kind", None) == "measured" else str(getattr(eng, "_cost_kind", "?")))
    return out
eng.descriptors = spy
if os.environ.get("PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
t0 = time.perf_counter()
ens.advance(S["proposer"], iters, 10, prefetch)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if os.environ.get("PROFILE"):
    pr.disable()
    pstats.Stats(pr).sort_stats("cumtime").print_stats(35)
print(f"{n_chains} chains, prefetch={prefetch}, groups={groups}: {dt / iters * 1e3:.3f} ms/iteration")
if groups > 1:
    kinds = ["-"] * len(eng.timing)
for (e0, e1, e2, n), k in zip(eng.timing, kinds):
    st = eng.d_stats[:8 * n].view(-1, 8)
    print(f"  n={n:5d} solve {e0.elapsed_time(e1):7.3f} ms  loglik {e1.elapsed_time(e2):6.3f} ms  order={k}")
steps = eng.d_stats[:8 * eng.n_solves].view(-1, 8)[:, 0].cpu().numpy()
print("last launch steps: mean", steps.mean(), "max", steps.max())
