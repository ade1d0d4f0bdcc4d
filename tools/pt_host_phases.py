"""Where the host's time goes in the prefetching 64-chain ensemble (dev tool): seconds per launch cycle in each phase
of pt.PTEnsemble (propose / enqueue = pack + upload + kernel launches / wait for the launch / decide / apply / exchange).
    python tools/pt_host_phases.py [chains] [iterations]"""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # as bench.py does: every chain group's stream on a hardware queue of its own
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import pt, pt_bench  # noqa: E402

n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
groups = int(os.environ.get("PT_GROUPS", "3"))  # (not GROUPS: bash owns that name)
small = groups > 1
on_device = os.environ.get("PT_DECIDE", "device") == "device"
dealt = groups > 2 and on_device  # as pt_bench._run sets the latency-bound 64-chain ensemble up
S = pt_bench.setup(n_chains, dev, 0, parts=1, groups=groups, dealt=groups - 1 if dealt else None)
ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluators"] if groups > 1 else S["evaluator"], rng_mode="counter",
                    streams=S["streams"], group=pt.LOCAL_ONLY, device_decisions=on_device, regroup_by_cost=dealt)
ens.initialise()
ens.advance(S["proposer"], 20, 10, True)
torch.cuda.synchronize()
if os.environ.get("NOGC") == "1":
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
rounds = []
_call = S["proposer"].__class__.__call__


def counting(self, states, temperatures, rng):  # how many redraw rounds does a proposer call take?
    class R:
        def __init__(self, r):
            self.r, self.n = r, 0

        def random(self, shape):
            self.n += 1
            return self.r.random(shape)
    r = R(rng)
    out = _call(self, states, temperatures, r)
    rounds.append(r.n)
    return out


S["proposer"].__class__.__call__ = counting
ens.prof = {}
for e in S["engines"]:
    e.prof = {}
marks = []
_ex = ens._exchange


def marked(it):
    marks.append(time.perf_counter())
    _ex(it)


ens._exchange = marked
t0 = time.perf_counter()
ens.advance(S["proposer"], iters, 10, True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
cycles = iters / 2 * groups
print(f"{n_chains} chains, {groups} groups: {dt / iters * 1e3:.3f} ms per iteration; host ms per launch cycle of one group:")
for k, v in ens.prof.items():
    if not k.endswith(".max"):
        print(f"  {k:10s} {v / (iters / 10 if k == 'exchange' else cycles) * 1e3:7.3f}" + (" (per exchange)" if k == "exchange" else "")
              + f"   longest {ens.prof[k + '.max'] * 1e3:.3f}")
up = sum(e.prof.get("upload", 0) for e in S["engines"]) / cycles * 1e3
la = sum(e.prof.get("launch", 0) for e in S["engines"]) / cycles * 1e3
print(f"  of enqueue: pack + upload {up:.3f}, kernel launches {la:.3f}")
print("redraw rounds per proposer call: mean", np.mean(rounds), "max", np.max(rounds), "calls", len(rounds), "calls with > 10 rounds", int(np.sum(np.array(rounds) > 10)))
print("ms per stretch of ten iterations:", " ".join(f"{(b - a) * 1e3:.1f}" for a, b in zip([t0] + marks[:-1], marks)))
print("sum of the phases above", round(sum(v for k, v in ens.prof.items() if not k.endswith('.max')) * 1e3, 1), "ms of", round(dt * 1e3, 1))
