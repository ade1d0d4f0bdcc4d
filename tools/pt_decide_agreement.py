"""Do the decisions taken on the device (mtrpl_pt_decide_dev: `u < exp(L_new - L_cur)` with the device's exp) ever differ
from numpy's on the host?  The same 64-chain ensemble walked both ways for N iterations; every state, log-likelihood and
swap is compared (dev tool).
    python tools/pt_decide_agreement.py [iterations] [chains]"""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import pt, pt_bench  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_chains = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
out = {}
for mode in ("host", "device"):
    groups = 2 if mode == "host" else 3
    S = pt_bench.setup(n_chains, dev, 0, parts=1, groups=groups)
    ens = pt.PTEnsemble(S["ens0"]["chain_states"], S["T"], S["evaluators"], rng_mode="counter", streams=S["streams"],
                        group=pt.LOCAL_ONLY, device_decisions=mode == "device")
    ens.initialise()
    n_acc = ens.advance(S["proposer"], iters, 10, True)
    torch.cuda.synchronize()
    out[mode] = (ens.states.copy(), ens.logll.copy(), ens.table.copy(), ens.swap_accepts.copy(), ens.acc_count.copy(), n_acc,
                 ens.n_pipelined)
h, d = out["host"], out["device"]
same = all(np.array_equal(a, b) for a, b in zip(h[:5], d[:5]))
print(f"{n_chains} chains, {iters} iterations: {h[5]} accepted moves (host), {d[5]} (device), {int(h[3].sum())} swaps; "
      f"{d[6]} launches decided on the device; states, log-likelihoods, table, swaps, accept counts identical: {same}")
sys.exit(0 if same else 1)
