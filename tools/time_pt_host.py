"""Host vs device time of one PT iteration (dev tool)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import _lib as L, laplace, pt, workloads as W, pt_bench
from metrotrpl_amd.likelihood import LikelihoodEngine
n_chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
times = W.default_times()
irf_tables = laplace.make_I_tables({745: pt_bench.synthetic_irf()})
vals, uncs = pt_bench.synthetic_data(irf_tables, 128, times)
eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, 128, ["TRPL"] * 6, [times] * 6, vals, uncs, {"TRPL": 1.0},
                       np.array(W.STAUB_FLUENCES), irf_tables=irf_tables, irf_convolution=[745] * 6)
ens0 = W.pt_ensemble(n_chains); T = ens0["temperatures"]
d_T = torch.from_numpy(np.ascontiguousarray(T)).cuda()
names = W.PARAM_NAMES; ix = W.indexes()
active = [n in W.PRIOR_BOX for n in names]; prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
do_log = np.ones(len(names), dtype=bool); base_move = np.where(active, 0.003, 0.0)
rng = np.random.default_rng(1)
states = ens0["chain_states"].copy()
acc = {k: 0.0 for k in ("propose", "pack", "h2d", "launch+gpu", "d2h", "accept+swap")}
N = 10
for it in range(N + 2):
    t0 = time.perf_counter()
    new = pt.propose(states, T, base_move, names, active, prior, do_log, ix, rng)
    t1 = time.perf_counter()
    d, ini_arr, ld = eng.descriptors(new)
    t2 = time.perf_counter()
    eng.n_solves = d.shape[0]
    eng.d_descs = torch.from_numpy(d.view(np.uint8).reshape(-1)).cuda(); eng.d_ini = torch.from_numpy(ini_arr).cuda(); eng.d_like = torch.from_numpy(ld.view(np.uint8).reshape(-1)).cuda()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    ll = eng.launch(d_T, len(T)); tab = ll.view(-1, eng.M, len(T)).sum(dim=1)
    torch.cuda.synchronize(); t4 = time.perf_counter()
    table = tab.cpu().numpy(); t5 = time.perf_counter()
    a = pt.metropolis_accept(np.zeros(n_chains), table[np.arange(n_chains), np.arange(n_chains)], rng)
    owner, att, accs = pt.replica_exchange(table, rng); t6 = time.perf_counter()
    if it >= 2:
        for k, v in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)): acc[k] += v
print(f"{n_chains} chains:", {k: round(v / N * 1e3, 2) for k, v in acc.items()}, "ms; total", round(sum(acc.values()) / N * 1e3, 2))
