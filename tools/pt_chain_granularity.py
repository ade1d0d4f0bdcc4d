"""tools/pt_chain_granularity.py -- what a chain-granular pipeline could buy a parallel-tempering stretch.

Between two replica exchanges the chains do not interact, so in principle every chain could run its launches at its own
pace: a stretch of K launches would then take max_chains sum_launches (slowest solve of the chain's launch) instead of
sum_launches max_chains (...), the kernel-boundary barrier of a chain group.  The two differ only if the slowest chain
CHANGES from launch to launch.  Measured here on the config-3 ensemble (64 chains x 6 measurements, the states of
workloads.pt_ensemble): a Metropolis walk of K iterations with every proposal accepted / rejected at random, the
integrator steps of every (chain, measurement) solve per iteration from the solve kernel's own statistics."""
import sys

import numpy as np

sys.path.insert(0, ".")
from metrotrpl_amd import batch, pt, workloads as W  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_chains = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ens0 = W.pt_ensemble(n_chains, 128)
T = ens0["temperatures"]
names, ix = W.PARAM_NAMES, W.indexes()
active = [n in W.PRIOR_BOX for n in names]
prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
prop = pt.Proposer(np.where(active, 0.003, 0.0), names, active, prior, np.ones(len(names), dtype=bool), ix)
rng = np.random.default_rng(5)
states = ens0["chain_states"].copy()
t = W.default_times()
steps = np.zeros((K, n_chains, 6), dtype=int)
for k in range(K):
    y = prop(states, T, rng)
    st = np.repeat(y, 6, axis=0)
    ini = np.tile(np.array(W.STAUB_FLUENCES, dtype=float), (n_chains, 1))
    th = np.tile(np.array(W.STAUB_LENGTHS, dtype=float), n_chains)
    _, status, stats = batch.solve_batch(st, ix, W.units(), ini, th, 128, t, return_stats=True)
    assert np.all(status == 0)
    steps[k] = stats[:, 0].reshape(n_chains, 6)
    acc = rng.random(n_chains) < 0.3
    states[acc] = y[acc]
per_chain = steps.max(axis=2)                      # [K, chains]: the slowest solve of a chain's launch
barrier = per_chain.max(axis=1).sum()              # sum over launches of the slowest chain
granular = per_chain.sum(axis=0).max()             # the slowest chain's own sum
for G in (1, 3, 8):
    grp = [per_chain[:, g::G].max(axis=1).sum() for g in range(G)]
    print(f"{G} chain group(s): stretch of {K} launches waits for {max(grp)} integrator steps (groups: {grp})")
print(f"chain-granular: {granular} steps = {granular / barrier:.3f} of the one-group barrier; mean solve {steps.mean():.0f}, "
      f"slowest solve of a launch {per_chain.max(axis=1).mean():.0f}; the slowest chain is chain {per_chain.sum(axis=0).argmax()} "
      f"in {int((per_chain.argmax(axis=1) == per_chain.sum(axis=0).argmax()).sum())} of {K} launches")
