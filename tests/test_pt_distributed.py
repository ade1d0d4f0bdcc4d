"""CPU-only: the N>1 path of the sharded parallel-tempering ensemble with the
`gloo` backend (world_size 2 and 3), plus unit tests of the host logic that
restates metropolis.py:42-90,:129-135 and trial_move_generation.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from metrotrpl_amd import pt, workloads as W  # noqa: E402


def _toy_evaluator(states_block, temps):
    """ll(state, T) = -sum((log10 state - mu)^2) / (T + 0.3): not 1/T-scaling, like the real one."""
    x = np.log10(np.abs(states_block[:, 1:5]) + 1e-300)
    e = np.sum((x - np.array([15.5, 1.2, 1.3, -10.3])) ** 2, axis=1)
    return torch.from_numpy(-e[:, None] * 50.0 / (np.asarray(temps)[None, :] + 0.3))


def _proposer():
    names, ix = W.PARAM_NAMES, W.indexes()
    active = [n in W.PRIOR_BOX for n in names]
    prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
    do_log = np.ones(len(names), dtype=bool)
    base = np.where(active, 0.05, 0.0)
    return lambda s, T, rng: pt.propose(s, T, base, names, active, prior, do_log, ix, rng)


def _run_serial(n_chains, iters):
    ens0 = W.pt_ensemble(n_chains)
    e = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], _toy_evaluator)
    e.initialise()
    prop = _proposer()
    for k in range(1, iters + 1):
        e.iterate(prop, k % 3 == 0)
    return e


def _worker(rank, world, port, n_chains, iters, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ens0 = W.pt_ensemble(n_chains)
    e = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], _toy_evaluator, rank=rank, world=world)
    e.initialise()
    prop = _proposer()
    for k in range(1, iters + 1):
        e.iterate(prop, k % 3 == 0)
    q.put((rank, e.states.copy(), e.logll.copy(), e.swap_accepts.copy(), (e.lo, e.hi)))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_chains", [(2, 8), (3, 8)])
def test_sharded_ensemble_equals_serial(world, n_chains):
    """Every rank ends with the same replicated ensemble, identical to the 1-rank run
    (even and uneven chain blocks)."""
    iters = 9
    ref = _run_serial(n_chains, iters)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chains, iters, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    blocks = sorted(r[4] for r in res)
    assert blocks == pt.partition(n_chains, world)
    for _, states, logll, swaps, _ in res:
        np.testing.assert_array_equal(states, ref.states)
        np.testing.assert_array_equal(logll, ref.logll)
        np.testing.assert_array_equal(swaps, ref.swap_accepts)
    assert ref.swap_attempts.sum() == 3 * (n_chains - 1)


class _PackedToy:
    """The toy evaluator with the two extra entry points of an engine-backed one (pt_bench.setup): `pack_rows` lays M
    'descriptors' per state row (here: the state's own bytes, and where a real one holds its output offset, the row's
    position in its set) and `launch_packed` evaluates whatever descriptors the caller hands back."""
    M = 2

    def __init__(self, calls, own_order=False):
        self.calls, self.own_order, self.order = calls, own_order, None
        if own_order:
            self.plan_packed = self._plan_packed
        else:
            self.row_cost = self._row_cost

    def _plan_packed(self, states):
        """An evaluator that issues its solves in an order of its own, fixed per stretch between two exchanges."""
        n = states.shape[0] * self.M
        self.prng = getattr(self, "prng", None) or np.random.default_rng(11)
        self.order = self.prng.permutation(n)
        return self.order // self.M

    def __call__(self, block, temps):
        self.calls.append(block.shape[0])
        return _toy_evaluator(block, temps)

    def pack_rows(self, states, sets=1):
        st = np.ascontiguousarray(states, dtype=np.float64)
        pos = np.tile(np.arange(st.shape[0] // sets, dtype=np.float64), sets)
        d = np.repeat(np.concatenate([st, pos[:, None]], axis=1), self.M, axis=0)
        if self.order is not None and self.order.shape[0] == d.shape[0] // sets:
            d = d.reshape(sets, d.shape[0] // sets, -1)[:, self.order].reshape(d.shape[0], -1)
        return np.ascontiguousarray(d).view(np.uint8).reshape(d.shape[0], -1)

    def _row_cost(self, n_rows):  # "the last launch's cost per state row": makes the pipe re-order its chains every segment
        self.rng = getattr(self, "rng", None) or np.random.default_rng(5)
        return self.rng.random(n_rows)

    def launch_packed(self, d, n_solves, temps):
        x = d.view(torch.float64).view(n_solves, -1).numpy()
        if self.order is not None and self.order.shape[0] == n_solves:
            back = np.empty_like(x)
            back[self.order] = x
            x = back
        np.testing.assert_array_equal(x[:, -1], np.repeat(np.arange(n_solves // self.M), self.M))  # every row in its place
        self.calls.append(("packed", n_solves // self.M))
        return _toy_evaluator(x[::self.M, :-1], temps)


def _run_counter(n_chains, iters, prefetch, rank=0, world=1, temper_freq=4, groups=1, device_decisions=False,
                 own_order=False, regroup=False):
    ens0 = W.pt_ensemble(n_chains)
    calls = []

    def ev(block, temps):
        calls.append(block.shape[0])
        return _toy_evaluator(block, temps)
    if device_decisions:
        ev = [_PackedToy(calls, own_order) for _ in range(groups)]
        ev = ev[0] if groups == 1 else ev
    e = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], ev if groups == 1 or device_decisions else [ev] * groups, rank=rank,
                      world=world, rng_mode="counter", device_decisions=device_decisions, regroup_by_cost=regroup)
    e.initialise()
    n_acc = e.advance(_proposer(), iters, temper_freq, prefetch=prefetch)
    return e, n_acc, calls


def test_prefetched_iterations_walk_the_same_chain():
    """iterate2 (both outcomes of iteration k evaluated ahead) == one iteration at a time, bit for bit,
    with half the evaluator launches between replica exchanges."""
    n_chains, iters = 8, 23
    seq, acc_s, calls_s = _run_counter(n_chains, iters, False)
    pre, acc_p, calls_p = _run_counter(n_chains, iters, True)
    assert acc_s == acc_p and 0 < acc_s < n_chains * iters
    np.testing.assert_array_equal(pre.states, seq.states)
    np.testing.assert_array_equal(pre.logll, seq.logll)
    np.testing.assert_array_equal(pre.table, seq.table)
    np.testing.assert_array_equal(pre.swap_accepts, seq.swap_accepts)
    assert seq.swap_attempts.sum() == (iters // 4) * (n_chains - 1) and seq.swap_accepts.sum() > 0
    assert pre.it == seq.it == iters
    assert len(calls_s) == 1 + iters
    # every replica exchange is timed in three parts (bench.py reports them as exchange_ms), and a library call leaves the
    # process's garbage collector as it found it unless asked (gc_freeze is opt-in)
    for e in (seq, pre):
        assert e.exchange_s["n"] == iters // 4 and all(e.exchange_s[k] >= 0.0 for k in ("wait", "gather", "replay"))
        assert e.exchange_s["wait"] + e.exchange_s["gather"] + e.exchange_s["replay"] > 0.0
        assert e.gc_freeze is False
    import gc
    assert gc.isenabled() and gc.get_freeze_count() == 0
    # the exchange falls after even iterations, so every pair (1,2), (3,4), ... goes out as one launch
    assert calls_p.count(3 * n_chains) == iters // 2 and len(calls_p) == 1 + iters // 2 + 1
    odd, _, calls_o = _run_counter(n_chains, 9, True, temper_freq=3)  # an exchange between 3 and 4: no pair there
    ref3, _, _ = _run_counter(n_chains, 9, False, temper_freq=3)
    np.testing.assert_array_equal(odd.states, ref3.states)
    assert calls_o[1:] == [3 * n_chains, n_chains, 3 * n_chains, n_chains, 3 * n_chains, n_chains]
    with pytest.raises(RuntimeError):
        pt.PTEnsemble(seq.states, seq.T, _toy_evaluator).iterate2(_proposer(), False)


@pytest.mark.parametrize("groups,prefetch", [(2, False), (2, True), (3, True)])
def test_chain_groups_walk_the_same_chain(groups, prefetch):
    """The ensemble cut into interleaved chain groups, each with its own evaluator launches between two
    replica exchanges (the pipelined form), == the whole ensemble one iteration at a time."""
    n_chains, iters = 8, 19
    seq, acc_s, _ = _run_counter(n_chains, iters, False)
    grp, acc_g, calls = _run_counter(n_chains, iters, prefetch, groups=groups)
    assert acc_g == acc_s and grp.it == iters and grp.git == [iters] * groups
    np.testing.assert_array_equal(grp.states, seq.states)
    np.testing.assert_array_equal(grp.logll, seq.logll)
    np.testing.assert_array_equal(grp.table, seq.table)
    np.testing.assert_array_equal(grp.swap_accepts, seq.swap_accepts)
    sizes = sorted({len(range(g, n_chains, groups)) for g in range(groups)})
    assert set(calls[1:]) <= {k * s_ for s_ in sizes for k in (1, 3)}
    # iterate() / iterate2() on a grouped ensemble: every group, then the exchange
    a = pt.PTEnsemble(seq.states, seq.T, [_toy_evaluator] * groups, rng_mode="counter")
    b = pt.PTEnsemble(seq.states, seq.T, _toy_evaluator, rng_mode="counter")
    a.initialise(), b.initialise()
    acc_a = a.iterate2(_proposer(), True)
    acc_b = (b.iterate(_proposer(), False), b.iterate(_proposer(), True))
    for x, y in zip(acc_a, acc_b):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(a.states, b.states)
    with pytest.raises(ValueError):
        pt.PTEnsemble(seq.states, seq.T, [_toy_evaluator] * 2)  # groups need counter-mode draws


@pytest.mark.parametrize("groups,temper_freq,iters,prefetch,own_order",
                         [(1, 4, 23, True, False), (1, 10, 31, True, False), (2, 10, 27, True, False), (3, 6, 19, True, True),
                          (1, 3, 9, True, False), (1, 5, 17, False, False), (2, 4, 11, False, True), (1, 10, 20, False, True)])
def test_device_side_decisions_walk_the_same_chain(groups, temper_freq, iters, prefetch, own_order):
    """Prefetching with the decisions taken where the evaluator's results are (`_DecidePipe`: the host prepares the
    four proposal sets the two decisions of the running launch can ask for, the device picks; two launches in flight,
    segments of two to five launches, odd leftovers through the host path) == one iteration at a time, bit for bit."""
    n_chains = 8
    seq, acc_s, _ = _run_counter(n_chains, iters, False, temper_freq=temper_freq)
    dev, acc_d, calls = _run_counter(n_chains, iters, prefetch, temper_freq=temper_freq, groups=groups, device_decisions=True,
                                     own_order=own_order)
    assert acc_d == acc_s and dev.it == iters and dev.git == [iters] * groups
    np.testing.assert_array_equal(dev.states, seq.states)
    np.testing.assert_array_equal(dev.logll, seq.logll)
    np.testing.assert_array_equal(dev.table, seq.table)
    np.testing.assert_array_equal(dev.acc_count, seq.acc_count)
    np.testing.assert_array_equal(dev.swap_accepts, seq.swap_accepts)
    packed = [c for c in calls if isinstance(c, tuple)]
    depth = 2 if prefetch else 1
    pairs = sum((min(iters, (k + 1) * temper_freq) - k * temper_freq) // depth for k in range(-(-iters // temper_freq)))
    assert len(packed) == dev.n_pipelined == groups * pairs  # every whole launch of a segment went through the pipe
    assert all(isinstance(c, tuple) or c in (n_chains, len(range(0, n_chains, groups)), len(range(groups - 1, n_chains, groups)))
               for c in calls)  # ... and nothing else was a three-set launch


def test_an_evaluator_that_cannot_pack_walks_the_host_path():
    """`packable = False` (a LikelihoodEngine with a fitted scale factor): device_decisions is asked for, the group's
    launches go through `_begin` / `_end` all the same, and the chain is the same."""
    n_chains, iters = 8, 12
    seq, acc_s, _ = _run_counter(n_chains, iters, False)
    ens0 = W.pt_ensemble(n_chains)
    calls = []
    ev = _PackedToy(calls)
    ev.packable = False
    e = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], ev, rng_mode="counter", device_decisions=True)
    e.initialise()
    acc = e.advance(_proposer(), iters, 4, prefetch=True)
    assert acc == acc_s and e.n_pipelined == 0 and not any(isinstance(c, tuple) for c in calls)
    np.testing.assert_array_equal(e.states, seq.states)
    np.testing.assert_array_equal(e.table, seq.table)


@pytest.mark.parametrize("groups,n_chains,prefetch", [(2, 8, True), (3, 8, True), (3, 11, False)])
def test_chain_groups_dealt_by_cost_walk_the_same_chain(groups, n_chains, prefetch):
    """`regroup_by_cost`: at every replica exchange the chains are dealt to the groups again, the most expensive ones (what
    the evaluators' `row_cost` said of the stretch before) to the group that starts first.  Which group evaluates a chain is
    no part of its walk: the one-iteration-at-a-time chain, bit for bit -- and the groups did change hands."""
    iters, temper_freq = 26, 6
    seq, acc_s, _ = _run_counter(n_chains, iters, False, temper_freq=temper_freq)
    seen = []
    keep = pt._DecidePipe.__init__

    def spy(self, ens, g, *a, **k):
        keep(self, ens, g, *a, **k)
        seen.append((g, tuple(sorted(self.sel.tolist()))))
    pt._DecidePipe.__init__ = spy
    try:
        dev, acc_d, _ = _run_counter(n_chains, iters, prefetch, temper_freq=temper_freq, groups=groups, device_decisions=True,
                                     regroup=True)
    finally:
        pt._DecidePipe.__init__ = keep
    assert acc_d == acc_s and dev.it == iters and dev.git == [iters] * groups
    np.testing.assert_array_equal(dev.states, seq.states)
    np.testing.assert_array_equal(dev.logll, seq.logll)
    np.testing.assert_array_equal(dev.table, seq.table)
    np.testing.assert_array_equal(dev.acc_count, seq.acc_count)
    np.testing.assert_array_equal(dev.swap_accepts, seq.swap_accepts)
    static = {g: tuple(range(g, n_chains, groups)) for g in range(groups)}
    stretches = [seen[k:k + groups] for k in range(0, len(seen), groups)]
    assert all(sorted(c for _, m in st for c in m) == list(range(n_chains)) for st in stretches)  # every chain, once
    assert all(len(m) == len(static[g]) for st in stretches for g, m in st)  # the groups keep their sizes
    assert [m for g, m in stretches[0]] == [static[g] for g in range(groups)]  # no cost known yet: as constructed
    assert any(m != static[g] for st in stretches[1:] for g, m in st)


def _prefetch_worker(rank, world, port, n_chains, iters, groups, q, device_decisions=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    e, n_acc, _ = _run_counter(n_chains, iters, True, rank, world, groups=groups, device_decisions=device_decisions)
    q.put((rank, e.states.copy(), e.logll.copy(), e.swap_accepts.copy(), n_acc, e.n_collectives, e.table.copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_chains,groups,device_decisions", [(2, 8, 1, False), (3, 8, 1, False), (2, 8, 2, False),
                                                                   (3, 7, 2, False), (2, 8, 2, True), (3, 7, 1, True)])
def test_prefetched_sharded_ensemble_equals_serial(world, n_chains, groups, device_decisions):
    """Counter-mode draws: every rank proposes and decides for its own chain block only (three proposal sets per
    launch when prefetching) and the ranks exchange ONCE per replica exchange -- even and uneven chain blocks,
    whole and in chain groups, decisions on the host or where the evaluator's results are: the walk of the one-rank,
    one-iteration-at-a-time run, bit for bit."""
    iters = 11
    ref, acc_ref, _ = _run_counter(n_chains, iters, False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_prefetch_worker, args=(r, world, port, n_chains, iters, groups, q, device_decisions))
             for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, states, logll, swaps, n_acc, n_coll, table in res:
        np.testing.assert_array_equal(states, ref.states)
        np.testing.assert_array_equal(logll, ref.logll)
        np.testing.assert_array_equal(table, ref.table)
        np.testing.assert_array_equal(swaps, ref.swap_accepts)
        assert n_acc == acc_ref
        # one all-gather at initialise, one per replica exchange (iterations 4 and 8), one to complete the
        # replicated view after the last iteration: NOT one per iteration
        assert n_coll == 1 + iters // 4 + 1
    # what the walk carries is what the evaluator says about the states it ended on
    want = _toy_evaluator(ref.states, ref.T).numpy()
    np.testing.assert_array_equal(ref.table, want)
    np.testing.assert_array_equal(ref.logll, np.diag(want))
    assert acc_ref == int(ref.acc_count.sum()) > 0


def test_partition():
    assert pt.partition(512, 8) == [(64 * r, 64 * (r + 1)) for r in range(8)]
    assert pt.partition(8, 3) == [(0, 3), (3, 6), (6, 8)]
    assert pt.partition(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]


def test_replica_exchange_matches_reference_rule():
    """Replay of metropolis.py:129-135 / :66-90 against a direct transcription that
    carries explicit per-chain ll(T) callables, with the same generator stream."""
    rng_t = np.random.default_rng(7)
    n = 6
    T = W.ladder(n, 50.0)
    e = rng_t.uniform(10, 200, n)
    u2 = 0.05
    ll = lambda c, k: -e[c] / (T[k] + u2)  # noqa: E731
    table = np.array([[ll(c, k) for k in range(n)] for c in range(n)])
    owner, att, acc = pt.replica_exchange(table, np.random.default_rng(3))
    rng = np.random.default_rng(3)
    who = list(range(n))
    logll = [ll(c, c) for c in range(n)]
    for _ in range(n - 1):
        i = int(rng.integers(0, n - 1))
        bi_ui, bj_ui = ll(who[i], i), ll(who[i], i + 1)
        bi_uj, bj_uj = ll(who[i + 1], i), ll(who[i + 1], i + 1)
        if rng.random() < np.exp(-(bi_ui + bj_uj - bi_uj - bj_ui)):
            logll[i], logll[i + 1] = bi_uj, bj_ui
            who[i], who[i + 1] = who[i + 1], who[i]
    assert list(owner) == who
    np.testing.assert_array_equal(table[owner, np.arange(n)], logll)
    assert att.sum() == n - 1 and acc.sum() <= att.sum()


def test_replica_exchange_with_batched_draws_applies_the_same_rule():
    """The counter-mode form (pair indices and accept rolls drawn as two arrays, math.exp) takes, attempt by attempt, the
    decision np.exp takes on the same pair and roll -- NaN (inf - inf), +-inf and overflowing ratios included."""
    for trial in range(60):
        rng = np.random.default_rng(trial)
        n = int(rng.integers(2, 40))
        table = -rng.random((n, n)) * 10.0 ** rng.integers(0, 4)
        for _ in range(3):
            table[rng.integers(0, n), rng.integers(0, n)] = rng.choice([np.nan, -np.inf, np.inf, 1e308, -1e308])
        owner, att, acc = pt.replica_exchange(table, np.random.default_rng(7), batched_draws=True)
        g = np.random.default_rng(7)
        pairs, rolls = g.integers(0, n - 1, size=n - 1), g.random(n - 1)
        who, n_acc = list(range(n)), [0] * n
        with np.errstate(all="ignore"):
            for i, u in zip(pairs, rolls):
                a, b = who[i], who[i + 1]
                if u < np.exp(-(table[a, i] + table[b, i + 1] - table[b, i] - table[a, i + 1])):
                    who[i], who[i + 1] = b, a
                    n_acc[i] += 1
        assert who == owner.tolist() and n_acc == acc.tolist() and att.sum() == n - 1
    o1, _, _ = pt.replica_exchange(np.zeros((1, 1)), np.random.default_rng(0), batched_draws=True)
    assert o1.tolist() == [0]


def test_propose_respects_box_and_physics():
    names, ix = W.PARAM_NAMES, W.indexes()
    active = [n in W.PRIOR_BOX for n in names]
    prior = {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}
    states, _ = W.draw_states(64, with_fluence=False)
    T = W.ladder(64)
    new = pt.propose(states, T, np.where(active, 0.003, 0.0), names, active, prior, np.ones(len(names), bool), ix,
                     np.random.default_rng(0))
    for n_, (lo, hi) in W.PRIOR_BOX.items():
        assert np.all((new[:, ix[n_]] > lo) & (new[:, ix[n_]] < hi))
    assert np.all(np.abs(np.log10(new[:, ix["tauN"]]) - np.log10(new[:, ix["tauP"]])) <= 2 + 1e-12)
    for n_ in ("n0", "eps", "Tm"):
        np.testing.assert_allclose(new[:, ix[n_]], states[:, ix[n_]], rtol=1e-14)
    assert np.mean(np.abs(np.log10(new[-1] / states[-1]))[1:11]) > np.mean(np.abs(np.log10(new[0] / states[0]))[1:11])
    # NaN proposal likelihood is rejected (metropolis.py:49-51)
    acc = pt.metropolis_accept(np.array([-5.0, -5.0, -5.0]), np.array([np.nan, -np.inf, 10.0]), np.random.default_rng(1))
    assert list(acc) == [False, False, True]


# ---- the reference-order driver (metropolis.Walk) with chains sharded over ranks ------------------
def _walk_fields(n_chains):
    names = W.PARAM_NAMES
    active = np.array([n in W.PRIOR_BOX for n in names])
    ens0 = W.pt_ensemble(n_chains)
    sf = {"names": list(names), "do_log": np.ones(len(names), dtype=bool), "active": active,
          "prior_dist": {n: W.PRIOR_BOX.get(n, (0, np.inf)) for n in names}, "_param_indexes": W.indexes(),
          "base_trial_move": np.where(active, 0.05, 0.0), "hard_bounds": 1, "_n_chains": n_chains,
          "_T": list(ens0["temperatures"]), "do_parallel_tempering": True, "temper_freq": 2}
    uf = [{"_T": t} for t in ens0["temperatures"]]
    return sf, uf, ens0["chain_states"]


def _walk_run(n_chains, iters, mode, rank=0, world=1):
    from metrotrpl_amd import metropolis as M
    sf, uf, st0 = _walk_fields(n_chains)
    blocks = pt.partition(n_chains, world)

    def gather(states):
        return pt.gather_ll_table(_toy_evaluator, states, np.asarray(sf["_T"]), blocks, rank, world)
    states = np.zeros((n_chains, st0.shape[1], iters))
    states[:, :, 0] = st0
    logll = np.zeros((n_chains, iters))
    accept = np.zeros((n_chains, iters), dtype=int)
    walk = M.Walk(sf, uf, np.random.default_rng(M.SEED), gather, rng_mode=mode)
    walk.run(states, logll, accept, 0, iters, need_initial_state=True)
    return states, logll, accept, walk.swap_accept.copy()


def _walk_worker(rank, world, port, n_chains, iters, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank,) + _walk_run(n_chains, iters, mode, rank, world))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["serial", "mpi"])
def test_reference_order_walk_sharded_equals_one_rank(mode):
    n_chains, iters, world = 5, 9, 2
    ref = _walk_run(n_chains, iters, mode)
    assert ref[2].sum() > 0 and ref[3].sum() > 0  # moves and swaps do happen
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_walk_worker, args=(r, world, port, n_chains, iters, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=90) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        for got, want in zip(r[1:], ref):
            np.testing.assert_array_equal(got, want)


def _local_only_worker(rank, world, port, q):
    """bench.py at N > 1: the ranks run a sharded ensemble together, THEN rank 0 runs one alone while the others wait
    at a barrier.  The lone ensemble (group=pt.LOCAL_ONLY) must issue no collective on the job's process group -- it
    would pair up with the others' barrier."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ens0 = W.pt_ensemble(8)
    e = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], _toy_evaluator, rank=rank, world=world, rng_mode="counter")
    e.initialise()
    e.advance(_proposer(), 8, 4, prefetch=True)
    out = None
    if rank == 0:
        solo = pt.PTEnsemble(ens0["chain_states"], ens0["temperatures"], _toy_evaluator, rank=0, world=1, rng_mode="counter",
                             group=pt.LOCAL_ONLY)
        solo.initialise()
        solo.advance(_proposer(), 8, 4, prefetch=True)
        out = (solo.states.copy(), solo.logll.copy(), solo.n_collectives, e.states.copy(), e.n_collectives)
    dist.barrier()
    q.put((rank, out))
    dist.destroy_process_group()


def test_a_lone_ensemble_inside_a_two_rank_job_issues_no_collective():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_local_only_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    solo_states, solo_logll, solo_coll, shard_states, shard_coll = res[0]
    assert solo_coll == 0 and shard_coll == 1 + 2  # initialise + two replica exchanges through gloo
    ref, _, _ = _run_counter(8, 8, True)
    np.testing.assert_array_equal(solo_states, ref.states)
    np.testing.assert_array_equal(solo_logll, ref.logll)
    np.testing.assert_array_equal(shard_states, ref.states)  # and the sharded walk is the same chain
