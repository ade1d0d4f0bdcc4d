"""Parallel-tempering ensemble sharded over the GPUs of one node.

What the reference does with one MPI rank per chain and point-to-point sends
(metropolis.py:140-267: bcast of the pair index, send/recv of -log_ri, of the
accept flag, of two log-likelihoods and two state vectors per accepted swap) is
done here with ONE collective per replica exchange (history-free "counter" draws, the mode the ensembles of
bench.py run in) or per iteration (draws in the reference's program order, "stream"):

  * chains are block-partitioned over the ranks (`partition`), all measurements
    of a chain stay on one GPU, so the per-chain log-likelihood is local;
  * every rank evaluates its block at ALL ladder temperatures -- ll(T) is not
    ll(1)/T when measurement uncertainties are non-zero
    (trial_move_evaluation.py:150-156), so a scalar per chain is not enough for a
    sequence of swaps -- and an all-gather (RCCL over xGMI; `gloo` in the CPU
    tests) assembles the [n_chains x n_T] table on every rank;
  * the `n_chains - 1` sequential random adjacent swap attempts
    (metropolis.py:66-90, :129-135) are REPLAYED identically on every rank from a
    shared-seed generator: no further traffic, swapped states move as index
    permutations of the gathered state array, and the new log-likelihood of a
    swapped chain is looked up in the table instead of being re-simulated as the
    reference does (metropolis.py:87-88);
  * BETWEEN two exchanges the chains do not interact.  With counter-mode draws chain c only ever
    sees row c of a draw, so a rank proposes and decides (metropolis.py:42-63) for its OWN block only:
    the log-likelihood rows of its current states stay on the device, one number per proposal (its
    log-likelihood at the chain's own temperature) comes back to the host, and nothing crosses the
    ranks until the next exchange gathers [own rows x (n_T + n_params)] once.  In "stream" mode every
    draw comes out of one generator in program order, so every rank has to replay every chain to stay
    in step with it, and the table is gathered at every iteration as before.

  * with evaluators that can launch from descriptors kept on the device, the decisions of a launch are taken there
    too (`_DecidePipe`): the host prepares the proposals of every outcome one launch ahead, and between two exchanges
    no launch waits for it.

Host logic only (numpy + torch tensors / torch.distributed); the evaluator -- and with it every kernel -- is injected.
"""
import gc
import math
import time

import numpy as np

MAX_PROPOSALS = 100  # sim_utils.py:20


def partition(n_chains, world):
    """Contiguous blocks, sizes differing by at most one -> [(start, stop)] per rank."""
    base, rem = divmod(n_chains, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


class Proposer:
    """make_trial_move (trial_move_generation.py:51-96) + approve_move (:4-49) for every chain at once,
    with the per-parameter tables laid out once: uniform box of half-width sqrt(T)*base_move
    (metropolis.py:43-44) in (log-)space, re-drawn up to MAX_PROPOSALS times for the chains whose proposal
    fails the checks.  Chain c only ever sees row c of each round's draw, so its proposal does not depend on
    the other chains."""

    def __init__(self, base_move, names, active, prior, do_log, ix, hard_bounds=True):
        self.do_log = np.asarray(do_log, dtype=bool)
        self.all_log = bool(self.do_log.all())
        self.base = np.asarray(base_move, dtype=np.float64)
        self.active = np.asarray(active, dtype=bool)
        self.lo = np.array([prior[p][0] if self.active[i] else -np.inf for i, p in enumerate(names)], dtype=np.float64)
        self.hi = np.array([prior[p][1] if self.active[i] else np.inf for i, p in enumerate(names)], dtype=np.float64)
        self.skip = ~self.active
        self.pn = (ix["p0"], ix["n0"]) if "p0" in ix and "n0" in ix else None
        self.tau = (ix["tauN"], ix["tauP"]) if "tauN" in ix and "tauP" in ix else None
        self.rounds = MAX_PROPOSALS if hard_bounds else 1

    def to_lin(self, x):
        return 10.0 ** x if self.all_log else np.where(self.do_log, 10.0 ** x, x)

    def approve(self, new_log):
        """True = all checks of approve_move pass; new_log: [C, n] in proposal space (log10 where do_log)."""
        return self._approve(new_log, self.to_lin(new_log))

    def _approve(self, new_log, lin):
        ok = (((self.lo < lin) & (lin < self.hi)) | self.skip).all(axis=1)
        if self.pn is not None:
            ok &= new_log[:, self.pn[0]] > new_log[:, self.pn[1]]
        if self.tau is not None:
            tn, tp = self.tau
            ltn = new_log[:, tn] if self.do_log[tn] else np.log10(new_log[:, tn])
            ltp = new_log[:, tp] if self.do_log[tp] else np.log10(new_log[:, tp])
            ok &= np.abs(ltn - ltp) <= 2
        return ok

    def __call__(self, states, temperatures, rng):
        cur = np.log10(states) if self.all_log else np.where(self.do_log, np.log10(np.where(self.do_log, states, 1.0)), states)
        width = np.sqrt(np.asarray(temperatures))[:, None] * self.base[None, :]
        out = np.empty_like(cur)
        todo = None  # rows still without an approved proposal (None: all of them)
        for _ in range(self.rounds):
            u = rng.random(states.shape)  # every round draws for the whole ensemble; a row uses its own numbers
            if todo is None:
                cand = cur + width * (2 * u - 1)
            else:
                cand = cur[todo] + width[todo] * (2 * u[todo] - 1)
            lin = self.to_lin(cand)
            ok = self._approve(cand, lin)
            if todo is None:
                out[:] = lin
                if ok.all():
                    break
                todo = np.flatnonzero(~ok)
            else:
                out[todo] = lin  # (a row that fails every round keeps its last draw, trial_move_generation.py:88-96)
                todo = todo[~ok]
                if todo.size == 0:
                    break
        return out


def approve(new_state_log, names, active, prior, do_log, ix):
    """trial_move_generation.py:4-49, vectorised over chains; True = all checks pass."""
    return Proposer(np.zeros(len(names)), names, active, prior, do_log, ix).approve(np.asarray(new_state_log))


def propose(states, temperatures, base_move, names, active, prior, do_log, ix, rng, hard_bounds=True):
    """One-off form of `Proposer` (builds the tables on every call)."""
    return Proposer(base_move, names, active, prior, do_log, ix, hard_bounds)(states, temperatures, rng)


def metropolis_accept(logll_cur, logll_new, rng, rows=None):
    """metropolis.py:48-53: accept if u < exp(logll_new - logll_cur); NaN -> reject.
    rows = (n, sel): the chains are rows `sel` of an n-chain ensemble whose accept roll is drawn whole."""
    lr = logll_new - logll_cur
    lr = np.where(np.isnan(lr), -np.inf, lr)
    u = rng.random(lr.shape[0]) if rows is None else rng.random(rows[0])[rows[1]]
    with np.errstate(over="ignore"):
        return u < np.exp(lr)


def replica_exchange(table, rng, batched_draws=False):
    """metropolis.py:129-135 + :66-90 on the gathered table.

    table[c, k] = log-likelihood of the state currently held by chain c at ladder
    temperature k.  Returns (owner, attempts, accepts): owner[k] = index of the
    state that sits at temperature slot k afterwards.
    batched_draws: the n - 1 pair indices and the n - 1 accept rolls as two draws instead of alternating one by one
    as the reference's loop does -- another, equally random sequence of attempts, three times cheaper to produce (the
    generator calls are most of this loop); for the history-free "counter" draws, which are nobody's stream to follow."""
    n = table.shape[0]
    owner = list(range(n))
    attempts, accepts = [0] * n, [0] * n
    item, integers, random, exp = table.item, rng.integers, rng.random, np.exp
    if batched_draws and n > 1:
        pairs = iter(rng.integers(0, n - 1, size=n - 1).tolist())
        rolls = iter(rng.random(n - 1).tolist())
        integers, random = (lambda lo, hi: next(pairs)), (lambda: next(rolls))
        exp = lambda x: math.exp(x) if x < 700.0 else (math.inf if x >= 700.0 else x)  # noqa: E731 (a NaN ratio stays NaN: rejected)
    with np.errstate(over="ignore", invalid="ignore"):
        for _ in range(n - 1):
            i = int(integers(0, n - 1))
            attempts[i] += 1
            a, b = owner[i], owner[i + 1]
            logratio = item(a, i) + item(b, i + 1) - item(b, i) - item(a, i + 1)  # bi_ui + bj_uj - bi_uj - bj_ui
            if random() < exp(-logratio):
                owner[i], owner[i + 1] = b, a
                accepts[i] += 1
    return np.array(owner), np.array(attempts, dtype=np.int64), np.array(accepts, dtype=np.int64)


LOCAL_ONLY = object()  # `group` of a ONE-rank ensemble that must not touch the process group it happens to live in (bench.py's
#                        one-GPU legs inside an N-rank job): no collective is issued, whatever torch.distributed has set up


def all_gather_rows(recv, send, group=None):
    """dist.all_gather_into_tensor -- through RCCL on device tensors; staged through the host where the process group is
    `gloo` and the tensors live on a GPU (the two-rank rehearsal of bench.py on a one-GPU box: gloo gathers host memory)."""
    import torch.distributed as dist
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = recv.cpu()
        dist.all_gather_into_tensor(r, send.cpu(), group=group)
        recv.copy_(r)
        return
    dist.all_gather_into_tensor(recv, send, group=group)


def coll_device(dev, group=None):
    """Where the small tensors of a barrier-side collective (a maximum over the ranks, a failure flag) should live: on the GPU
    for RCCL, on the host for `gloo` (bench.py's one-GPU rehearsal of the N > 1 path)."""
    import torch.distributed as dist
    return "cpu" if (dist.is_initialized() and dist.get_backend(group) == "gloo") else dev


def _single(world, group):
    """One rank and nothing to gather through: no process group initialised (or the caller said LOCAL_ONLY).  A one-rank
    group that WAS set up (MTRPL_FORCE_DIST=1, tests/gpu_ensemble_child.py --dist 1) does go through the collective."""
    import torch.distributed as dist
    return world == 1 and (group is LOCAL_ONLY or not (dist.is_available() and dist.is_initialized()))


class TableGather:
    """One evaluator call on this rank's block of every state array in `states_list` (each [n_chains,
    n_params]) at all ladder temperatures, and ONE all-gather of the results (RCCL on GPUs, gloo in the CPU
    tests).  Construction only enqueues (on `stream` when given): the evaluator's kernels, the collective and
    the copy to pinned host memory; `tables()` waits and returns the list of [n_chains, nT] numpy arrays."""

    def __init__(self, evaluator, states_list, temperatures, blocks, rank=0, world=1, group=None, stream=None):
        import torch
        import torch.distributed as dist
        lo, hi = blocks[rank]
        self.c, self.G, self.world = hi - lo, len(states_list), world
        self.sizes = [b[1] - b[0] for b in blocks]
        mx = max(self.sizes)
        block = states_list[0][lo:hi] if self.G == 1 else np.concatenate([s[lo:hi] for s in states_list], axis=0)
        self.event = None
        with (torch.cuda.stream(stream) if stream is not None else _Nothing()):
            local = evaluator(block, temperatures)
            self.nT = local.shape[1]
            self.single = _single(world, group)
            if self.single:
                dev_result = local
            else:
                if len(set(self.sizes)) == 1:
                    send = local.contiguous()
                else:
                    # uneven blocks: collectives want equal shapes, so every group is padded to the largest block
                    send = torch.zeros((self.G, mx, self.nT), dtype=local.dtype, device=local.device)
                    send[:, :self.c] = local.view(self.G, self.c, self.nT)
                    send = send.view(self.G * mx, self.nT)
                dev_result = torch.empty((world * self.G * mx, self.nT), dtype=local.dtype, device=local.device)
                all_gather_rows(dev_result, send, group)
            if dev_result.is_cuda and stream is not None:
                self.pool_key = (tuple(dev_result.shape), dev_result.dtype)
                free = _PINNED.setdefault(self.pool_key, [])
                self.host = free.pop() if free else torch.empty(dev_result.shape, dtype=dev_result.dtype, pin_memory=True)
                self.host.copy_(dev_result, non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record(stream)
                self.keep = dev_result  # alive until the copy has run
            else:
                self.host = dev_result.detach().cpu()

    def tables(self):
        if self.event is not None:
            self.event.synchronize()
            self.keep = None
            arr = self.host.numpy().copy()
            _PINNED[self.pool_key].append(self.host)  # the pinned buffer goes back for the next launch
        else:
            arr = self.host.numpy()
        if self.single:
            return [arr[g * self.c:(g + 1) * self.c] for g in range(self.G)]
        mx = max(self.sizes)
        arr = arr.reshape(self.world, self.G, mx, self.nT)
        return [np.concatenate([arr[r, g, :self.sizes[r]] for r in range(self.world)], axis=0) for g in range(self.G)]


_PINNED = {}  # (shape, dtype) -> idle pinned host buffers of TableGather


class _Nothing:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def gather_ll_tables(evaluator, states_list, temperatures, blocks, rank=0, world=1, group=None):
    """Several state arrays through one evaluator call and one collective -> list of [n_chains, nT] tables."""
    return TableGather(evaluator, states_list, temperatures, blocks, rank, world, group).tables()


def gather_ll_table(evaluator, states, temperatures, blocks, rank=0, world=1, group=None):
    """One state array -> its [n_chains, nT] table (see TableGather)."""
    return gather_ll_tables(evaluator, [states], temperatures, blocks, rank, world, group)[0]


PROPOSE, ACCEPT, SWAP = 0, 1, 2  # what a counter-mode draw is for


class _Rows:
    """Rows `sel` of the [n, ...] uniforms a whole-ensemble draw would produce (proposals of a chain group)."""

    def __init__(self, rng, n, sel):
        self.rng, self.n, self.sel = rng, n, sel

    def random(self, shape):
        return self.rng.random((self.n,) + tuple(shape[1:]))[self.sel]


class _TiledRows(_Rows):
    """... the same rows once per candidate state of the group, stacked (`_DecidePipe`)."""

    def __init__(self, rng, n, sel, reps):
        super().__init__(rng, n, sel)
        self.reps = reps

    def random(self, shape):
        u = super().random(shape)
        return np.tile(u, (self.reps,) + (1,) * (u.ndim - 1))


class PTEnsemble:
    """One rank's view of the sharded ensemble.

    evaluator(states_block [c, n_params], temperatures [nT]) -> torch tensor [c, nT] (sum over
    measurements) on the device the process group works on; or a LIST of such evaluators, one per chain
    group (below).

    rng_mode "stream": every draw comes from one shared-seed generator in program order, like the
    reference's walk (metropolis.py:296); every rank replays every chain and the table is all-gathered at
    every iteration.  rng_mode "counter": the draws of iteration k for purpose w (proposal / accept roll /
    swaps) come from a Philox generator keyed by (seed, k, w) -- they do not depend on how many numbers
    earlier iterations consumed, and chain c only ever uses row c of a draw.  That buys three things, all
    walking exactly the chain the one-iteration-at-a-time, one-rank run walks:
      * local decisions: a rank proposes and accepts / rejects for its own chain block only; the ranks
        exchange [own rows x (n_T + n_params)] ONCE per replica exchange (`sync`), not per iteration, and
        between exchanges only one number per proposal leaves the device;
      * prefetching: the proposals of iteration k+1 for BOTH outcomes of iteration k are evaluated in the
        same launch as iteration k's (a GPU with idle SIMDs does two iterations in one solve latency);
      * chain groups: between two replica exchanges the chains do not interact, so the ensemble is cut into
        `groups` interleaved sets (chain c in group c % groups), each with its own evaluator and HIP stream;
        while the host decides and re-proposes for one group, the other group's launch is on the GPU.

      * device-side decisions (`device_decisions`, evaluators that offer `pack_rows` / `launch_packed`): the accept /
        reject of a launch, the choice of the next launch's proposals among the sets the host has prepared while
        the launch ran (one per outcome: two, or four when prefetching) and the update of the table rows all happen
        on the device, behind the launch on its stream; the host runs one launch ahead and reads the decisions back
        when it needs the states.  Between two replica exchanges no launch waits for the host (`_DecidePipe`).

    In counter mode `states`, `logll` and `table` are complete on every rank after `initialise`, after a
    replica exchange and after `advance` / `sync`; in between a rank keeps only its own rows current."""

    _frozen = False  # gc.freeze() has been called by an ensemble of this process (advance)

    def __init__(self, states, temperatures, evaluator, rank=0, world=1, seed=235817049752375780, group=None,
                 rng_mode="stream", streams=None, device_decisions=False, gc_freeze=False, regroup_by_cost=False):
        if rng_mode not in ("stream", "counter"):
            raise ValueError("rng_mode must be 'stream' or 'counter'")
        self.states = np.array(states, dtype=np.float64)
        self.T = np.asarray(temperatures, dtype=np.float64)
        self.n = self.states.shape[0]
        self.rank, self.world, self.group = rank, world, group
        self.blocks = partition(self.n, world)
        self.lo, self.hi = self.blocks[rank]
        self.evaluators = list(evaluator) if isinstance(evaluator, (list, tuple)) else [evaluator]
        self.groups = len(self.evaluators)
        if self.groups > 1 and rng_mode != "counter":
            raise ValueError("chain groups need rng_mode='counter'")
        self.streams = list(streams) if streams is not None else [None] * self.groups
        # group g = chains g, g + groups, ...; gblocks[g][r] = rank r's rows of the group's state array
        self.sel = [np.arange(g, self.n, self.groups) for g in range(self.groups)]
        self.gblocks = [[(int(np.searchsorted(sel, lo)), int(np.searchsorted(sel, hi))) for lo, hi in self.blocks]
                        for sel in self.sel]
        # ... and this rank's own chains of group g (global indices)
        self.own = [sel[a:b] for sel, (a, b) in zip(self.sel, (gb[rank] for gb in self.gblocks))]
        self.rng_mode = rng_mode
        self.local = rng_mode == "counter"  # propose / decide for the own block only, gather at exchanges
        self.seed = int(seed)
        self.rng = np.random.default_rng(seed)  # metropolis.py:296 (same seed on every rank, :410-411)
        self._philox = None
        self.it = 0  # iterations done (by every group)
        self.git = [0] * self.groups
        self.pending = {}
        self.table = None
        self.logll = None
        self.cur_tab = None                       # device: table rows of this rank's current states [hi - lo, nT]
        self.acc_count = np.zeros(self.n, dtype=np.int64)  # accepted moves per chain (own rows current)
        self.n_collectives = 0                    # all-gathers issued so far
        self.swap_attempts = np.zeros(self.n, dtype=np.int64)
        self.swap_accepts = np.zeros(self.n, dtype=np.int64)
        self.prof = None  # a dict: host seconds per phase of the loop (tools/pt_host_phases.py)
        self.chain_cost = np.zeros(self.n)  # what the evaluator said a chain's last evaluation cost (own rows; 0 = unknown)
        self.device_decisions = bool(device_decisions)
        self.gc_freeze = bool(gc_freeze)
        # Latency-bound ensembles (a launch of every group fits the device at once): at every replica exchange the own
        # chains are dealt to the groups again BY COST -- the most expensive chains to the group that starts first -- so
        # that one group carries the slow chains and the others finish early, instead of every group waiting for a slow
        # chain of its own in every launch (`_pipelined`; needs evaluators with `row_cost`).  A chain's walk does not
        # depend on the group it is in (counter mode).  Throughput-bound ensembles want groups of EQUAL cost: off.
        self.regroup_by_cost = bool(regroup_by_cost)
        self.pipe_own = {}  # group -> the own chains its pipe walks in the current stretch (default: own[g])
        self.n_pipelined = 0  # launches decided on the device so far
        # host seconds of the replica exchanges so far: waiting for the groups' last launches | the all-gather and its
        # copy to the host | swap replay, permutation and the upload of the own rows; and how many there were
        self.exchange_s = {"wait": 0.0, "gather": 0.0, "replay": 0.0, "n": 0}

    def _tick(self, key, t0):
        if self.prof is not None:
            dt = time.perf_counter() - t0
            self.prof[key] = self.prof.get(key, 0.0) + dt
            self.prof[key + ".max"] = max(self.prof.get(key + ".max", 0.0), dt)

    @property
    def evaluator(self):
        return self.evaluators[0]

    @evaluator.setter
    def evaluator(self, ev):
        self.evaluators[0] = ev

    def draws(self, it, what):
        """The generator the draws of iteration `it` for purpose `what` come from."""
        if self.rng_mode == "stream":
            return self.rng
        # ONE generator object, re-keyed in place (an order of magnitude cheaper than building a Philox per
        # draw): whoever asks uses it up before the next call of draws()
        if self._philox is None:
            key = [self.seed & (2 ** 64 - 1), (self.seed >> 64) & (2 ** 64 - 1)]
            bg = np.random.Philox(key=key, counter=[0, 0, 0, 0])
            self._philox = (bg, np.random.Generator(bg), bg.state)
        bg, gen, st = self._philox
        st["state"]["counter"][:] = [0, 0, what, it]
        st["buffer"][:] = 0
        st["buffer_pos"], st["has_uint32"], st["uinteger"] = 4, 0, 0
        bg.state = st
        return gen

    def _collective(self):
        """Does a gather go through torch.distributed (more than one rank, or a one-rank group that was set up)?"""
        return not _single(self.world, self.group)

    def gather_table(self, states):
        """Evaluate the local block at all ladder temperatures, all-gather -> [n, nT] numpy."""
        self.n_collectives += int(self._collective())
        return gather_ll_table(self.evaluators[0], states, self.T, self.blocks, self.rank, self.world, self.group)

    def initialise(self):
        if not self.local:
            self.table = self.gather_table(self.states)
            self.logll = self.table[np.arange(self.n), np.arange(self.n)].copy()
            return
        self.cur_tab = self.evaluators[0](self.states[self.lo:self.hi], self.T).clone()
        self.logll = np.zeros(self.n)
        self.sync()

    # ---- counter mode: what crosses the ranks, once per replica exchange ----------------------------
    def sync(self):
        """All-gather [own rows x (n_T + n_params + 1)] = the log-likelihood rows of the current states (from the
        device), the states and the per-chain accept counts: afterwards `table`, `states`, `logll` and
        `acc_count` are complete and identical on every rank.  One collective."""
        import torch
        import torch.distributed as dist
        if not self.local:
            return
        t_wait = time.perf_counter()
        for st in self.streams:  # the decisions of every group have been applied to cur_tab on its side stream
            if st is not None:
                st.synchronize()
        for c_ in getattr(self, "_devc", {}).values():
            if c_.get("side") is not None:
                c_["side"].synchronize()
        t_gather = time.perf_counter()
        self.exchange_s["wait"] += t_gather - t_wait
        try:
            self._sync_gather()
        finally:
            self.exchange_s["gather"] += time.perf_counter() - t_gather

    def _sync_gather(self):
        import torch
        import torch.distributed as dist
        c, nT, npar = self.hi - self.lo, self.T.shape[0], self.states.shape[1]
        dev = self.cur_tab.device
        host = np.concatenate([self.states[self.lo:self.hi], self.acc_count[self.lo:self.hi, None].astype(np.float64)], axis=1)
        single = _single(self.world, self.group)
        if single:
            self.table = self.cur_tab.detach().cpu().numpy().copy()
            self._sync_logll()
            return
        mx = max(b[1] - b[0] for b in self.blocks)
        send = torch.zeros((mx, nT + npar + 1), dtype=torch.float64, device=dev)
        send[:c, :nT] = self.cur_tab
        send[:c, nT:] = torch.from_numpy(host).to(dev)
        recv = torch.empty((self.world * mx, nT + npar + 1), dtype=torch.float64, device=dev)
        all_gather_rows(recv, send, self.group)
        self.n_collectives += 1
        arr = recv.cpu().numpy().reshape(self.world, mx, nT + npar + 1)
        full = np.concatenate([arr[r, :b[1] - b[0]] for r, b in enumerate(self.blocks)], axis=0)
        self.table = np.ascontiguousarray(full[:, :nT])
        self.states = np.ascontiguousarray(full[:, nT:nT + npar])
        self.acc_count = np.rint(full[:, -1]).astype(np.int64)
        self.logll = self.table[np.arange(self.n), np.arange(self.n)].copy()

    def _dev_cache(self, g, G):
        """Index tensors and mask staging of group g's launches, built ONCE on the device: nothing in the hot
        path copies from pageable host memory (such a copy makes the host wait for the stream's queued kernels,
        i.e. for the whole solve, and the launches of the chain groups no longer overlap)."""
        import torch
        key = (g, G)
        c = getattr(self, "_devc", None)
        if c is None:
            c = self._devc = {}
        if key not in c:
            dev = self.cur_tab.device
            sel = self.own[g]
            pin = dev.type == "cuda"
            side = c.get(("side", g))
            if side is None and pin and self.streams[g] is not None:
                # the accepted rows move into cur_tab on a stream of their own: nothing reads cur_tab before the next
                # exchange, so that work has no place in front of the group's next launch
                side = c[("side", g)] = {"side": torch.cuda.Stream(device=dev)}
            # (made on the group's stream: `zeros` is a fill kernel, and everything that touches these tensors later runs on
            # that stream or behind an event of it -- torch's side streams do not wait for the default stream)
            with (torch.cuda.stream(self.streams[g]) if (pin and self.streams[g] is not None) else _Nothing()):
                c[key] = {"col": torch.from_numpy(np.tile(sel, G)).to(dev),
                          "rows": torch.from_numpy(np.ascontiguousarray(sel - self.lo)).to(dev),
                          "mask_host": torch.zeros((2, max(1, sel.shape[0])), dtype=torch.bool, pin_memory=pin),
                          "mask_dev": torch.zeros((2, max(1, sel.shape[0])), dtype=torch.bool, device=dev),
                          "side": side["side"] if side else None, "side_done": None}
                if pin and self.streams[g] is not None:
                    self.streams[g].synchronize()  # (built once per ensemble: the side stream may use them at once)
        return c[key]

    def _sync_logll(self):
        self.logll = self.table[np.arange(self.n), np.arange(self.n)].copy()

    # ---- one group, one launch: propose + enqueue, then decide --------------------------------------
    def _begin(self, proposer, g, depth):
        """Propose the next `depth` (1 or 2) iterations of group g and enqueue their evaluation: per chain
        the proposal of iteration it+1 and, for depth 2, the two proposals it+2 can make -- from the current
        state if it+1 rejects, from the it+1 proposal if it accepts."""
        t0 = time.perf_counter()
        it = self.git[g]
        sel = self.own[g] if self.local else self.sel[g]
        xs, Ts = self.states[sel], self.T[sel]
        whole = self.groups == 1 and not self.local

        def rows(i):
            rng = self.draws(i, PROPOSE)
            return rng if whole else _Rows(rng, self.n, sel)
        y1 = proposer(xs, Ts, rows(it + 1)) if sel.shape[0] else xs.copy()
        props = [y1]
        if depth == 2:
            if self.rng_mode != "counter":
                raise RuntimeError("prefetching needs rng_mode='counter'")
            props += [proposer(xs, Ts, rows(it + 2)), proposer(y1, Ts, rows(it + 2))] if sel.shape[0] else [xs.copy(), xs.copy()]
        self._tick("propose", t0)
        if self.local:
            t0 = time.perf_counter()
            self.pending[g] = (depth, props, _LocalEval(self.evaluators[g], props, self.T, sel, self.streams[g],
                                                        self._dev_cache(g, len(props))))
            self._tick("enqueue", t0)
            return
        self.n_collectives += int(self._collective())
        gather = TableGather(self.evaluators[g], props, self.T, self.gblocks[g], self.rank, self.world, self.group,
                             self.streams[g])
        self.pending[g] = (depth, props, gather)

    def _displace(self, g, it, new_states, new_table):
        """Accept/reject of iteration `it` for group g (metropolis.py:48-63) given the evaluated proposals."""
        sel = self.sel[g]
        new_ll = new_table[np.arange(sel.shape[0]), sel]  # chain c sits at ladder slot c
        rows = None if self.groups == 1 else (self.n, sel)
        acc = metropolis_accept(self.logll[sel], new_ll, self.draws(it, ACCEPT), rows)
        idx = sel[acc]
        self.states[idx] = new_states[acc]
        self.table[idx] = new_table[acc]
        self.logll[idx] = new_ll[acc]
        self.acc_count[idx] += 1
        return acc

    def _end(self, g):
        depth, props, ev = self.pending.pop(g)
        it = self.git[g]
        if not self.local:
            tabs = ev.tables()
            accs = [self._displace(g, it + 1, props[0], tabs[0])]
            if depth == 2:
                a = accs[0][:, None]
                accs.append(self._displace(g, it + 2, np.where(a, props[2], props[1]), np.where(a, tabs[2], tabs[1])))
            self.git[g] = it + depth
            return accs
        # own block only: the proposals' log-likelihoods at the chains' own temperatures decide; the table rows
        # of the accepted ones replace the current rows ON THE DEVICE
        sel = self.own[g]
        t0 = time.perf_counter()
        lls = ev.own_ll()  # [depth == 2 ? 3 : 1][c]
        self._tick("wait", t0)
        t0 = time.perf_counter()
        a1 = metropolis_accept(self.logll[sel], lls[0], self.draws(it + 1, ACCEPT), (self.n, sel))
        self.states[sel[a1]] = props[0][a1]
        self.logll[sel[a1]] = lls[0][a1]
        self.acc_count[sel[a1]] += 1
        accs, masks = [a1], [a1]
        if depth == 2:
            ll2 = np.where(a1, lls[2], lls[1])
            p2 = np.where(a1[:, None], props[2], props[1])
            a2 = metropolis_accept(self.logll[sel], ll2, self.draws(it + 2, ACCEPT), (self.n, sel))
            self.states[sel[a2]] = p2[a2]
            self.logll[sel[a2]] = ll2[a2]
            self.acc_count[sel[a2]] += 1
            accs.append(a2)
            masks.append(a2)
        self._tick("decide", t0)
        t0 = time.perf_counter()
        ev.apply(self.cur_tab, sel - self.lo, masks)
        self._tick("apply", t0)
        self.git[g] = it + depth
        full = []
        for a in accs:  # in the shape the replicated mode returns: one flag per chain of the group (own rows set)
            f = np.zeros(self.sel[g].shape[0], dtype=bool)
            f[self.gblocks[g][self.rank][0]:self.gblocks[g][self.rank][1]] = a
            full.append(f)
        return full

    def _exchange(self, it):
        t0 = time.perf_counter()
        before = self.exchange_s["wait"] + self.exchange_s["gather"]
        self._exchange_(it)
        dt = time.perf_counter() - t0
        self.exchange_s["replay"] += dt - (self.exchange_s["wait"] + self.exchange_s["gather"] - before)
        self.exchange_s["n"] += 1
        self._tick("exchange", t0)

    def _exchange_(self, it):
        self.sync()
        owner, att, accs = replica_exchange(self.table, self.draws(it, SWAP), batched_draws=self.rng_mode == "counter")
        self.states = self.states[owner]
        self.table = self.table[owner]
        self.chain_cost = self.chain_cost[owner]
        self._sync_logll()
        self.swap_attempts += att
        self.swap_accepts += accs
        if self.local:
            import torch
            self.cur_tab.copy_(torch.from_numpy(np.ascontiguousarray(self.table[self.lo:self.hi])))
            if self.cur_tab.is_cuda:
                torch.cuda.current_stream().synchronize()  # the groups' streams read and write cur_tab from here on

    def _all_groups(self, proposer, depth, do_swap):
        for g in range(self.groups):
            self._begin(proposer, g, depth)
        accs = [self._end(g) for g in range(self.groups)]
        self.it += depth
        if do_swap:
            self._exchange(self.it)
        out = []
        for d in range(depth):  # back to chain order
            acc = np.zeros(self.n, dtype=bool)
            for g in range(self.groups):
                acc[self.sel[g]] = accs[g][d]
            out.append(acc)
        return out

    def iterate(self, proposer, do_swap):
        """One Metropolis iteration for every chain (+ replica exchange if do_swap).  -> accept flags per chain
        (counter mode: of this rank's own chains, False elsewhere)."""
        return self._all_groups(proposer, 1, do_swap)[0]

    def iterate2(self, proposer, do_swap):
        """Two iterations from ONE evaluator launch per group (no replica exchange between them; `do_swap`:
        exchange after the second); one of the two speculative evaluations per chain is discarded.  Same chain
        as two `iterate` calls (counter mode only)."""
        if self.rng_mode != "counter":
            raise RuntimeError("iterate2 needs rng_mode='counter'")
        return tuple(self._all_groups(proposer, 2, do_swap))

    def _pipelined(self, proposer, seg_end, depth):
        """Every whole launch of `depth` iterations up to `seg_end`, for the groups whose evaluator can launch from
        descriptors kept on the device, with the decisions taken there (`_DecidePipe`); the groups' host work is
        interleaved."""
        todo = []
        for g in range(self.groups):
            ev = self.evaluators[g]
            n_launches = (seg_end - self.git[g]) // depth
            if n_launches > 0 and hasattr(ev, "pack_rows") and hasattr(ev, "launch_packed") and getattr(ev, "packable", True):
                todo.append((g, n_launches))
        # The first launch of a stretch is what the device waits for (it idles from the last decision before the exchange
        # until then): its proposals are made for all the groups at once, and a group's pipe is built right before its
        # first launch goes out, not before everybody's.
        first = None
        self.pipe_own = {}
        if len(todo) > 1 and len({self.git[g] for g, _ in todo}) == 1:
            allsel = np.concatenate([self.own[g] for g, _ in todo])
            if (self.regroup_by_cost and len(todo) == self.groups and len({n for _, n in todo}) == 1
                    and np.any(self.chain_cost[allsel] > 0)):
                by_cost = allsel[np.argsort(-self.chain_cost[allsel], kind="stable")]
                cuts = np.cumsum([0] + [self.own[g].shape[0] for g, _ in todo])  # (the groups keep their sizes)
                self.pipe_own = {g: by_cost[a:b] for (g, _), a, b in zip(todo, cuts[:-1], cuts[1:])}
            if allsel.shape[0]:
                it = self.git[todo[0][0]]
                xs, Ts = self.states[allsel], self.T[allsel]
                rows = lambda i: _Rows(self.draws(i, PROPOSE), self.n, allsel)  # noqa: E731
                ys = [proposer(xs, Ts, rows(it + 1))]
                if depth == 2:
                    ys += [proposer(xs, Ts, rows(it + 2)), proposer(ys[0], Ts, rows(it + 2))]
                first = (dict(zip(allsel.tolist(), range(allsel.shape[0]))), ys)
        pipes, plans = [], []
        for step in range(2 * max((n for _, n in todo), default=0) + 2):
            for k, (g, n_launches) in enumerate(todo):
                if step == 0:
                    pipes.append(_DecidePipe(self, g, proposer, n_launches, depth, first))
                    plans.append(pipes[-1].schedule())
                if step < len(plans[k]):
                    getattr(pipes[k], plans[k][step][0])(plans[k][step][1])

    def advance(self, proposer, n_iters, temper_freq, prefetch=False):
        """`n_iters` iterations with a replica exchange after every `temper_freq`-th (metropolis.py:129).
        Between two exchanges every chain group runs at its own pace: with `prefetch` two iterations per
        launch, and with several groups the launch of one overlaps the host work of the others.
        -> accepted moves of the whole ensemble (counter mode: `states`, `logll`, `table` are complete again on
        return)."""
        # (no cyclic garbage collection between two launches: a full collection walks every object of the process --
        # tens of milliseconds with torch loaded, a dozen launches' worth -- and this loop creates no cycles.  Switching
        # it off only postpones that collection to the moment it is switched on again, so before the first stretch
        # everything that exists -- the imported modules, the engines: nothing of it is garbage -- is moved out of the
        # collector's sight once, gc.freeze(); later full collections look at what was created since and take a
        # millisecond.  That is a process-wide, lasting side effect of a library call -- every object the host application
        # holds at that moment moves to the permanent generation -- so it is OPT-IN (`gc_freeze=True`); the
        # default only switches the collector off for the duration of this call.)
        if self.gc_freeze and not PTEnsemble._frozen:
            gc.collect()
            gc.freeze()
            PTEnsemble._frozen = True
        gc_was = gc.isenabled()
        gc.disable()
        try:
            return self._advance(proposer, n_iters, temper_freq, prefetch)
        finally:
            if gc_was:
                gc.enable()

    def _advance(self, proposer, n_iters, temper_freq, prefetch):
        n_acc, stop = 0, self.it + n_iters
        acc0 = int(self.acc_count.sum())
        while self.it < stop:
            seg_end = min(stop, (self.it // temper_freq + 1) * temper_freq)  # next exchange or the end
            depth = lambda g: 2 if prefetch and seg_end - self.git[g] >= 2 else 1  # noqa: E731
            if self.local and self.device_decisions:
                self._pipelined(proposer, seg_end, 2 if prefetch else 1)  # every whole launch up to seg_end
            queue = [g for g in range(self.groups) if self.git[g] < seg_end]
            for g in queue:
                self._begin(proposer, g, depth(g))
            while queue:
                g = queue.pop(0)
                n_acc += sum(int(a.sum()) for a in self._end(g))
                if self.git[g] < seg_end:
                    self._begin(proposer, g, depth(g))
                    queue.append(g)
            self.it = seg_end
            if seg_end % temper_freq == 0:
                self._exchange(seg_end)
        if self.local:
            if stop % temper_freq != 0:
                self.sync()
            return int(self.acc_count.sum()) - acc0
        return n_acc


class _DecidePipe:
    """Counter mode, one chain group on this rank, from its current iteration through `n_launches` launches of `depth`
    iterations each (1, or 2: prefetching) with NO host decision between the launches.

    depth 2: launch j evaluates, per chain, the proposal of iteration 2j+1 and the two proposals 2j+2 can make.  Which
    state it starts from depends on the two decisions of launch j-1 -- four outcomes: (reject, reject) keeps s,
    (accept, reject) y1, (reject, accept) y2r, (accept, accept) y2a -- and those four states are known on the host as
    soon as launch j-1 has been PROPOSED.  So while launch j-1 runs the host proposes from all four (stacked: one
    `proposer` call per draw), packs the four descriptor sets and sends them up; behind launch j-1 on the group's
    stream sit its decisions (`u < exp(ll_new - ll_cur)`, the uniforms went up with the package), the per-chain pick of
    the next descriptors, the move of the accepted table rows into `cur_tab`, launch j and the copy of the outcome and
    the proposals' log-likelihoods per chain to pinned memory.  The host reads the outcome of launch j-2 before it
    prepares launch j (it needs the actual state s_{j-2}); the device never waits for it.  depth 1: one proposal per
    chain and launch, two outcomes -- the throughput-bound ensemble, whose launches otherwise wait for the host to
    decide, propose and pack between them.  Same chain as `_begin` / `_end`, bit for bit, as long as `exp` on the
    device and in numpy agree on which side of a uniform a ratio lies.

    What the group's evaluator has to offer: `pack_rows(states [R, n_params], sets)` -> descriptor bytes
    [R * M, B] (`sets` alternative sets of R / sets state rows, each laid out as if launched alone, in the order the
    evaluator issues them) and `launch_packed(descriptors on the device, n_solves, temperatures)` -> table [R / sets,
    nT]; optionally `plan_packed(states of one set)` -> the state row of every descriptor row (an evaluator that issues
    its solves in an order of its own; default: M consecutive rows per state row), `select` / `decide` (the two device
    steps as one kernel each instead of torch operations), `row_cost(R)` (cost per state row of the last launch) and
    `packable` (False: this evaluator's descriptors cannot be chosen on the device -- its group walks the host path)."""

    def __init__(self, ens, g, proposer, n_launches, depth, first=None):
        import torch
        self.torch, self.ens, self.g, self.proposer, self.L, self.depth = torch, ens, g, proposer, n_launches, depth
        self.first = first  # (chain -> row, [y1, y2r, y2a] of all the groups' chains): the first launch's proposals, made already
        self.G = 3 if depth == 2 else 1  # proposals per chain and launch
        self.ev = ens.evaluators[g]
        self.stream = ens.streams[g]
        self.sel = ens.pipe_own.get(g, ens.own[g])
        if hasattr(self.ev, "row_cost") and self.sel.shape[0]:
            # the chains that were expensive in the last launch go first: launches of two groups share the device, and
            # what has to wait for a SIMD should be the cheapest solves (the hardware starts workgroups in index order)
            self.sel = self.sel[np.argsort(-ens.chain_cost[self.sel], kind="stable")]
        self.c = self.sel.shape[0]
        self.it0 = ens.git[g]
        self.s = ens.states[self.sel].copy()  # the state after the last decision the host has read
        self.Ts = ens.T[self.sel]
        self.pk = []  # per launch: candidate proposals, pinned buffers, event
        self.done = 0  # launches whose decisions the host has read
        self.k_of = {}  # launch -> outcome index (a1 + 2 a2) per chain, once read
        dev = ens.cur_tab.device
        self.dev = dev
        if self.c:
            c, G = self.c, self.G
            # the state row (g * c + chain) of every descriptor row of a launch, in the evaluator's issue order
            if hasattr(self.ev, "plan_packed"):
                srow = np.asarray(self.ev.plan_packed(np.tile(self.s, (G, 1))), dtype=np.int64)
            else:
                srow = None
            self.fast = hasattr(self.ev, "decide") and hasattr(self.ev, "select")
            self.srow = srow
            with self._on():
                # (the index tensors are made ON THE GROUP'S STREAM: `.long()` is a kernel, and the gather below -- like every
                # later launch of this group -- runs on that stream, which does not wait for torch's default stream.  Made
                # on the default stream, as they were until round 5, the gather could read them before the conversion had
                # written them: an out-of-range index and a memory fault, seen once two ranks shared a device in the
                # rehearsal of bench.py's N > 1 path)
                idx = torch.from_numpy(np.stack([self.sel - ens.lo, self.sel]).astype(np.int32)).to(dev)
                self.row32, self.col32 = idx[0], idx[1]
                self.rows, self.col = idx[0].long(), idx[1].long()
                self.cur_ll = ens.cur_tab[self.rows, self.col].clone()  # the chains' log-likelihoods at their own temperatures
                self.outcome = None
                self.chain_of_row = None  # built with the first package (the evaluator's M is not known here)
                if self.fast:
                    self.outcome = torch.zeros(c, dtype=torch.int32, device=dev)
                    self.result = torch.zeros((4, c), dtype=torch.float64, device=dev)

    def _on(self):
        return self.torch.cuda.stream(self.stream) if self.stream is not None else _Nothing()

    def schedule(self):
        """The host's actions in order: ('enqueue', j) / ('read', j)."""
        out = [("enqueue", 0)]
        if self.L > 1:
            out.append(("enqueue", 1))
        for j in range(2, self.L):
            out += [("read", j - 2), ("enqueue", j)]
        if self.L > 1:
            out.append(("read", self.L - 2))
        out.append(("read", self.L - 1))
        return out

    def _actual(self, j):
        """The proposals launch j really evaluated [G, c, n_params] (known once launch j-1 has been read)."""
        pk = self.pk[j]
        if "actual" not in pk:
            p = pk["props"]  # [n_cand, G, c, n_params]
            pk["actual"] = p[0] if p.shape[0] == 1 else np.ascontiguousarray(
                p[self.k_of[j - 1], :, np.arange(self.c)].transpose(1, 0, 2))
        return pk["actual"]

    def enqueue(self, j):
        torch, ens, c, G = self.torch, self.ens, self.c, self.G
        if c == 0:
            self.pk.append({})
            return
        t0 = time.perf_counter()
        assert self.done >= j - 1 and len(self.pk) == j
        if j == 0:
            cands = self.s[None]
        else:  # the states launch j-1 can leave behind, from the state the host knows (after launch j-2)
            cands = np.stack([self.s] + list(self._actual(j - 1)))
        nc = cands.shape[0]
        it = self.it0 + self.depth * j
        S = cands.reshape(nc * c, -1)
        Tt = np.tile(self.Ts, nc)
        rows = lambda i: _TiledRows(ens.draws(i, PROPOSE), ens.n, self.sel, nc)  # noqa: E731
        u = np.zeros((2, c))
        u[0] = ens.draws(it + 1, ACCEPT).random(ens.n)[self.sel]
        if G == 3:
            u[1] = ens.draws(it + 2, ACCEPT).random(ens.n)[self.sel]
        if j == 0 and self.first is not None:
            pos = np.array([self.first[0][ch] for ch in self.sel.tolist()])
            ys = [y[pos] for y in self.first[1]]
            self.first = None
        else:
            ys = [self.proposer(S, Tt, rows(it + 1))]
            if G == 3:
                ys += [self.proposer(S, Tt, rows(it + 2)), self.proposer(ys[0], Tt, rows(it + 2))]
        props = np.stack([y.reshape(nc, c, -1) for y in ys], axis=1)  # [nc, G, c, n_params]
        ens._tick("propose", t0)
        t0 = time.perf_counter()
        raw = self.ev.pack_rows(props.reshape(nc * G * c, -1), nc)  # [nc * G c M, B] bytes, each set placed as if alone
        ens._tick("pack", t0)
        t0 = time.perf_counter()
        n_solves = raw.shape[0] // nc
        nbytes = raw.size
        pin = self.dev.type == "cuda"
        key = ("pipe", nbytes + 16 * c)
        free = _PINNED.setdefault(key, [])
        stage = free.pop() if free else torch.empty(nbytes + 16 * c, dtype=torch.uint8, pin_memory=pin)
        stage[:nbytes].numpy()[:] = raw.reshape(-1)
        stage[nbytes:].numpy()[:] = u.view(np.uint8).reshape(-1)
        with self._on():
            if self.chain_of_row is None:
                M = n_solves // (G * c)
                srow = self.srow if self.srow is not None else np.repeat(np.arange(G * c), M)
                self.chain_of_row = torch.from_numpy((srow % c).astype(np.int32)).to(self.dev)
                self.chain_of_row_l, self.ar = self.chain_of_row.long(), torch.arange(n_solves, device=self.dev)
            up = torch.empty(nbytes + 16 * c, dtype=torch.uint8, device=self.dev)
            up.copy_(stage, non_blocking=True)
            u_dev = up[nbytes:].view(torch.float64)  # [2, c] flat
            if self.fast:  # the evaluator brings kernels for both steps (pt_bench.setup: mtrpl_pt_select_dev / _decide_dev)
                d = torch.empty(nbytes // nc, dtype=torch.uint8, device=self.dev)
                self.ev.select(up[:nbytes], nc, n_solves, self.chain_of_row, self.outcome, d)
            else:
                cand_dev = up[:nbytes].view(nc, n_solves, -1)
                d = cand_dev[0] if nc == 1 else cand_dev[self.outcome[self.chain_of_row_l], self.ar]
                d = d.contiguous().view(-1)
            tab = self.ev.launch_packed(d, n_solves, ens.T)  # [G c, nT]
            if self.fast:
                self.ev.decide(tab, G, self.col32, self.row32, u_dev, self.cur_ll, ens.cur_tab, self.outcome, self.result)
                res = self.result
            else:
                t = tab.view(G, c, -1)
                diag = tab.gather(1, self.col.repeat(G)[:, None])[:, 0].view(G, c)
                a1 = u_dev[:c] < torch.exp(diag[0] - self.cur_ll)  # NaN compares False: rejected, like metropolis_accept
                cur1 = torch.where(a1, diag[0], self.cur_ll)
                m1 = a1[:, None]
                new_rows = torch.where(m1, t[0], ens.cur_tab[self.rows])
                if G == 3:
                    ll2 = torch.where(a1, diag[2], diag[1])
                    a2 = u_dev[c:] < torch.exp(ll2 - cur1)
                    self.cur_ll = torch.where(a2, ll2, cur1)
                    self.outcome = a1.long() + 2 * a2.long()
                    new_rows = torch.where(a2[:, None], torch.where(m1, t[2], t[1]), new_rows)
                    res = torch.cat([self.outcome.double()[None], diag], dim=0)  # [4, c]
                else:
                    self.cur_ll, self.outcome = cur1, a1.long()
                    res = torch.cat([self.outcome.double()[None], diag, torch.zeros_like(diag), torch.zeros_like(diag)], dim=0)
                ens.cur_tab[self.rows] = new_rows
            rkey = ("pipe_res", 4 * c)
            rfree = _PINNED.setdefault(rkey, [])
            host = rfree.pop() if rfree else torch.empty((4, c), dtype=torch.float64, pin_memory=pin)
            host.copy_(res, non_blocking=True)
            event = None
            if pin:
                event = torch.cuda.Event()
                event.record(self.stream if self.stream is not None else torch.cuda.current_stream())
        self.pk.append({"props": props, "host": host, "event": event, "stage": (key, stage), "rkey": rkey})
        ens.n_pipelined += 1
        ens._tick("enqueue", t0)

    def read(self, j):
        """The decisions of launch j -> the host's copy of the group's states, log-likelihoods and accept counts."""
        ens, c = self.ens, self.c
        assert self.done == j
        self.done = j + 1
        if c == 0:
            ens.git[self.g] += self.depth
            return
        pk = self.pk[j]
        t0 = time.perf_counter()
        if pk["event"] is not None:
            pk["event"].synchronize()
        ens._tick("wait", t0)
        t0 = time.perf_counter()
        res = pk["host"].numpy().copy()
        _PINNED[pk["rkey"]].append(pk["host"])
        _PINNED[pk["stage"][0]].append(pk["stage"][1])
        k = np.rint(res[0]).astype(np.int64)
        a = self._actual(j)  # needs k_of[j - 1]
        self.k_of[j] = k
        a1, a2 = (k & 1).astype(bool), (k >> 1).astype(bool)
        sel = self.sel
        s, ll = self.s, ens.logll[sel]
        s = np.where(a1[:, None], a[0], s)
        ll = np.where(a1, res[1], ll)
        if self.G == 3:
            s = np.where(a2[:, None], np.where(a1[:, None], a[2], a[1]), s)
            ll = np.where(a2, np.where(a1, res[3], res[2]), ll)
        self.s = s
        ens.states[sel] = s
        ens.logll[sel] = ll
        ens.acc_count[sel] += a1.astype(np.int64) + a2.astype(np.int64)
        ens.git[self.g] += self.depth
        if j == self.L - 1 and hasattr(self.ev, "row_cost"):
            ens.chain_cost[sel] = np.asarray(self.ev.row_cost(self.G * c), dtype=np.float64).reshape(self.G, c).max(axis=0)
        ens._tick("decide", t0)


class _LocalEval:
    """Counter mode, one launch of one chain group on this rank: the evaluator's kernels for the proposals of the
    rank's own chains (enqueued on `stream` when given) and the copy of ONE number per proposal -- its
    log-likelihood at the chain's own temperature -- to pinned host memory.  The [rows x nT] tables stay on the
    device; `apply` moves the accepted rows into the rank's current-table tensor there.  `cache`: the index
    tensors and the pinned mask staging of this group (PTEnsemble._dev_cache)."""

    def __init__(self, evaluator, props, temperatures, sel, stream, cache):
        import torch
        self.c, self.G, self.stream, self.cache = sel.shape[0], len(props), stream, cache
        self.event, self.dev_tab = None, None
        if self.c == 0:
            self.host = None
            return
        block = props[0] if self.G == 1 else np.concatenate(props, axis=0)
        with (torch.cuda.stream(stream) if stream is not None else _Nothing()):
            tab = evaluator(block, temperatures)  # [G * c, nT]
            self.dev_tab = tab
            diag = tab.gather(1, cache["col"][:, None])[:, 0]
            if diag.is_cuda:
                key = ((self.G * self.c,), diag.dtype)
                free = _PINNED.setdefault(key, [])
                self.pool_key = key
                self.host = free.pop() if free else torch.empty(key[0], dtype=diag.dtype, pin_memory=True)
                self.host.copy_(diag, non_blocking=True)
                self.event = torch.cuda.Event()
                self.event.record(stream if stream is not None else torch.cuda.current_stream())
            else:
                self.host = diag

    def own_ll(self):
        if self.c == 0:
            return [np.zeros(0)] * self.G
        if self.event is not None:
            self.event.synchronize()
            arr = self.host.numpy().copy()
            _PINNED[self.pool_key].append(self.host)
        else:
            arr = self.host.numpy()
        return [arr[g * self.c:(g + 1) * self.c] for g in range(self.G)]

    def apply(self, cur_tab, rows, masks):
        """cur_tab[rows] <- table row of the accepted proposal: iteration 1 takes props[0] where masks[0]; iteration
        2 (prefetched) takes props[2] where masks[0] else props[1], where masks[1].  On the group's side stream when
        it has one (the launch this decides has finished -- its event was waited for -- so the side stream has
        nothing to wait for; `PTEnsemble.sync` waits for the side streams before anyone reads cur_tab)."""
        import torch
        if self.c == 0:
            return
        side = self.cache.get("side")
        st = side if side is not None else self.stream
        with (torch.cuda.stream(st) if st is not None else _Nothing()):
            mh, md, idx = self.cache["mask_host"], self.cache["mask_dev"], self.cache["rows"]
            # (the previous launch of this group has been decided, i.e. its mask copy has been consumed: on the
            # group's own stream the event its decision waited for sits behind that copy; on the side stream the
            # event recorded behind the previous apply is waited for here -- it fired a whole launch ago)
            if self.cache.get("side_done") is not None:
                self.cache["side_done"].synchronize()
            mh[0, :self.c] = torch.from_numpy(np.ascontiguousarray(masks[0]))
            if len(masks) == 2:
                mh[1, :self.c] = torch.from_numpy(np.ascontiguousarray(masks[1]))
            md.copy_(mh, non_blocking=True)
            t = self.dev_tab.view(self.G, self.c, -1)
            m1 = md[0, :self.c, None]
            cur = torch.where(m1, t[0], cur_tab[idx])
            if len(masks) == 2:
                cur = torch.where(md[1, :self.c, None], torch.where(m1, t[2], t[1]), cur)
            cur_tab[idx] = cur
            if side is not None:
                self.dev_tab.record_stream(side)  # allocated on the group's stream, last read here
                if self.cache.get("side_done") is None:
                    self.cache["side_done"] = torch.cuda.Event()
                self.cache["side_done"].record(side)
