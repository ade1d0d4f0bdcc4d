"""Parallel-tempering ensemble sharded over the GPUs of one node.

What the reference does with one MPI rank per chain and point-to-point sends
(metropolis.py:140-267: bcast of the pair index, send/recv of -log_ri, of the
accept flag, of two log-likelihoods and two state vectors per accepted swap) is
done here with ONE collective per iteration:

  * chains are block-partitioned over the ranks (`partition`), all measurements
    of a chain stay on one GPU, so the per-chain log-likelihood is local;
  * every rank evaluates its block at ALL ladder temperatures -- ll(T) is not
    ll(1)/T when measurement uncertainties are non-zero
    (trial_move_evaluation.py:150-156), so a scalar per chain is not enough for a
    sequence of swaps -- and an all-gather (RCCL over xGMI; `gloo` in the CPU
    tests) assembles the [n_chains x n_T] table on every rank;
  * proposals, Metropolis accept/reject (metropolis.py:42-63) and the
    `n_chains - 1` sequential random adjacent swap attempts
    (metropolis.py:66-90, :129-135) are REPLAYED identically on every rank from a
    shared-seed generator: no further traffic, swapped states move as index
    permutations of the (replicated, tiny) host-side state array, and the new
    log-likelihood of a swapped chain is looked up in the table instead of being
    re-simulated as the reference does (metropolis.py:87-88).

Host logic only (numpy + torch.distributed); the evaluator is injected.
"""
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


def approve(new_state_log, names, active, prior, do_log, ix):
    """trial_move_generation.py:4-49, vectorised over chains; True = all checks pass.
    new_state_log: [C, n] in proposal space (log10 where do_log)."""
    lin = np.where(do_log, 10.0 ** new_state_log, new_state_log)
    ok = np.ones(new_state_log.shape[0], dtype=bool)
    for i, p in enumerate(names):
        if not active[i]:
            continue
        lo, hi = prior[p]
        ok &= (lo < lin[:, i]) & (lin[:, i] < hi)
    if "p0" in ix and "n0" in ix:
        ok &= new_state_log[:, ix["p0"]] > new_state_log[:, ix["n0"]]
    if "tauN" in ix and "tauP" in ix:
        ltn = new_state_log[:, ix["tauN"]] if do_log[ix["tauN"]] else np.log10(new_state_log[:, ix["tauN"]])
        ltp = new_state_log[:, ix["tauP"]] if do_log[ix["tauP"]] else np.log10(new_state_log[:, ix["tauP"]])
        ok &= np.abs(ltn - ltp) <= 2
    return ok


def propose(states, temperatures, base_move, names, active, prior, do_log, ix, rng, hard_bounds=True):
    """make_trial_move (trial_move_generation.py:51-96) for every chain at once:
    uniform box of half-width sqrt(T)*base_move (metropolis.py:43-44) in (log-)space,
    re-drawn up to MAX_PROPOSALS times for chains whose proposal fails approve_move."""
    do_log = np.asarray(do_log, dtype=bool)
    cur = np.where(do_log, np.log10(np.where(do_log, states, 1.0)), states)
    width = np.sqrt(np.asarray(temperatures))[:, None] * np.asarray(base_move)[None, :]
    new = cur.copy()
    todo = np.ones(states.shape[0], dtype=bool)
    for _ in range(MAX_PROPOSALS if hard_bounds else 1):
        u = rng.random(states.shape)
        cand = cur + width * (2 * u - 1)
        new[todo] = cand[todo]
        todo &= ~approve(new, names, active, prior, do_log, ix)
        if not todo.any():
            break
    return np.where(do_log, 10.0 ** new, new)


def metropolis_accept(logll_cur, logll_new, rng):
    """metropolis.py:48-53: accept if u < exp(logll_new - logll_cur); NaN -> reject."""
    lr = logll_new - logll_cur
    lr = np.where(np.isnan(lr), -np.inf, lr)
    u = rng.random(lr.shape[0])
    with np.errstate(over="ignore"):
        return u < np.exp(lr)


def replica_exchange(table, rng):
    """metropolis.py:129-135 + :66-90 on the gathered table.

    table[c, k] = log-likelihood of the state currently held by chain c at ladder
    temperature k.  Returns (owner, attempts, accepts): owner[k] = index of the
    state that sits at temperature slot k afterwards."""
    n = table.shape[0]
    owner = np.arange(n)
    attempts = np.zeros(n, dtype=np.int64)
    accepts = np.zeros(n, dtype=np.int64)
    for _ in range(n - 1):
        i = int(rng.integers(0, n - 1))
        attempts[i] += 1
        a, b = owner[i], owner[i + 1]
        bi_ui, bj_ui = table[a, i], table[a, i + 1]
        bi_uj, bj_uj = table[b, i], table[b, i + 1]
        logratio = bi_ui + bj_uj - bi_uj - bj_ui
        with np.errstate(over="ignore", invalid="ignore"):
            acc = rng.random() < np.exp(-logratio)
        if acc:
            owner[i], owner[i + 1] = b, a
            accepts[i] += 1
    return owner, attempts, accepts


def gather_ll_table(evaluator, states, temperatures, blocks, rank=0, world=1, group=None):
    """Evaluate this rank's block of chain states at all ladder temperatures and
    all-gather the [n_chains, nT] table (one collective; RCCL on GPUs, gloo in CPU tests)."""
    import torch
    import torch.distributed as dist
    lo, hi = blocks[rank]
    n = blocks[-1][1]
    local = evaluator(states[lo:hi], temperatures)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local.detach().cpu().numpy()
    sizes = [b[1] - b[0] for b in blocks]
    if len(set(sizes)) == 1:
        full = torch.empty((n, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(full, local.contiguous(), group=group)
    else:
        # uneven blocks: collectives want equal shapes, so pad to the largest block
        mx = max(sizes)
        padded = torch.zeros((mx, local.shape[1]), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
        buf = torch.empty((world * mx, local.shape[1]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(buf, padded, group=group)
        full = torch.cat([buf[r * mx:r * mx + s] for r, s in enumerate(sizes)], dim=0)
    return full.cpu().numpy()


class PTEnsemble:
    """One rank's view of the sharded ensemble.

    evaluator(states_block [c, n_params], temperatures [nT]) -> torch tensor
    [c, nT] (sum over measurements) on the device the process group works on."""

    def __init__(self, states, temperatures, evaluator, rank=0, world=1, seed=235817049752375780, group=None):
        self.states = np.array(states, dtype=np.float64)
        self.T = np.asarray(temperatures, dtype=np.float64)
        self.n = self.states.shape[0]
        self.rank, self.world, self.group = rank, world, group
        self.blocks = partition(self.n, world)
        self.lo, self.hi = self.blocks[rank]
        self.evaluator = evaluator
        self.rng = np.random.default_rng(seed)  # metropolis.py:296 (same seed on every rank, :410-411)
        self.table = None
        self.logll = None
        self.swap_attempts = np.zeros(self.n, dtype=np.int64)
        self.swap_accepts = np.zeros(self.n, dtype=np.int64)

    def gather_table(self, states):
        """Evaluate the local block at all ladder temperatures, all-gather -> [n, nT] numpy."""
        return gather_ll_table(self.evaluator, states, self.T, self.blocks, self.rank, self.world, self.group)

    def initialise(self):
        self.table = self.gather_table(self.states)
        self.logll = self.table[np.arange(self.n), np.arange(self.n)].copy()

    def iterate(self, proposer, do_swap):
        """One Metropolis iteration for every chain (+ replica exchange if do_swap)."""
        new_states = proposer(self.states, self.T, self.rng)
        new_table = self.gather_table(new_states)
        new_ll = new_table[np.arange(self.n), np.arange(self.n)]
        acc = metropolis_accept(self.logll, new_ll, self.rng)
        self.states[acc] = new_states[acc]
        self.table[acc] = new_table[acc]
        self.logll[acc] = new_ll[acc]
        if do_swap:
            owner, att, accs = replica_exchange(self.table, self.rng)
            self.states = self.states[owner]
            self.table = self.table[owner]
            self.logll = self.table[np.arange(self.n), np.arange(self.n)].copy()
            self.swap_attempts += att
            self.swap_accepts += accs
        return acc
