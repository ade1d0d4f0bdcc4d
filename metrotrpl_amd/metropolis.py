"""Metropolis / parallel-tempering driver with the reference's call signature
(`metro`, metropolis.py:283-476) and its random-number draw order, evaluated in
batches on the GPUs.

What the reference does one chain and one measurement at a time
(`trial_displacement_move` :42-63 -> `eval_trial_move` -> LSODA solve per
measurement) is done here per ITERATION for the whole ensemble:

  1. proposals for every chain are drawn on the host in the reference's order --
     the log-likelihood of a proposal never influences a draw, only how many
     `RNG.random(n_params)` tries a chain needs (trial_move_generation.py:71-94)
     and one `RNG.random()` for its accept roll (:39), so all draws of an
     iteration can be made BEFORE anything is simulated;
  2. one launch evaluates n_chains x num_meas solves + IRF convolutions +
     likelihoods at ALL ladder temperatures (`likelihood.LikelihoodEngine`); with
     several ranks each evaluates its block of chains and one all-gather
     assembles the [n_chains, n_T] table (`pt.gather_ll_table`);
  3. accept/reject and the replica-exchange attempts are replayed on the host of
     every rank from the table: no re-simulation after a swap (the reference
     re-runs both chains, :87-88 / :249,:259).

`rng_mode` selects whose draw order is reproduced:
  "serial" -- `main_metro_loop_serial` (:93-137): ONE generator, chains in order;
  "mpi"    -- `main_metro_loop` (:140-267): every chain owns a copy of the same
              generator (the reference broadcasts one RNG state to all ranks,
              :410-411); chain 0's copy also picks the swap pairs (:209), chain
              i's copy rolls the swap acceptance (:232).

Host logic only (numpy + torch.distributed); the evaluator is injected, so the
same code is exercised on CPU by tests/ with the oracle as evaluator.
"""
import copy
import logging
import os
from time import perf_counter

import numpy as np

from . import pt
from .sim_utils import Ensemble, load_pickle
from .trial_move_generation import make_trial_move

MSG_FREQ = 100     # metropolis.py:31
MSG_COOLDOWN = 3   # metropolis.py:32
SEED = 235817049752375780  # metropolis.py:296


def roll_acceptance(rng, logratio):
    """metropolis.py:35-39."""
    with np.errstate(over="ignore"):
        if isinstance(logratio, np.ndarray):
            return rng.random(len(logratio)) < np.exp(logratio)
        return rng.random() < np.exp(logratio)


class Walk:
    """The ensemble's mutable walk state between iterations: the per-chain rows of
    the log-likelihood table (the reference's non-picklable `ll_funcs`) and the
    generator(s)."""

    def __init__(self, shared_fields, unique_fields, RNG, gather, rng_mode="serial", logger=None,
                 serial_swap_aliasing=False):
        """gather(states [n_chains, n_params]) -> numpy [n_chains, n_T]:
        table[c, k] = log-likelihood of states[c] (summed over measurements) at ladder
        temperature k (`shared_fields["_T"][k]`)."""
        if rng_mode not in ("serial", "mpi"):
            raise ValueError("rng_mode must be 'serial' or 'mpi'")
        self.sf, self.uf = shared_fields, unique_fields
        self.n = shared_fields["_n_chains"]
        self.T = np.asarray(shared_fields["_T"], dtype=float)
        self.gather = gather
        self.mode = rng_mode
        self.logger = logger or logging.getLogger("metrotrpl_amd")
        # reproduces `states[i,:,k], states[i+1,:,k] = states[i+1,:,k], states[i,:,k]` of
        # swap_move_serial (:85): a tuple swap of numpy VIEWS, after which both chains hold the
        # state of chain i+1.  Only for bit-for-bit comparisons with the serial reference.
        self.alias = bool(serial_swap_aliasing)
        if rng_mode == "serial":
            self.rngs = [RNG] * self.n
        else:
            self.rngs = [RNG] + [copy.deepcopy(RNG) for _ in range(self.n - 1)]
        self.table = None
        self.swap_attempts = np.zeros(self.n, dtype=int)
        self.swap_accept = np.zeros(self.n, dtype=int)

    # -- random_state of a checkpoint: ONE bit-generator state dict in both modes, as the reference
    # stores it (`MS_list.random_state = RNG.bit_generator.state` on rank 0, metropolis.py:362,:446) and
    # loads it (`RNG.bit_generator.state = RNG_state` after a bcast to every rank, :410-411).  In "mpi"
    # mode that is chain 0's generator; a RESUMED walk therefore restarts every chain from chain 0's
    # stream position -- the per-chain streams that had drifted apart are re-synchronised, exactly what
    # a resumed reference run does -- so a pickle written here resumes in the reference and vice versa.
    def random_state(self):
        return self.rngs[0].bit_generator.state

    def set_random_state(self, state):
        if isinstance(state, list):  # checkpoints of metrotrpl_amd 0.1: one state per chain
            state = state[0]
        for r in self.rngs:
            r.bit_generator.state = state

    def evaluate_current(self, states_k):
        self.table = self.gather(np.ascontiguousarray(states_k))
        return self.table[np.arange(self.n), np.arange(self.n)].copy()

    def displacement(self, k, states, logll, accept):
        """metropolis.py:42-63 for every chain; states [n, n_params, iters]."""
        n, sf = self.n, self.sf
        new = np.empty((n, states.shape[1]))
        u = np.empty(n)
        for m in range(n):
            move = self.uf[m]["_T"] ** 0.5 * sf["base_trial_move"]
            new[m] = make_trial_move(states[m, :, k - 1], move, sf, self.rngs[m], None)
            u[m] = self.rngs[m].random()
        new_table = self.gather(new)
        for m in range(n):
            _logll = new_table[m, m]
            logratio = _logll - logll[m, k - 1]
            if np.isnan(logratio):
                logratio = -np.inf
            with np.errstate(over="ignore"):
                accepted = u[m] < np.exp(logratio)
            if accepted:
                logll[m, k] = _logll
                states[m, :, k] = new[m]
                accept[m, k] = 1
                self.table[m] = new_table[m]
            else:
                logll[m, k] = logll[m, k - 1]
                states[m, :, k] = states[m, :, k - 1]

    def swaps(self, k, states, logll):
        """metropolis.py:129-135 + :66-90 (serial) / :203-263 (mpi) from the table."""
        n, tab = self.n, self.table
        for _ in range(n - 1):
            i = int(self.rngs[0].integers(0, n - 1))
            self.swap_attempts[i] += 1
            bi_ui, bj_ui = tab[i, i], tab[i, i + 1]
            bi_uj, bj_uj = tab[i + 1, i], tab[i + 1, i + 1]
            logratio = bi_ui + bj_uj - bi_uj - bj_ui
            roller = self.rngs[0] if self.mode == "serial" else self.rngs[i]
            with np.errstate(over="ignore", invalid="ignore"):
                accepted = roll_acceptance(roller, -logratio)
            if not accepted:
                continue
            self.swap_accept[i] += 1
            logll[i, k] = bi_uj
            logll[i + 1, k] = bj_ui
            if self.alias:
                states[i, :, k] = states[i + 1, :, k]
                tab[i] = tab[i + 1]
            else:
                states[[i, i + 1], :, k] = states[[i + 1, i], :, k]
                tab[[i, i + 1]] = tab[[i + 1, i]]

    def run(self, states, logll, accept, starting_iter, num_iters, need_initial_state=True):
        """main_metro_loop_serial (:93-137) over iterations [starting_iter, num_iters)."""
        sf = self.sf
        if need_initial_state:
            self.logger.info("Simulating initial state:")
            logll[:, 0] = self.evaluate_current(states[:, :, 0])
            starting_iter += 1
        else:
            self.evaluate_current(states[:, :, starting_iter - 1])
        for k in range(starting_iter, num_iters):
            if k % MSG_FREQ == 0 or k < starting_iter + MSG_COOLDOWN:
                for m in range(self.n):
                    self.logger.info(f"Iter {k} MetroState #{m} Current state: {states[m, :, k - 1]} "
                                     f"logll {logll[m, k - 1]}")
            self.displacement(k, states, logll, accept)
            if sf["do_parallel_tempering"] and k % sf["temper_freq"] == 0:
                self.swaps(k, states, logll)
        return states, logll, accept, self.swap_attempts, self.swap_accept


def _start_logging(log_dir, name, verbose):
    logger = logging.getLogger(name)
    logger.setLevel(logging.DEBUG if verbose else logging.INFO)
    handler = logging.FileHandler(os.path.join(log_dir, f"{name}.log"), mode="a")
    handler.setFormatter(logging.Formatter("%(asctime)s %(levelname)s: %(message)s"))
    logger.addHandler(handler)
    return logger, handler


def _stretches(start, stop, every):
    """(a, b, last) for the stretches [a, b) a run from `start` to `stop` is cut into when it checkpoints after every
    `every` iterations: b = min(a + every, stop); the stretch that reaches `stop` is the last (metropolis.py:388-412)."""
    a = start
    while True:
        b = stop if every <= 0 else min(a + every, stop)
        yield a, b, b == stop
        if b == stop:
            return
        a = b


def metro(sim_info, iniPar, e_data, MCMC_fields, param_info, verbose=False, export_path="", **kwargs):
    """Drop-in for metropolis.metro (:283-476).  Extra keyword arguments:
    rng_mode ("serial" | "mpi", default "mpi" -- the reference's production mode is one MPI
    rank per chain; serial_fallback=True selects "serial" as in the reference),
    gather (an injected table evaluator; default: the GPU LikelihoodEngine of this rank),
    irf_dir (where irf_{wavelength}nm.csv live, default "IRFs" as metropolis.py:335).
    Launch under torch.distributed.run to shard the chains over several GPUs."""
    clock0 = perf_counter()
    serial_fallback = kwargs.get("serial_fallback", False)
    rng_mode = kwargs.get("rng_mode", "serial" if serial_fallback else "mpi")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    out_dir = MCMC_fields["output_path"]
    os.makedirs(out_dir, exist_ok=True)
    load_checkpoint = MCMC_fields.get("load_checkpoint", None)
    num_iters = MCMC_fields["num_iters"]
    checkpoint_freq = MCMC_fields.get("checkpoint_freq", num_iters)
    RNG = np.random.default_rng(SEED)
    name = kwargs.get("logger_name", "Ensemble0") + f"-rank{rank}-"
    logger, handler = _start_logging(out_dir, name, verbose)

    starting_iter = 0
    if load_checkpoint is None:
        MS_list = Ensemble(param_info, sim_info, MCMC_fields, num_iters, verbose)
        ef = MS_list.ensemble_fields
        if rank == 0:
            MS_list.checkpoint(os.path.join(ef["output_path"], export_path))
        ef["_init_params"] = iniPar
        ef["_times"], ef["_vals"], ef["_uncs"] = e_data
        MS_list.random_state = RNG.bit_generator.state
        if ef.get("irf_convolution", None) is not None:
            from . import laplace
            irfs = {}
            for w in ef["irf_convolution"]:
                if w > 0 and int(w) not in irfs:
                    irfs[int(w)] = np.loadtxt(os.path.join(kwargs.get("irf_dir", "IRFs"), f"irf_{int(w)}nm.csv"),
                                              delimiter=",")
            ef["_IRF_tables"] = laplace.make_I_tables(irfs)
        else:
            ef["_IRF_tables"] = {}
    else:
        MS_list = load_pickle(os.path.join(out_dir, load_checkpoint))
        ef = MS_list.ensemble_fields
        if "starting_iter" in MCMC_fields and MCMC_fields["starting_iter"] < MS_list.latest_iter:
            starting_iter = MCMC_fields["starting_iter"]
            MS_list.H.extend(starting_iter)
        else:
            starting_iter = MS_list.latest_iter
            MS_list.H.extend(num_iters)
            ef["num_iters"] = MCMC_fields["num_iters"]

    states, logll, accept = MS_list.H.states, MS_list.H.loglikelihood, MS_list.H.accept
    uf = MS_list.unique_fields
    n_chains = ef["_n_chains"]

    gather = kwargs.get("gather", None)
    use_dist = world > 1
    if gather is None:
        import torch
        from .likelihood import engine_from_fields
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        if use_dist:
            import torch.distributed as dist
            if not dist.is_initialized():
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                dist.init_process_group("nccl", device_id=dev)
        eng = engine_from_fields(ef, uf[0], device=local)
        d_T = torch.from_numpy(np.ascontiguousarray(ef["_T"], dtype=np.float64)).to(dev)
        blocks = pt.partition(n_chains, world)

        def evaluator(block, temps):
            eng.upload(block)
            ll = eng.launch(d_T, len(temps))
            return ll.view(-1, eng.M, len(temps)).sum(dim=1)

        def gather(all_states):
            return pt.gather_ll_table(evaluator, all_states, ef["_T"], blocks, rank, world)

    walk = Walk(ef, uf, RNG, gather, rng_mode=rng_mode, logger=logger,
                serial_swap_aliasing=kwargs.get("serial_swap_aliasing", False))
    walk.set_random_state(MS_list.random_state)
    target = os.path.join(ef["output_path"], export_path)

    def fold_swap_counts():
        MS_list.H.swap_attempts = MS_list.H.swap_attempts + walk.swap_attempts
        MS_list.H.swap_accept = MS_list.H.swap_accept + walk.swap_accept
        walk.swap_attempts[:] = 0
        walk.swap_accept[:] = 0

    def snapshot(upto):
        """What a checkpoint holds (metropolis.py:402-412, :431-446, :464-472): history, the iteration reached, ONE
        generator state."""
        MS_list.latest_iter = upto
        MS_list.H.pack(states, logll, accept)
        MS_list.random_state = walk.random_state()

    # The run is cut into stretches of checkpoint_freq iterations (the last one shorter); every stretch but the last
    # is followed by a checkpoint, the last by the export.  In "mpi" mode a stretch's swap counts are folded in
    # before its checkpoint (:431-437); the serial loop's only after the last stretch (:464-466).
    fresh = load_checkpoint is None
    for a, b, last in _stretches(starting_iter, num_iters, checkpoint_freq):
        logger.info(f"stretch {a} -> {b}")
        walk.run(states, logll, accept, a, b, need_initial_state=fresh and a == starting_iter)
        if last:
            break
        if rng_mode == "mpi":
            fold_swap_counts()
        snapshot(b)
        if rank == 0:
            logger.info(f"checkpoint after iteration {b}")
            MS_list.checkpoint(target)
    logger.info(f"rank {rank}: {perf_counter() - clock0:.3f} s")

    fold_swap_counts()
    snapshot(num_iters)
    if rank == 0 and export_path:
        logger.info(f"export to {ef['output_path']}")
        MS_list.checkpoint(target)
    logger.removeHandler(handler)
    handler.close()
    return MS_list
