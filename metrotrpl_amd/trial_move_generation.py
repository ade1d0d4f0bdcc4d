"""Proposal generation with the reference's exact random-draw order
(trial_move_generation.py:4-96): one `RNG.random(n_params)` per try, proposals
re-drawn from the CURRENT state up to MAX_PROPOSALS times under hard bounds, the
last candidate returned even if it still fails.  Host logic only.

`pt.propose` is the vectorised all-chains-at-once variant of the same rule (its
draw order differs); the drivers in `metropolis.py` use this module whenever the
walk has to be reproducible against the reference.
"""
import numpy as np

from .sim_utils import MAX_PROPOSALS

MU_SEED = 235817049752375781  # the walk's seed (metropolis.py:296) + 1


def approve_move(new_state, shared_fields):
    """Names of the failed checks (empty list = approved), trial_move_generation.py:4-49.
    new_state is in proposal space (log10 where do_log)."""
    names = shared_fields["names"]
    do_log = shared_fields["do_log"]
    ix = shared_fields["_param_indexes"]
    prior = shared_fields["prior_dist"]
    lin = np.where(do_log, 10 ** new_state, new_state)
    failed = []
    for i, p in enumerate(names):
        if shared_fields["active"][i] and not (prior[p][0] < lin[i] < prior[p][1]):
            failed.append(f"{p}_size")
    if "p0" in names and "n0" in names and not (new_state[ix["p0"]] > new_state[ix["n0"]]):
        failed.append("p0_greater")
    if "tauN" in names and "tauP" in names:
        ltn = new_state[ix["tauN"]] if do_log[ix["tauN"]] else np.log10(new_state[ix["tauN"]])
        ltp = new_state[ix["tauP"]] if do_log[ix["tauP"]] else np.log10(new_state[ix["tauP"]])
        if not (np.abs(ltn - ltp) <= 2):
            failed.append("tn_tp_close")
    return failed


def make_trial_move(current_state, trial_move, shared_fields, RNG, logger=None):
    """Uniform box proposal around current_state (half-width trial_move, in log10 space
    where do_log), trial_move_generation.py:51-96."""
    do_log = shared_fields["do_log"]
    cur = np.array(current_state, dtype=float)
    with np.errstate(divide="ignore", invalid="ignore"):  # entries that are not log-scaled may be <= 0
        cur = np.where(do_log, np.log10(cur), cur)
    mu_constraint = shared_fields.get("do_mu_constraint", None)
    max_tries = MAX_PROPOSALS if shared_fields.get("hard_bounds", 0) else 1
    new = np.array(cur)
    for tries in range(1, max_tries + 1):
        new = cur + trial_move * (2 * RNG.random(cur.shape) - 1)
        if mu_constraint is not None:
            # ambipolar-mobility constraint (trial_move_generation.py:79-85).  The reference draws it from
            # numpy's GLOBAL, unseeded generator; here every rank replays every chain's proposals and
            # must see the same value, so it comes from a generator of its own with a fixed seed (kept in
            # shared_fields, hence in the checkpoints): RNG's stream stays the reference's.
            ambi, ambi_std = mu_constraint[0], mu_constraint[1]
            mu_rng = shared_fields.get("_mu_rng")
            if mu_rng is None:
                mu_rng = shared_fields["_mu_rng"] = np.random.default_rng(MU_SEED)
            mu_a = mu_rng.uniform(ambi - ambi_std, ambi + ambi_std)
            ix = shared_fields["_param_indexes"]
            new[ix["mu_p"]] = np.log10((2 / mu_a - 1 / 10 ** new[ix["mu_n"]]) ** -1)
        failed = approve_move(new, shared_fields)
        if not failed:
            if logger is not None:
                logger.debug(f"Found params in {tries} tries")
            break
        if logger is not None:
            logger.warning(f"Failed checks: {failed}")
    return np.where(do_log, 10 ** new, new)
