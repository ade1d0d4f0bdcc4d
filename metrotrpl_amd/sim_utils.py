"""Grid value object -- same fields and errors as the reference's
sim_utils.Grid (sim_utils.py:246-277)."""
from sys import float_info

import numpy as np

# sim_utils.py:17-23
DEFAULT_HMAX = 4
DEFAULT_TEMPER_FREQ = 10
NEGATIVE_FRAC_TOL = 0.2


class Grid:
    """Collection of values describing the grid dimensions for the simulation."""

    def __init__(self, thickness, nx, tSteps, hmax):
        self.thickness = thickness
        self.nx = nx
        self.dx = self.thickness / self.nx
        self.xSteps = np.linspace(self.dx / 2, self.thickness - self.dx / 2, self.nx)
        if tSteps[0] != 0:
            raise ValueError("Grid error - times must start at t=0")
        self.tSteps = tSteps
        self.start_time = 0
        self.nt = len(tSteps) - 1
        self.hmax = hmax
        self.final_time = self.tSteps[-1]
        self.min_y = float_info.min


MAX_PROPOSALS = 100  # sim_utils.py:20


class History:
    """Record of the states an ensemble has visited -- same attributes as the
    reference's History (sim_utils.py:25-76), so that pickles are interchangeable."""

    def __init__(self, n_chains, num_iters, names):
        self.states_are_one_array = True
        self.states = np.zeros((n_chains, len(names), num_iters), dtype=float)
        self.accept = np.zeros((n_chains, num_iters), dtype=int)
        self.loglikelihood = np.zeros((n_chains, num_iters), dtype=float)
        self.swap_attempts = np.zeros(n_chains, dtype=int)
        self.swap_accept = np.zeros(n_chains, dtype=int)

    def update(self, names):
        """mean_{param} views of states[:, i] (sim_utils.py:37-41; what the reference GUI reads)."""
        for i, param in enumerate(names):
            setattr(self, f"mean_{param}", self.states[:, i])

    def pack(self, states, logll, accept):
        self.states, self.loglikelihood, self.accept = states, logll, accept

    def truncate(self, k):
        self.states = self.states[:, :, :k]
        self.accept = self.accept[:, :k]
        self.loglikelihood = self.loglikelihood[:, :k]

    def extend(self, new_num_iters):
        """Grow (zero-padded) or cut the record to new_num_iters iterations (sim_utils.py:56-76)."""
        have = self.accept.shape[1]
        if new_num_iters <= have:
            if new_num_iters < have:
                self.truncate(new_num_iters)
            return
        add = new_num_iters - have
        n_chains, n_par = self.states.shape[0], self.states.shape[1]
        self.accept = np.concatenate((self.accept, np.zeros((n_chains, add))), axis=1)
        self.loglikelihood = np.concatenate((self.loglikelihood, np.zeros((n_chains, add))), axis=1)
        self.states = np.concatenate((self.states, np.zeros((n_chains, n_par, add))), axis=2)


def _as_array(table, names, dtype, default=None):
    if default is None:
        return np.array([table[p] for p in names], dtype=dtype)
    return np.array([table.get(p, default) for p in names], dtype=dtype)


class Ensemble:
    """Settings + history of a parallel-tempering ensemble: the same
    `ensemble_fields` / `unique_fields` / `H` / `random_state` / `latest_iter`
    layout the reference builds in sim_utils.py:98-199 (it consumes -- pops from --
    param_info and MCMC_fields exactly as the reference does)."""

    ESSENTIAL = ("output_path", "init_cond_path", "measurement_path", "checkpoint_freq", "ini_mode", "solver",
                 "model", "num_iters", "log_y")
    OPTIONAL = ("rtol", "atol", "scale_factor", "load_checkpoint", "fittable_fluences", "fittable_absps",
                "irf_convolution", "do_mu_constraint")

    def __init__(self, param_info, sim_info, MCMC_fields, num_iters, verbose=False):
        ef = {}
        for key in self.ESSENTIAL:
            ef[key] = MCMC_fields.pop(key)
        for key in self.OPTIONAL:
            ef[key] = MCMC_fields.pop(key, None)
        ef["temper_freq"] = MCMC_fields.pop("temper_freq", DEFAULT_TEMPER_FREQ)
        if "model_uncertainty" in MCMC_fields and "likel2move_ratio" in MCMC_fields:
            MCMC_fields.pop("likel2move_ratio")
        if "likel2move_ratio" in MCMC_fields:
            ef["likel2move_ratio"] = MCMC_fields.pop("likel2move_ratio")
        ef["hard_bounds"] = MCMC_fields.pop("hard_bounds", 0)
        ef["hmax"] = MCMC_fields.pop("hmax", DEFAULT_HMAX)
        ef["force_min_y"] = MCMC_fields.pop("force_min_y", 0)

        names = param_info["names"]
        ef["prior_dist"] = param_info.pop("prior_dist")
        ef["do_log"] = _as_array(param_info.pop("do_log"), names, bool)
        active = param_info.pop("active")
        move = param_info.pop("trial_move")
        ef["base_trial_move"] = np.array([move[p] if active[p] else 0 for p in names], dtype=float)
        ef["active"] = _as_array(active, names, bool)
        ef["units"] = _as_array(param_info.pop("unit_conversions"), names, float, default=1)
        ef["_param_indexes"] = {name: i for i, name in enumerate(names)}
        ef["_T"] = MCMC_fields.pop("parallel_tempering", [1])
        ef["_n_chains"] = len(ef["_T"])
        ef["names"] = param_info.pop("names")

        self.ensemble_fields = ef
        self.H = History(ef["_n_chains"], num_iters, ef["names"])
        self.H.states[:, :, 0] = _as_array(param_info["init_guess"], ef["names"], float)

        self.unique_fields = []
        for i in range(ef["_n_chains"]):
            u = dict(MCMC_fields)
            u["_T"] = ef["_T"][i]
            if "likel2move_ratio" in ef:
                u["model_uncertainty"] = {m: max(ef["base_trial_move"]) * ef["likel2move_ratio"][m]
                                          for m in sim_info["meas_types"]}
            self.unique_fields.append(u)
        ef["do_parallel_tempering"] = ef["_n_chains"] > 1
        ef["_sim_info"] = sim_info
        self.latest_iter = 0
        self.random_state = None

    def checkpoint(self, fname, reference_compatible=True):
        """Pickle the ensemble (sim_utils.py:91-96).  With reference_compatible the
        classes are recorded as `sim_utils.Ensemble` / `sim_utils.History`, the names
        the reference's own loaders and GUI resolve."""
        self.H.update(self.ensemble_fields["names"])
        dump_pickle(self, fname, reference_compatible)


def dump_pickle(obj, fname, reference_compatible=True):
    import io
    import pickle
    import sys
    mine = sys.modules[__name__]
    if not reference_compatible:
        with open(fname, "wb+") as f:
            pickle.dump(obj, f)
        return
    # pickle writes classes by (module, qualname) and checks that the name resolves to the same
    # object: for the duration of the dump this module answers to the reference's module name
    prev = sys.modules.get("sim_utils")
    olds = [(c, c.__module__) for c in (Ensemble, History)]
    try:
        sys.modules["sim_utils"] = mine
        for c, _ in olds:
            c.__module__ = "sim_utils"
        buf = io.BytesIO()
        pickle.dump(obj, buf)
    finally:
        for c, m in olds:
            c.__module__ = m
        if prev is None:
            sys.modules.pop("sim_utils", None)
        else:
            sys.modules["sim_utils"] = prev
    with open(fname, "wb+") as f:
        f.write(buf.getvalue())


def load_pickle(fname):
    """Load an Ensemble written by this package or by the reference (classes recorded
    as sim_utils.*) without needing the reference on sys.path."""
    import pickle
    mine = {"Ensemble": Ensemble, "History": History, "EnsembleTemplate": Ensemble}

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            if module in ("sim_utils", __name__) and name in mine:
                return mine[name]
            return super().find_class(module, name)

    with open(fname, "rb") as f:
        return _Unpickler(f).load()
