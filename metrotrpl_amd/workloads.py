"""
Synthetic workloads for tests and bench.py.

Pure numpy; no reference code, no oracle, no GPU.  Values follow the
reference's example generator MCMC_script_writer.py:45-115 (parameter names,
unit conversions, prior box, un-randomised initial guesses) and
Inputs/staub_MAPI_threepower_twothick_fluences.csv:1-6 (power scan).
"""
import numpy as np

# MCMC_script_writer.py:56-57
PARAM_NAMES = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Sf", "Sb", "tauN", "tauP", "eps", "Tm"]

# MCMC_script_writer.py:59-66
UNIT_CONVERSIONS = {
    "n0": ((1e-7) ** 3), "p0": ((1e-7) ** 3),
    "mu_n": ((1e7) ** 2) / (1e9), "mu_p": ((1e7) ** 2) / (1e9),
    "ks": ((1e7) ** 3) / (1e9),
    "Cn": ((1e7) ** 6) / (1e9), "Cp": ((1e7) ** 6) / (1e9),
    "Sf": 1e-2, "Sb": 1e-2, "Tm": 1,
}

# MCMC_script_writer.py:72-85 (active parameters only)
PRIOR_BOX = {
    "p0": (1e14, 1e16), "mu_n": (1e0, 1e2), "mu_p": (1e0, 1e2), "ks": (1e-11, 1e-9),
    "Cn": (1e-29, 1e-27), "Cp": (1e-29, 1e-27), "Sf": (1e-1, 1e4), "Sb": (1e-1, 1e4),
    "tauN": (1, 2e3), "tauP": (1, 1e4),
}

# MCMC_script_writer.py:87-100
INITIAL_GUESS = {
    "n0": 1e8, "p0": 3e15, "mu_n": 20, "mu_p": 20, "ks": 4.8e-11, "Cn": 4.4e-29, "Cp": 4.4e-29,
    "Sf": 10, "Sb": 10, "tauN": 511, "tauP": 871, "eps": 10, "Tm": 300,
}

# Inputs/staub_MAPI_threepower_twothick_fluences.csv:1-6
STAUB_FLUENCES = [[2.12e11, 6e4, 1], [2.12e11, 6e4, 1], [1.92e12, 6e4, 1],
                  [1.92e12, 6e4, 1], [2.75e13, 6e4, 1], [2.75e13, 6e4, 1]]
# MCMC_script_writer.py:47
STAUB_LENGTHS = [311, 2000, 311, 2000, 311, 2000]

SEED = 20261004


def indexes():
    return {name: i for i, name in enumerate(PARAM_NAMES)}


def units():
    return np.array([UNIT_CONVERSIONS.get(name, 1) for name in PARAM_NAMES], dtype=float)


def initial_state():
    return np.array([INITIAL_GUESS[n] for n in PARAM_NAMES], dtype=float)


def default_times(t_end=2000.0, nt=1000):
    return np.linspace(0, t_end, nt + 1)


def _approved(state, ix):
    """trial_move_generation.py:22-45 (physics checks only: the box is met by construction)."""
    if not state[ix["p0"]] > state[ix["n0"]]:
        return False
    return abs(np.log10(state[ix["tauN"]]) - np.log10(state[ix["tauP"]])) <= 2


def draw_states(n, rng=None, with_fluence=True):
    """Config-2 recipe: log-uniform inside the prior box for every active
    parameter (PARAM_NAMES order), then the fluence (log-uniform over the staub
    power scan); whole draws failing the physics checks are rejected.

    Returns (states [n, 13] in user units, fluences [n])."""
    rng = np.random.default_rng(SEED) if rng is None else rng
    ix = indexes()
    states = np.empty((n, len(PARAM_NAMES)))
    flu = np.empty(n)
    k = 0
    while k < n:
        s = initial_state()
        for name in PARAM_NAMES:
            if name in PRIOR_BOX:
                lo, hi = PRIOR_BOX[name]
                s[ix[name]] = 10 ** rng.uniform(np.log10(lo), np.log10(hi))
        f = 10 ** rng.uniform(np.log10(2.12e11), np.log10(2.75e13)) if with_fluence else 0.0
        if not _approved(s, ix):
            continue
        states[k] = s
        flu[k] = f
        k += 1
    return states, flu


def config2(n=1024, nx=128, thickness=2000.0):
    """1024-solve synthetic batch, single-layer absorber (BASELINE.json configs[1])."""
    states, flu = draw_states(n)
    ini = np.stack([flu, np.full(n, 6e4), np.ones(n)], axis=1)
    return dict(states=states, iniPar=ini, nx=nx, thickness=thickness, tSteps=default_times(),
                hmax=4.0, meas="TRPL", model="std", ini_mode="fluence")


def ladder(n_chains, t_max=1000.0):
    """Config 3/4 temperature ladder T_k = t_max^(k/(n-1))."""
    if n_chains == 1:
        return np.ones(1)
    return t_max ** (np.arange(n_chains) / (n_chains - 1))


def pt_ensemble(n_chains, nx=128, rng=None):
    """Config 3/4: n_chains parallel-tempering chains x the 6-measurement staub
    power scan.  Chain states are config-2 draws (a spread-out ensemble, as
    after burn-in at high temperature).  Returns per-solve arrays, chain-major."""
    states, _ = draw_states(n_chains, rng=rng, with_fluence=False)
    n_meas = len(STAUB_LENGTHS)
    st = np.repeat(states, n_meas, axis=0)
    ini = np.tile(np.array(STAUB_FLUENCES, dtype=float), (n_chains, 1))
    thick = np.tile(np.array(STAUB_LENGTHS, dtype=float), n_chains)
    return dict(states=st, chain_states=states, iniPar=ini, thickness=thick, nx=nx,
                tSteps=default_times(), hmax=4.0, meas="TRPL", model="std", ini_mode="fluence",
                n_chains=n_chains, n_meas=n_meas, temperatures=ladder(n_chains))


def config5(n_states=16, nx=512):
    """BASELINE.json configs[4] / SURVEY.md 8d config 5: nx=512 drift-diffusion on a thick (2000 nm)
    and a thin (311 nm) film, TRPL and TRTS alternating: every state of the seeded config-2 stream
    is simulated as (2000 nm, TRPL), (2000 nm, TRTS), (311 nm, TRPL), (311 nm, TRTS)."""
    states, flu = draw_states(n_states)
    st = np.repeat(states, 4, axis=0)
    ini = np.stack([np.repeat(flu, 4), np.full(4 * n_states, 6e4), np.ones(4 * n_states)], axis=1)
    thick = np.tile([2000.0, 2000.0, 311.0, 311.0], n_states)
    meas = ["TRPL", "TRTS", "TRPL", "TRTS"] * n_states
    return dict(states=st, iniPar=ini, nx=nx, thickness=thick, tSteps=default_times(), hmax=4.0, meas=meas,
                model="std", ini_mode="fluence", sigma={"TRPL": 1.0, "TRTS": 10.0})
