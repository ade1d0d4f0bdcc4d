"""
make_mcmc_golden.py -- golden Metropolis / parallel-tempering WALKS produced by the
reference's own serial driver (`metropolis.main_metro_loop_serial`, imported from
/root/reference in the build container; same temp-dir numba stand-in as
make_golden.py).  Only data is committed: tests/golden/mcmc.npz holds the inputs
(as JSON + arrays) and the reference's states / log-likelihoods / accept flags /
swap counters for

  single : 1 chain, 2 measurements (311 nm + 2000 nm film), 40 iterations;
  pt     : 3 chains (T = 1, 3, 9), swaps every 2 iterations, 24 iterations, one
           measurement convolved with a 189-tap IRF, a fittable scale factor `_s1`.

The solver runs at RTOL=1e-10 / ATOL=1e-14 so that the stored log-likelihoods are the
converged ones to ~1e-8.

    python tests/golden/make_mcmc_golden.py
"""
import json
import logging
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
from make_golden import _import_reference  # noqa: E402

SEED = 235817049752375780
NX = 32


def base_config(n_chains, num_iters, with_scale, irf):
    names = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Sf", "Sb", "tauN", "tauP", "eps", "Tm"]
    units = {"n0": 1e-21, "p0": 1e-21, "mu_n": 1e5, "mu_p": 1e5, "ks": 1e12, "Cn": 1e33, "Cp": 1e33,
             "Sf": 1e-2, "Sb": 1e-2, "Tm": 1}
    prior = {"n0": (0, np.inf), "p0": (1e14, 1e16), "mu_n": (1, 100), "mu_p": (1, 100), "ks": (1e-11, 1e-9),
             "Cn": (1e-29, 1e-27), "Cp": (1e-29, 1e-27), "Sf": (0.1, 1e4), "Sb": (0.1, 1e4), "tauN": (1, 2000),
             "tauP": (1, 10000), "eps": (0, np.inf), "Tm": (0, np.inf)}
    guess = {"n0": 1e8, "p0": 3e15, "mu_n": 20, "mu_p": 20, "ks": 4.8e-11, "Cn": 4.4e-29, "Cp": 4.4e-29,
             "Sf": 10, "Sb": 10, "tauN": 511, "tauP": 871, "eps": 10, "Tm": 300}
    active = {n: int(n not in ("n0", "eps", "Tm")) for n in names}
    move = {n: 0.04 for n in names}
    do_log = {n: 1 for n in names}
    if with_scale:
        names = names + ["_s1"]
        do_log["_s1"] = 1
        prior["_s1"] = (0, np.inf)
        guess["_s1"] = 1.3
        move["_s1"] = 0.02
        active["_s1"] = 1
    param_info = {"names": names, "active": active, "unit_conversions": units, "do_log": do_log,
                  "prior_dist": prior, "init_guess": guess, "trial_move": move}
    sim_info = {"lengths": [311.0, 2000.0], "nx": [NX, NX], "meas_types": ["TRPL", "TRPL"], "num_meas": 2}
    T = [float(3 ** k) for k in range(n_chains)]
    MCMC_fields = {"init_cond_path": None, "measurement_path": None, "output_path": "mcmc-out",
                   "num_iters": num_iters, "solver": ("solveivp",), "model": "std",
                   "model_uncertainty": {"TRPL": 0.05}, "ini_mode": "fluence", "log_y": 1, "hard_bounds": 1,
                   "checkpoint_freq": num_iters, "load_checkpoint": None, "rtol": 1e-10, "atol": 1e-14,
                   "hmax": 4, "temper_freq": 2, "parallel_tempering": T,
                   "irf_convolution": [745, 0] if irf else None,
                   "scale_factor": [0.02, [1], None, [1.0, 1.3]] if with_scale else None}
    return param_info, sim_info, MCMC_fields


def jsonable(o):
    if isinstance(o, dict):
        return {k: jsonable(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [jsonable(v) for v in o]
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, (np.floating, np.integer)):
        return o.item()
    return o


def run_case(name, n_chains, num_iters, with_scale, irf, out):
    fs, laplace, sim_utils, tme = _import_reference()
    import metropolis as ref_metro
    from metrotrpl_amd.pt_bench import synthetic_irf
    param_info, sim_info, MCMC_fields = base_config(n_chains, num_iters, with_scale, irf)
    cfg_json = json.dumps(jsonable({"param_info": param_info, "sim_info": sim_info, "MCMC_fields": MCMC_fields}))
    times = np.linspace(0, 100, 51)
    iniPar = np.array([[2.75e13, 6e4, 1.0], [1.92e12, 6e4, 1.0]])
    # synthetic "measurements": the model at a state displaced from the initial guess + noise
    names = param_info["names"]
    truth = {**param_info["init_guess"], "tauN": 300.0, "tauP": 600.0, "Sf": 30.0, "mu_n": 12.0}
    state_true = np.array([truth[n] for n in names], dtype=float)
    un = np.array([param_info["unit_conversions"].get(n, 1) for n in names], dtype=float)
    idx = {n: i for i, n in enumerate(names)}
    rng = np.random.default_rng(7)
    vals, uncs = [], []
    for m in range(2):
        g = sim_utils.Grid(sim_info["lengths"][m], NX, times, 4)
        sig = fs.solve(iniPar[m].copy(), g, state_true.copy(), idx, meas="TRPL", units=un, ini_mode="fluence",
                       RTOL=1e-10, ATOL=1e-14)
        vals.append(np.log10(sig) + (3.0 if (irf and m == 0) else 0.0) + rng.normal(0, 0.02, len(times)))
        uncs.append(np.full(len(times), 0.02))
    irf_raw = synthetic_irf()
    tables = laplace.make_I_tables({745: irf_raw}) if irf else {}

    E = sim_utils.Ensemble(param_info, sim_info, MCMC_fields, num_iters)
    ef = E.ensemble_fields
    ef["_init_params"] = iniPar.copy()
    ef["_times"], ef["_vals"], ef["_uncs"] = [times.copy(), times.copy()], vals, uncs
    ef["_IRF_tables"] = tables
    RNG = np.random.default_rng(SEED)
    logger = logging.getLogger("golden")
    logger.addHandler(logging.NullHandler())
    logger.propagate = False
    t0 = time.time()
    states, logll, accept, att, acc = ref_metro.main_metro_loop_serial(
        E.H.states, E.H.loglikelihood, E.H.accept, 0, num_iters, ef, E.unique_fields, RNG, logger,
        need_initial_state=True)
    print(f"{name}: {time.time() - t0:.1f} s, accept rate {accept.mean():.2f}, swaps {acc.sum()}/{att.sum()}",
          flush=True)
    out[f"{name}__config"] = np.array(cfg_json)
    out[f"{name}__times"] = times
    out[f"{name}__iniPar"] = iniPar
    out[f"{name}__vals"] = np.array(vals)
    out[f"{name}__uncs"] = np.array(uncs)
    out[f"{name}__irf"] = irf_raw
    out[f"{name}__states"] = states
    out[f"{name}__logll"] = logll
    out[f"{name}__accept"] = accept
    out[f"{name}__swap_attempts"] = att
    out[f"{name}__swap_accept"] = acc


if __name__ == "__main__":
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.chdir(os.environ.get("TMPDIR", "/tmp"))
    out = {}
    run_case("single", 1, 40, False, False, out)
    run_case("pt", 3, 24, True, True, out)
    np.savez_compressed(os.path.join(HERE, "mcmc.npz"), **out)
    print("wrote", os.path.join(HERE, "mcmc.npz"))
