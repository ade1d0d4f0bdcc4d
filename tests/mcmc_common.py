"""Shared set-up for the Metropolis-walk parity tests: rebuild the ensemble of a
golden case (tests/golden/mcmc.npz, produced by the reference's serial driver --
see tests/golden/make_mcmc_golden.py) with this package's own Ensemble."""
import json
import os

import numpy as np

from metrotrpl_amd.sim_utils import Ensemble

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name, make_I_tables):
    z = np.load(os.path.join(GOLDEN, "mcmc.npz"))
    cfg = json.loads(str(z[f"{name}__config"]))
    pi, si, mf = cfg["param_info"], cfg["sim_info"], cfg["MCMC_fields"]
    pi["prior_dist"] = {k: tuple(v) for k, v in pi["prior_dist"].items()}
    mf["solver"] = tuple(mf["solver"])
    num_iters = mf["num_iters"]
    E = Ensemble(pi, si, mf, num_iters)
    ef = E.ensemble_fields
    times = z[f"{name}__times"]
    ef["_init_params"] = z[f"{name}__iniPar"].copy()
    ef["_times"] = [times.copy(), times.copy()]
    ef["_vals"] = [v for v in z[f"{name}__vals"]]
    ef["_uncs"] = [u for u in z[f"{name}__uncs"]]
    ef["_IRF_tables"] = make_I_tables({745: z[f"{name}__irf"]}) if ef.get("irf_convolution") is not None else {}
    ref = {k: z[f"{name}__{k}"] for k in ("states", "logll", "accept", "swap_attempts", "swap_accept")}
    return E, num_iters, ref


def check_walk(E, walk, ref, rtol_ll):
    H = E.H
    # proposals are RNG-driven: as long as every accept/swap decision agrees the walks coincide -- up to
    # the 1-ulp drift the reference's solve() leaves in its stored states by scaling `state` with
    # `units` in place and dividing back (forward_solver.py:119,:188; SURVEY 8b: the drop-in leaves
    # the caller's state bit-unchanged)
    np.testing.assert_array_equal(H.accept, ref["accept"])
    np.testing.assert_array_equal(walk.swap_attempts, ref["swap_attempts"])
    np.testing.assert_array_equal(walk.swap_accept, ref["swap_accept"])
    np.testing.assert_allclose(H.states, ref["states"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(H.loglikelihood, ref["logll"], rtol=rtol_ll)
