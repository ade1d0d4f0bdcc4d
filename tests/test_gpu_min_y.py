"""GPU: force_min_y (utils.set_min_y, trial_move_evaluation.py:136-140) in the fused
likelihood kernel -- the reference's carrier-depletion test
(Tests/test_eval_trial_move.py:82-143) and parity with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(tau):
    names = ["n0", "p0", "mu_n", "mu_p", "ks", "Cn", "Cp", "Tm", "Sf", "Sb", "tauN", "tauP", "eps", "m"]
    uc = {"n0": ((1e-7) ** 3), "p0": ((1e-7) ** 3), "mu_n": ((1e7) ** 2) / (1e9), "mu_p": ((1e7) ** 2) / (1e9),
          "ks": ((1e7) ** 3) / (1e9), "Sf": 1e-2, "Sb": 1e-2}
    guess = {"n0": 1e8, "p0": 1e17, "mu_n": 0, "mu_p": 0, "ks": 1e-13, "Sf": 0, "Sb": 0, "Cn": 0, "Cp": 0,
             "Tm": 300, "tauN": tau, "tauP": tau, "eps": 10, "m": 1}
    idx = {n: names.index(n) for n in names}
    state = np.array([guess[n] for n in names], dtype=float)
    units = np.array([uc.get(n, 1) for n in names], dtype=float)
    nt = 1000
    times = [np.linspace(0, 100, nt + 1)]
    vals = [np.log10(2e14 * np.exp(-times[0] / 8))]
    uncs = [np.ones(nt + 1) * 1e-99]
    shared = {"units": units, "solver": ("solveivp",), "model": "std", "hmax": 4, "rtol": 1e-10, "atol": 1e-14,
              "force_min_y": True, "_sim_info": {"lengths": [2000], "nx": [128], "meas_types": ["TRPL"], "num_meas": 1},
              "_param_indexes": idx, "_init_params": np.array([1e15 * np.ones(128)]), "ini_mode": "density",
              "_times": times, "_vals": vals, "_uncs": uncs}
    return state, {"model_uncertainty": {"TRPL": 1}}, shared


def test_depletion_with_force_min_y():
    from metrotrpl_amd.likelihood import LikelihoodEngine
    from oracle import oracle as O
    out = []
    for tau in (4.0, 4.01):
        state, uniq, shared = _setup(tau)
        si = shared["_sim_info"]
        eng = LikelihoodEngine(shared["_param_indexes"], shared["units"], si["lengths"], 128, si["meas_types"],
                               shared["_times"], shared["_vals"], shared["_uncs"], uniq["model_uncertainty"],
                               shared["_init_params"], ini_mode="density", force_min_y=True)
        ll = eng.eval(state[None, :], [1.0, 5.0])[0]          # [M=1, nT=2]
        _, ref = O.eval_trial_move(state, dict(uniq, _T=1.0), shared, temperatures=[1.0, 5.0])
        assert np.all(np.isfinite(ll))
        # the clamp index sits where the trace crosses min_y: a 1e-7 relative difference in the
        # trace can move it by one sample (measured: 6e-5 relative on the sum)
        np.testing.assert_allclose(ll, ref, rtol=1e-4)
        out.append(ll[0, 0])
        eng_off = LikelihoodEngine(shared["_param_indexes"], shared["units"], si["lengths"], 128, si["meas_types"],
                                   shared["_times"], shared["_vals"], shared["_uncs"], uniq["model_uncertainty"],
                                   shared["_init_params"], ini_mode="density", force_min_y=False)
        off = eng_off.eval(state[None, :], [1.0])[0, 0, 0]
        assert off < ll[0, 0]  # without the clamp the depleted tail dominates the misfit
    assert out[1] > out[0]     # Tests/test_eval_trial_move.py:143
