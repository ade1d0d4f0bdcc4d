"""GPU parity of the fused likelihood path (solve -> IRF -> trim -> ll(T)) against
(a) the reference's golden log-likelihoods (Tests/test_eval_trial_move.py set-ups,
captured at the reference's rtol=1e-5: its own pin is decimal=0) and (b) the CPU
oracle run at tight tolerance on this machine (gate: 1e-6 relative)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
TEMPS = [1.0, 3.0, 7.5, 1000.0]


def _setup(which):
    from tests.test_oracle_solve import _eval_setup
    ev = np.load(os.path.join(G, "eval.npz"))
    state, uniq, shared = _eval_setup(ev, which)
    return ev, state, uniq, shared


def _engine(state, uniq, shared):
    from metrotrpl_amd.likelihood import LikelihoodEngine
    from metrotrpl_amd import laplace
    si = shared["_sim_info"]
    irf_tables, irf_conv = None, shared.get("irf_convolution")
    if irf_conv is not None:
        raw = np.load(os.path.join(G, "irf.npz"))["irf745_raw"]
        irf_tables = laplace.make_I_tables({745: raw})  # built on the device too
    scale_shift = None
    fs = shared.get("scale_factor")
    if fs is not None:
        pidx = shared["_param_indexes"]
        scale_shift = []
        for m in range(si["num_meas"]):
            g0 = m
            for grp in fs[2]:
                if m in grp:
                    g0 = grp[0]
            scale_shift.append(np.log10(state[pidx[f"_s{g0}"]]))
    return LikelihoodEngine(shared["_param_indexes"], shared["units"], si["lengths"], si["nx"][0], si["meas_types"],
                            shared["_times"], shared["_vals"], shared["_uncs"], uniq["model_uncertainty"],
                            shared["_init_params"], ini_mode=shared["ini_mode"], hmax=shared["hmax"],
                            rtol=shared["rtol"], atol=shared["atol"], irf_tables=irf_tables,
                            irf_convolution=irf_conv, scale_shift=scale_shift)


@pytest.mark.parametrize("which", ["base", "cutoff", "scale", "mixed", "irf"])
def test_loglik_matches_reference_and_oracle(which):
    from oracle import oracle as O
    ev, state, uniq, shared = _setup(which)
    eng = _engine(state, uniq, shared)
    table = eng.eval(np.asarray(state)[None, :], TEMPS)[0]  # [M, nT]
    # (a) the reference's golden values (rtol=1e-5 LSODA): its own tests pin decimal=0
    np.testing.assert_allclose(table[:, 0].sum(), float(ev[which + "__logll_T1"]), rtol=2e-5, atol=1.5)
    np.testing.assert_allclose(table[:, 1], ev[which + "__ll_meas_T3"], rtol=2e-5, atol=1.0)
    np.testing.assert_allclose(table[:, 2].sum(), float(ev[which + "__logll_T7p5"]), rtol=2e-5, atol=0.5)
    # (b) oracle at tight tolerance, every temperature: 1e-6 relative
    tight = dict(shared, rtol=1e-10, atol=1e-14)
    _, ref = O.eval_trial_move(state, dict(uniq, _T=1.0), tight, temperatures=TEMPS)
    scale = np.max(np.abs(ref))
    # atol: n * (d log10)^2 with d log10 ~ 4e-8 (1e-7 relative on the signal) ~ 2e-12 -- only the
    # 'scale' set-up (ll ~ 1e-20, i.e. a perfect fit) is anywhere near it
    np.testing.assert_allclose(table, ref, rtol=1e-6, atol=1e-9 * scale + 1e-10)


def test_loglik_batch_of_chains_and_failures():
    """Many chains at once == one at a time == the oracle; then the failure paths of
    trial_move_evaluation.py:117-123,:159-165 on hand-made signals."""
    import ctypes as C
    import torch
    from metrotrpl_amd import _lib as L
    from oracle import oracle as O
    ev, state, uniq, shared = _setup("irf")
    eng = _engine(state, uniq, shared)
    state = np.asarray(state, dtype=float)
    pidx = shared["_param_indexes"]
    states = np.stack([state] * 4)
    states[1, pidx["ks"]] *= 3.0
    states[2, pidx["ks"]] *= -1.0   # negative PL: clamped to min_y from the first sample on (forward_solver.py:190-192)
    states[3, pidx["ks"]] *= 0.5
    t_all = eng.eval(states, TEMPS)
    tight = dict(shared, rtol=1e-10, atol=1e-14)
    for c in range(4):
        one = eng.eval(states[c:c + 1], TEMPS)[0]
        np.testing.assert_array_equal(one, t_all[c])
        assert np.all(np.isfinite(one))
        if c == 2:
            # constant (clamped) signal: the convolved trace is a plateau of mathematically equal
            # samples, so np.argmax -- and with it the time shift -- is decided by summation
            # round-off (numpy's pairwise sums, the oracle's sequential sums and the kernel's
            # prefix sums all differ); only finiteness is meaningful here
            continue
        _, ref = O.eval_trial_move(states[c], dict(uniq, _T=1.0), tight, temperatures=TEMPS)
        np.testing.assert_allclose(t_all[c], ref, rtol=1e-6)
    # ll(T) is increasing in T (denominator grows), and ll(T) != ll(1)/T because uncs > 0
    assert np.all(np.diff(t_all[0], axis=1) > 0)
    assert not np.allclose(t_all[0][:, 1], t_all[0][:, 0] / 3.0, rtol=1e-5)

    # ---- failure paths, no-IRF engine, signals written by hand into the device buffer ----
    ev, state, uniq, shared = _setup("base")
    eng = _engine(state, uniq, shared)
    eng.upload(np.stack([np.asarray(state, dtype=float)] * 3))
    T = torch.tensor(TEMPS, dtype=torch.float64, device=eng.dev)
    eng.launch(T, len(TEMPS))
    torch.cuda.synchronize()
    nt = eng.nts[0]
    sig = eng.d_sig.cpu().numpy().reshape(3, 2, nt).copy()
    good = sig[0].copy()
    sig[0, 0, :nt // 4] *= -1          # 25 % negative  -> -inf   (NEGATIVE_FRAC_TOL = 0.2)
    sig[1, 0, :nt // 10] *= -1         # 10 % negative  -> flipped, finite, same value as the positive trace
    sig[2, 1, 5] = np.nan              # NaN likelihood -> -inf
    eng.d_sig.copy_(torch.from_numpy(sig.reshape(-1)))
    eng.ctx.check(L.lib().mtrpl_loglik_batch_dev(eng.ctx.handle, eng.n_solves, eng.d_like.data_ptr(),
                                                 eng.d_sig.data_ptr(), eng.d_t.data_ptr(), eng.d_exp_t.data_ptr(),
                                                 eng.d_exp_y.data_ptr(), eng.d_exp_u.data_ptr(), T.data_ptr(),
                                                 len(TEMPS), eng.max_nt, eng.d_ll.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ll = eng.d_ll.cpu().numpy().reshape(3, 2, len(TEMPS))
    assert np.all(np.isneginf(ll[0, 0])) and np.all(np.isfinite(ll[0, 1]))
    assert np.all(np.isneginf(ll[2, 1])) and np.all(np.isfinite(ll[2, 0]))
    ref = O.loglik_tail(good[0], shared["_vals"][0], shared["_uncs"][0], 0.0, 1.0, TEMPS)
    np.testing.assert_allclose(ll[1, 0], ref, rtol=1e-12)


@pytest.mark.parametrize("kind", ["power", "jitter", "coarse"])
def test_loglik_irf_on_unaligned_time_grids(kind):
    """The prefix-sum form of the FIR must agree with the definition when the simulation
    samples do NOT fall on the IRF half-step grid (intervals straddling a sample are
    evaluated directly) and when the grid is non-uniform."""
    from oracle import oracle as O
    ev, state, uniq, shared = _setup("irf")
    rng = np.random.default_rng(5)
    if kind == "power":
        t = 100.0 * np.linspace(0, 1, 701) ** 1.3
    elif kind == "jitter":
        t = np.linspace(0, 100, 1001)
        t[1:-1] += rng.uniform(-0.03, 0.03, 999)
    else:
        t = np.linspace(0, 60.0, 41)  # 1.5 ns steps: not a multiple of 0.025/2 * 2
        t[1:] += 0.00371
    n = len(t)
    shared = dict(shared, _times=[t, t], _vals=[23 - np.linspace(0, 1, n), 24 - np.linspace(0, 2, n)],
                  _uncs=[np.full(n, 0.02), np.full(n, 0.02)])
    eng = _engine(state, uniq, shared)
    table = eng.eval(np.asarray(state)[None, :], TEMPS)[0]
    tight = dict(shared, rtol=1e-10, atol=1e-14)
    _, ref = O.eval_trial_move(state, dict(uniq, _T=1.0), tight, temperatures=TEMPS)
    assert np.all(np.isfinite(ref))
    np.testing.assert_allclose(table, ref, rtol=1e-6)


@pytest.mark.parametrize("n,t_end", [(2501, 250.0), (4001, 200.0), (6001, 150.0)])
def test_loglik_long_traces(n, t_end):
    """Real TRPL files run to thousands of samples and the reference has no limit
    (trial_move_evaluation.py:96-158).  2501 samples: per-sample arrays in LDS beyond the 64 KB default
    (one block per CU); 4001 and 6001: per-block work space in device memory.  With IRF convolution, against
    the oracle's definition-order sums; the no-IRF case of the same traces as well."""
    from oracle import oracle as O
    ev, state, uniq, shared = _setup("irf")
    t = np.linspace(0, t_end, n)
    shared = dict(shared, _times=[t, t], _vals=[23 - np.linspace(0, 1, n), 24 - np.linspace(0, 2, n)],
                  _uncs=[np.full(n, 0.02), np.full(n, 0.02)])
    for irf_on in (True, False):
        sh = dict(shared) if irf_on else dict(shared, irf_convolution=None)
        eng = _engine(state, uniq, sh)
        table = eng.eval(np.asarray(state)[None, :], TEMPS)[0]
        tight = dict(sh, rtol=1e-10, atol=1e-14)
        _, ref = O.eval_trial_move(state, dict(uniq, _T=1.0), tight, temperatures=TEMPS)
        assert np.all(np.isfinite(ref))
        np.testing.assert_allclose(table, ref, rtol=1e-6)
    # several chains at once through the strided-block variant: same values as one at a time
    eng = _engine(state, uniq, shared)
    states = np.stack([np.asarray(state, dtype=float)] * 3)
    states[1, shared["_param_indexes"]["ks"]] *= 2.0
    all3 = eng.eval(states, TEMPS)
    for c in range(3):
        np.testing.assert_array_equal(eng.eval(states[c:c + 1], TEMPS)[0], all3[c])


def test_loglik_argmax_of_hand_made_traces():
    """The fused kernel stops its argmax walk over the convolved trace once no later output can still win
    (bound from the remaining samples).  Hand-made signals whose maximum is NOT where a decay has it --
    late peak, two equal peaks, a ramp that tops out at the very end, a second peak one part in 10^9 taller,
    a fast decay, a plateau followed by a late spike -- against the oracle's convolution, argmax shift,
    trim and likelihood tail (laplace.py:44-126, trial_move_evaluation.py:117-165)."""
    import torch
    from metrotrpl_amd import _lib as L
    from oracle import oracle as O
    ev, state, uniq, shared = _setup("irf")
    eng = _engine(state, uniq, shared)
    n_tr = 7
    M = eng.M
    n_chains = -(-n_tr // M)
    eng.upload(np.stack([np.asarray(state, dtype=float)] * n_chains))
    T = torch.tensor(TEMPS, dtype=torch.float64, device=eng.dev)
    eng.launch(T, len(TEMPS))
    torch.cuda.synchronize()
    nt = eng.nts[0]
    assert all(n == nt for n in eng.nts)
    t = np.asarray(shared["_times"][0], dtype=float)
    span = t[-1]
    bump = lambda c, w: np.exp(-((t - c) / w) ** 2)  # noqa: E731
    traces = [
        1e20 * (bump(0.7 * span, 0.05 * span) + 0.01),                       # late peak
        1e20 * (bump(0.25 * span, 0.02 * span) + bump(0.75 * span, 0.02 * span) + 1e-3),   # twin peaks: first wins
        1e18 * (1.0 + t / span),                                              # ramp: argmax at the end region
        1e20 * (bump(0.2 * span, 0.03 * span) + (1 + 1e-9) * bump(0.8 * span, 0.03 * span) + 1e-3),  # second is taller
        1e21 * np.exp(-t / (0.01 * span)) + 1e12,                             # fast decay: the walk stops early
        1e19 * (1.0 + 50.0 * bump(0.97 * span, 0.004 * span)),                # plateau, spike near the end
        1e20 * np.exp(-t / (0.5 * span)),                                     # slow decay
    ]
    sig = eng.d_sig.cpu().numpy().reshape(n_chains * M, nt).copy()
    for k, y in enumerate(traces):
        sig[k] = y
    eng.d_sig.copy_(torch.from_numpy(sig.reshape(-1)))
    eng.ctx.check(L.lib().mtrpl_loglik_batch_dev(eng.ctx.handle, eng.n_solves, eng.d_like.data_ptr(),
                                                 eng.d_sig.data_ptr(), eng.d_t.data_ptr(), eng.d_exp_t.data_ptr(),
                                                 eng.d_exp_y.data_ptr(), eng.d_exp_u.data_ptr(), T.data_ptr(),
                                                 len(TEMPS), eng.max_nt, eng.d_ll.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    ll = eng.d_ll.cpu().numpy().reshape(n_chains * M, len(TEMPS))
    raw = np.load(os.path.join(G, "irf.npz"))["irf745_raw"]
    tab = O.make_I_tables({745: raw})[745]
    shifts = []
    for k, y in enumerate(traces):
        m = k % M
        ct, cy, ok = O.do_irf_convolution(t, y, tab, time_max_shift=True)
        shifts.append(-ct[0])
        ev_t = np.asarray(shared["_times"][m], dtype=float)
        try:
            sol, tc, vc, uc = O.post_conv_trim(ct, cy, ev_t, np.asarray(shared["_vals"][m]), np.asarray(shared["_uncs"][m]))
            ref = O.loglik_tail(sol, vc, uc, 0.0, uniq["model_uncertainty"]["TRPL"], TEMPS)
        except IndexError:  # nothing left of the experimental trace: the reference raises, the kernel says -inf
            ok = False
        if not ok:
            ref = np.full(len(TEMPS), -np.inf)
        if np.all(np.isfinite(ref)):
            np.testing.assert_allclose(ll[k], ref, rtol=1e-9, err_msg=f"trace {k}")
        else:
            np.testing.assert_array_equal(ll[k], ref, err_msg=f"trace {k}")
    # the cases really differ in where the maximum sits: early, middle, end
    assert shifts[4] < 0.01 * span and 0.6 * span < shifts[0] < 0.8 * span and 0.2 * span < shifts[1] < 0.3 * span
    assert shifts[3] > 0.7 * span and shifts[5] > 0.9 * span


@pytest.mark.parametrize("mode", ["density", "fluence"])
def test_fittable_fluence_and_absorption_factors(mode):
    """trial_move_evaluation.py:38-52: the fittable factors multiply elements 0 / 1 of the init-params row in both
    ini_modes -- in density mode the densities of nodes 0 and 1 (the engine dropped them there).
    Against the reference's golden log-likelihoods (LSODA at rtol 1e-7: 2e-6) and the tight oracle (1e-6)."""
    from oracle import oracle as O
    from tests.test_oracle_solve import _fittable_setup
    from metrotrpl_amd.likelihood import engine_from_fields
    ev, state, uniq, shared = _fittable_setup(mode)
    eng = engine_from_fields(shared, uniq)
    T = [1.0, 3.0, 7.5]
    states = np.stack([state, state])
    pidx = shared["_param_indexes"]
    states[1, pidx["_f0"]] = 1.0   # second chain: no change of measurement 0's first element
    states[1, pidx["_a0"]] = 0.8
    table = eng.eval(states, T)
    np.testing.assert_allclose(table[0], ev[mode + "__ll_meas"], rtol=3e-6)
    tight = dict(shared, rtol=1e-10, atol=1e-14)
    for c in range(2):
        _, ref = O.eval_trial_move(states[c], uniq, tight, temperatures=T)
        np.testing.assert_allclose(table[c], ref, rtol=1e-6)
    assert not np.allclose(table[0], table[1], rtol=1e-3)
    # the engine scales a copy: the caller's rows are untouched and a second evaluation gives the same table
    np.testing.assert_array_equal(shared["_init_params"], ev[mode + "__ini"])
    np.testing.assert_array_equal(eng.eval(states, T), table)
