"""Pin the CPU oracle (oracle/) against vectors captured from the reference
itself (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def rhs():
    return np.load(os.path.join(G, "rhs.npz"))


@pytest.fixture(scope="module")
def irf():
    return np.load(os.path.join(G, "irf.npz"))


def test_dydt_std_bit_exact(rhs):
    # forward_solver.py:332-372: same operation order -> same bits
    for k in range(int(rhs["n_cases"])):
        a = rhs[f"std{k}_args"]
        dy = O.dydt_std(0.0, rhs[f"std{k}_y"], int(a[0]), *a[1:])
        np.testing.assert_array_equal(dy, rhs[f"std{k}_dy"])
        # the numpy twin (forward_solver.py:277-329) groups mu*kB*T differently
        np.testing.assert_allclose(dy, rhs[f"std{k}_dy_np"], rtol=1e-9, atol=1e-18)


def test_dydt_traps_bit_exact(rhs):
    for k in range(int(rhs["n_cases"])):
        a = rhs[f"traps{k}_args"]
        dy = O.dydt_traps(0.0, rhs[f"traps{k}_y"], int(a[0]), *a[1:])
        np.testing.assert_array_equal(dy, rhs[f"traps{k}_dy"])


def test_E_field(rhs):
    n0, p0, eps, dx = rhs["ef_par"]
    np.testing.assert_array_equal(O.E_field(rhs["ef_N"], rhs["ef_P"], n0, p0, eps, dx), rhs["ef_E"])
    np.testing.assert_array_equal(O.E_field(rhs["ef_N"], rhs["ef_P"], n0, p0, eps, dx, corner_E=24),
                                  rhs["ef_E_corner"])
    np.testing.assert_array_equal(O.E_field(rhs["ef_N2"], rhs["ef_P2"], n0, p0, eps, dx), rhs["ef_E2"])
    with pytest.raises(NotImplementedError):
        O.E_field(np.zeros((2, 2, 2)), np.zeros((2, 2, 2)), 0, 0, 1, 1)


def test_integrals(rhs):
    assert O.integrate_1D(0.37, rhs["int_y"]) == float(rhs["int_1d"])
    dx = 2.4296875
    np.testing.assert_array_equal(O.calculate_PL(dx, rhs["ef_N2"], rhs["ef_P2"], 4.8e1, 1e-13, 3e-6),
                                  rhs["pl_2d"])
    np.testing.assert_array_equal(O.calculate_TRTS(dx, rhs["ef_N2"], rhs["ef_P2"], 2e6, 1.5e6, 1e-13, 3e-6),
                                  rhs["trts_2d"])
    assert O.calculate_PL(dx, rhs["ef_N2"][0], rhs["ef_P2"][0], 4.8e1, 1e-13, 3e-6) == float(rhs["pl_1d"])
    assert O.calculate_TRTS(dx, rhs["ef_N2"][0], rhs["ef_P2"][0], 2e6, 1.5e6, 1e-13, 3e-6) == float(rhs["trts_1d"])
    # Tests/test_integrals.py:10-22
    x = np.arange(0.005, 1, 0.01)
    np.testing.assert_almost_equal(O.integrate_1D(0.01, x), 0.5)


def test_I_tables(irf):
    for w in (520, 745):
        tab = O.make_I_tables({w: irf[f"irf{w}_raw"]})[w][0]
        # np.sum inside scipy.simpson is pairwise -> agreement to rounding
        np.testing.assert_allclose(tab, irf[f"irf{w}_table"], rtol=2e-13, atol=1e-13)
        assert np.all(tab[-1] == 0)
    t = np.linspace(0, 10, 1001)
    tab = O.make_I_tables({1: np.vstack((t, np.sin(t))).T})[1][0]
    np.testing.assert_allclose(tab, irf["sin_table"], rtol=1e-12, atol=1e-17)


def test_convolve_and_analytic(irf):
    h = O.convolve(irf["sin_f"], irf["sin_table"])
    np.testing.assert_allclose(h, irf["sin_conv"], rtol=1e-12, atol=1e-15)
    # Tests/test_convolution.py:81-99
    t = np.linspace(0, 10, 1001)
    np.testing.assert_almost_equal(h, 0.5 * (np.exp(-t) + np.sin(t) - np.cos(t)), decimal=5)
    ct, cy, ok = O.do_irf_convolution(t / 2, np.exp(-t / 2), (irf["sin_table"], t), time_max_shift=True)
    assert ok == bool(irf["sin_full_ok"])
    np.testing.assert_allclose(ct, irf["sin_full_t"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(cy, irf["sin_full_y"], rtol=1e-12, atol=1e-15)
    assert ct[np.argmax(cy)] == 0


@pytest.mark.parametrize("w,shift", [(745, 0), (745, 1), (520, 0), (520, 1)])
def test_do_irf_convolution_shipped(irf, w, shift):
    tag = f"pl{w}_{shift}"
    table = (irf[f"irf{w}_table"], irf[f"irf{w}_raw"][:, 0])
    ts, y = irf[tag + "_t"], irf[tag + "_y"]
    ct, cy, ok = O.do_irf_convolution(ts, y, table, time_max_shift=bool(shift))
    assert ok == bool(irf[tag + "_ok"])
    assert len(cy) == int(irf[tag + "_conv_len"])
    assert int(np.argmax(cy)) == int(irf[tag + "_conv_argmax"])
    np.testing.assert_allclose([ct[0], ct[-1]], irf[tag + "_conv_t_first_last"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(cy[:400], irf[tag + "_conv_head"], rtol=1e-12)
    np.testing.assert_allclose(cy[::97], irf[tag + "_conv_stride"], rtol=1e-12)
    np.testing.assert_allclose(np.sum(cy), float(irf[tag + "_conv_sum"]), rtol=1e-12)
    exp_y = np.log10(y) + 0.01
    ty, tt, tv, tu = O.post_conv_trim(ct, cy, ts, exp_y, np.full_like(ts, 0.02))
    assert len(tv) == int(irf[tag + "_trim_len"]) == len(ty) == len(tt) == len(tu)
    np.testing.assert_allclose(ty, irf[tag + "_trim_y"], rtol=1e-11)
    np.testing.assert_array_equal(tt, irf[tag + "_trim_t"])


def test_post_conv_trim(irf):
    t = np.linspace(0, 10, 1001)
    dbl = np.arange(0, 10 + (t[1] - t[0]) / 4, (t[1] - t[0]) / 2)
    conv_t = dbl[:801]
    y_c, times_c, vals_c, uncs_c = O.post_conv_trim(conv_t, np.arange(len(conv_t), dtype=float), t,
                                                    np.arange(len(t), dtype=float), np.arange(len(t)) * 0.1)
    np.testing.assert_allclose(y_c, irf["trim_yc"], rtol=1e-13)
    np.testing.assert_array_equal(times_c, irf["trim_tc"])
    assert len(y_c) == len(vals_c) == len(uncs_c)
    assert np.all(times_c < max(conv_t))
