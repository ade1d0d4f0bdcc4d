"""The small helper functions of the reference's forward_solver / laplace modules (forward_solver.py:250-329,
laplace.py:129-167) against vectors produced by the reference itself (tests/golden/make_helpers_golden.py):
the oracle on the CPU, the drop-in modules on the GPU."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def h():
    return np.load(os.path.join(G, "helpers.npz"))


def test_oracle_I_moment_any_spacing(h):
    from oracle import oracle as O
    irf = h["irf"]
    for (m, n, lo, hi, nu), ref in zip(h["moment_cases"], h["moment_vals"]):
        assert (lo, hi) == (0, 1)
        got = O.I_moment(irf[:, 0], irf[:, 1], int(m), int(n), u_spacing=int(nu))
        assert got == pytest.approx(ref, rel=1e-13, abs=1e-300), (m, n, nu)


@pytest.mark.gpu
def test_I_moment_any_spacing(h):
    """laplace.I_moment with the reference's default u_spacing=100, odd and tiny sample counts (SciPy's
    Simpson rule has separate branches for them) and the 1000 make_I_tables uses."""
    from metrotrpl_amd import laplace
    irf = h["irf"]
    for (m, n, lo, hi, nu), ref in zip(h["moment_cases"], h["moment_vals"]):
        got = laplace.I_moment(irf[:, 0], irf[:, 1], int(m), int(n), u_lower=int(lo), u_upper=int(hi), u_spacing=int(nu))
        assert got == pytest.approx(ref, rel=1e-12, abs=1e-300), (m, n, nu, got, ref)
    assert laplace.I_moment(irf[:, 0], irf[:, 1], 5, 1) == pytest.approx(h["moment_vals"][1], rel=1e-12)  # default spacing
    # y[m+1-u_upper] with a negative index wraps around as numpy does
    for (m, n, lo, hi, nu), ref in zip(h["moment_wrap_cases"], h["moment_wrap_vals"]):
        got = laplace.I_moment(irf[:, 0], irf[:, 1], int(m), int(n), u_lower=int(lo), u_upper=int(hi), u_spacing=int(nu))
        assert got == pytest.approx(ref, rel=1e-12, abs=1e-300), (m, n, lo, hi, nu, got, ref)
    with pytest.raises(ValueError):
        laplace.I_moment(irf[:, 0], irf[:, 1], 500, 1)  # IndexError in the reference: lag beyond the IRF
    with pytest.raises(ValueError):
        laplace.I_moment(irf[:, 0], irf[:, 1], 0, 1, u_upper=500)  # ... and below -len(y)


@pytest.mark.gpu
def test_density_helpers_and_integrals(h):
    from metrotrpl_amd import forward_solver as fs
    N, P = h["N"], h["P"]
    rr = fs.calculate_RR(N, P, *h["args_rr"])
    np.testing.assert_allclose(rr, h["rr"], rtol=1e-14)
    np.testing.assert_allclose(fs.calculate_RR(N[1], P[1], *h["args_rr"]), h["rr"][1], rtol=1e-14)
    np.testing.assert_allclose(fs.calculate_photoc(N, P, *h["args_pc"]), h["photoc"], rtol=1e-14)
    np.testing.assert_allclose(fs.integrate_2D(float(h["dx"]), h["rr"]), h["int2d"], rtol=1e-13)
    assert fs.integrate_1D(float(h["dx"]), h["rr"][2]) == pytest.approx(float(h["int1d"]), rel=1e-13)
    with pytest.raises(ValueError):
        fs.integrate_2D(1.0, h["rr"][0])


@pytest.mark.gpu
def test_dydt_numpy_twin():
    """forward_solver.dydt (the reference's numpy twin of dydt_numba, :277-329): the reference's own output."""
    from metrotrpl_amd import forward_solver as fs
    z = np.load(os.path.join(G, "rhs.npz"))
    n = 0
    for k in range(int(z["n_cases"])):
        a = z[f"std{k}_args"]
        L = int(a[0])
        if L > 128:
            continue
        dy = fs.dydt(0.0, z[f"std{k}_y"], L, *a[1:])
        ref = z[f"std{k}_dy_np"]
        scale = np.max(np.abs(ref[:2 * L]))
        np.testing.assert_allclose(dy[:2 * L], ref[:2 * L], rtol=1e-9, atol=1e-12 * scale)
        np.testing.assert_allclose(dy[2 * L:], ref[2 * L:], rtol=1e-12, atol=1e-14 * np.max(np.abs(ref[2 * L:])))
        n += 1
    assert n >= 4
