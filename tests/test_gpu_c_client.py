"""The drop-in boundary from a host that is neither Python nor PyTorch: tests/c_client/solve_client.c is compiled with
gcc against include/metrotrpl_amd.h, linked to the in-tree shared library and run as its own process; its traces are
the Python path's traces bit for bit (both go through mtrpl_solve_batch).  What a cgo / JNI / FFI binding of the
reference's maintainers would do."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_client_gets_the_same_traces(tmp_path):
    from metrotrpl_amd import _lib as L, batch, workloads as W
    c = W.config2(24)
    thickness = np.where(np.arange(24) % 2 == 0, 2000.0, 311.0)
    t = np.ascontiguousarray(c["tSteps"][:301])
    descs, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], thickness, 128, len(t),
                                   meas=["TRPL", "TRTS"] * 12)
    sig, status, _ = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], thickness, 128, t,
                                       meas=["TRPL", "TRTS"] * 12, return_stats=True)
    exe = str(tmp_path / "solve_client")
    libdir = os.path.dirname(L.LIB_PATH)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), "-o", exe,
                           os.path.join(REPO, "tests", "c_client", "solve_client.c"), "-L", libdir, "-lmetrotrpl_amd",
                           "-Wl,-rpath," + libdir])
    fin, fout = str(tmp_path / "batch.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        np.array([len(descs), len(t)], dtype=np.int64).tofile(f)
        descs.tofile(f)
        t.tofile(f)
    # (a C caller exports the hardware-queue setting itself: the library never touches the environment)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    assert "24 solves" in r.stdout and "bad-argument call -> 1" in r.stdout, r.stdout
    raw = np.fromfile(fout, dtype=np.uint8)
    st_c = raw[:4 * 24].view(np.int32)
    sig_c = raw[4 * 24:].view(np.float64).reshape(24, len(t))
    np.testing.assert_array_equal(st_c, status)
    np.testing.assert_array_equal(sig_c, sig)
    # ... and through the pipelined host path: a pinned output buffer from mtrpl_host_alloc, three sub-launches of 8 solves
    fout2 = str(tmp_path / "out2.bin")
    r = subprocess.run([exe, fin, fout2, "8"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr[-500:])
    np.testing.assert_array_equal(np.fromfile(fout2, dtype=np.uint8), raw)
