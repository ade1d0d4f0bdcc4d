"""tools/api_sweep.py -- mtrpl_solve_batch host to host at B = 8192 (config 2) over host_chunk x host_streams."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = L.Context(0)
c = W.config2(B)
t = np.ascontiguousarray(c["tSteps"])
nt = t.shape[0]
d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], nt, hmax=c["hmax"],
                           meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
status = np.zeros(B, dtype=np.int32)
buf = L.pinned_f64(ctx, B * nt)


def call():
    ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, B, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0], L.dptr(t), nt,
                                        L.dptr(buf), buf.shape[0], L.iptr(status), None))


for streams, chunk in ((0, 1024), (4, 1024), (4, 512), (4, 2048), (3, 1024), (6, 1024), (8, 1024), (8, 512), (6, 512), (2, 2048),
                       (2, 4096), (4, 1365)):
    ctx.set_option("host_streams", streams)
    ctx.set_option("host_chunk", chunk)
    call()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    print(f"B {B} streams {streams} chunk {chunk:5d}: {min(ts) * 1e3:7.2f} ms  {B / min(ts):9.0f} solves/s  (mean {np.mean(ts) * 1e3:.2f} ms)", flush=True)
