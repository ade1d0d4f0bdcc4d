"""Where the steps of a solve go along the time axis (dev tool): attempt-by-attempt traces of several solves of the
config-2 batch from ONE diagnostic build with a run-time selectable traced solve:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMTRPL_TRACE=0 -o /tmp/libtrace.so metrotrpl_amd/csrc/mtrpl_capi.hip
    MTRPL_LIB_PATH=/tmp/libtrace.so python tools/startup_trace.py [out.npz] [opt=value ...] [solves=860,467,998,631,79]
Per solve: accepted steps, rejections, mean order and mean h/t per decade of time."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
sel = [860, 467, 998, 631, 79, 6, 217]
out = None
for kv in sys.argv[1:]:
    if kv.startswith("solves="):
        sel = [int(x) for x in kv[7:].split(",")]
    elif "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, float(v))
    else:
        out = kv
lib = C.CDLL(L.LIB_PATH)
c = W.config2(1024)
ix, un = W.indexes(), W.units()
edges = [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 0.1, 0.5, 1, 2, 4, 10, 50, 200, 1000, 2001]
keep = {}
print("columns per decade of t: accepted steps / rejected+failed attempts / mean order / mean h/t (accepted)")
print("t <       " + " ".join(f"{e:>15g}" for e in edges[1:]))
for k in sel:
    lib.mtrpl_debug_trace_select(C.c_int(0))
    one = [k]
    sig, status, st = batch.solve_batch(c["states"][one], ix, un, c["iniPar"][one], 2000.0, 128, c["tSteps"], return_stats=True, ctx=ctx)
    buf = np.zeros((8192, 8), dtype=np.float32); n = C.c_int(0)
    lib.mtrpl_debug_trace(buf.ctypes.data_as(C.c_void_p), C.byref(n))
    tr = buf[:n.value]
    keep[f"s{k}"] = tr
    code, t, h, order, nit = tr[:, 0], tr[:, 1].astype(float), tr[:, 2].astype(float), tr[:, 3], tr[:, 4]
    acc = code == 0
    tend = t + h  # t is the time BEFORE the attempt (accepted rows are traced before t += h)
    row = []
    for a, b in zip(edges[:-1], edges[1:]):
        m = (tend > a) & (tend <= b)
        ma = m & acc
        if ma.sum():
            row.append(f"{ma.sum():4d}/{(m & ~acc).sum():2d}/{order[ma].mean():3.1f}/{np.mean(h[ma] / tend[ma]):5.3f}")
        else:
            row.append(f"{0:4d}/{(m & ~acc).sum():2d}/ - /  -  ")
    s = c["states"][k]
    print(f"#{k:4d} steps {st[0,0]:5d} rej {st[0,4]:3d} h0 {h[0]:.1e} nit/step {nit[acc].mean():.2f} | tauN {s[ix['tauN']]:.0f} tauP {s[ix['tauP']]:.0f} Sf {s[ix['Sf']]:.1f} Sb {s[ix['Sb']]:.1f} mu {s[ix['mu_n']]:.0f}/{s[ix['mu_p']]:.0f} p0 {s[ix['p0']]:.1e} flu {c['iniPar'][k,0]:.1e}")
    print("          " + " ".join(f"{r:>15s}" for r in row))
    # the order ramp: attempt index and t at which each order is first reached
    ramp = []
    for q in range(2, 6):
        w = np.nonzero(order >= q)[0]
        ramp.append(f"q{q}@{w[0]}:{t[w[0]]:.1e}" if len(w) else f"q{q}@-")
    print("          order ramp (attempt:t) " + " ".join(ramp))
if out:
    np.savez(out, **keep)
