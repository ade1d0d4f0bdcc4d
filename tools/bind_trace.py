"""tools/bind_trace.py -- which half of the goal-oriented error test sets the step size (dev tool, -DMTRPL_TRACE build:
field 6 of an attempt's record is rho = (test value) / (its RMS part): 1 / shape_factor where the SHAPE test binds, larger
where the MASS test does).  Per decade of time: accepted steps, share of them bound by the mass test, mean order, mean h / t,
mean test value of the accepted steps (1 = at the tolerance).
    MTRPL_LIB_PATH=metrotrpl_amd/lib/libmetrotrpl_amd_trace.so python tools/bind_trace.py [solves=..] [opt=value ...]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
ctx = L.default_context()
sel = [860, 467, 998, 631, 79, 6, 217]
for kv in sys.argv[1:]:
    if kv.startswith("solves="):
        sel = [int(x) for x in kv[7:].split(",")]
    elif "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, float(v))
sf = ctx.get_option("shape_factor")
lib = C.CDLL(L.LIB_PATH)
c = W.config2(1024)
ix, un = W.indexes(), W.units()
edges = [0, 1e-3, 1e-2, 0.1, 1, 2, 4, 10, 50, 200, 1000, 2001]
print("per decade of t: accepted steps / % bound by the mass test / mean order / mean h/t / mean test value")
print("t <       " + " ".join(f"{e:>22g}" for e in edges[1:]))
for k in sel:
    lib.mtrpl_debug_trace_select(C.c_int(0))
    one = [k]
    sig, status, st = batch.solve_batch(c["states"][one], ix, un, c["iniPar"][one], 2000.0, 128, c["tSteps"], return_stats=True, ctx=ctx)
    buf = np.zeros((8192, 8), dtype=np.float32); n = C.c_int(0)
    lib.mtrpl_debug_trace(buf.ctypes.data_as(C.c_void_p), C.byref(n))
    tr = buf[:n.value]
    code, t, h, order, nit, dsm, rho = tr[:, 0], tr[:, 1].astype(float), tr[:, 2].astype(float), tr[:, 3], tr[:, 4], tr[:, 5], tr[:, 6]
    acc = code == 0
    tend = t + h
    row = []
    for a, b in zip(edges[:-1], edges[1:]):
        ma = (tend > a) & (tend <= b) & acc
        if ma.sum():
            mass = np.mean(rho[ma] > 1.01 / sf)
            row.append(f"{ma.sum():4d}/{100 * mass:3.0f}%/{order[ma].mean():3.1f}/{np.mean(h[ma] / tend[ma]):5.3f}/{dsm[ma].mean():4.2f}")
        else:
            row.append("   0/  -/ - /  -  /  - ")
    s = c["states"][k]
    print(f"#{k:4d} steps {st[0,0]:5d} rej {st[0,4]:3d} fac {st[0,2]:3d} nit/step {nit[acc].mean():.2f} | tauN {s[ix['tauN']]:.0f} tauP {s[ix['tauP']]:.0f} Sf {s[ix['Sf']]:.1f} Sb {s[ix['Sb']]:.1f} mu {s[ix['mu_n']]:.0f}/{s[ix['mu_p']]:.0f} p0 {s[ix['p0']]:.1e} flu {c['iniPar'][k,0]:.1e}")
    print("          " + " ".join(f"{r:>22s}" for r in row))
