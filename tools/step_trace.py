"""Per-attempt trace (accepted / rejected / Newton failure: t, h, order, Newton iterations, error-test value) of ONE solve
of a small config-2 batch, from a diagnostic build (dev tool; the product library has no trace):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DMTRPL_TRACE=<solve index> -o /tmp/libtrace.so metrotrpl_amd/csrc/mtrpl_capi.hip
    MTRPL_LIB_PATH=/tmp/libtrace.so python tools/step_trace.py out.npy
This is how the permlane-swap reduction bug of round 3 was found (old and new integrator traced side by side)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
L.SIGNATURES = dict(L.SIGNATURES)
ctx = L.default_context()
NB = int(os.environ.get("TRACE_BATCH", "8"))
c = W.config2(NB)
sig, status, st = batch.solve_batch(c["states"], W.indexes(), W.units(), c["iniPar"], 2000.0, 128, c["tSteps"], return_stats=True, ctx=ctx)
lib = C.CDLL(L.LIB_PATH)
buf = np.zeros((8192, 8), dtype=np.float32); n = C.c_int(0)
lib.mtrpl_debug_trace(buf.ctypes.data_as(C.c_void_p), C.byref(n))
print("status", status[:8], "steps", st[:8, 0], "trace rows", n.value, "batch", NB)
np.save(sys.argv[1], buf[:n.value])
