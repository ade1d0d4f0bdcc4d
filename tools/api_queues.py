"""tools/api_queues.py -- does the pipelined host path keep its rate inside a process that already owns other streams?
usage: GPU_MAX_HW_QUEUES=n python tools/api_queues.py n_torch_streams n_contexts"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch  # noqa: E402
from metrotrpl_amd import _lib as L, batch, workloads as W  # noqa: E402

n_streams, n_ctx = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
x = torch.zeros(1 << 20, device=dev)
side = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]
for s in side:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
others = [L.Context(0) for _ in range(n_ctx)]
B = 8192
ctx = L.Context(0)
c = W.config2(B)
t = np.ascontiguousarray(c["tSteps"])
nt = t.shape[0]
d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], c["nx"], nt, hmax=c["hmax"],
                           meas=c["meas"], model=c["model"], ini_mode=c["ini_mode"])
status = np.zeros(B, dtype=np.int32)
buf = L.pinned_f64(ctx, B * nt)
for o in others:  # their streams have been used
    small = np.empty(4 * nt)
    st4 = np.zeros(4, dtype=np.int32)
    o.check(L.lib().mtrpl_solve_batch(o.handle, 4, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0], L.dptr(t), nt, L.dptr(small),
                                      small.shape[0], L.iptr(st4), None))


def call():
    ctx.check(L.lib().mtrpl_solve_batch(ctx.handle, B, d.ctypes.data_as(C.c_void_p), L.dptr(ini), ini.shape[0], L.dptr(t), nt,
                                        L.dptr(buf), buf.shape[0], L.iptr(status), None))


for streams, chunk in ((0, 1024), (4, 1024), (4, 2048), (2, 2048)):
    ctx.set_option("host_streams", streams)
    ctx.set_option("host_chunk", chunk)
    call()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        call()
        ts.append(time.perf_counter() - t0)
    print(f"queues {os.environ.get('GPU_MAX_HW_QUEUES')} torch streams {n_streams} contexts {n_ctx} | streams {streams} chunk {chunk:5d}: "
          f"{min(ts) * 1e3:7.2f} ms  {B / min(ts):9.0f} solves/s", flush=True)
