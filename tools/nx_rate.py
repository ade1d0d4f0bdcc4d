"""dev: solves per second of ONE launch of B config-5-style solves (thick / thin film, TRPL / TRTS) at a given nx:
    python tools/nx_rate.py nx B [opt=value ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metrotrpl_amd import _lib as L, batch, workloads as W
nx, B = int(sys.argv[1]), int(sys.argv[2])
ctx = L.Context(0)
for kv in sys.argv[3:]:
    k, v = kv.split("="); ctx.set_option(k, float(v))
dev = torch.device("cuda", 0)
c = W.config5(B // 4, nx)
t = np.ascontiguousarray(c["tSteps"]); nt = len(t)
d, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], nx, nt, meas=c["meas"])
dd = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev); di = torch.from_numpy(ini).to(dev); dt = torch.from_numpy(t).to(dev)
out = torch.zeros(B * nt, dtype=torch.float64, device=dev); st = torch.zeros(B, dtype=torch.int32, device=dev); sa = torch.zeros(B * 8, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream()
def launch():
    ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, B, dd.data_ptr(), di.data_ptr(), dt.data_ptr(), out.data_ptr(), st.data_ptr(), sa.data_ptr(), nx, 0, s.cuda_stream))
launch(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): launch()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
stats = sa.cpu().numpy().reshape(B, 8)
print(f"nx {nx} B {B}: {ms:.2f} ms per launch = {B / ms:.1f} k solves/s | ok {(st == 0).all().item()} steps {stats[:,0].mean():.0f}/{stats[:,0].max()}")
