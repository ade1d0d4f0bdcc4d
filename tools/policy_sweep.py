"""One line per integrator-option setting: launch time and step counts of the config-2 batch, worst relative error
per decade group on the converged traces of tests/golden/tail_decades.npz (dev tool).
usage: python tools/policy_sweep.py "early_factor=30" "early_factor=30,e_weight=10" ..."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metrotrpl_amd import _lib as L, batch, workloads as W

ctx = L.Context(0)
dev = torch.device("cuda", 0)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tail_decades.npz"))
idx = {n: i for i, n in enumerate(str(z["names"]).split(","))}
ref = z["sig"]
dec = np.floor(np.log10(ref[:, :1]) - np.log10(ref)).astype(int)
c = W.config2(1024)
t_host = np.ascontiguousarray(c["tSteps"])
nt = len(t_host)
descs, ini = batch.build_batch(c["states"], W.indexes(), W.units(), c["iniPar"], c["thickness"], 128, nt)
d_descs = torch.from_numpy(descs.view(np.uint8).reshape(-1)).to(dev)
d_ini, d_t = torch.from_numpy(ini).to(dev), torch.from_numpy(t_host).to(dev)
d_out = torch.zeros(1024 * nt, dtype=torch.float64, device=dev)
d_status = torch.zeros(1024, dtype=torch.int32, device=dev)
d_stats = torch.zeros(1024 * 8, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream()


def launch():
    ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, 1024, d_descs.data_ptr(), d_ini.data_ptr(), d_t.data_ptr(), d_out.data_ptr(),
                                            d_status.data_ptr(), d_stats.data_ptr(), 128, 0, s.cuda_stream))


for setting in ["shipped"] + sys.argv[1:]:
    saved = {}
    if setting != "shipped":
        for kv in setting.split(","):
            k, v = kv.split("=")
            saved[k] = ctx.get_option(k)
            ctx.set_option(k, float(v))
    launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        launch()
    e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    st = d_stats.cpu().numpy().reshape(-1, 8)
    ok = int((d_status.cpu().numpy() == 0).sum())
    sig, status, stats = batch.solve_batch(z["states"], idx, z["units"], z["iniPar"], z["thickness"], 128, z["tSteps"], return_stats=True, ctx=ctx)
    rel = np.abs(sig - ref) / ref
    grp = [rel[(dec >= a) & (dec < b)].max() for a, b in ((0, 4), (4, 6), (6, 9), (9, 11))]
    for k, v in saved.items():
        ctx.set_option(k, v)
    print(f"{setting:45s} {ms:6.3f} ms  steps {st[:, 0].mean():5.0f}/{st[:, 0].max():4d} ok {ok:4d} rej {st[:, 4].mean():4.1f} | fixture steps {stats[:, 0].mean():5.0f} "
          f"err d0-3 {grp[0]:.1e} d4-5 {grp[1]:.1e} d6-8 {grp[2]:.1e} d9-10 {grp[3]:.1e}", flush=True)
