import sys, time
import numpy as np
sys.path.insert(0, ".")
import torch
from metrotrpl_amd import _lib as L, batch, workloads as W
dev = torch.device("cuda", 0)
ctx = L.Context(0)
B = 1024
c = W.config2(B)
names = W.PARAM_NAMES + ["kC", "Nt", "tauE"]
ix = {n: i for i, n in enumerate(names)}
un = np.array(list(W.units()) + [((1e7) ** 3) / (1e9), ((1e-7) ** 3), 1.0])
rng = np.random.default_rng(1)
st = np.zeros((B, len(names))); st[:, :13] = c["states"]
st[:, ix["kC"]] = 10 ** rng.uniform(-11, -9, B); st[:, ix["Nt"]] = 10 ** rng.uniform(14, 16, B); st[:, ix["tauE"]] = 10 ** rng.uniform(0.5, 3, B)
t = np.ascontiguousarray(c["tSteps"]); nt = len(t)
for model, states, idx, units in (("traps", st, ix, un), ("std", c["states"], W.indexes(), W.units())):
    for variant in ((0,) if model == "traps" else (1,)):
        ctx.set_option("kernel_variant", variant)
        d, ini = batch.build_batch(states, idx, units, c["iniPar"], c["thickness"], 128, nt, meas="TRPL", model=model)
        dd = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev); di = torch.from_numpy(ini).to(dev); dt = torch.from_numpy(t).to(dev)
        o = torch.zeros(B * nt, dtype=torch.float64, device=dev); s = torch.zeros(B, dtype=torch.int32, device=dev); sa = torch.zeros(B * 8, dtype=torch.int32, device=dev)
        def step():
            ctx.check(L.lib().mtrpl_solve_batch_dev(ctx.handle, B, dd.data_ptr(), di.data_ptr(), dt.data_ptr(), o.data_ptr(), s.data_ptr(), sa.data_ptr(), 128, L.MODEL[model], torch.cuda.current_stream().cuda_stream))
        step(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / 5
        print(model, "team kernel, one 1024-solve launch at a time: %.2f ms, %.0f solves/s, steps mean %.0f, ok %s" % (dtm * 1e3, B / dtm, sa.cpu().numpy().reshape(B, 8)[:, 0].mean(), bool((s == 0).all())))
