"""Time solve vs fused likelihood for a PT block (dev tool)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from metrotrpl_amd import _lib as L, laplace, workloads as W, pt_bench
from metrotrpl_amd.likelihood import LikelihoodEngine
nch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
times = W.default_times()
irf_tables = laplace.make_I_tables({745: pt_bench.synthetic_irf()})
vals, uncs = pt_bench.synthetic_data(irf_tables, 128, times)
eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, 128, ["TRPL"] * 6, [times] * 6, vals, uncs, {"TRPL": 1.0},
                       np.array(W.STAUB_FLUENCES), irf_tables=irf_tables, irf_convolution=[745] * 6)
ens = W.pt_ensemble(nch)
T = torch.from_numpy(np.ascontiguousarray(ens["temperatures"])).cuda()
eng.upload(ens["chain_states"])
lib, h = L.lib(), eng.ctx.handle
s = torch.cuda.current_stream().cuda_stream
eng.launch(T, len(ens["temperatures"])); torch.cuda.synchronize()
for rep in range(2):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    eng.ctx.check(lib.mtrpl_solve_batch_dev(h, eng.n_solves, eng.d_descs.data_ptr(), eng.d_ini.data_ptr(), eng.d_t.data_ptr(), eng.d_sig.data_ptr(), eng.d_status.data_ptr(), eng.d_stats.data_ptr(), 128, 0, s))
    e[1].record()
    eng.ctx.check(lib.mtrpl_loglik_batch_dev(h, eng.n_solves, eng.d_like.data_ptr(), eng.d_sig.data_ptr(), eng.d_t.data_ptr(), eng.d_exp_t.data_ptr(), eng.d_exp_y.data_ptr(), eng.d_exp_u.data_ptr(), T.data_ptr(), len(ens["temperatures"]), eng.max_nt, eng.d_ll.data_ptr(), s))
    e[2].record(); torch.cuda.synchronize()
    print(f"chains={nch} solves={eng.n_solves}: solve {e[0].elapsed_time(e[1]):.1f} ms, loglik {e[1].elapsed_time(e[2]):.1f} ms", flush=True)
ll = eng.d_ll.cpu().numpy().reshape(nch, 6, -1)
print("finite frac", np.isfinite(ll).mean(), "ll[0,:,0]", ll[0, :, 0])
