"""Where does the fused likelihood kernel spend its time: IRF convolution vs temperature ladder (dev tool)."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metrotrpl_amd import _lib as L, laplace, workloads as W, pt_bench
from metrotrpl_amd.likelihood import LikelihoodEngine
nch = 512
times = W.default_times()
irf_tables = laplace.make_I_tables({745: pt_bench.synthetic_irf()})
vals, uncs = pt_bench.synthetic_data(irf_tables, 128, times)
ens = W.pt_ensemble(nch)
for irf in (True, False):
    eng = LikelihoodEngine(W.indexes(), W.units(), W.STAUB_LENGTHS, 128, ["TRPL"] * 6, [times] * 6, vals, uncs, {"TRPL": 1.0},
                           np.array(W.STAUB_FLUENCES), irf_tables=irf_tables if irf else None, irf_convolution=[745] * 6 if irf else None)
    eng.upload(ens["chain_states"])
    for nT in (512, 32, 1):
        T = torch.from_numpy(np.ascontiguousarray(ens["temperatures"][:nT])).cuda()
        eng.launch(T, nT); torch.cuda.synchronize()
        lib, h = L.lib(), eng.ctx.handle
        s = torch.cuda.current_stream().cuda_stream
        e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        e[0].record()
        for _ in range(3):
            eng.ctx.check(lib.mtrpl_loglik_batch_dev(h, eng.n_solves, eng.d_like.data_ptr(), eng.d_sig.data_ptr(), eng.d_t.data_ptr(), eng.d_exp_t.data_ptr(), eng.d_exp_y.data_ptr(), eng.d_exp_u.data_ptr(), T.data_ptr(), nT, eng.max_nt, eng.d_ll.data_ptr(), s))
        e[1].record(); torch.cuda.synchronize()
        print(f"irf={irf} nT={nT}: loglik {e[0].elapsed_time(e[1])/3:.2f} ms", flush=True)
