"""Batched likelihood evaluation on the device: the counterpart of the reference's
eval_trial_move / one_sim_likelihood (trial_move_evaluation.py:9-166) for a whole
ensemble of chains at once.

For every chain c and measurement m: solve (forward_solver.py:41-203) -> optional
IRF convolution + trim (laplace.py:44-126) -> negative-fraction check, sign flip,
log10, scale shift, err^2 and ll(T) = -sum err^2 / (sigma^2 T + 2 u^2) for every
ladder temperature (trial_move_evaluation.py:117-158).  Simulated traces never
leave HBM; only the [C, M, nT] log-likelihood table is produced.

torch is used for device buffers only (allocation, H2D/D2H, the current stream).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import batch


class LikelihoodEngine:
    def __init__(self, indexes, units, lengths, nx, meas_types, times, vals, uncs, sigma, iniPar, ini_mode="fluence",
                 model="std", hmax=4.0, rtol=None, atol=None, irf_tables=None, irf_convolution=None,
                 scale_shift=None, force_min_y=False, device=0, ctx=None):
        """times/vals/uncs: per-measurement 1-D arrays (as _times/_vals/_uncs of the
        reference's shared fields); sigma: {meas_type: model_uncertainty};
        irf_tables: {wavelength: (table [k,3], t_irf [k])} from make_I_tables;
        irf_convolution: per-measurement wavelength or 0/None."""
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.ctx = ctx or L.default_context(device)
        self.indexes, self.units = indexes, np.asarray(units, dtype=np.float64)
        self.M = len(lengths)
        self.lengths = np.asarray(lengths, dtype=np.float64)
        self.nx, self.meas_types = int(nx), list(meas_types)
        self.iniPar = np.atleast_2d(np.asarray(iniPar, dtype=np.float64))
        self.ini_mode, self.model, self.hmax, self.rtol, self.atol = ini_mode, model, hmax, rtol, atol
        times = [np.ascontiguousarray(t, dtype=np.float64) for t in times]
        for t in times:
            if t[0] != 0:
                raise ValueError("Grid error - times must start at t=0")
        self.nts = [len(t) for t in times]
        self.t_off = np.concatenate([[0], np.cumsum(self.nts)[:-1]]).astype(np.int64)
        self.max_nt = max(self.nts)
        self.d_t = torch.from_numpy(np.concatenate(times)).to(self.dev)
        self.d_exp_t = self.d_t  # experimental times == simulation times (trial_move_evaluation.py:215)
        self.d_exp_y = torch.from_numpy(np.concatenate([np.asarray(v, dtype=np.float64) for v in vals])).to(self.dev)
        self.d_exp_u = torch.from_numpy(np.concatenate([np.asarray(u, dtype=np.float64) for u in uncs])).to(self.dev)
        self.sigma = [float(sigma[m]) for m in self.meas_types]
        self.scale_shift = np.zeros(self.M) if scale_shift is None else np.asarray(scale_shift, dtype=np.float64)
        # IRF tables -> device (mtrpl_irf_tables_upload)
        self.irf_index = [-1] * self.M
        self.force_min_y = bool(force_min_y)  # shared_fields["force_min_y"], trial_move_evaluation.py:136
        if irf_tables and irf_convolution is not None:
            waves = sorted({int(w) for w in irf_convolution if w})
            nk = np.array([irf_tables[w][0].shape[0] for w in waves], dtype=np.int32)
            tabs = np.concatenate([L.f64(irf_tables[w][0]).reshape(-1) for w in waves])
            dts = np.array([float(np.mean(np.diff(irf_tables[w][1]))) for w in waves])  # laplace.py:66
            self.ctx.check(L.lib().mtrpl_irf_tables_upload(self.ctx.handle, len(waves), L.iptr(nk), L.dptr(tabs),
                                                           L.dptr(dts)))
            self.irf_index = [waves.index(int(w)) if w else -1 for w in irf_convolution]
        self._cap = 0

    def _ensure(self, n_solves, nT):
        torch = self.torch
        if n_solves > self._cap:
            self._cap = n_solves
            tot = int(np.sum(self.nts)) * (n_solves // self.M)
            self.d_sig = torch.zeros(tot, dtype=torch.float64, device=self.dev)
            self.d_status = torch.zeros(n_solves, dtype=torch.int32, device=self.dev)
            self.d_stats = torch.zeros(n_solves * 8, dtype=torch.int32, device=self.dev)
        self.d_ll = torch.zeros(n_solves * nT, dtype=torch.float64, device=self.dev)

    def descriptors(self, states):
        """Host-side packing for C chains x M measurements (chain-major)."""
        states = np.atleast_2d(np.asarray(states, dtype=np.float64))
        Cn = states.shape[0]
        st = np.repeat(states, self.M, axis=0)
        ini = np.tile(self.iniPar, (Cn, 1))
        thick = np.tile(self.lengths, Cn)
        meas = self.meas_types * Cn
        per_chain = int(np.sum(self.nts))
        out_off = (np.repeat(np.arange(Cn, dtype=np.int64), self.M) * per_chain + np.tile(self.t_off, Cn))
        d, ini_arr = batch.build_batch(st, self.indexes, self.units, ini, thick, self.nx, 1, hmax=self.hmax, meas=meas,
                                       model=self.model, ini_mode=self.ini_mode, RTOL=self.rtol, ATOL=self.atol)
        d["nt"] = np.tile(self.nts, Cn)
        d["t_off"] = np.tile(self.t_off, Cn)
        d["out_off"] = out_off
        ld = np.zeros(Cn * self.M, dtype=L.LIKE_DESC_DTYPE)
        ld["sig_off"] = out_off
        ld["t_off"] = np.tile(self.t_off, Cn)
        ld["exp_off"] = np.tile(self.t_off, Cn)
        ld["nt"] = d["nt"]
        ld["n_exp"] = d["nt"]
        ld["irf"] = np.tile(self.irf_index, Cn)
        ld["force_min_y"] = int(self.force_min_y)
        ld["scale_shift"] = np.tile(self.scale_shift, Cn)
        ld["sigma"] = np.tile(self.sigma, Cn)
        return d, ini_arr, ld

    def upload(self, states):
        torch = self.torch
        d, ini_arr, ld = self.descriptors(states)
        self.n_solves = d.shape[0]
        self.d_descs = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(self.dev)
        self.d_ini = torch.from_numpy(ini_arr).to(self.dev)
        self.d_like = torch.from_numpy(ld.view(np.uint8).reshape(-1)).to(self.dev)

    def launch(self, d_T, nT, stream=None):
        """Enqueue solve + likelihood for the uploaded states on `stream` (torch stream)."""
        torch = self.torch
        self._ensure(self.n_solves, nT)
        s = (stream or torch.cuda.current_stream()).cuda_stream
        lib, h = L.lib(), self.ctx.handle
        self.ctx.check(lib.mtrpl_solve_batch_dev(h, self.n_solves, self.d_descs.data_ptr(), self.d_ini.data_ptr(),
                                                 self.d_t.data_ptr(), self.d_sig.data_ptr(), self.d_status.data_ptr(),
                                                 self.d_stats.data_ptr(), self.nx, L.MODEL[self.model], s))
        self.ctx.check(lib.mtrpl_loglik_batch_dev(h, self.n_solves, self.d_like.data_ptr(), self.d_sig.data_ptr(),
                                                  self.d_t.data_ptr(), self.d_exp_t.data_ptr(), self.d_exp_y.data_ptr(),
                                                  self.d_exp_u.data_ptr(), d_T.data_ptr(), nT, self.max_nt,
                                                  self.d_ll.data_ptr(), s))
        return self.d_ll

    def eval(self, states, temperatures=(1.0,)):
        """-> ll table [C, M, nT] (numpy).  logll of chain c at T_k = table[c, :, k].sum()."""
        torch = self.torch
        T = np.ascontiguousarray(np.atleast_1d(temperatures), dtype=np.float64)
        d_T = torch.from_numpy(T).to(self.dev)
        self.upload(states)
        ll = self.launch(d_T, len(T))
        torch.cuda.synchronize()
        return ll.cpu().numpy().reshape(-1, self.M, len(T))
