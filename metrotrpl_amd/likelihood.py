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

import numpy as np

from . import _lib as L
from . import batch


class LikelihoodEngine:
    def __init__(self, indexes, units, lengths, nx, meas_types, times, vals, uncs, sigma, iniPar, ini_mode="fluence",
                 model="std", hmax=4.0, rtol=None, atol=None, irf_tables=None, irf_convolution=None,
                 scale_shift=None, force_min_y=False, device=0, ctx=None, fittables=None, sort_by_cost=True, parts=2,
                 split_min=2048, sort_min=1024):
        """times/vals/uncs: per-measurement 1-D arrays (as _times/_vals/_uncs of the
        reference's shared fields); sigma: {meas_type: model_uncertainty};
        irf_tables: {wavelength: (table [k,3], t_irf [k])} from make_I_tables;
        irf_convolution: per-measurement wavelength or 0/None; nx: one value or one per
        measurement; fittables: {"fluences"|"absps"|"scale": per-measurement index of the
        state entry that multiplies iniPar[0] / iniPar[1] / sets scale_shift = log10(entry),
        -1 = none} (trial_move_evaluation.py:38-60, see `fittable_indexes`).
        parts / split_min: launches of more than split_min solves run as `parts` chain blocks on separate
        streams; sort_by_cost / sort_min: launches of more than sort_min solves issue the expensive solves
        of a block first."""
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device)
        self.ctx = ctx or L.default_context(device)
        self.indexes, self.units = indexes, np.asarray(units, dtype=np.float64)
        self.M = len(lengths)
        self.lengths = np.asarray(lengths, dtype=np.float64)
        self.nx_list = np.broadcast_to(np.asarray(nx, dtype=np.int32), (self.M,)).copy()
        self.nx, self.meas_types = int(self.nx_list.max()), list(meas_types)
        if ini_mode == "density" and np.any(self.nx_list != self.nx):
            raise NotImplementedError("density-mode initial conditions need one nx for all measurements")
        self.fit = {k: np.asarray((fittables or {}).get(k, [-1] * self.M), dtype=np.int64)
                    for k in ("fluences", "absps", "scale")}
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
        self.sort_by_cost = bool(sort_by_cost)
        self.parts = max(1, int(parts))
        self.split_min, self.sort_min = int(split_min), int(sort_min)
        self.packed_sort_min = self.sort_min  # plan_packed's threshold; a caller that wants EVERY packed launch ordered sets 0
        self._stage = []  # pinned staging buffers of the descriptor upload: [tensor, event] pairs, used in turn
        self._side = []
        self._cap = 0
        self._tmpl = {}
        self._ticks = None
        self.timing = None  # a list: launch() appends (stream, e0, e1, e2, n_solves) HIP-event triples around
        #                     the solve and the likelihood kernel of every block (bench.py: roofline)
        self.per_simd = 0   # 1 / 2: this engine's launches run the one- / two-solves-per-SIMD build of the wave kernel
        #                     (mtrpl_solve_batch_dev2: a caller with several engines in flight knows what shares the
        #                     device); 0: by the size of the launch

    def _ensure(self, n_solves, nT):
        torch = self.torch
        if n_solves > self._cap:
            self._cap = n_solves
            tot = int(np.sum(self.nts)) * (n_solves // self.M)
            self.d_sig = torch.zeros(tot, dtype=torch.float64, device=self.dev)
            self.d_status = torch.zeros(n_solves, dtype=torch.int32, device=self.dev)
            self.d_stats = torch.zeros(n_solves * 8, dtype=torch.int32, device=self.dev)
        # (the likelihood kernel writes every entry of the table -- NaN for a descriptor it refuses, -inf for a failed trace --
        # so the buffer is neither cleared nor made anew per launch: a 4 MB fill kernel took its turn among the solves of
        # three chain groups, half a millisecond of every launch cycle of the 512-chain ensemble)
        if getattr(self, "d_ll", None) is None or self.d_ll.shape[0] != n_solves * nT:
            self.d_ll = torch.empty(n_solves * nT, dtype=torch.float64, device=self.dev)

    def _template(self, Cn):
        """Everything in the descriptors of C chains x M measurements (chain-major) that does not depend on the
        states: grids, offsets, tolerances, measurement kinds, validation -- built once per ensemble size."""
        tm = self._tmpl.get(Cn)
        if tm is not None:
            return tm
        ini = np.tile(self.iniPar, (Cn, 1))
        thick = np.tile(self.lengths, Cn)
        meas = self.meas_types * Cn
        per_chain = int(np.sum(self.nts))
        out_off = (np.repeat(np.arange(Cn, dtype=np.int64), self.M) * per_chain + np.tile(self.t_off, Cn))
        dummy = np.ones((Cn * self.M, self.units.shape[0]))
        d, ini_arr = batch.build_batch(dummy, self.indexes, self.units, ini, thick, self.nx, 1, hmax=self.hmax, meas=meas,
                                       model=self.model, ini_mode=self.ini_mode, RTOL=self.rtol, ATOL=self.atol)
        d["nx"] = np.tile(self.nx_list, Cn)
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
        rows = np.repeat(np.arange(Cn), self.M)
        fit = {}
        for key in ("fluences", "absps", "scale"):
            idx = np.tile(self.fit[key], Cn)
            use = idx >= 0
            fit[key] = (use, rows[use], idx[use]) if use.any() else None
        tm = self._tmpl[Cn] = dict(d=d, ini_arr=ini_arr, ld=ld, ini3=d["ini"].copy(), fit=fit)
        return tm

    def descriptors(self, states):
        """Host-side packing for C chains x M measurements (chain-major)."""
        states = np.atleast_2d(np.asarray(states, dtype=np.float64))
        Cn = states.shape[0]
        tm = self._template(Cn)
        d, ld, ini_arr, fit = tm["d"].copy(), tm["ld"], tm["ini_arr"], tm["fit"]
        d["par"] = np.repeat(batch.pack_params(states * self.units, self.indexes), self.M, axis=0)
        if fit["fluences"] is not None or fit["absps"] is not None:
            # a COPY of the init row is scaled (the reference compounds in place, :44/:51).  The factors act on
            # elements 0 and 1 of the measurement's init-params row whatever the row means: [fluence, alpha, ..]
            # in fluence mode, the densities of nodes 0 and 1 in density mode (trial_move_evaluation.py:44,:51
            # index iniPar[0] / iniPar[1] without looking at ini_mode).
            if self.ini_mode == "density":
                ini_arr = ini_arr.reshape(d.shape[0], -1).copy()  # per launch: bypasses the _const cache
                for key, col in (("fluences", 0), ("absps", 1)):
                    if fit[key] is not None:
                        use, r, i = fit[key]
                        ini_arr[use, col] *= states[r, i]
                ini_arr = ini_arr.reshape(-1)
            else:
                ini3 = tm["ini3"].copy()
                for key, col in (("fluences", 0), ("absps", 1)):
                    if fit[key] is not None:
                        use, r, i = fit[key]
                        ini3[use, col] *= states[r, i]
                d["ini"] = ini3
        if fit["scale"] is not None:
            use, r, i = fit["scale"]
            ld = ld.copy()
            with np.errstate(divide="ignore", invalid="ignore"):
                ld["scale_shift"][use] = np.log10(states[r, i])
        # Over-subscribed launches (more solves than SIMDs) are cut into `parts` chain blocks that
        # run on separate HIP streams: the likelihood kernel of one block overlaps the solve tail of
        # the next.  Inside a block the expensive solves go first (the hardware hands out workgroups
        # in index order): the tightened-tolerance ones, tauP >> tauN (DESIGN.md "error control").
        # Only the solve descriptors move; outputs keep their places (out_off), the likelihood
        # descriptors and the ll table stay chain-major.
        n = d.shape[0]
        parts = self.parts if n > self.split_min else 1
        cuts = [(Cn * p // parts) * self.M for p in range(parts + 1)]
        self._order = None
        if self.sort_by_cost and n > self.sort_min:
            cost = self._measured_cost(n, states)
            self._cost_kind = "measured" if cost is not None else "heuristic"
            if cost is None:  # first launch: the tightened-tolerance solves (tauP >> tauN) are the slow ones
                tn, tp = states[:, self.indexes["tauN"]], states[:, self.indexes["tauP"]]
                cost = np.repeat(1.0 + tp / tn, self.M)
            order = np.concatenate([a + np.argsort(-cost[a:b], kind="stable") for a, b in zip(cuts[:-1], cuts[1:])])
            d = d[order]
            self._order = order
        self._cuts = cuts
        return d, ini_arr, ld

    def _row_cost(self, n):
        """Residency of every solve in the previous launch, row for row (chain-major order); None when the previous
        launch had another size or was not ordered (its residencies were not copied back)."""
        prev = getattr(self, "_prev_order", None)
        if prev is None or prev.shape[0] != n or self._ticks is None or self._ticks[1] != n:
            return None
        self._ticks[2].synchronize()
        cost = np.empty(n)
        cost[prev] = self._ticks[0][:n].numpy()
        return cost if np.any(cost > 0) else None

    def _measured_cost(self, n, states):
        """Residency of every solve in the previous launch (stats[7]), in chain-major order: Metropolis
        proposals are small moves, so the last iteration's cost is the best predictor of this one's.  A
        replica exchange moves states along the ladder, so each row takes the cost of the closest of the
        previous rows (within two slots of it, else anywhere).  Not used when some state has no previous state nearby
        (dense sampling)."""
        prev = getattr(self, "_prev_order", None)
        last = getattr(self, "_prev_states", None)
        self._prev_states = np.array(states, dtype=np.float64, copy=True)
        if prev is None or prev.shape[0] != n or getattr(self, "d_stats", None) is None:
            return None
        if last is None or last.shape != self._prev_states.shape:
            return None
        Cn = states.shape[0]
        with np.errstate(all="ignore"):
            cur, old = np.log(np.abs(self._prev_states)), np.log(np.abs(last))
            best, src = np.full(Cn, np.inf), np.arange(Cn)
            for o in (0, 1, -1, 2, -2):
                dist = np.nan_to_num(np.abs(cur - np.roll(old, -o, axis=0)), nan=0.0, posinf=np.inf).max(axis=1)
                upd = dist < best
                best[upd] = dist[upd]
                src[upd] = (np.arange(Cn)[upd] + o) % Cn
            far = best >= 2.0
            if far.any() and int(far.sum()) * Cn <= 200000:  # moved further than two slots: look everywhere
                dall = np.nan_to_num(np.abs(cur[far][:, None, :] - old[None, :, :]), nan=0.0, posinf=np.inf).max(axis=2)
                j = dall.argmin(axis=1)
                best[far], src[far] = dall[np.arange(j.shape[0]), j], j
        if not np.all(best < 2.0):
            return None
        if self._ticks is not None and self._ticks[1] == n:  # copied back right behind the launch
            self._ticks[2].synchronize()
            ticks = self._ticks[0][:n].numpy().astype(np.float64)
        else:
            ticks = self.d_stats[:8 * n].view(-1, 8)[:, 7].cpu().numpy().astype(np.float64)
        if not np.any(ticks > 0):
            return None
        cost = np.empty(n)
        cost[prev] = ticks
        return cost.reshape(Cn, self.M)[src].reshape(-1)

    def _const(self, tm, key, arr):
        """Device copy of a descriptor array; the template's own (state-independent) arrays go up once."""
        raw = arr.view(np.uint8).reshape(-1) if arr.dtype.names else arr
        if arr is not tm[key]:
            return self.torch.from_numpy(raw).to(self.dev)
        if "dev_" + key not in tm:
            tm["dev_" + key] = self.torch.from_numpy(raw).to(self.dev)
        return tm["dev_" + key]

    def _to_device(self, raw):
        """The descriptor array through a pinned staging buffer, asynchronously on the current stream (a copy from
        pageable memory makes the host wait for the copy engine; two buffers in turn, each behind the event of its
        last copy)."""
        torch = self.torch
        n = raw.shape[0]
        if len(self._stage) < 2:
            self._stage.append([torch.empty(max(n, 1 << 16), dtype=torch.uint8, pin_memory=True), None])
            slot = self._stage[-1]
        else:
            slot = self._stage[0]
            self._stage.reverse()
            if slot[1] is not None:
                slot[1].synchronize()
            if slot[0].shape[0] < n:
                slot[0] = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        slot[0][:n].numpy()[:] = raw
        out = torch.empty(n, dtype=torch.uint8, device=self.dev)
        out.copy_(slot[0][:n], non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream())
        return out

    # ---- packed launches: descriptors the caller keeps (and may choose between) on the device ------------------
    def plan_packed(self, states):
        """Fix, for the packed launches of `states.shape[0]` state rows that follow, the order the solves are issued in
        (`sort_by_cost` / `sort_min`: the residencies of the previous launch of this size, row for row -- the rows of
        a Metropolis ensemble between two replica exchanges keep their chains -- or, without any, tauP / tauN of
        `states`) and the split into `parts` streams.  -> the state row of every descriptor row, in issue order."""
        states = np.atleast_2d(np.asarray(states, dtype=np.float64))
        R = states.shape[0]
        n = R * self.M
        parts = self.parts if n > self.split_min else 1
        cuts = [(R * p // parts) * self.M for p in range(parts + 1)]
        order = None
        if self.sort_by_cost and n > self.packed_sort_min:
            cost = self._row_cost(n)
            if cost is None:
                cost = np.repeat(1.0 + states[:, self.indexes["tauP"]] / states[:, self.indexes["tauN"]], self.M)
            order = np.concatenate([a + np.argsort(-cost[a:b], kind="stable") for a, b in zip(cuts[:-1], cuts[1:])])
        self._packed = (n, cuts, order)
        return (np.arange(n) if order is None else order) // self.M

    def packable(self):
        """Can launches run from descriptors picked on the device (`pack_rows` / `use_packed`)?  Only while the likelihood
        descriptors and the initial-condition array do not depend on the state: no fitted scale factor, no density-mode
        fittables."""
        return not (self.fit["scale"].max() >= 0 or
                    (self.ini_mode == "density" and max(self.fit["fluences"].max(), self.fit["absps"].max()) >= 0))

    def pack_rows(self, states, sets=1):
        """Solve descriptors of `states` [R, n_params] x M measurements as bytes [R * M, desc size]; `sets` > 1: the
        rows are that many alternative sets of R / sets state rows, each placed (output and initial-condition offsets)
        and ordered (`plan_packed`; default: as given, M rows per state row) as if it were launched alone -- for callers
        that hold several candidate sets on the device and pick rows there (pt.PTEnsemble's device-side decisions).
        Only for ensembles whose likelihood descriptors and initial-condition array do not depend on the state (no
        fitted scale factor, no density-mode fittables)."""
        if not self.packable():
            raise NotImplementedError("packed launches need state-independent likelihood descriptors")
        keep = self.sort_by_cost, self.parts
        self.sort_by_cost, self.parts = False, 1
        try:
            d, _, _ = self.descriptors(states)
        finally:
            self.sort_by_cost, self.parts = keep
        one = d.shape[0] // sets
        if sets > 1:
            for key in ("out_off", "ini_off"):
                d[key] = np.tile(d[key][:one], sets)
        plan = getattr(self, "_packed", None)
        if plan is not None and plan[0] == one and plan[2] is not None:
            d = d.reshape(sets, one)[:, plan[2]].reshape(-1)
        return d.view(np.uint8).reshape(d.shape[0], -1)

    def use_packed(self, d_descs, n_solves):
        """The next `launch` runs the `n_solves` descriptors of the device tensor `d_descs` (bytes, as `pack_rows`
        lays one set out)."""
        tm = self._template(n_solves // self.M)
        self.n_solves, self.d_descs = int(n_solves), d_descs
        self.d_ini = self._const(tm, "ini_arr", tm["ini_arr"])
        self.d_like = self._const(tm, "ld", tm["ld"])
        plan = getattr(self, "_packed", None)
        if plan is not None and plan[0] == n_solves:
            self._cuts, self._order = plan[1], plan[2]
        else:
            self._cuts, self._order = [0, int(n_solves)], None

    def upload(self, states):
        d, ini_arr, ld = self.descriptors(states)
        self.n_solves = d.shape[0]
        tm = self._template(self.n_solves // self.M)
        self.d_descs = self._to_device(d.view(np.uint8).reshape(-1))
        self.d_ini = self._const(tm, "ini_arr", ini_arr)
        self.d_like = self._const(tm, "ld", ld)

    def launch(self, d_T, nT, stream=None):
        """Enqueue solve + likelihood for the uploaded states on `stream` (torch stream); with
        several parts the extra streams fork from and join back into it."""
        torch = self.torch
        self._ensure(self.n_solves, nT)
        main = stream or torch.cuda.current_stream()
        lib, h = L.lib(), self.ctx.handle
        cuts = self._cuts
        n_parts = len(cuts) - 1
        if len(self._side) < n_parts - 1:
            self._side = list(reversed(L.shared_streams(self.dev, max(4, n_parts - 1))))[:n_parts - 1]  # from the far end
            #                  of the process's pool: the streams callers launch engines on come from its near end
        # (the library issues the solves of a launch in the order of the descriptor array: the measured-cost order
        # this engine packed them in is the order they run in)
        fork = None
        if n_parts > 1:
            fork = torch.cuda.Event()
            fork.record(main)
        joins = []
        sd, sl = L.SOLVE_DESC_DTYPE.itemsize, L.LIKE_DESC_DTYPE.itemsize
        for p in range(n_parts):
            a, b = cuts[p], cuts[p + 1]
            if b <= a:
                continue
            st = main if p == 0 else self._side[p - 1]
            if p > 0:
                st.wait_event(fork)
            s = st.cuda_stream
            if self.timing is not None:
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                ev[0].record(st)
            self.ctx.check(lib.mtrpl_solve_batch_dev2(h, b - a, self.d_descs.data_ptr() + a * sd, self.d_ini.data_ptr(),
                                                      self.d_t.data_ptr(), self.d_sig.data_ptr(),
                                                      self.d_status.data_ptr() + 4 * a, self.d_stats.data_ptr() + 32 * a,
                                                      self.nx, L.MODEL[self.model], s, int(self.per_simd)))
            if self.timing is not None:
                ev[1].record(st)
            self.ctx.check(lib.mtrpl_loglik_batch_dev(h, b - a, self.d_like.data_ptr() + a * sl, self.d_sig.data_ptr(),
                                                      self.d_t.data_ptr(), self.d_exp_t.data_ptr(),
                                                      self.d_exp_y.data_ptr(), self.d_exp_u.data_ptr(), d_T.data_ptr(),
                                                      nT, self.max_nt, self.d_ll.data_ptr() + 8 * a * nT, s))
            if self.timing is not None:
                ev[2].record(st)
                self.timing.append((ev[0], ev[1], ev[2], b - a))
            if p > 0:
                ev = torch.cuda.Event()
                ev.record(st)
                joins.append(ev)
        for ev in joins:
            main.wait_event(ev)
        self._prev_order = self._order  # which solve sat where, for the cost read-back of the next launch
        if self._order is not None:
            # the residencies of this launch, for the ordering of the next: copied to pinned memory behind the
            # kernels instead of a blocking read when the next launch is being packed
            n = self.n_solves
            if self._ticks is None or self._ticks[0].shape[0] < n:
                self._ticks = [torch.empty(n, dtype=torch.int32, pin_memory=True), 0, torch.cuda.Event()]
            self._ticks[0][:n].copy_(self.d_stats[:8 * n].view(-1, 8)[:, 7], non_blocking=True)
            self._ticks[1] = n
            self._ticks[2].record(main)
        else:
            self._ticks = None
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

    # ---- several evaluations in flight: lanes ---------------------------------------------------------------
    def clone(self):
        """A second LANE of the same problem: shares everything that is constant on the device (time grids, data, IRF
        tables, descriptor templates) and owns what one launch writes (simulated traces, status, table, staging), so that
        evaluations on different lanes may be in flight on different streams (`eval_async`)."""
        import copy
        c = copy.copy(self)
        c._stage, c._side, c._cap, c._ticks, c.timing = [], [], 0, None, None
        for k in ("d_sig", "d_status", "d_stats", "d_ll", "_prev_order", "_prev_states", "_packed", "_lane_out", "_lane_T"):
            c.__dict__.pop(k, None)
        return c

    def eval_async(self, states, temperatures=(1.0,), stream=None):
        """`eval` without the wait: upload, solve, likelihood and the copy of the table to pinned host memory are
        enqueued on `stream` (a torch stream; default: the current one) and a handle comes back at once;
        `handle.result()` waits for that lane only and returns the [C, M, nT] table.  One evaluation per lane at a time:
        call `result()` before the next `eval_async` of the same engine, use `clone()`s for more in flight."""
        torch = self.torch
        T = np.ascontiguousarray(np.atleast_1d(temperatures), dtype=np.float64)
        st = stream or torch.cuda.current_stream()
        with torch.cuda.stream(st):
            cached = getattr(self, "_lane_T", None)
            if cached is None or cached[0].shape != T.shape or not np.array_equal(cached[0], T):
                cached = self._lane_T = (T.copy(), torch.from_numpy(T).to(self.dev))
            self.upload(states)
            ll = self.launch(cached[1], len(T), st)
            n = ll.shape[0]
            host = getattr(self, "_lane_out", None)
            if host is None or host.shape[0] < n:
                host = self._lane_out = torch.empty(n, dtype=torch.float64, pin_memory=True)
            host[:n].copy_(ll, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st)
        return _Pending(done, host, n, (-1, self.M, len(T)))


class _Pending:
    """An evaluation in flight (LikelihoodEngine.eval_async)."""

    def __init__(self, done, host, n, shape):
        self._done, self._host, self._n, self._shape = done, host, n, shape

    def ready(self):
        return self._done.query()

    def result(self):
        self._done.synchronize()
        return self._host[:self._n].numpy().reshape(self._shape).copy()


def fittable_indexes(shared_fields):
    """Resolve the reference's fittable_fluences / fittable_absps / scale_factor specs
    ([move size, measurement indices, constraint groups, guesses]) to, per measurement, the
    index of the `_f{g}` / `_a{g}` / `_s{g}` state entry it uses (g = first member of the
    measurement's constraint group, utils.py:4-14), -1 where the measurement has none."""
    M = shared_fields["_sim_info"]["num_meas"]
    pidx = shared_fields["_param_indexes"]
    out = {}
    for key, field, prefix in (("fluences", "fittable_fluences", "_f"), ("absps", "fittable_absps", "_a"),
                               ("scale", "scale_factor", "_s")):
        spec = shared_fields.get(field, None)
        idx = [-1] * M
        if spec is not None:
            for m in range(M):
                if m not in spec[1]:
                    continue
                g = m
                for grp in (spec[2] or []):
                    if m in grp:
                        g = grp[0]
                        break
                idx[m] = pidx[f"{prefix}{g}"]
        out[key] = idx
    return out


def engine_from_fields(shared_fields, unique_fields, device=0, ctx=None):
    """A LikelihoodEngine for the ensemble described by the reference's shared/unique
    field dictionaries (sim_utils.py:98-199 + the `_times/_vals/_uncs/_init_params/
    _IRF_tables` entries metro() adds, metropolis.py:325-342)."""
    sf = shared_fields
    si = sf["_sim_info"]
    if "solver" in sf and sf["solver"][0] not in ("solveivp", "odeint", "diagnostic"):
        raise NotImplementedError("No solver available")  # forward_solver.py:225 (the NN surrogate is out of scope)
    # read_config_script_file leaves sim_info["nx"] at its un-selected length (bayes_io.py:531-536); the
    # reference indexes it by measurement number (utils.py:35), i.e. uses the leading entries
    nx = np.asarray(si["nx"]).reshape(-1)
    nx = nx[:len(si["lengths"])] if nx.size > 1 else nx
    return LikelihoodEngine(sf["_param_indexes"], sf["units"], si["lengths"], nx, si["meas_types"],
                            sf["_times"], sf["_vals"], sf["_uncs"], unique_fields["model_uncertainty"],
                            sf["_init_params"], ini_mode=sf["ini_mode"], model=sf["model"], hmax=sf["hmax"],
                            rtol=sf.get("rtol"), atol=sf.get("atol"), irf_tables=sf.get("_IRF_tables") or None,
                            irf_convolution=sf.get("irf_convolution"), force_min_y=sf.get("force_min_y", False),
                            device=device, ctx=ctx, fittables=fittable_indexes(sf))
