// solve_kernel_blk_nq.hpp -- the team kernel (W wavefronts per solve, nx <= 128*W, std or traps) with the
// integrator in Nordsieck form: set-up and device functions of solve_kernel_blk.hpp / solver_block.hpp,
// time stepping of solve_kernel_nq.hpp (fixed-coefficient BDF as LSODE).  fp64 only; the fp32 sweep
// (mtrpl_fp32_pass.hpp) keeps the NDF-form team kernel.
#pragma once
#include "solve_kernel_blk.hpp"
#include "solve_kernel_nq.hpp"

namespace mtrpl {

template <int W, bool TRAPS, bool GBLK>
__global__ __launch_bounds__(64 * W) void solve_kernel_blk_nq(const mtrpl_solve_desc *__restrict__ descs,
                                                           const f64 *__restrict__ ini,
                                                           const f64 *__restrict__ tsteps,
                                                           f64 *__restrict__ out, int32_t *__restrict__ status,
                                                           int32_t *__restrict__ stats, int64_t B, SolveTuning tune,
                                                           double2 *__restrict__ gblk)
{
    constexpr int R = 2, V = TRAPS ? 8 : 6, T = Team<W>::T;
    const int64_t sidx = blockIdx.x;
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if (sidx >= B) return;
    const int lane = threadIdx.x & 63, tid = threadIdx.x;
    __shared__ double2 xbuf_lds[team_xbuf_bytes<W>() / 16];
    __shared__ double rbuf_lds[4 * W + W + 4];
    __shared__ double2 blk_lds[GBLK ? 1 : team_blk_bytes<W>() / 16];
    Team<W> tm;
    tm.tid = tid; tm.lane = lane; tm.wave = tid >> 6;
    tm.xbuf = xbuf_lds; tm.rbuf = rbuf_lds; tm.xpar = 0; tm.rpar = 0;
    auto tnorm = [&](const double (&v)[V], double inv_S_) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < V; ++k) s += v[k] * v[k];
        return norm_sqrt(team_sum<W>(tm, s) * inv_S_);
    };
    const mtrpl_solve_desc &ds = descs[sidx];
    const int L = ds.nx, nt = ds.nt, meas = ds.meas;
    const double dx = ds.thickness / (double)L;  // sim_utils.py:265
    const Par p = make_par(ds.par, dx);
    const f64 *ts = tsteps + ds.t_off;
    f64 *o = out + ds.out_off;
    const double sig_scale = (meas == MTRPL_MEAS_TRPL) ? 1e23 : 1e9;  // forward_solver.py:189,:197
    const double min_y = ds.min_y;
    const double hmax = tune.honor_hmax ? ds.hmax : 1e300;
    const double inv_S = 1.0 / (double)((TRAPS ? 4 : 3) * L + 1);

    // ---- initial condition, forward_solver.py:100-122 ----
    double y[V];
    double dNmax = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = tid * R + r;
        double dN = 0.0;
        if (i < L) {
            if (ds.ini_mode == MTRPL_INI_DENSITY) {
                dN = ini[ds.ini_off + i] * 1e-21;  // :102
            } else {
                const double fluence = ds.ini[0] * 1e-14, alpha = ds.ini[1] * 1e-7;  // :107-108
                const int ii = ((int)ds.ini[2] < 0) ? (L - 1 - i) : i;               // :111
                // np.linspace(dx/2, thickness - dx/2, nx)[ii]  (sim_utils.py:266)
                const double x0 = dx / 2, x1 = ds.thickness - dx / 2;
                const double step = (x1 - x0) / (double)(L - 1);
                const double x = (ii == L - 1) ? x1 : x0 + (double)ii * step;
                dN = fluence * alpha * exp(-alpha * x);  // :109
            }
        }
        y[r] = (i < L) ? dN + p.n0 : 1.0;       // :120
        y[R + r] = (i < L) ? dN + p.p0 : 1.0;   // :121
        if (TRAPS) y[6 + r] = 0.0;              // :133 np.zeros_like(N)
        dNmax = fmax(dNmax, (i < L) ? fmax(y[r], y[R + r]) : 0.0);
    }
    {
        // E_field (:26-38, :122): E_i = Lambda*dx*cumsum((P-p0)-(N-n0))[i-1]; prefix scan over the team
        double own[R], run = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            own[r] = (i < L) ? ((y[R + r] - p.p0) - (y[r] - p.n0)) : 0.0;
            run += own[r];
        }
        double incl = run;  // inclusive scan of per-thread totals inside the wave
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double t = shf(incl, (lane - s) & 63);
            incl += (lane >= s) ? t : 0.0;
        }
        double excl = incl - run;
        if (W > 1) {  // add the totals of the waves to the left
            double *r_ = tm.rbuf + 2 * W * 2;
            if (lane == 63) r_[tm.wave] = incl;
            __syncthreads();
            for (int w_ = 0; w_ < W; ++w_)
                if (w_ < tm.wave) excl += r_[w_];
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            y[2 * R + r] = (i > 0 && i < L) ? p.Lam * dx * excl : 0.0;
            excl += own[r];
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) dNmax = fmax(dNmax, __shfl_xor(dNmax, s, 64));
    if (W > 1) {
        double *r_ = tm.rbuf + 2 * W * 2;
        if (lane == 0) r_[tm.wave] = dNmax;
        __syncthreads();
        for (int w_ = 0; w_ < W; ++w_) dNmax = fmax(dNmax, r_[w_]);
        __syncthreads();
    }
    auto sigof = [&](const double (&yy)[V]) {
        // dx * sum ks (N P - n0 p0) (forward_solver.py:228-236,:258-269) or
        // dx * sum q_C (mu_n (N-n0) + mu_p (P-p0)) (:239-247,:272-274)
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            const double N = yy[r], P = yy[R + r];
            const double v = (meas == MTRPL_MEAS_TRPL) ? p.ks * (N * P - p.n0p0)
                                                       : kQC * (p.mun * (N - p.n0) + p.mup * (P - p.p0));
            s += (i < L) ? v : 0.0;
        }
        return team_sum<W>(tm, s) * p.dx * sig_scale;
    };

    // ---- tolerances (DESIGN.md "error control") ----
    double rtol_ = fmax(fmin(tune.rtol_scale * ds.rtol, tune.rtol_cap), tune.rtol_floor);
    // TRTS is linear in the carrier density: its 4-decade window spans twice the e-folds of a
    // high-injection PL trace, and the relative error of a monotone decay accumulates per e-fold
    if (meas == MTRPL_MEAS_TRTS) rtol_ *= tune.trts_rtol_factor;
    // SRH with tauP >> tauN (p-type): the decay rate rises from 1/(tauN+tauP) at high injection to 1/tauN
    // at low injection, and a relative error made before that transition is amplified by the ratio of the
    // two rates (d ln(dN/N)/dt = -r'(N) N).  The local tolerance is tightened by that a-priori factor.
    if (tune.amp_ref > 0.0) rtol_ /= fmax(1.0, (1.0 + p.tp / p.tn) / tune.amp_ref);
    rtol_ = fmax(rtol_, tune.rtol_floor);
    const double rtol = rtol_;
    const double atol = rtol * tune.atol_rel * dNmax;
    // the field is a fast slaved variable (dielectric relaxation): its local error is damped, not
    // accumulated, so it gets a looser relative weight than the densities (DESIGN.md "error control")
    // (TRPL only: the photoconductivity weighs dP directly, and its p99 census error doubles with a looser field)
    const double rtol_E = rtol * ((meas == MTRPL_MEAS_TRPL) ? tune.e_weight : 1.0);
    auto rtol_of = [&](int v) { return (v >= 2 * R && v < 3 * R) ? rtol_E : rtol; };
    const double early_until = (nt > 1) ? tune.early_frac * ts[1] : 0.0;
    const double early_inv = 1.0 / fmax(tune.early_factor, 1.0);
    const double newton_tol = (tune.newton_tol > 0.0)
                                  ? tune.newton_tol
                                  : fmax(10.0 * 2.220446049250313e-16 / rtol, fmin(0.03, sqrt(rtol)));

    int n_steps = 0, n_rhs = 0, n_fac = 0, n_jac = 0, n_rej = 0, n_nfail = 0;
    bool clamped = false;
    int st = MTRPL_SOLVE_OK;

    auto emit = [&](int k, double v) {
        // forward_solver.py:190-192: from the first sample below min_y onward -> min_y
        if (!clamped && v < min_y) clamped = true;
        if (clamped) v = min_y;
        if (tid == 0) o[k] = v;
    };

    emit(0, sigof(y));

    double z[6][V];
    double f[V], w[V];
    team_rhs<W, TRAPS>(tm, p, y, f, L);
    n_rhs++;

    // ---- initial step (Hairer, Norsett & Wanner I, II.4, first-order variant) ----
    double h;
    {
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = y[v] / (atol + rtol_of(v) * fabs(y[v]));
        const double d0 = tnorm(w, inv_S);
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = f[v] / (atol + rtol_of(v) * fabs(y[v]));
        const double d1 = tnorm(w, inv_S);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        double y1[V], f1[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y1[v] = y[v] + h0 * f[v];
        team_rhs<W, TRAPS>(tm, p, y1, f1, L);
        n_rhs++;
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = (f1[v] - f[v]) / (atol + rtol_of(v) * fabs(y[v]));
        const double d2 = tnorm(w, inv_S) / h0;
        const double dm = fmax(d1, d2);
        const double h1 = (dm <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : sqrt(0.01 / dm);
        h = fmin(fmin(100.0 * h0, h1), hmax);
        if (nt > 1) h = fmin(h, ts[nt - 1]);
        if (!(h > 0.0)) h = 1e-9;
    }


#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) z[k][v] = (k == 0) ? y[v] : ((k == 1) ? h * f[v] : 0.0);

    int order = 1, n_equal = 0;
    double t = 0.0;
    n_jac++;
    TeamFactors<W, TRAPS> F;
    F.blk = GBLK ? (gblk + (size_t)sidx * (team_blk_bytes<W>() / 16)) : blk_lds;
    bool have_fac = false;
    double rate_prev = 1.0;
    int rate_age = 0;
    int io = 1;
    const double t_end = (nt > 1) ? ts[nt - 1] : 0.0;
    double tq_next = (nt > 1) ? ts[1] : 1e300;
    int since_rej = 1000;  // accepted steps since the last rejected one
    double acp[V];  // the correction of the previous step (order-increase estimate)
#pragma unroll
    for (int v = 0; v < V; ++v) acp[v] = 0.0;

    while (io < nt) {
        if (n_steps >= tune.max_steps) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
        bool accepted = false;
        double acor[V], yn[V], isc[V];  // isc: 1/(atol + rtol*|y_predicted|), the error weights of this step
        double dsm = 0.0, safety = 0.9;
        int attempts = 0;
        while (!accepted) {
            if (++attempts > 60 || !(h > 1e-13 * fmax(t, 1e-3))) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
            // predictor: z <- A z; z[0] is the predicted state, z[1] the predicted h*y'
            double el0 = 1.0, inv_t2 = 1.0;
            MTRPL_ORDER_SWITCH(order, (nq_predict<K, V>(z), el0 = NqC<K>::el0, inv_t2 = NqC<K>::inv_t2))
            const double c = h * el0;
            const double early_scale = (tune.early_factor > 1.0 && t + h <= early_until) ? early_inv : 1.0;
#pragma unroll
            for (int v = 0; v < V; ++v) isc[v] = ctl_rcp(atol + rtol_of(v) * fabs(z[0][v])) * early_scale;
            // simplified Newton on acor:  (I - h l0 J) dx = h f(y) - z[1] - acor,  y = z[0] + l0 acor
            bool converged = false, fresh = false;
            int n_iter = 0;
            while (!converged) {
                if (!have_fac) {
                    // the analytic Jacobian is evaluated inside the factorisation: always at the predictor
                    team_build<W, TRAPS>(tm, p, z[0], c, L, F);
                    have_fac = true;
                    fresh = true;
                    n_fac++;
                    n_jac++;
                    rate_prev = 1.0;  // new matrix: measure the rate afresh
                    rate_age = 0;
                }
#pragma unroll
                for (int v = 0; v < V; ++v) { acor[v] = 0.0; yn[v] = z[0][v]; }
                double dn_old = 0.0;
                for (int k = 0; k < 4; ++k) {
                    n_iter = k + 1;
                    team_rhs<W, TRAPS>(tm, p, yn, f, L);
                    n_rhs++;
                    double b[V], dxv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) b[v] = h * f[v] - z[1][v] - acor[v];
                    team_apply<W, TRAPS>(tm, F, b, dxv);
#pragma unroll
                    for (int v = 0; v < V; ++v) w[v] = el0 * dxv[v] * isc[v];
                    const double dn = tnorm(w, inv_S);
                    if (!(dn < 1e300)) break;  // NaN / inf
                    double rate = rate_prev;
                    const bool negligible = dn < 1e-2 * newton_tol;
                    if (k > 0) {
                        rate = dn * ctl_rcp(dn_old);
                        if (!negligible) {
                            double rp = rate;  // rate^(4-k)
                            for (int q = 1; q < 4 - k; ++q) rp *= rate;
                            if (rate >= 1.0 || rp * dn > newton_tol * (1.0 - rate)) break;
                        }
                        if (rate < 1.0) {
                            rate_prev = (rate_prev >= 1.0 && tune.rate_first_safety > 0.0)
                                            ? fmin(fmax(tune.rate_first_safety * rate, 1e-4), 0.9)
                                            : fmax(0.3 * rate_prev, rate);
                            rate_age = 0;
                        }
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) { acor[v] += dxv[v]; yn[v] += el0 * dxv[v]; }
                    const bool rate_ok = (k > 0) || (rate_prev < 0.9 && rate_age < tune.rate_max_age);
                    if (negligible || (rate_ok && rate < 1.0 && rate * dn < newton_tol * (1.0 - rate))) { converged = true; break; }
                    dn_old = dn;
                }
                if (converged && n_iter == 1) rate_age++;
                if (!converged) {
                    if (fresh) break;
                    have_fac = false;  // refactor with the Jacobian at this predictor, then try once more
                }
            }
            if (!converged) {
                n_nfail++;
                MTRPL_ORDER_SWITCH(order, (nq_retract<K, V>(z), nq_rescale<K, V>(z, 0.5)))
                h *= 0.5;
                n_equal = 0;
                have_fac = false;
                continue;
            }
            // 0.9 * (2*4 + 1) / (2*4 + n_iter), n_iter = 1..4
            safety = (n_iter == 1) ? 0.9 : ((n_iter == 2) ? 0.81 : ((n_iter == 3) ? 8.1 / 11.0 : 0.675));
#pragma unroll
            for (int v = 0; v < V; ++v) w[v] = acor[v] * isc[v];
            dsm = tnorm(w, inv_S) * inv_t2;
            if (dsm > 1.0) {
                n_rej++;
                since_rej = 0;
                const double fac = fmax(0.2, safety * inv_root(dsm, order + 1));
                MTRPL_ORDER_SWITCH(order, (nq_retract<K, V>(z), nq_rescale<K, V>(z, fac)))
                h *= fac;
                n_equal = 0;
                have_fac = false;
            } else {
                accepted = true;
            }
        }
        if (!accepted) break;
        n_steps++;
        n_equal++;
        since_rej++;
        t += h;
        MTRPL_ORDER_SWITCH(order, (nq_correct<K, V>(z, acor)))
        // dense output (Taylor polynomial of the Nordsieck array), forward_solver.py:167
        const double ih = frcp(h);
        while (io < nt) {
            const double tq = tq_next;  // = ts[io], loaded one output ahead
            if (tq > t * (1.0 + 1e-15)) break;
            double yq[V];
            const double s_ = (tq - t) * ih;
            MTRPL_ORDER_SWITCH(order, (nq_dense<K, V>(z, s_, yq)))
            emit(io, sigof(yq));
            io++;
            tq_next = (io < nt) ? ts[io] : 1e300;
        }
        if (n_equal < order + 1) {
#pragma unroll
            for (int v = 0; v < V; ++v) acp[v] = acor[v];
            continue;
        }
        // order / step-size selection (stode: ddn from the last Nordsieck row, dup from acor - acor_prev)
        double sdn = 0.0, sup = 0.0;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const double a_ = (acor[v] - acp[v]) * isc[v];
            sup += a_ * a_;
            acp[v] = acor[v];
        }
        double inv_t1 = 1.0, inv_t3 = 1.0, el_up = 0.0;
        float e0 = -0.5f, e1 = -0.5f, e2 = -0.5f;
        switch (order) {
            case 1: { constexpr int K = 1; for (int v = 0; v < V; ++v) sdn += (z[K][v] * isc[v]) * (z[K][v] * isc[v]);
                      inv_t1 = NqC<K>::inv_t1; inv_t3 = NqC<K>::inv_t3; el_up = NqC<K>::el_up; e0 = NqC<K>::nik0; e1 = NqC<K>::nik1; e2 = NqC<K>::nik2; } break;
            case 2: { constexpr int K = 2; for (int v = 0; v < V; ++v) sdn += (z[K][v] * isc[v]) * (z[K][v] * isc[v]);
                      inv_t1 = NqC<K>::inv_t1; inv_t3 = NqC<K>::inv_t3; el_up = NqC<K>::el_up; e0 = NqC<K>::nik0; e1 = NqC<K>::nik1; e2 = NqC<K>::nik2; } break;
            case 3: { constexpr int K = 3; for (int v = 0; v < V; ++v) sdn += (z[K][v] * isc[v]) * (z[K][v] * isc[v]);
                      inv_t1 = NqC<K>::inv_t1; inv_t3 = NqC<K>::inv_t3; el_up = NqC<K>::el_up; e0 = NqC<K>::nik0; e1 = NqC<K>::nik1; e2 = NqC<K>::nik2; } break;
            case 4: { constexpr int K = 4; for (int v = 0; v < V; ++v) sdn += (z[K][v] * isc[v]) * (z[K][v] * isc[v]);
                      inv_t1 = NqC<K>::inv_t1; inv_t3 = NqC<K>::inv_t3; el_up = NqC<K>::el_up; e0 = NqC<K>::nik0; e1 = NqC<K>::nik1; e2 = NqC<K>::nik2; } break;
            default: { constexpr int K = 5; for (int v = 0; v < V; ++v) sdn += (z[K][v] * isc[v]) * (z[K][v] * isc[v]);
                      inv_t1 = NqC<K>::inv_t1; inv_t3 = NqC<K>::inv_t3; el_up = NqC<K>::el_up; e0 = NqC<K>::nik0; e1 = NqC<K>::nik1; e2 = NqC<K>::nik2; } break;
        }
        team_sum2<W>(tm, sdn, sup);
        const double ddn = (order > 1) ? inv_t1 * norm_sqrt(sdn * inv_S) : 1e300;
        const double dup = (order < 5) ? inv_t3 * norm_sqrt(sup * inv_S) : 1e300;
        const double fm = (order > 1) ? pow_neg_inv(ddn, e0) : 0.0;
        const double f0 = pow_neg_inv(dsm, e1);
        const double fp = (order < 5) ? pow_neg_inv(dup, e2) : 0.0;
        // stode's biases: a change of order has to promise more than staying (1.2 / 1.3 / 1.4)
        int dorder = 0;
        double fbest = f0 * tune.nq_bias_same;
        if (fm * tune.nq_bias_down > fbest) { fbest = fm * tune.nq_bias_down; dorder = -1; }
        if (fp * tune.nq_bias_up > fbest) { fbest = fp * tune.nq_bias_up; dorder = 1; }
        double fac = fmin(10.0, (tune.nq_bias_same < 1.0 ? 1.0 : safety) * fbest);
        // a step that was rejected a moment ago says the estimates behind `fac` are optimistic here (stiff
        // surface terms): grow cautiously for a while instead of hunting (reject / shrink / regrow / reject)
        if (since_rej < tune.nq_rej_memory) fac = fmin(fac, tune.nq_rej_growth);
        if (h * fac > hmax) fac = hmax * ih;
        if (dorder == 0 && fac >= 0.9 && fac < tune.nq_deadband_hi) {
            n_equal = order - 2;  // keep h and the factors; test again three steps on
            continue;
        }
        if (dorder == 1) {
            // new last row z[q+1] = acor * l_q / (q+1)
            switch (order) {
                case 1: for (int v = 0; v < V; ++v) z[2][v] = acor[v] * el_up; break;
                case 2: for (int v = 0; v < V; ++v) z[3][v] = acor[v] * el_up; break;
                case 3: for (int v = 0; v < V; ++v) z[4][v] = acor[v] * el_up; break;
                default: for (int v = 0; v < V; ++v) z[5][v] = acor[v] * el_up; break;
            }
        }
        order += dorder;
        MTRPL_ORDER_SWITCH(order, (nq_rescale<K, V>(z, fac)))
        h *= fac;
        n_equal = 0;
        have_fac = false;
    }

    if (st != MTRPL_SOLVE_OK) {
        // forward_solver.py:168: samples after the last solved time come from a zeroed state
        double zero_[V];
#pragma unroll
        for (int v = 0; v < V; ++v) zero_[v] = 0.0;
        const double zs = sigof(zero_);
        for (; io < nt; ++io) emit(io, zs);
    }
    (void)t_end;
    if (tid == 0) {
        status[sidx] = st;
        if (stats) {
            int32_t *s = stats + 8 * sidx;
            s[0] = n_steps; s[1] = n_rhs; s[2] = n_fac; s[3] = n_jac;
            s[4] = n_rej; s[5] = n_nfail; s[6] = order;
            s[7] = (int32_t)((__builtin_amdgcn_s_memtime() - clk0_) >> 4);  // residency of this solve, 16-tick units of the constant-rate counter
        }
    }
}

}  // namespace mtrpl
