// solve_kernel_blk.hpp -- the batched forward-solve kernel, generalised: W wavefronts
// per solve (nx <= 128*W), std or traps model.  Same algorithm and the same
// reference mapping as solve_kernel.hpp (forward_solver.py:41-203); the device
// functions are in solver_block.hpp.
// (no #pragma once: mtrpl_fp32_pass.hpp compiles this file a second time with `double` read as `float`)
#if (!defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVE_KERNEL_BLK_HPP_F64)) || (defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVE_KERNEL_BLK_HPP_F32))
#ifdef MTRPL_FP32_PASS
#define MTRPL_SOLVE_KERNEL_BLK_HPP_F32
#else
#define MTRPL_SOLVE_KERNEL_BLK_HPP_F64
#endif
#include "solve_kernel.hpp"
#include "solver_block.hpp"

namespace mtrpl {

template <int W, bool TRAPS, bool GBLK>
__global__ __launch_bounds__(64 * W) void solve_kernel_blk(const mtrpl_solve_desc *__restrict__ descs,
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

    double D[8][V];
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
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) D[k][v] = (k == 0) ? y[v] : ((k == 1) ? h * f[v] : 0.0);

    int order = 1, n_equal = 0;
    double t = 0.0;
    double yJ[V];
#pragma unroll
    for (int v = 0; v < V; ++v) yJ[v] = y[v];
    n_jac++;
    TeamFactors<W, TRAPS> F;
    F.blk = GBLK ? (gblk + (size_t)sidx * (team_blk_bytes<W>() / 16)) : blk_lds;
    bool have_fac = false;
    double rate_prev = 1.0;
    int rate_age = 0;
    int io = 1;
    const double t_end = (nt > 1) ? ts[nt - 1] : 0.0;
    double tq_next = (nt > 1) ? ts[1] : 1e300;

    while (io < nt) {
        if (n_steps >= tune.max_steps) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
        bool jac_current = (n_steps == 0);
        bool accepted = false;
        double d[V], yn[V], isc[V];  // isc: 1/(atol + rtol*|y_predicted|), the error weights of this step
        double err_norm = 0.0, safety = 0.9;
        int attempts = 0;
        while (!accepted) {
            if (++attempts > 60 || !(h > 1e-13 * fmax(t, 1e-3))) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
            // predictor
            double yp[V], psi[V];
            double ialpha = 1.0, ec = 1.0;
            // before the first requested sample only the slow content of the solution matters (fast spatial
            // modes are damped by then): optional looser local tolerance for steps that end before it
            const double early_scale = (tune.early_factor > 1.0 && t + h <= early_until) ? early_inv : 1.0;
            MTRPL_ORDER_SWITCH(order, (ndf_predict<K, V>(D, yp, psi), ialpha = NdfC<K>::ialpha, ec = NdfC<K>::errc))
            const double c = h * ialpha;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                psi[v] *= ialpha;
                isc[v] = ctl_rcp(atol + rtol_of(v) * fabs(yp[v])) * early_scale;
            }
            // simplified Newton
            bool converged = false;
            int n_iter = 0;
            while (!converged) {
                if (!have_fac) {
                    if (tune.fresh_jac && !jac_current) {
                        // the analytic Jacobian is evaluated inside the factorisation anyway: take it at
                        // the current predictor instead of the point of the last convergence failure
#pragma unroll
                        for (int v = 0; v < V; ++v) yJ[v] = yp[v];
                        n_jac++;
                        jac_current = true;
                    }
                    team_build<W, TRAPS>(tm, p, yJ, c, L, F);
                    have_fac = true;
                    n_fac++;
                    rate_prev = 1.0;  // new matrix: measure the rate afresh
                    rate_age = 0;
                }
#pragma unroll
                for (int v = 0; v < V; ++v) { d[v] = 0.0; yn[v] = yp[v]; }
                double dn_old = 0.0;
                for (int k = 0; k < 4; ++k) {
                    n_iter = k + 1;
                    team_rhs<W, TRAPS>(tm, p, yn, f, L);
                    n_rhs++;
                    double b[V], dy[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) b[v] = c * f[v] - psi[v] - d[v];
                    team_apply<W, TRAPS>(tm, F, b, dy);
#pragma unroll
                    for (int v = 0; v < V; ++v) w[v] = dy[v] * isc[v];
                    const double dn = tnorm(w, inv_S);
                    if (!(dn < 1e300)) break;  // NaN / inf
                    // contraction rate: measured from the 2nd iteration on; for the 1st
                    // iteration the rate of the previous steps with these same factors
                    // is carried over (Brown, Byrne & Hindmarsh 1989, VODE "crate")
                    double rate = rate_prev;
                    // a correction two orders below the Newton tolerance ends the iteration whatever the
                    // rate: near equilibrium (deep tail of a thin film at low fluence) successive
                    // corrections are rounding noise of equal size, their ratio ~1 reads as divergence and
                    // the step was halved and refactorised up to 200 times per solve for nothing
                    const bool negligible = dn < 1e-2 * newton_tol;
                    if (k > 0) {
                        rate = dn * ctl_rcp(dn_old);
                        if (!negligible) {
                            double rp = rate;  // rate^(4-k)
                            for (int q = 1; q < 4 - k; ++q) rp *= rate;
                            if (rate >= 1.0 || rp * dn > newton_tol * (1.0 - rate)) break;
                        }
                        if (rate < 1.0) {  // (a ratio of two noise-level corrections is not a measurement)
                            // first measurement with new factors: take it (x rate_first_safety) instead
                            // of smoothing down from 1 over several two-iteration steps
                            rate_prev = (rate_prev >= 1.0 && tune.rate_first_safety > 0.0)
                                            ? fmin(fmax(tune.rate_first_safety * rate, 1e-4), 0.9)
                                            : fmax(0.3 * rate_prev, rate);
                            rate_age = 0;
                        }
                    }
#pragma unroll
                    for (int v = 0; v < V; ++v) { yn[v] += dy[v]; d[v] += dy[v]; }
                    const bool rate_ok = (k > 0) || (rate_prev < 0.9 && rate_age < tune.rate_max_age);
                    if (negligible || (rate_ok && rate < 1.0 && rate * dn < newton_tol * (1.0 - rate))) { converged = true; break; }
                    dn_old = dn;
                }
                if (converged && n_iter == 1) rate_age++;
                if (!converged) {
                    if (jac_current) break;
#pragma unroll
                    for (int v = 0; v < V; ++v) yJ[v] = yp[v];
                    n_jac++;
                    jac_current = true;
                    have_fac = false;
                }
            }
            if (!converged) {
                n_nfail++;
                h *= 0.5;
                change_D<V>(D, order, 0.5, lane);
                n_equal = 0;
                have_fac = false;
                continue;
            }
            safety = 0.9 * (2.0 * 4 + 1) / (2.0 * 4 + n_iter);
#pragma unroll
            for (int v = 0; v < V; ++v) w[v] = ec * d[v] * isc[v];
            err_norm = tnorm(w, inv_S);
            if (err_norm > 1.0) {
                n_rej++;
                const double fac = fmax(0.2, safety * inv_root(err_norm, order + 1));
                h *= fac;
                change_D<V>(D, order, fac, lane);
                n_equal = 0;
                have_fac = false;
            } else {
                accepted = true;
            }
        }
        if (!accepted) break;
        n_steps++;
        n_equal++;
        t += h;
        // update differences: D[order+2] = d - D[order+1]; D[order+1] = d; D[i] += D[i+1]
        MTRPL_ORDER_SWITCH(order, (ndf_update<K, V>(D, d)))
        // dense output (polynomial through the last order+1 steps), forward_solver.py:167
        const double ih = frcp(h);
        while (io < nt) {
            const double tq = tq_next;  // = ts[io], loaded one output ahead
            if (tq > t * (1.0 + 1e-15)) break;
            double yq[V];
            MTRPL_ORDER_SWITCH(order, (ndf_dense<K, V>(D, tq, t, h, ih, yq)))
            emit(io, sigof(yq));
            io++;
            tq_next = (io < nt) ? ts[io] : 1e300;
        }
        if (n_equal < order + 1) continue;
        // order / step-size selection
        double sm = 0.0, sp = 0.0;
        double ec_lo = 0.0, ec_hi = 0.0;
        MTRPL_ORDER_SWITCH(order, (ndf_sel<K, V>(D, isc, sm, sp), ec_lo = NdfC<K>::errc_lo, ec_hi = NdfC<K>::errc_hi))
        team_sum2<W>(tm, sm, sp);
        const double em = (order > 1) ? ec_lo * norm_sqrt(sm * inv_S) : 1e300;
        const double ep = (order < 5) ? ec_hi * norm_sqrt(sp * inv_S) : 1e300;
        const double fm = (order > 1) ? inv_root(em, order) : 0.0;
        const double f0 = inv_root(err_norm, order + 1);
        const double fp = (order < 5) ? inv_root(ep, order + 2) : 0.0;
        int dorder = 0;
        double fbest = f0;
        if (fm > fbest) { fbest = fm; dorder = -1; }
        if (fp > fbest) { fbest = fp; dorder = 1; }
        double fac = fmin(10.0, safety * fbest);
        if (h * fac > hmax) fac = hmax * ih;
        if (dorder == 0 && fac >= 0.9 && fac < 1.2) {
            n_equal = order - 2;  // keep h and the factors; test again three steps on
            continue;
        }
        order += dorder;
        h *= fac;
        change_D<V>(D, order, fac, lane);
        n_equal = 0;
        have_fac = false;
    }

    if (st != MTRPL_SOLVE_OK) {
        // forward_solver.py:168: samples after the last solved time come from a zeroed state
        double z[V];
#pragma unroll
        for (int v = 0; v < V; ++v) z[v] = 0.0;
        const double zs = sigof(z);
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

#endif  // MTRPL_SOLVE_KERNEL_BLK_HPP
