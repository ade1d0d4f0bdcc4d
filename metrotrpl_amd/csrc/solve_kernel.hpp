// solve_kernel.hpp -- the batched forward-solve kernel (one wave64 per solve).
// Mirrors forward_solver.solve (forward_solver.py:41-203): initial condition
// (:100-122), stiff integration sampled at g.tSteps (:163-168), PL / TRTS
// spatial integral (:186-197, :228-274), unit scale and min_y clamp (:189-192).
// (no #pragma once: mtrpl_fp32_pass.hpp compiles this file a second time with `double` read as `float`)
#if (!defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVE_KERNEL_HPP_F64)) || (defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVE_KERNEL_HPP_F32))
#ifdef MTRPL_FP32_PASS
#define MTRPL_SOLVE_KERNEL_HPP_F32
#else
#define MTRPL_SOLVE_KERNEL_HPP_F64
#endif
#include "solver_device.hpp"

// Diagnostic build only (-DMTRPL_STAMPS): s_memtime per phase; the real kernel has no stamps.
#ifndef MTRPL_STAMP
#ifdef MTRPL_STAMPS
#define MTRPL_STAMP(acc)                                        \
    do {                                                        \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc += now_ - tlast_;                                   \
        tlast_ = now_;                                          \
    } while (0)
#else
#define MTRPL_STAMP(acc) do { } while (0)
#endif
#endif

namespace mtrpl {

struct SolveTuning {
    double rtol_scale;  // internal rtol = clamp(rtol_scale * desc.rtol, rtol_floor, rtol_cap)
    double rtol_cap;
    double rtol_floor;  // below ~3e-9 the Newton increments reach the fp64 noise of the flux differences
    double atol_rel;    // internal atol = rtol_int * atol_rel * max(N0, P0 over the film)
    int max_steps;
    double newton_tol;  // <= 0: SciPy's rule max(10 eps/rtol, min(0.03, sqrt(rtol))); else this value
    int rate_max_age;   // consecutive 1-iteration Newton solves allowed on a carried-over rate (0: always >= 2)
    double rate_first_safety;  // > 0: after a refactorisation carry (this x first measured rate); 0: VODE smoothing only
    double e_weight;    // relative error weight of the field components = e_weight * rtol_int
    int honor_hmax;     // 1: desc.hmax caps the step (the reference's max_step); 0: error control alone
    double trts_rtol_factor;  // internal rtol of TRTS solves = this x that of TRPL solves (not below rtol_floor)
    double amp_ref;           // > 0: internal rtol /= max(1, (1 + tauP/tauN) / amp_ref); 0: off
    int fresh_jac;            // 1: every refactorisation re-evaluates the Jacobian at the current predictor
    double nq_deadband_hi;  // Nordsieck kernels: a step factor in [0.9, this) keeps h and the factors
    int nq_rej_memory; double nq_rej_growth;  // Nordsieck kernel: growth cap for this many steps after a rejection
    double nq_bias_same, nq_bias_down, nq_bias_up;  // Nordsieck kernel: multipliers of the step factors (stode: 1/1.2, 1/1.3, 1/1.4)
    double early_factor;      // > 1: local tolerance x this for steps ending before early_frac * (first sample time)
    double early_frac;
};

// Norms only steer step-size, order and Newton decisions: the f32 hardware square root (1 ulp of
// f32) is plenty and replaces the fp64 v_sqrt + refinement sequence on the critical path.
__device__ inline double norm_sqrt(double x) { return (double)__builtin_amdgcn_sqrtf((float)x); }

// 1/x to f32 accuracy for quantities that only steer the controller (error weights, contraction rates)
__device__ inline double ctl_rcp(double x) { return (double)__builtin_amdgcn_rcpf((float)x); }

// NDF constants of order K as compile-time values (Shampine & Reichelt 1997): inside
// MTRPL_ORDER_SWITCH they cost nothing, where the run-time select chains of ndf_kappa/ndf_gamma and the
// fp64 division in ndf_errc sat on the scalar critical path of every step.
template <int K>
struct NdfC {
    static constexpr double kp[7] = {0.0, -0.1850, -1.0 / 9.0, -0.0823, -0.0415, 0.0, 0.0};
    static constexpr double gm[7] = {0.0, 1.0, 1.5, 1.5 + 1.0 / 3.0, 1.5 + 1.0 / 3.0 + 0.25, 1.5 + 1.0 / 3.0 + 0.25 + 0.2, 0.0};
    static constexpr double alpha = (1.0 - kp[K]) * gm[K];
    static constexpr double ialpha = 1.0 / alpha;
    static constexpr double errc = kp[K] * gm[K] + 1.0 / (double)(K + 1);
    static constexpr double errc_lo = (K > 1) ? kp[K > 1 ? K - 1 : 1] * gm[K > 1 ? K - 1 : 1] + 1.0 / (double)K : 0.0;
    static constexpr double errc_hi = (K < 5) ? kp[K + 1] * gm[K + 1] + 1.0 / (double)(K + 2) : 0.0;
    static constexpr float nik0 = -1.0f / (float)K, nik1 = -1.0f / (float)(K + 1), nik2 = -1.0f / (float)(K + 2);
};

// x^e (e = -1/k as a compile-time constant): f32 hardware log2 / exp2
__device__ inline double pow_neg_inv(double x, float e)
{
    const float xf = (float)fmin(fmax(x, 1e-30), 1e30);
    return (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf(xf) * e);
}

// RMS norm over the 3nx+1 ODE components; v is already divided by the weights.
template <int V>
__device__ inline double wnorm(const double (&v)[V], double inv_S)
{
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < V; ++k) s += v[k] * v[k];
    return uniform(norm_sqrt(wave_sum(s) * inv_S));
}

// value of lane `src` (a compile-time constant after unrolling) as a wave-uniform scalar
__device__ inline double lane_value(double v, int src)
{
    union { double d; int i[sizeof(double) / 4]; } u;
    u.d = v;
#pragma unroll
    for (unsigned k = 0; k < sizeof(double) / 4; ++k) u.i[k] = __builtin_amdgcn_readlane(u.i[k], src);
    return u.d;
}

// D[0..order] <- (R(factor) U)^T D[0..order]  (rescale the backward differences
// to a step factor*h; Shampine & Reichelt eq. for quasi-constant step size).
// The (order+1)^2 coefficients are wave-uniform: lane 6*j+i computes RU[j][i],
// readlane broadcasts it as a scalar operand.
template <int V>
__device__ inline void change_D(double (&D)[8][V], int order, double factor, int lane)
{
    const int j = lane / 6, i = lane % 6;  // coefficient (j,i), lanes 0..35
    double ru = 0.0;
    if (lane < 36 && j <= order && i <= order) {
        // R[j][m] = prod_{k=1..j} (k-1-factor*m)/k, U[m][i] = prod_{k=1..m} (k-1-i)/k ; column/row 0 special.
        // Fully unrolled with predicates and reciprocal constants: the run-time loop bounds and fp64
        // divisions of the straightforward form cost ~10k cycles per call.
        constexpr double inv[6] = {0.0, 1.0, 0.5, 1.0 / 3.0, 0.25, 0.2};
        const double fi = (double)i;
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            double rjm = 1.0, umi = 1.0;
            const double fm_ = factor * (double)m;
#pragma unroll
            for (int k = 1; k < 6; ++k) {
                rjm *= (k <= j) ? ((double)(k - 1) - fm_) * inv[k] : 1.0;
                umi *= (k <= m) ? ((double)(k - 1) - fi) * inv[k] : 1.0;
            }
            if (m == 0 && j > 0) rjm = 0.0;
            if (i == 0 && m > 0) umi = 0.0;
            ru += (m <= order) ? rjm * umi : 0.0;
        }
    }
    double Dn[6][V];
#pragma unroll
    for (int ii = 0; ii < 6; ++ii) {
#pragma unroll
        for (int v = 0; v < V; ++v) Dn[ii][v] = 0.0;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) {
            const double cf = lane_value(ru, jj * 6 + ii);
#pragma unroll
            for (int v = 0; v < V; ++v) Dn[ii][v] += cf * D[jj][v];
        }
    }
#pragma unroll
    for (int ii = 0; ii < 6; ++ii)
        if (ii <= order) {
#pragma unroll
            for (int v = 0; v < V; ++v) D[ii][v] = Dn[ii][v];
        }
}

// ---- order-specialised pieces of the NDF step -----------------------------------
// `order` is wave-uniform, so a scalar switch picks one of five fully static
// bodies: no per-row predication, and every D[k][v] has a fixed register.
#ifndef MTRPL_ORDER_SWITCH
#define MTRPL_ORDER_SWITCH(order_, CALL)      \
    switch (order_) {                          \
        case 1: { constexpr int K = 1; CALL; } break; \
        case 2: { constexpr int K = 2; CALL; } break; \
        case 3: { constexpr int K = 3; CALL; } break; \
        case 4: { constexpr int K = 4; CALL; } break; \
        default: { constexpr int K = 5; CALL; } break; \
    }
#endif

template <int K, int V>
__device__ inline void ndf_predict(const double (&D)[8][V], double (&yp)[V], double (&psi)[V])
{
    constexpr double g[6] = {0.0, 1.0, 1.5, 1.5 + 1.0 / 3.0, 1.5 + 1.0 / 3.0 + 0.25, 1.5 + 1.0 / 3.0 + 0.25 + 0.2};
#pragma unroll
    for (int v = 0; v < V; ++v) { yp[v] = D[0][v]; psi[v] = 0.0; }
#pragma unroll
    for (int j = 1; j <= K; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) { yp[v] += D[j][v]; psi[v] += g[j] * D[j][v]; }
}

// D[K+2] = d - D[K+1]; D[K+1] = d; D[i] += D[i+1], i = K..0
template <int K, int V>
__device__ inline void ndf_update(double (&D)[8][V], const double (&d)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) {
        D[K + 2][v] = d[v] - D[K + 1][v];
        D[K + 1][v] = d[v];
    }
#pragma unroll
    for (int i = K; i >= 0; --i)
#pragma unroll
        for (int v = 0; v < V; ++v) D[i][v] += D[i + 1][v];
}

// value of the order-K interpolating polynomial at tq (t = end of the step just taken)
template <int K, int V>
__device__ inline void ndf_dense(const double (&D)[8][V], double tq, double t, double h, double ih, double (&yq)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) yq[v] = D[0][v];
    double pj = 1.0;
#pragma unroll
    for (int j = 1; j <= K; ++j) {
        pj *= (tq - (t - (double)(j - 1) * h)) * ih * (1.0 / (double)j);
#pragma unroll
        for (int v = 0; v < V; ++v) yq[v] += D[j][v] * pj;
    }
}

// squared weighted norms of D[K] (order-1 candidate) and D[K+2] (order+1 candidate)
template <int K, int V>
__device__ inline void ndf_sel(const double (&D)[8][V], const double (&isc)[V], double &sm, double &sp)
{
    sm = 0.0;
    sp = 0.0;
#pragma unroll
    for (int v = 0; v < V; ++v) {
        sm += (D[K][v] * isc[v]) * (D[K][v] * isc[v]);
        sp += (D[K + 2][v] * isc[v]) * (D[K + 2][v] * isc[v]);
    }
}

// signal of one state: dx * sum_i ks (N_i P_i - n0 p0)  (forward_solver.py:228-236,:258-269)
// or dx * sum_i q_C (mu_n (N_i-n0) + mu_p (P_i-p0))      (:239-247,:272-274)
template <int R>
__device__ inline double signal_of(const Par &p, const double (&y)[3 * R], int meas, int lane, int L, double scale)
{
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
        const double N = y[r], P = y[R + r];
        const double v = (meas == MTRPL_MEAS_TRPL) ? p.ks * (N * P - p.n0p0)
                                                   : kQC * (p.mun * (N - p.n0) + p.mup * (P - p.p0));
        s += (i < L) ? v : 0.0;
    }
    return uniform(wave_sum(s) * p.dx * scale);
}

#ifndef MTRPL_FP32_PASS  // the tuned wave kernel exists in fp64 only
__global__ __launch_bounds__(64, 1) void solve_kernel_r2(const mtrpl_solve_desc *__restrict__ descs,
                                                          const double *__restrict__ ini,
                                                          const double *__restrict__ tsteps,
                                                          double *__restrict__ out, int32_t *__restrict__ status,
                                                          int32_t *__restrict__ stats, int64_t B, SolveTuning tune)
{
    constexpr int R = 2, V = 6;
    const int64_t sidx = blockIdx.x;
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if (sidx >= B) return;
    const int lane = threadIdx.x & 63;
    const mtrpl_solve_desc &ds = descs[sidx];
    const int L = ds.nx, nt = ds.nt, meas = ds.meas;
    const double dx = ds.thickness / (double)L;  // sim_utils.py:265
    const Par p = make_par(ds.par, dx);
    const double *ts = tsteps + ds.t_off;
    double *o = out + ds.out_off;
    const double sig_scale = (meas == MTRPL_MEAS_TRPL) ? 1e23 : 1e9;  // forward_solver.py:189,:197
    const double min_y = ds.min_y;
    const double hmax = tune.honor_hmax ? ds.hmax : 1e300;
    const double inv_S = 1.0 / (double)(3 * L + 1);

    // ---- initial condition, forward_solver.py:100-122 ----
    double y[V];
    double dNmax = 0.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = lane * R + r;
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
        dNmax = fmax(dNmax, (i < L) ? fmax(y[r], y[R + r]) : 0.0);
    }
    {
        // E_field (:26-38, :122): E_i = Lambda*dx*cumsum((P-p0)-(N-n0))[i-1]; prefix scan over the wave
        double own[R], run = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            own[r] = (i < L) ? ((y[R + r] - p.p0) - (y[r] - p.n0)) : 0.0;
            run += own[r];
        }
        double incl = run;  // inclusive scan of per-lane totals
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double t = shf(incl, (lane - s) & 63);
            incl += (lane >= s) ? t : 0.0;
        }
        double excl = incl - run;  // sum over nodes < lane*R
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            y[2 * R + r] = (i > 0 && i < L) ? p.Lam * dx * excl : 0.0;
            excl += own[r];
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) dNmax = fmax(dNmax, __shfl_xor(dNmax, s, 64));

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
        if (lane == 0) o[k] = v;
    };

    emit(0, signal_of<R>(p, y, meas, lane, L, sig_scale));

    double D[8][V];
    double f[V], w[V];
    rhs_std<R>(p, y, f, lane, L);
    n_rhs++;

    // ---- initial step (Hairer, Norsett & Wanner I, II.4, first-order variant) ----
    double h;
    {
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = y[v] / (atol + rtol_of(v) * fabs(y[v]));
        const double d0 = wnorm<V>(w, inv_S);
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = f[v] / (atol + rtol_of(v) * fabs(y[v]));
        const double d1 = wnorm<V>(w, inv_S);
        double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        double y1[V], f1[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y1[v] = y[v] + h0 * f[v];
        rhs_std<R>(p, y1, f1, lane, L);
        n_rhs++;
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = (f1[v] - f[v]) / (atol + rtol_of(v) * fabs(y[v]));
        const double d2 = wnorm<V>(w, inv_S) / h0;
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
    __shared__ double2 fac_lds[kFactorLdsBytes / 16];
    Factors<R> F;
    F.lds = fac_lds;
    bool have_fac = false;
    double rate_prev = 1.0;
    int rate_age = 0;
    int io = 1;
    const double t_end = (nt > 1) ? ts[nt - 1] : 0.0;
    double tq_next = (nt > 1) ? ts[1] : 1e300;

#ifdef MTRPL_STAMPS
    unsigned long long tlast_ = __builtin_amdgcn_s_memtime(), ph_sel = 0, ph_pred = 0, ph_fac = 0, ph_newt = 0,
                       ph_err = 0, ph_upd = 0, ph_dense = 0;
#endif
    while (io < nt) {
        MTRPL_STAMP(ph_sel);
        if (n_steps >= tune.max_steps) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
        bool jac_current = (n_steps == 0);
        bool accepted = false;
        double d[V], yn[V], isc[V];  // isc: 1/(atol + rtol*|y_predicted|), the error weights of this step
        double err_norm = 0.0, safety = 0.9;
        int attempts = 0;
        while (!accepted) {
            if (++attempts > 60 || !(h > 1e-13 * fmax(t, 1e-3))) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
            // predictor
            MTRPL_STAMP(ph_err);
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
            MTRPL_STAMP(ph_pred);
            bool converged = false;
            int n_iter = 0;
            while (!converged) {
                if (!have_fac) {
                    MTRPL_STAMP(ph_newt);
                    if (tune.fresh_jac && !jac_current) {
                        // the analytic Jacobian is evaluated inside the factorisation anyway: take it at
                        // the current predictor instead of the point of the last convergence failure
#pragma unroll
                        for (int v = 0; v < V; ++v) yJ[v] = yp[v];
                        n_jac++;
                        jac_current = true;
                    }
                    build_factors(p, yJ, c, lane, L, F);
                    have_fac = true;
                    n_fac++;
                    rate_prev = 1.0;  // new matrix: measure the rate afresh
                    rate_age = 0;
                    MTRPL_STAMP(ph_fac);
                }
#pragma unroll
                for (int v = 0; v < V; ++v) { d[v] = 0.0; yn[v] = yp[v]; }
                double dn_old = 0.0;
                for (int k = 0; k < 4; ++k) {
                    n_iter = k + 1;
                    rhs_std<R>(p, yn, f, lane, L);
                    n_rhs++;
                    double b[V], dy[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) b[v] = c * f[v] - psi[v] - d[v];
                    apply_factors(F, b, dy, lane, L);
#pragma unroll
                    for (int v = 0; v < V; ++v) w[v] = dy[v] * isc[v];
                    const double dn = wnorm<V>(w, inv_S);
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
            MTRPL_STAMP(ph_newt);
            // 0.9 * (2*4 + 1) / (2*4 + n_iter), n_iter = 1..4
            safety = (n_iter == 1) ? 0.9 : ((n_iter == 2) ? 0.81 : ((n_iter == 3) ? 8.1 / 11.0 : 0.675));
#pragma unroll
            for (int v = 0; v < V; ++v) w[v] = ec * d[v] * isc[v];
            err_norm = wnorm<V>(w, inv_S);
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
        MTRPL_STAMP(ph_err);
        n_steps++;
        n_equal++;
        t += h;
        // update differences: D[order+2] = d - D[order+1]; D[order+1] = d; D[i] += D[i+1]
        MTRPL_ORDER_SWITCH(order, (ndf_update<K, V>(D, d)))
        // dense output (polynomial through the last order+1 steps), forward_solver.py:167
        MTRPL_STAMP(ph_upd);
        const double ih = frcp(h);
        while (io < nt) {
            const double tq = tq_next;  // = ts[io], loaded one output ahead (its latency hides behind a step)
            if (tq > t * (1.0 + 1e-15)) break;
            double yq[V];
            MTRPL_ORDER_SWITCH(order, (ndf_dense<K, V>(D, tq, t, h, ih, yq)))
            emit(io, signal_of<R>(p, yq, meas, lane, L, sig_scale));
            io++;
            tq_next = (io < nt) ? ts[io] : 1e300;
        }
        MTRPL_STAMP(ph_dense);
        if (n_equal < order + 1) continue;
        // order / step-size selection
        double sm = 0.0, sp = 0.0;
        double ec_lo = 0.0, ec_hi = 0.0;
        float e0 = -0.5f, e1 = -0.5f, e2 = -0.5f;
        MTRPL_ORDER_SWITCH(order, (ndf_sel<K, V>(D, isc, sm, sp), ec_lo = NdfC<K>::errc_lo, ec_hi = NdfC<K>::errc_hi,
                                   e0 = NdfC<K>::nik0, e1 = NdfC<K>::nik1, e2 = NdfC<K>::nik2))
        const double em = (order > 1) ? ec_lo * uniform(norm_sqrt(wave_sum(sm) * inv_S)) : 1e300;
        const double ep = (order < 5) ? ec_hi * uniform(norm_sqrt(wave_sum(sp) * inv_S)) : 1e300;
        const double fm = (order > 1) ? pow_neg_inv(em, e0) : 0.0;
        const double f0 = pow_neg_inv(err_norm, e1);
        const double fp = (order < 5) ? pow_neg_inv(ep, e2) : 0.0;
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
        const double zs = signal_of<R>(p, z, meas, lane, L, sig_scale);
        for (; io < nt; ++io) emit(io, zs);
    }
    (void)t_end;
    if (lane == 0) {
        status[sidx] = st;
        if (stats) {
            int32_t *s = stats + 8 * sidx;
            s[0] = n_steps; s[1] = n_rhs; s[2] = n_fac; s[3] = n_jac;
            s[4] = n_rej; s[5] = n_nfail; s[6] = order;
            s[7] = (int32_t)((__builtin_amdgcn_s_memtime() - clk0_) >> 4);  // residency of this solve, 16-tick units of the constant-rate counter
#ifdef MTRPL_STAMPS
            s[1] = (int)(ph_sel >> 10); s[2] = (int)(ph_pred >> 10); s[3] = (int)(ph_fac >> 10);
            s[4] = (int)(ph_newt >> 10); s[5] = (int)(ph_err >> 10); s[6] = (int)(ph_upd >> 10);
            s[7] = (int)(ph_dense >> 10);
#endif
        }
    }
}

#endif  // !MTRPL_FP32_PASS

}  // namespace mtrpl

#endif  // MTRPL_SOLVE_KERNEL_HPP
