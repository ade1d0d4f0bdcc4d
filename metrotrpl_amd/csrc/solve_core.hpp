// solve_core.hpp -- the ONE time integrator of the forward-solve kernels.
//
// forward_solver.solve (forward_solver.py:41-203) = initial condition (:100-122), stiff integration
// sampled at g.tSteps (:162-168), PL / TRTS spatial integral (:186-197, :228-274), unit scale and
// min_y clamp (:189-192).  The reference integrates with SciPy's LSODA; here the stiff half of LSODA
// is re-built around the structure of this system: a variable-order (1..5) fixed-coefficient BDF in
// NORDSIECK form, as Hindmarsh's LSODE (`cfode` coefficients, `stode` step): history z[j] = h^j y^(j)/j!,
// predicted in place by the Pascal triangle, corrected by z += l*acor, rescaled by eta^j on a step
// change, interpolated by Horner's rule for the dense output; simplified Newton with an analytic
// Jacobian whose linear algebra belongs to the backend.
//
// integrate<BK>() is the whole per-solve program.  A backend BK owns the distribution of the grid
// nodes over lanes / waves and everything that needs neighbour or whole-grid data:
//
//   BK::real, BK::R (nodes per thread), BK::V (values per thread = 3R std / 4R traps), BK::TRAPS
//   bk.set_problem(par, L)          model parameters and grid size of this solve
//   bk.node(r)                      global node index of slot r of this thread
//   bk.publish(y, zE)               make this thread group's boundary state (y and, per node slot, zE = z1_E + acor_E)
//                                   visible to rhs()/build()/apply()
//                                   of the neighbouring group (no-op where shuffles reach everything)
//   bk.rhs(y, f)                    f = dy/dt                              (forward_solver.py:332-418)
//   bk.build(y, c, h)               factorise I - c*J(y)
//   bk.apply(b, x)                  x = (I - c*J)^-1 b with the kept factors
//   bk.reduce<NS>(s, y, zE)         s[k] <- sum over the whole grid, identical in every thread;
//                                   doubles as publish(y, zE) for the next rhs()
//   bk.scan_excl(v), bk.max_all(v)  exclusive prefix over the threads before this one / maximum
//   bk.out_begin / out_sample / out_finish   emission of the output samples (scale, min_y clamp)
//   bk.leader()                     the one thread that writes status / stats
//
// Value layout of a thread: y[r] = N, y[R+r] = P, y[2R+r] = E on the left face of node r,
// y[3R+r] = N_trap (traps model).
#pragma once
#include "model_physics.hpp"

// Diagnostic build only (-DMTRPL_STAMPS): s_memtime per phase; the real kernel has no stamps.
#ifdef MTRPL_STAMPS
#define MTRPL_STAMP(acc)                                              \
    do {                                                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
        acc += now_ - tlast_;                                         \
        tlast_ = now_;                                                \
    } while (0)
#elif defined(MTRPL_MARKS)  // analysis build: phase boundaries as comments in the assembly listing
#define MTRPL_STAMP(acc) asm volatile("; MARK " #acc)
#else
#define MTRPL_STAMP(acc) do { } while (0)
#endif

// `order` is wave-uniform.  The Nordsieck rows above the current order are kept at exactly zero, so every
// VECTOR operation on the history (Pascal predictor, retraction, rescaling, Horner interpolation) is the
// order-5 one whatever the order is -- one body, no joins of 36-register arrays behind a five-way switch --
// and the order only selects SCALAR coefficients, through this switch:
#define MTRPL_ORDER_SWITCH(order_, CALL)               \
    switch (order_) {                                   \
        case 1: { constexpr int K = 1; CALL; } break;   \
        case 2: { constexpr int K = 2; CALL; } break;   \
        case 3: { constexpr int K = 3; CALL; } break;   \
        case 4: { constexpr int K = 4; CALL; } break;   \
        default: { constexpr int K = 5; CALL; } break;  \
    }

namespace mtrpl {

struct SolveTuning {
    double rtol_scale;  // internal rtol = min(rtol_cap, rtol_scale * desc.rtol), then the policy factors below
    double rtol_cap;    // upper end of the internal rtol: the accuracy contract is the converged answer
    double rtol_floor;  // the policy factors (TRTS, tauP/tauN) never tighten below this ...
    double rtol_min;    // ... but a caller's own RTOL does, down to this noise floor (then status 2)
    double rtol_force;  // > 0: internal rtol = exactly this for every solve (tolerance sweeps); 0: the policy
    double atol_rel;    // absolute tolerance of the FIELD components = rtol_int * atol_rel * max(N0, P0 over the film)
    double atol_rel_np; // ... and of the densities = min(rtol_int * atol_rel_np * max(N0, P0), desc.atol)
    int max_steps;
    double newton_tol;  // <= 0: SciPy's rule max(10 eps/rtol, min(0.03, sqrt(rtol))); else this value
    int rate_max_age;   // consecutive 1-iteration Newton solves allowed on a carried-over rate (0: always >= 2)
    double rate_first_safety;  // > 0: after a refactorisation carry (this x first measured rate); 0: VODE smoothing only
    double e_weight;    // relative error weight of the field components = e_weight * rtol_int (TRPL solves)
    int honor_hmax;     // 1: desc.hmax caps the step (the reference's max_step); 0: error control alone
    double hmax_deep;   // ... until the signal has fallen below this fraction of its first sample (0: always)
    double trts_rtol_factor;  // internal rtol of TRTS solves = this x that of TRPL solves
    double amp_ref;           // > 0: internal rtol /= max(1, (1 + tauP/tauN) / amp_ref); 0: off
    double nq_deadband_hi;    // a step factor in [0.9, this) keeps h and the factors
    int nq_rej_memory; double nq_rej_growth;  // growth cap for this many steps after a rejection
    double nq_bias_same, nq_bias_down, nq_bias_up;  // multipliers of the step factors (stode: 1/1.2, 1/1.3, 1/1.4)
    double early_factor;      // > 1: local tolerance x this for steps ending before early_frac * (first sample time)
    double early_frac;
};

// The shipped settings as compile-time constants: a kernel instantiated on this type instead of on the
// run-time struct keeps none of them in (spilled) scalar registers.  Single source of the defaults.
struct DefaultTune {
    static constexpr double rtol_scale = 0.1, rtol_cap = 3e-9, rtol_floor = 3e-10, rtol_min = 3e-11, rtol_force = 0.0;
    static constexpr double atol_rel = 1e-4, atol_rel_np = 1e-8;
    static constexpr int max_steps = 2000000;
    static constexpr double newton_tol = 0.1;
    static constexpr int rate_max_age = 8;
    static constexpr double rate_first_safety = 4.0, e_weight = 3.0;
    static constexpr int honor_hmax = 1;
    static constexpr double hmax_deep = 1e-11, trts_rtol_factor = 0.5, amp_ref = 16.0, nq_deadband_hi = 1.2;
    static constexpr int nq_rej_memory = 80;
    static constexpr double nq_rej_growth = 1.2, nq_bias_same = 1.0, nq_bias_down = 1.0, nq_bias_up = 1.0;
    static constexpr double early_factor = 10.0, early_frac = 0.5;
    static SolveTuning make()
    {
        return SolveTuning{rtol_scale, rtol_cap, rtol_floor, rtol_min, rtol_force, atol_rel, atol_rel_np, max_steps,
                           newton_tol, rate_max_age, rate_first_safety, e_weight, honor_hmax, hmax_deep, trts_rtol_factor,
                           amp_ref, nq_deadband_hi, nq_rej_memory, nq_rej_growth, nq_bias_same, nq_bias_down, nq_bias_up,
                           early_factor, early_frac};
    }
};

// ODEPACK cfode, meth = 2: l-vector and error-test constants of the fixed-step BDF of order Q
struct NqTable {
    double el[6][6];   // el[q][j], q = 1..5
    double t1[6], t2[6], t3[6];
};
constexpr NqTable nq_make_table()
{
    NqTable T{};
    double pc[8] = {1.0, 0, 0, 0, 0, 0, 0, 0};
    double rq1fac = 1.0;
    for (int nq = 1; nq <= 5; ++nq) {
        const double fnq = (double)nq;
        pc[nq] = 0.0;
        for (int i = nq; i >= 1; --i) pc[i] = pc[i - 1] + fnq * pc[i];
        pc[0] = fnq * pc[0];
        for (int j = 0; j <= nq; ++j) T.el[nq][j] = pc[j] / pc[1];
        T.el[nq][1] = 1.0;
        T.t1[nq] = rq1fac;
        T.t2[nq] = (double)(nq + 1) / T.el[nq][0];
        T.t3[nq] = (double)(nq + 2) / T.el[nq][0];
        rq1fac = rq1fac / fnq;
    }
    return T;
}
template <int K>
struct NqC {
    static constexpr NqTable T = nq_make_table();
    static constexpr double el0 = T.el[K][0];
    static constexpr double elK = T.el[K][K];  // coefficient of the last row
    static constexpr double inv_t1 = 1.0 / T.t1[K], inv_t2 = 1.0 / T.t2[K], inv_t3 = 1.0 / T.t3[K];
    static constexpr double el_up = T.el[K][K] / (double)(K + 1);  // new last row on an order increase
    static constexpr float nik0 = -1.0f / (float)K, nik1 = -1.0f / (float)(K + 1), nik2 = -1.0f / (float)(K + 2);
};

// z <- A z (Pascal triangle), in place; and its inverse
template <int K, int V, class T>
__device__ inline void nq_predict(T (&z)[6][V])
{
#pragma unroll
    for (int k = 1; k <= K; ++k)
#pragma unroll
        for (int j = K; j >= k; --j)
#pragma unroll
            for (int v = 0; v < V; ++v) z[j - 1][v] += z[j][v];
}
template <int K, int V, class T>
__device__ inline void nq_retract(T (&z)[6][V])
{
#pragma unroll
    for (int k = 1; k <= K; ++k)
#pragma unroll
        for (int j = K; j >= k; --j)
#pragma unroll
            for (int v = 0; v < V; ++v) z[j - 1][v] -= z[j][v];
}
// z[j] += l_j * acor
template <int K, int V, class T>
__device__ inline void nq_correct(T (&z)[6][V], const T (&acor)[V])
{
    constexpr NqTable Tb = nq_make_table();
#pragma unroll
    for (int j = 0; j <= K; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) z[j][v] += (T)Tb.el[K][j] * acor[v];
}
// the same with the l-vector of a run-time order (zero above it): rows above the order stay zero
template <int V, class T>
__device__ inline void nq_correct_rt(T (&z)[6][V], const T (&acor)[V], const T (&el)[6])
{
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int v = 0; v < V; ++v) z[j][v] += el[j] * acor[v];
}
// z[j] *= eta^j
template <int K, int V, class T>
__device__ inline void nq_rescale(T (&z)[6][V], T eta)
{
    T r = (T)1;
#pragma unroll
    for (int j = 1; j <= K; ++j) {
        r *= eta;
#pragma unroll
        for (int v = 0; v < V; ++v) z[j][v] *= r;
    }
}
// y(t_n + s h) = sum_j z[j] s^j
template <int K, int V, class T>
__device__ inline void nq_dense(const T (&z)[6][V], T s, T (&yq)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) yq[v] = z[K][v];
#pragma unroll
    for (int j = K - 1; j >= 0; --j)
#pragma unroll
        for (int v = 0; v < V; ++v) yq[v] = yq[v] * s + z[j][v];
}

// Output samples: unit scale (forward_solver.py:189,:197) and the min_y clamp "from the first sample
// below min_y onward" (:190-192).  One instance per solve, held by the thread(s) that store.
struct Emitter {
    double *o;
    double min_y, scale;
    double deep_below;  // deep_frac x the first sample
    bool clamped, deep;
    __device__ inline void begin(double *o_, double min_y_, double scale_, double deep_frac)
    {
        o = o_; min_y = min_y_; scale = scale_; clamped = false; deep = false;
        deep_below = -deep_frac;  // negative until the first sample fixes the scale
    }
    // total = dx-less grid sum; writer = this thread stores.  Every thread that calls sees the same v,
    // so `deep` is uniform among the callers.
    __device__ inline void put(int k, double total, double dx, bool writer)
    {
        double v = total * dx * scale;
        if (deep_below < 0.0) deep_below = -deep_below * fabs(v);
        else if (!(fabs(v) >= deep_below)) deep = true;
        if (!clamped && v < min_y) clamped = true;
        if (clamped) v = min_y;
        if (writer) o[k] = v;
    }
};

template <class BK, class TUNE>  // TUNE: SolveTuning (run-time options) or DefaultTune (the shipped ones, constants)
__device__ inline void integrate(BK &bk, const mtrpl_solve_desc &ds, const double *__restrict__ ini,
                                 const double *__restrict__ tsteps, double *__restrict__ out,
                                 int32_t *__restrict__ status, int32_t *__restrict__ stats, int64_t sidx,
                                 const TUNE &tune, unsigned long long clk0_)
{
#ifdef MTRPL_SCHED
    const unsigned long long sched_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
    using real = typename BK::real;
    constexpr int R = BK::R, V = BK::V;
    constexpr bool TRAPS = BK::TRAPS;
    const int L = ds.nx, nt = ds.nt, meas = ds.meas;
    const double dx = ds.thickness / (double)L;  // sim_utils.py:265
    bk.set_problem(make_par<real>(ds.par, dx), L);
    const ParT<real> &p = bk.p;
    const double *ts = tsteps + ds.t_off;
    const real hmax = tune.honor_hmax ? (real)ds.hmax : big<real>();
    const real inv_S = (real)(1.0 / (double)((TRAPS ? 4 : 3) * L + 1));
    bk.out_begin(out + ds.out_off, ds.min_y, (meas == MTRPL_MEAS_TRPL) ? 1e23 : 1e9, tune.hmax_deep);  // forward_solver.py:189,:197

    // ---- initial condition, forward_solver.py:100-122 ----
    real y[V];
    real dNmax = (real)0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = bk.node(r);
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
        y[r] = (i < L) ? (real)(dN + ds.par[MTRPL_P_N0]) : (real)1;       // :120
        y[R + r] = (i < L) ? (real)(dN + ds.par[MTRPL_P_P0]) : (real)1;   // :121
        if (TRAPS) y[3 * R + r] = (real)0;                                  // :133 np.zeros_like(N)
        dNmax = fmax(dNmax, (i < L) ? fmax(y[r], y[R + r]) : (real)0);
    }
    {
        // E_field (:26-38, :122): E_i = Lambda*dx*cumsum((P-p0)-(N-n0))[i-1]; prefix scan over the grid
        real own[R], run = (real)0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = bk.node(r);
            own[r] = (i < L) ? ((y[R + r] - p.p0) - (y[r] - p.n0)) : (real)0;
            run += own[r];
        }
        real excl = bk.scan_excl(run);  // sum over the nodes of all threads before this one
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = bk.node(r);
            y[2 * R + r] = (i > 0 && i < L) ? p.Lam * p.dx * excl : (real)0;
            excl += own[r];
        }
    }
    dNmax = bk.max_all(dNmax);

    // ---- tolerances (DESIGN.md "error control") ----
    // The contract is the converged solution, so a loose RTOL does not loosen the integration
    // (rtol_cap); a tight one tightens it, below the policy floor too, down to the fp64 noise floor
    // rtol_min -- a request below that is raised to it and reported as MTRPL_SOLVE_OK_TOL_CLAMPED.
    bool tol_clamped = false;
    real rtol, atol, atol_E;
    {
        const double req = tune.rtol_scale * ds.rtol;
        const double base = fmin(req, tune.rtol_cap);
        double r_;
        if (base >= tune.rtol_floor) {
            r_ = base;
            // TRTS is linear in the carrier density: its 4-decade window spans twice the e-folds of a
            // high-injection PL trace, and the relative error of a monotone decay accumulates per e-fold
            if (meas == MTRPL_MEAS_TRTS) r_ *= tune.trts_rtol_factor;
            // SRH with tauP >> tauN (p-type): the decay rate rises from 1/(tauN+tauP) at high injection to
            // 1/tauN at low injection, and a relative error made before that transition is amplified by the
            // ratio of the two rates (d ln(dN/N)/dt = -r'(N) N): tighten by that a-priori factor.
            if (tune.amp_ref > 0.0) r_ /= fmax(1.0, (1.0 + ds.par[MTRPL_P_TAUP] / ds.par[MTRPL_P_TAUN]) / tune.amp_ref);
            r_ = fmax(r_, tune.rtol_floor);
        } else {
            r_ = base;  // the caller asked for more than the policy would ever do
            if (r_ < tune.rtol_min) { r_ = tune.rtol_min; tol_clamped = true; }
        }
        if (tune.rtol_force > 0.0) { r_ = tune.rtol_force; tol_clamped = false; }
        rtol = (real)r_;
        // Absolute terms.  The densities are resolved RELATIVELY deep into the tail of a trace (a
        // low-injection decay takes the excess density many decades down; the signal follows it), so their
        // absolute term is tiny; desc.atol (solver units, nm^-3) bounds it from above -- the default 1e-10
        // never does.  The field passes through zero and keeps an absolute floor in the scale of the
        // problem (a tiny one makes every BDF code crawl there, LSODA included).
        atol = (real)fmin(r_ * tune.atol_rel_np * (double)dNmax, ds.atol);
        atol_E = (real)(r_ * tune.atol_rel * (double)dNmax);
    }
    // the field is a fast slaved variable (dielectric relaxation): its local error is damped, not
    // accumulated, so it gets a looser relative weight than the densities (DESIGN.md "error control")
    // (TRPL only: the photoconductivity weighs dP directly, and its p99 census error doubles with a looser field)
    const real rtol_E = rtol * ((meas == MTRPL_MEAS_TRPL) ? (real)tune.e_weight : (real)1);
    auto rtol_of = [&](int v) { return (v >= 2 * R && v < 3 * R) ? rtol_E : rtol; };
    auto atol_of = [&](int v) { return (v >= 2 * R && v < 3 * R) ? atol_E : atol; };
    const real early_until = (nt > 1) ? (real)(tune.early_frac * ts[1]) : (real)0;
    const real early_inv = (real)(1.0 / fmax(tune.early_factor, 1.0));
    const real newton_tol = (tune.newton_tol > 0.0)
                                ? (real)tune.newton_tol
                                : (real)fmax(10.0 * 2.220446049250313e-16 / (double)rtol, fmin(0.03, sqrt((double)rtol)));

    int n_steps = 0, n_rhs = 0, n_fac = 0, n_jac = 0, n_rej = 0, n_nfail = 0;
    int st = MTRPL_SOLVE_OK;

    // dx-less grid sum of the signal density of one state
    auto sig_partial = [&](const real (&yy)[V]) {
        real s = (real)0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const real v = signal_density(p, meas, yy[r], yy[R + r]);
            s += (bk.node(r) < L) ? v : (real)0;
        }
        return s;
    };
    bk.out_sample(0, sig_partial(y));

    real z[6][V];
    real f[V], w[V], zE[R];
#pragma unroll
    for (int r = 0; r < R; ++r) zE[r] = (real)0;
    bk.publish(y, zE);
    bk.rhs(y, f);
    n_rhs++;

    // ---- initial step (Hairer, Norsett & Wanner I, II.4, first-order variant) ----
    real h;
    {
        real s3[1];
        auto tnorm = [&](const real (&vv)[V]) {
            s3[0] = (real)0;
#pragma unroll
            for (int k = 0; k < V; ++k) s3[0] += vv[k] * vv[k];
            bk.template reduce<1>(s3, y, zE);
            return norm_sqrt(s3[0] * inv_S);
        };
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = y[v] / (atol_of(v) + rtol_of(v) * fabs(y[v]));
        const real d0 = tnorm(w);
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = f[v] / (atol_of(v) + rtol_of(v) * fabs(y[v]));
        const real d1 = tnorm(w);
        real h0 = (d0 < (real)1e-5 || d1 < (real)1e-5) ? (real)1e-6 : (real)0.01 * d0 / d1;
        real y1[V], f1[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y1[v] = y[v] + h0 * f[v];
        bk.publish(y1, zE);
        bk.rhs(y1, f1);
        n_rhs++;
#pragma unroll
        for (int v = 0; v < V; ++v) w[v] = (f1[v] - f[v]) / (atol_of(v) + rtol_of(v) * fabs(y[v]));
        const real d2 = tnorm(w) / h0;
        const real dm = fmax(d1, d2);
        const real h1 = (dm <= (real)1e-15) ? fmax((real)1e-6, h0 * (real)1e-3) : (real)sqrt((real)0.01 / dm);
        h = fmin(fmin((real)100.0 * h0, h1), hmax);
        if (nt > 1) h = fmin(h, (real)ts[nt - 1]);
        if (!(h > (real)0)) h = (real)1e-9;
    }

#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int v = 0; v < V; ++v) z[k][v] = (k == 0) ? y[v] : ((k == 1) ? h * f[v] : (real)0);

    int order = 1, n_equal = 0;
    real t = (real)0;
    n_jac++;
    bool have_fac = false;
    real rate_prev = (real)1;
    int rate_age = 0;
    int io = 1;
    real tq_next = (nt > 1) ? (real)ts[1] : big<real>();
    int since_rej = 1000;  // accepted steps since the last rejected one
    real acp[V];           // the correction of the previous step (order-increase estimate)
#pragma unroll
    for (int v = 0; v < V; ++v) acp[v] = (real)0;

#ifdef MTRPL_STAMPS
    // controller vector work | factorisation | rhs | linear re-solve | reductions + publishes | Newton glue | dense output
    unsigned long long tlast_ = __builtin_amdgcn_s_memtime(), ph_ctl = 0, ph_fac = 0, ph_rhs = 0, ph_app = 0,
                       ph_red = 0, ph_glue = 0, ph_dense = 0;
#endif
    while (io < nt) {
        MTRPL_STAMP(ph_ctl);
        if (n_steps >= tune.max_steps) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
        bool accepted = false;
        real acor[V], yn[V], isc[V];  // isc: 1/(atol + rtol*|y_predicted|), the error weights of this step
        real dsm = (real)0, safety = (real)0.9;
        real ssel0 = (real)0, ssel1 = (real)0;  // selection sums of the accepted iteration
        int attempts = 0;
        while (!accepted) {
            if (++attempts > 60 || !(h > (real)1e-13 * fmax(t, (real)1e-3))) { st = MTRPL_SOLVE_STOPPED_EARLY; break; }
            MTRPL_STAMP(ph_ctl);
            // the order / step-size selection runs after this step if it is accepted with n_equal >= order:
            // its two grid sums then travel with the Newton norms instead of costing a reduction of their own
            const bool will_select = n_equal >= order;
            // predictor: z <- A z; z[0] is the predicted state, z[1] the predicted h*y'
            real el0 = (real)1, inv_t2 = (real)1, elK = (real)0;
            nq_predict<5, V>(z);
            MTRPL_ORDER_SWITCH(order, (el0 = (real)NqC<K>::el0, inv_t2 = (real)NqC<K>::inv_t2, elK = (real)NqC<K>::elK))
            const real c = h * el0;
            const real early_scale = (tune.early_factor > 1.0 && t + h <= early_until) ? early_inv : (real)1;
#pragma unroll
            for (int v = 0; v < V; ++v) isc[v] = ctl_rcp(atol_of(v) + rtol_of(v) * fabs(z[0][v])) * early_scale;
            MTRPL_STAMP(ph_ctl);
            // simplified Newton on acor:  (I - h l0 J) dx = h f(y) - z[1] - acor,  y = z[0] + l0 acor
            bool converged = false, fresh = false;
            int n_iter = 0;
            while (!converged) {
#pragma unroll
                for (int v = 0; v < V; ++v) { acor[v] = (real)0; yn[v] = z[0][v]; }
#pragma unroll
                for (int r = 0; r < R; ++r) zE[r] = z[1][2 * R + r];
                MTRPL_STAMP(ph_glue);
                bk.publish(yn, zE);
                MTRPL_STAMP(ph_red);
                if (!have_fac) {
                    // the analytic Jacobian is evaluated inside the factorisation: always at the predictor
                    bk.build(z[0], c, h);
                    MTRPL_STAMP(ph_fac);
                    have_fac = true;
                    fresh = true;
                    n_fac++;
                    n_jac++;
                    rate_prev = (real)1;  // new matrix: measure the rate afresh
                    rate_age = 0;
                }
                real dn_old = (real)0;
                for (int k = 0; k < 4; ++k) {
                    n_iter = k + 1;
                    MTRPL_STAMP(ph_glue);
                    bk.rhs(yn, f);
                    MTRPL_STAMP(ph_rhs);
                    n_rhs++;
                    real b[V], dxv[V];
#pragma unroll
                    for (int v = 0; v < V; ++v) b[v] = h * f[v] - z[1][v] - acor[v];
                    MTRPL_STAMP(ph_glue);
                    bk.apply(b, dxv);
                    MTRPL_STAMP(ph_app);
                    // the update is applied before the convergence decision (a rejected iterate is simply
                    // discarded): the norms of the increment and of the whole correction, and the boundary
                    // state the next rhs() needs, then share one reduction
                    real s4[4] = {(real)0, (real)0, (real)0, (real)0};
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const real wv = el0 * dxv[v] * isc[v];
                        s4[0] += wv * wv;
                        acor[v] += dxv[v];
                        yn[v] += el0 * dxv[v];
                        const real av = acor[v] * isc[v];
                        s4[1] += av * av;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) zE[r] = z[1][2 * R + r] + acor[2 * R + r];
                    if (will_select) {
                        // stode: ddn from the last Nordsieck row (after the correction), dup from acor - acor_prev
                        switch (order) {
                            case 1: for (int v = 0; v < V; ++v) { const real q_ = (z[1][v] + elK * acor[v]) * isc[v]; s4[2] += q_ * q_; } break;
                            case 2: for (int v = 0; v < V; ++v) { const real q_ = (z[2][v] + elK * acor[v]) * isc[v]; s4[2] += q_ * q_; } break;
                            case 3: for (int v = 0; v < V; ++v) { const real q_ = (z[3][v] + elK * acor[v]) * isc[v]; s4[2] += q_ * q_; } break;
                            case 4: for (int v = 0; v < V; ++v) { const real q_ = (z[4][v] + elK * acor[v]) * isc[v]; s4[2] += q_ * q_; } break;
                            default: for (int v = 0; v < V; ++v) { const real q_ = (z[5][v] + elK * acor[v]) * isc[v]; s4[2] += q_ * q_; } break;
                        }
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const real a_ = (acor[v] - acp[v]) * isc[v];
                            s4[3] += a_ * a_;
                        }
                        MTRPL_STAMP(ph_glue);
                        bk.template reduce<4>(s4, yn, zE);
                    } else {
                        real s2[2] = {s4[0], s4[1]};
                        MTRPL_STAMP(ph_glue);
                        bk.template reduce<2>(s2, yn, zE);
                        s4[0] = s2[0];
                        s4[1] = s2[1];
                    }
                    MTRPL_STAMP(ph_red);
                    const real dn = norm_sqrt(s4[0] * inv_S);
                    if (!(dn < big<real>())) break;  // NaN / inf
                    real rate = rate_prev;
                    const bool negligible = dn < (real)1e-2 * newton_tol;
                    if (k > 0) {
                        rate = dn * ctl_rcp(dn_old);
                        if (!negligible) {
                            real rp = rate;  // rate^(4-k)
                            for (int q = 1; q < 4 - k; ++q) rp *= rate;
                            if (rate >= (real)1 || rp * dn > newton_tol * ((real)1 - rate)) break;
                        }
                        if (rate < (real)1) {
                            rate_prev = (rate_prev >= (real)1 && tune.rate_first_safety > 0.0)
                                            ? fmin(fmax((real)tune.rate_first_safety * rate, (real)1e-4), (real)0.9)
                                            : fmax((real)0.3 * rate_prev, rate);
                            rate_age = 0;
                        }
                    }
                    const bool rate_ok = (k > 0) || (rate_prev < (real)0.9 && rate_age < tune.rate_max_age);
                    if (negligible || (rate_ok && rate < (real)1 && rate * dn < newton_tol * ((real)1 - rate))) {
                        converged = true;
                        dsm = norm_sqrt(s4[1] * inv_S) * inv_t2;
                        ssel0 = s4[2];
                        ssel1 = s4[3];
                        break;
                    }
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
                nq_retract<5, V>(z);
                nq_rescale<5, V>(z, (real)0.5);
                h *= (real)0.5;
                n_equal = 0;
                have_fac = false;
                continue;
            }
            MTRPL_STAMP(ph_glue);
            // 0.9 * (2*4 + 1) / (2*4 + n_iter), n_iter = 1..4
            safety = (n_iter == 1) ? (real)0.9 : ((n_iter == 2) ? (real)0.81 : ((n_iter == 3) ? (real)(8.1 / 11.0) : (real)0.675));
            if (dsm > (real)1) {
                n_rej++;
                since_rej = 0;
                const real fac = fmax((real)0.2, safety * inv_root(dsm, order + 1));
                nq_retract<5, V>(z);
                nq_rescale<5, V>(z, fac);
                h *= fac;
                n_equal = 0;
                have_fac = false;
            } else {
                accepted = true;
            }
        }
        if (!accepted) break;
        MTRPL_STAMP(ph_ctl);
        const bool selecting = n_equal >= order;  // = will_select of the accepted attempt
        n_steps++;
        n_equal++;
        since_rej++;
        t += h;
        {
            real el[6];
            MTRPL_ORDER_SWITCH(order, (el[0] = (real)NqC<K>::T.el[K][0], el[1] = (real)NqC<K>::T.el[K][1], el[2] = (real)NqC<K>::T.el[K][2],
                                       el[3] = (real)NqC<K>::T.el[K][3], el[4] = (real)NqC<K>::T.el[K][4], el[5] = (real)NqC<K>::T.el[K][5]))
            nq_correct_rt<V>(z, acor, el);
        }
        MTRPL_STAMP(ph_ctl);
        // dense output (Taylor polynomial of the Nordsieck array), forward_solver.py:167
        const real ih = frcp(h);
        while (io < nt) {
            const real tq = tq_next;  // = ts[io], loaded one output ahead
            if (tq > t * (real)(1.0 + 1e-15)) break;
            real yq[V];
            const real s_ = (tq - t) * ih;
            nq_dense<5, V>(z, s_, yq);
            bk.out_sample(io, sig_partial(yq));
            io++;
            tq_next = (io < nt) ? (real)ts[io] : big<real>();
        }
        MTRPL_STAMP(ph_dense);
#pragma unroll
        for (int v = 0; v < V; ++v) acp[v] = acor[v];
        if (!selecting) continue;
        // order / step-size selection (stode)
        real inv_t1 = (real)1, inv_t3 = (real)1, el_up = (real)0;
        float e0 = -0.5f, e1 = -0.5f, e2 = -0.5f;
        MTRPL_ORDER_SWITCH(order, (inv_t1 = (real)NqC<K>::inv_t1, inv_t3 = (real)NqC<K>::inv_t3, el_up = (real)NqC<K>::el_up,
                                   e0 = NqC<K>::nik0, e1 = NqC<K>::nik1, e2 = NqC<K>::nik2))
        const real ddn = (order > 1) ? inv_t1 * norm_sqrt(ssel0 * inv_S) : big<real>();
        const real dup = (order < 5) ? inv_t3 * norm_sqrt(ssel1 * inv_S) : big<real>();
        const real fm = (order > 1) ? pow_neg_inv(ddn, e0) : (real)0;
        const real f0 = pow_neg_inv(dsm, e1);
        const real fp = (order < 5) ? pow_neg_inv(dup, e2) : (real)0;
        // stode's biases: a change of order has to promise more than staying (1.2 / 1.3 / 1.4)
        int dorder = 0;
        real fbest = f0 * (real)tune.nq_bias_same;
        if (fm * (real)tune.nq_bias_down > fbest) { fbest = fm * (real)tune.nq_bias_down; dorder = -1; }
        if (fp * (real)tune.nq_bias_up > fbest) { fbest = fp * (real)tune.nq_bias_up; dorder = 1; }
        real fac = fmin((real)10, (tune.nq_bias_same < 1.0 ? (real)1 : safety) * fbest);
        // a step that was rejected a moment ago says the estimates behind `fac` are optimistic here (stiff
        // surface terms): grow cautiously for a while instead of hunting (reject / shrink / regrow / reject)
        if (since_rej < tune.nq_rej_memory) fac = fmin(fac, (real)tune.nq_rej_growth);
        // max_step guards LSODA against stepping over structure; once the signal is down in the rounding noise
        // of N P - n0 p0 (hmax_deep, 11 decades by default) there is none left and the error test alone decides
        if (h * fac > hmax && !bk.deep()) fac = hmax * ih;
        if (dorder == 0 && fac >= (real)0.9 && fac < (real)tune.nq_deadband_hi) {
            n_equal = order - 2;  // keep h and the factors; test again three steps on
            continue;
        }
        if (dorder != 0) {
            // order up: the new last row is z[q+1] = acor * l_q / (q+1); order down: the last row is dropped,
            // i.e. zeroed.  Row j = 2..5 is scaled by keep_j and receives add_j * acor (both wave-uniform).
            const int qn = order + dorder;
#pragma unroll
            for (int j = 2; j < 6; ++j) {
                const real keep = (j <= qn && j <= order) ? (real)1 : (real)0;
                const real add = (dorder == 1 && j == qn) ? el_up : (real)0;
#pragma unroll
                for (int v = 0; v < V; ++v) z[j][v] = keep * z[j][v] + add * acor[v];
            }
            order = qn;
        }
        nq_rescale<5, V>(z, fac);
        h *= fac;
        n_equal = 0;
        have_fac = false;
    }

    if (st != MTRPL_SOLVE_OK) {
        // forward_solver.py:168: samples after the last solved time come from a zeroed state
        real zero_[V];
#pragma unroll
        for (int v = 0; v < V; ++v) zero_[v] = (real)0;
        const real zp = sig_partial(zero_);
        for (; io < nt; ++io) bk.out_sample(io, zp);
    } else if (tol_clamped) {
        st = MTRPL_SOLVE_OK_TOL_CLAMPED;
    }
    bk.out_finish();
    if (bk.leader()) {
        status[sidx] = st;
        if (stats) {
            int32_t *s = stats + 8 * sidx;
            s[0] = n_steps; s[1] = n_rhs; s[2] = n_fac; s[3] = n_jac;
            s[4] = n_rej; s[5] = n_nfail; s[6] = order;
            s[7] = (int32_t)((__builtin_amdgcn_s_memtime() - clk0_) >> 4);  // residency of this solve, 16-tick units
#ifdef MTRPL_SCHED  // diagnostic build (tools/sched_trace.py): when (100 MHz real-time counter) and where the solve ran
            s[5] = (int32_t)(sched_t0_ & 0x7fffffff);
            s[3] = (int32_t)(__builtin_amdgcn_s_memrealtime() & 0x7fffffff);
            s[6] = (int32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));  // HW_ID, 32 bits
            s[4] = (int32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));  // XCC_ID
#endif
#ifdef MTRPL_STAMPS
            s[1] = (int)(ph_ctl >> 10); s[2] = (int)(ph_fac >> 10); s[3] = (int)(ph_rhs >> 10);
            s[4] = (int)(ph_app >> 10); s[5] = (int)(ph_red >> 10); s[6] = (int)(ph_glue >> 10);
            s[7] = (int)(ph_dense >> 10);
#endif
        }
    }
}

}  // namespace mtrpl
