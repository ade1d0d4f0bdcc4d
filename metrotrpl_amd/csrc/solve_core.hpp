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
//   bk.rhs(y, f)                    f = dy/dt (forward_solver.py:332-418); padded slots (node >= L) may
//                                   hold anything finite: integrate() multiplies them by a zero step
//   bk.build(y, c)                  factorise I - c*J(y)
//   bk.apply(b, x)                  x = (I - c*J)^-1 b with the kept factors
//   bk.reduce_f<NS>(s)              s[k] <- sum over the whole grid in fp32 (the sums only steer the
//                                   controller), identical in every thread
//   bk.scan_excl(v), bk.max_all(v)  exclusive prefix over the threads before this one / maximum
//   bk.out_begin / out_first / out_sample / out_flush / deep   emission of the output samples (scale, min_y clamp)
//   bk.leader()                     the one thread that writes status / stats
//
// Value layout of a thread: y[r] = N, y[R+r] = P, y[2R+r] = E on the left face of node r,
// y[3R+r] = N_trap (traps model).
//
// What is wave-uniform and only steers the integration -- error norms, Newton contraction rates, step and
// order factors -- is fp32 throughout (`ctl`): there is no scalar floating point on this ISA, so each such
// operation is a vector instruction with one useful lane, and a conversion to and from fp64 around every
// hardware sqrt / rcp / log / exp tripled their number.  Everything order-dependent comes from one 128-byte row
// of a table in constant memory (scalar loads), not from a switch over the order.
#pragma once
#include <cstddef>
#include <type_traits>

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

namespace mtrpl {

#ifdef MTRPL_TRACE
__device__ float g_trace[8 * 8192];
__device__ int g_trace_n;
__device__ int g_trace_sel = MTRPL_TRACE;  // the traced solve (mtrpl_debug_trace_select)
#define TRACE(code, n_iter_, dsm_) do { if (sidx == g_trace_sel && bk.leader() && g_trace_n < 8192) { float *q_ = g_trace + 8 * g_trace_n++; \
    q_[0] = (float)(code); q_[1] = (float)t; q_[2] = (float)h; q_[3] = (float)order; q_[4] = (float)(n_iter_); q_[5] = (float)(dsm_); q_[6] = (float)rho; q_[7] = (float)io;  /* rho: tools/bind_trace.py */ } } while (0)
#else
#define TRACE(code, n_iter_, dsm_) do { } while (0)
#endif

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
    int honor_hmax;     // 1: desc.hmax caps the step (the reference's max_step); 0 (default since the goal-oriented error
                        // test: with it the cap changes no sample of any fixture): error control alone
    double hmax_deep;   // ... until the signal has fallen below this fraction of its first sample (0: always)
    double deep_factor; // ... from where on the local tolerance is multiplied by this as well (the state sits at
                        // equilibrium to rounding: no sample down there is resolved or gated)
    double hmax_scale;  // the cap is hmax_scale x desc.hmax
    double tail_frac;   // once the signal has fallen below this fraction of its first sample ...
    double tail_factor; // ... the local tolerance is multiplied by this (< 1: the relative error of a long decay
                        // accumulates per e-fold; the deep decades of a trace need the error budget of the shallow ones)
    double far_frac;    // ... and below this fraction by far_factor instead (> 1: nine decades down nothing resolves
    double far_factor;  // the signal any more -- measured traces span four to five -- and the slowest solves of a
                        // batch are the ones that crawl through decades 9 to 17)
    double trts_rtol_factor;  // internal rtol of TRTS solves = this x that of TRPL solves
    double amp_ref;           // > 0: local tolerance /= max(1, (1 + tauP/tauN) / amp_ref) ...; 0: off
    double amp_post;          // ... until the excess density has fallen below amp_post x p0 (judged by the emitted signal):
                              // from there on the decay is low-injection and nothing amplifies an error any more
    double nq_deadband_hi;    // a step factor in [0.9, this) keeps h and the factors
    int nq_rej_memory; double nq_rej_growth;  // growth cap for this many steps after a rejection
    double nq_bias_same, nq_bias_down, nq_bias_up;  // multipliers of the step factors (stode: 1/1.2, 1/1.3, 1/1.4)
    double early_factor;      // > 1: local tolerance x this for steps ending before early_frac * (first sample time)
    double early_frac;
    int rtol_mode;            // 0: the contract is the converged trace (rtol_cap holds); 1: the CALLER's RTOL decides, as it does
                              // in the reference (forward_solver.py:163-165): internal rtol = rtol_scale x RTOL, capped at 1e-3 only
    double shape_factor;      // > 1: goal-oriented error test (DESIGN.md "error control"): the RMS norm of the local error
    double mass_floor;        // (see fun_weight) ... relative to sum (N - n0) + mass_floor x n0 x nx
    double fun_weight;        // estimate is allowed shape_factor x the tolerance as long as its MASS component -- the
                              // signed grid sum of the estimate for N against the grid sum of N, weighted fun_weight --
                              // stays inside the tolerance itself
};

// The shipped settings as compile-time constants: a kernel instantiated on this type instead of on the
// run-time struct keeps none of them in (spilled) scalar registers.  Single source of the defaults.
struct DefaultTune {
    static constexpr double rtol_scale = 0.1, rtol_cap = 2e-9, rtol_floor = 2e-10, rtol_min = 3e-11, rtol_force = 0.0;
    static constexpr double atol_rel = 1e-4, atol_rel_np = 1e-9;
    static constexpr int max_steps = 2000000;
    static constexpr double newton_tol = 0.1;
    static constexpr int rate_max_age = 8;
    static constexpr double rate_first_safety = 4.0, e_weight = 3.0;
    static constexpr int honor_hmax = 0;
    static constexpr double hmax_deep = 1e-11, deep_factor = 1000.0, hmax_scale = 1.0, tail_frac = 1e-4, tail_factor = 1.0, far_frac = 1e-9, far_factor = 30.0, trts_rtol_factor = 0.5, amp_ref = 10.0, amp_post = 0.03, nq_deadband_hi = 1.2;
    static constexpr int nq_rej_memory = 0;  // (rounds 1-4: 80.  Under the goal-oriented test the cap buys nothing: 4 % more steps, same accuracy)
    static constexpr double nq_rej_growth = 1.2, nq_bias_same = 1.0, nq_bias_down = 1.0, nq_bias_up = 1.0;
    static constexpr double early_factor = 10.0, early_frac = 0.5;
    static constexpr int rtol_mode = 0;
    static constexpr double shape_factor = 30.0, mass_floor = 1e-3, fun_weight = 3.0;
    static SolveTuning make()
    {
        return SolveTuning{rtol_scale, rtol_cap, rtol_floor, rtol_min, rtol_force, atol_rel, atol_rel_np, max_steps,
                           newton_tol, rate_max_age, rate_first_safety, e_weight, honor_hmax, hmax_deep, deep_factor, hmax_scale, tail_frac, tail_factor, far_frac, far_factor, trts_rtol_factor,
                           amp_ref, amp_post, nq_deadband_hi, nq_rej_memory, nq_rej_growth, nq_bias_same, nq_bias_down, nq_bias_up,
                           early_factor, early_frac, rtol_mode, shape_factor, mass_floor, fun_weight};
    }
};

// ODEPACK cfode, meth = 2: l-vector and error-test constants of the fixed-step BDF of order q = 1..5, one
// 128-byte row per order.  The rows sit in constant memory: the order is wave-uniform, so a row is read with
// scalar loads where it is needed and no order-dependent constant lives across a step.
struct NqRow {
    double el[6];       // l-vector, zero above the order
    double inv_t1, inv_t2, inv_t3;  // 1 / error-test constants for order q-1, q, q+1 (stode's tesco)
    double el_up;       // l_q / (q+1): the new last row on an order increase
    double elK;         // l_q: coefficient of the last row
    float nik0, nik1, nik2;         // -1/q, -1/(q+1), -1/(q+2): exponents of the three step factors
    float inv_qp1;                  // 1/(q+1)
    double pad_[3];
};
static_assert(sizeof(NqRow) == 128, "one row = 128 bytes");
struct NqRows { NqRow row[6]; };
constexpr NqRows nq_make_rows()
{
    NqRows T{};
    double pc[8] = {1.0, 0, 0, 0, 0, 0, 0, 0};
    double rq1fac = 1.0;
    for (int nq = 1; nq <= 5; ++nq) {
        const double fnq = (double)nq;
        pc[nq] = 0.0;
        for (int i = nq; i >= 1; --i) pc[i] = pc[i - 1] + fnq * pc[i];
        pc[0] = fnq * pc[0];
        NqRow &r = T.row[nq];
        for (int j = 0; j <= nq; ++j) r.el[j] = pc[j] / pc[1];
        r.el[1] = 1.0;
        const double t1 = rq1fac, t2 = (double)(nq + 1) / r.el[0], t3 = (double)(nq + 2) / r.el[0];
        r.inv_t1 = 1.0 / t1; r.inv_t2 = 1.0 / t2; r.inv_t3 = 1.0 / t3;
        r.elK = r.el[nq];
        r.el_up = r.el[nq] / (double)(nq + 1);
        r.nik0 = -1.0f / (float)nq; r.nik1 = -1.0f / (float)(nq + 1); r.nik2 = -1.0f / (float)(nq + 2);
        r.inv_qp1 = 1.0f / (float)(nq + 1);
        rq1fac = rq1fac / fnq;
    }
    return T;
}
static __constant__ NqRows kNq = nq_make_rows();

// The row of the current order, two ways.  OrderRow<false>: scalar loads from constant memory where a value is needed --
// free for the vector ALU, which is what a SIMD shared by two solves is short of.  OrderRow<true>: the 16 doubles of the row sit
// in lanes 0-15 of one register pair (one vector load per change of order) and a value reaches the scalar unit by
// v_readlane -- no scalar-memory round trip (100-200 cycles, three or four of them exposed per step) for a wave that has its
// SIMD to itself and nothing else to do while it waits.  Same values either way.
template <bool IN_VGPR> struct OrderRow;
template <> struct OrderRow<false> {
    const NqRow *r;
    __device__ inline void load(int order, int) { r = &kNq.row[order]; }
    __device__ inline double el(int j) const { return r->el[j]; }
    __device__ inline double inv_t1() const { return r->inv_t1; }
    __device__ inline double inv_t2() const { return r->inv_t2; }
    __device__ inline double inv_t3() const { return r->inv_t3; }
    __device__ inline double el_up() const { return r->el_up; }
    __device__ inline double elK() const { return r->elK; }
    __device__ inline float nik0() const { return r->nik0; }
    __device__ inline float nik1() const { return r->nik1; }
    __device__ inline float nik2() const { return r->nik2; }
    __device__ inline float inv_qp1() const { return r->inv_qp1; }
};
template <> struct OrderRow<true> {
    double v;  // lane j < 16: double j of the row (el[0..5], inv_t1..3, el_up, elK, (nik0, nik1), (nik2, inv_qp1), pad)
    __device__ inline void load(int order, int lane) { v = reinterpret_cast<const double *>(&kNq.row[order])[lane & 15]; }
    __device__ inline double d(int j) const { return lane_value(v, j); }
    __device__ inline float f(int j, int half) const
    {
        union { double d; float q[2]; } u;
        u.d = lane_value(v, j);
        return u.q[half];
    }
    __device__ inline double el(int j) const { return d(j); }
    __device__ inline double inv_t1() const { return d(6); }
    __device__ inline double inv_t2() const { return d(7); }
    __device__ inline double inv_t3() const { return d(8); }
    __device__ inline double el_up() const { return d(9); }
    __device__ inline double elK() const { return d(10); }
    __device__ inline float nik0() const { return f(11, 0); }
    __device__ inline float nik1() const { return f(11, 1); }
    __device__ inline float nik2() const { return f(12, 0); }
    __device__ inline float inv_qp1() const { return f(12, 1); }
};
static_assert(offsetof(NqRow, inv_t1) == 48 && offsetof(NqRow, el_up) == 72 && offsetof(NqRow, elK) == 80 && offsetof(NqRow, nik0) == 88 &&
              offsetof(NqRow, nik2) == 96, "OrderRow<true> reads the row by these slots");

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

// Output samples: unit scale (forward_solver.py:189,:197) and the min_y clamp "from the first sample
// below min_y onward" (:190-192).  One instance per solve, held by the thread(s) that store.
struct Emitter {
    double *o;
    double min_y, scale;
    // The zones of the error-control policy are fractions of the first sample.  Per sample ONE ratio (fp32: the zone
    // boundaries are policy, not results) folded into a running minimum; the four zone tests are made where a step
    // asks for them, against fractions that are literals in the kernels with the shipped options compiled in.  (Four
    // fp64 thresholds were eight registers of a kernel that spills, and four sticky integer flags three vector
    // instructions each per sample: compare, select 0/1, or.)
    double inv_first;   // 1 / |first sample|
    float rmin;         // smallest |sample| / |first sample| so far
    float deep_frac, tail_frac, far_frac;
    float post_frac;    // output level below which the decay is low-injection, as a fraction of the first sample
    double post_level;  // ... absolute, until the first sample is known
    int clamped;
    __device__ inline void begin(double *o_, double min_y_, double scale_, double deep_frac_, double tail_frac_, double far_frac_,
                                 double post_level_)
    {
        o = o_; min_y = min_y_; scale = scale_; clamped = 0;
        deep_frac = (float)deep_frac_;
        tail_frac = (float)tail_frac_;
        far_frac = (float)far_frac_;
        post_level = post_level_;
        post_frac = 0.0f;
        inv_first = 0.0;
        rmin = 1.0f;
    }
    __device__ inline bool deep() const { return !(rmin >= deep_frac); }
    __device__ inline bool tail() const { return !(rmin >= tail_frac); }
    __device__ inline bool far() const { return !(rmin >= far_frac); }
    __device__ inline bool post() const { return !(rmin >= post_frac); }
    // total = dx-less grid sum; writer = this thread stores.  Every thread that calls sees the same v,
    // so the zones are uniform among the callers.
    __device__ inline void first(double total, double dx, bool writer)
    {
        double v = total * dx * scale;
        inv_first = 1.0 / fmax(fabs(v), 1e-300);  // the first sample fixes the scale
        post_frac = (float)fmin(post_level * inv_first, 1e30);
        rmin = 1.0f;
        clamped = v < min_y;
        if (clamped) v = min_y;
        if (writer) o[0] = v;
    }
    // the sample as it is stored (scale, clamp) and its trace in the zone bookkeeping
    __device__ inline double value(double total, double dx)
    {
        double v = total * dx * scale;
        rmin = fminf(rmin, (float)(fabs(v) * inv_first));
        clamped |= v < min_y;
        if (clamped) v = min_y;
        return v;
    }
    __device__ inline void put(int k, double total, double dx, bool writer)
    {
        const double v = value(total, dx);
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
    using ctl = float;  // wave-uniform controller arithmetic
    constexpr int R = BK::R, V = BK::V;
    constexpr bool TRAPS = BK::TRAPS;
    const int L = ds.nx, nt = ds.nt, meas = ds.meas;
    const double dx = ds.thickness / (double)L;  // sim_utils.py:265
    bk.set_problem(make_par<real>(ds.par, dx), L);
    const ParT<real> &p = bk.p;
    const double *ts = tsteps + ds.t_off;
    const real hmax = tune.honor_hmax ? (real)(ds.hmax * tune.hmax_scale) : big<real>();
    const ctl inv_S = (ctl)(1.0 / (double)((TRAPS ? 4 : 3) * L + 1));
    const double out_scale = (meas == MTRPL_MEAS_TRPL) ? 1e23 : 1e9;  // forward_solver.py:189,:197
    // output level of a film whose excess density is amp_post x p0 everywhere: ks dn (dn + p0 + n0) L (PL) or
    // q (mu_n + mu_p) dn L (TRTS); below it the decay is low-injection (amp_post, DESIGN.md "error control")
    double post_level = 0.0;
    {
        const double dn = tune.amp_post * ds.par[MTRPL_P_P0], n0_ = ds.par[MTRPL_P_N0], p0_ = ds.par[MTRPL_P_P0];
        post_level = ((meas == MTRPL_MEAS_TRPL) ? ds.par[MTRPL_P_KS] * dn * (dn + p0_ + n0_)
                                                : kQC * (ds.par[MTRPL_P_MU_N] + ds.par[MTRPL_P_MU_P]) * dn) * ds.thickness * out_scale;
    }
    bk.out_begin(out + ds.out_off, ds.min_y, out_scale, tune.hmax_deep, tune.tail_frac, tune.far_frac, post_level);
    const int lane64 = (int)(threadIdx.x & 63);

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
    dNmax = uniform(bk.max_all(dNmax));

    // ---- tolerances (DESIGN.md "error control") ----
    // The contract is the converged solution, so a loose RTOL does not loosen the integration
    // (rtol_cap); a tight one tightens it, below the policy floor too, down to the fp64 noise floor
    // rtol_min -- a request below that is raised to it and reported as MTRPL_SOLVE_OK_TOL_CLAMPED.
    bool tol_clamped = false;
    real rtol, atol, atol_E;
    double amp_mul_d = 1.0;
    {
        const double req = tune.rtol_scale * ds.rtol;
        const double base = fmin(req, tune.rtol_mode == 1 ? 1e-3 : tune.rtol_cap);
        double r_;
        if (base >= tune.rtol_floor) {
            r_ = base;
            // TRTS is linear in the carrier density: its 4-decade window spans twice the e-folds of a
            // high-injection PL trace, and the relative error of a monotone decay accumulates per e-fold
            if (meas == MTRPL_MEAS_TRTS) r_ *= tune.trts_rtol_factor;
            // SRH with tauP >> tauN (p-type): the decay rate rises from 1/(tauN+tauP) at high injection to
            // 1/tauN at low injection, and a relative error made before that transition is amplified by the
            // ratio of the two rates (d ln(dN/N)/dt = -r'(N) N): tighten by that a-priori factor.
            // (applied as a factor on the error norms of the steps taken while the decay is still high-injection:
            // amp_mul below; here only its part that the floor allows)
            if (tune.amp_ref > 0.0) {
                const double amp = fmax(1.0, (1.0 + ds.par[MTRPL_P_TAUP] / ds.par[MTRPL_P_TAUN]) / tune.amp_ref);
                amp_mul_d = fmin(amp, fmax(1.0, r_ / tune.rtol_floor));
            }
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
        // (never exactly zero: the trapped density of the traps model starts at 0, and 0 / (0 + rtol * 0) in the
        // norms of the initial step would be NaN for a caller's ATOL = 0)
        atol = (real)fmax(fmin(r_ * tune.atol_rel_np * (double)dNmax, ds.atol), sizeof(real) == 8 ? 1e-300 : 1e-37);
        atol_E = (real)(r_ * tune.atol_rel * (double)dNmax);
    }
    // the field is a fast slaved variable (dielectric relaxation): its local error is damped, not
    // accumulated, so it gets a looser relative weight than the densities (DESIGN.md "error control")
    // (TRPL only: the photoconductivity weighs dP directly, and its p99 census error doubles with a looser field)
    const real rtol_E = rtol * ((meas == MTRPL_MEAS_TRPL) ? (real)tune.e_weight : (real)1);
    auto is_E = [](int v) { return v >= 2 * R && v < 3 * R; };
    // Backends may leave the field components out of the controller's norms (BK::kFieldInNorms = false): E is slaved to the
    // charge (Gauss's law is a linear invariant of the system: dE_i/dt = -Lam (Jn_i + Jp_i) is its time derivative), its
    // correction follows the densities' through the back-substitution, and its weight is the loosest of the three.
    constexpr bool kNormE = BK::kFieldInNorms;
    auto in_norm = [&](int v) { return kNormE || !is_E(v); };
    auto rtol_of = [&](int v) { return is_E(v) ? rtol_E : rtol; };
    auto atol_of = [&](int v) { return is_E(v) ? atol_E : atol; };
    // the same as fp32 values: what the weights 1/(atol + rtol |y|) are built from.  Densities (1e-4 ... 1e-22 nm^-3)
    // and tolerances are far inside the fp32 range; the absolute terms are kept above its smallest normal number,
    // so that a caller's ATOL = 0 (or 1e-45) next to a component that has gone to zero gives a finite weight and
    // not 1/0 (the reference's LSODA integrates such a case).
    // (a backend short of vector registers keeps these wave-uniform values in scalar ones: BK::kScalarPolicy.  They come out
    // of floating-point arithmetic, which this ISA has on the vector unit only, so the compiler leaves them where they were
    // computed -- and, in a kernel that spills, spills them.)
    auto sc = [](auto x) {
        if constexpr (BK::kScalarPolicy) return uniform(x);
        else return x;
    };
    const ctl rtol_f = sc((ctl)rtol), rtol_E_f = sc((ctl)rtol_E), atol_f = sc(fmaxf((ctl)atol, 1e-37f)), atol_E_f = sc(fmaxf((ctl)atol_E, 1e-37f));
    const real early_until = sc((nt > 1) ? (real)(tune.early_frac * ts[1]) : (real)0);
    const ctl early_mul = (ctl)(1.0 / fmax(tune.early_factor, 1.0));  // norms of an early step are scaled by this
    const ctl tail_mul = (ctl)(1.0 / tune.tail_factor), far_mul = (ctl)(1.0 / tune.far_factor), deep_mul = (ctl)(1.0 / (tune.far_factor * tune.deep_factor)), amp_mul = sc((ctl)amp_mul_d);
    const ctl newton_tol = sc((tune.newton_tol > 0.0)
                               ? (ctl)tune.newton_tol
                               : (ctl)fmax(10.0 * 2.220446049250313e-16 / (double)rtol, fmin(0.03, sqrt((double)rtol))));

    const bool goal = tune.shape_factor > 1.0;  // goal-oriented error test on
    int n_steps = 0, n_rhs = 0, n_fac = 0, n_bad = 0;  // n_bad: rejected steps | Newton failures << 16
    int st = MTRPL_SOLVE_OK;

    // dx-less grid sum of the signal density of one state
    auto sig_partial = [&](const real (&nn)[R], const real (&pp)[R], const real (&tr)[R]) {
        real s = (real)0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const real v = TRAPS ? signal_density(p, meas, nn[r], pp[r], tr[r]) : signal_density(p, meas, nn[r], pp[r]);
            s += (bk.node(r) < L) ? v : (real)0;
        }
        return s;
    };
    {
        real nn[R], pp[R], tr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { nn[r] = y[r]; pp[r] = y[R + r]; tr[r] = (real)0; }  // (:133: the traps start empty)
        bk.out_first(sig_partial(nn, pp, tr));
    }

    real z[6][V];
    real f[V];
    bk.rhs(y, f);
    n_rhs++;
#pragma unroll
    for (int v = 0; v < V; ++v) f[v] = (bk.node(v % R) < L) ? f[v] : (real)0;  // padded slots: rhs() leaves anything

    // ---- initial step (Hairer, Norsett & Wanner I, II.4, first-order variant) ----
    real h;
    {
        // weighted RMS norm; scaled by its largest entry so that the fp32 sum of squares cannot overflow
        auto tnorm = [&](const real (&vv)[V]) {
            real m = (real)0;
#pragma unroll
            for (int k = 0; k < V; ++k) m = fmax(m, fabs(vv[k]));
            m = uniform(bk.max_all(m));  // (readfirstlane: the compiler must know that the step size is wave-uniform)
            if (!(m > (real)0) || !(m < big<real>())) return m;
            const real im = (real)1 / m;
            float s1[1] = {0.0f};
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float q_ = (float)(vv[k] * im);
                s1[0] = fmaf(q_, q_, s1[0]);
            }
            bk.template reduce_f<1>(s1);
            return m * (real)__builtin_amdgcn_sqrtf(s1[0] * inv_S);
        };
        real w[V];
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
        bk.rhs(y1, f1);
        n_rhs++;
#pragma unroll
        for (int v = 0; v < V; ++v)
            w[v] = (bk.node(v % R) < L) ? (f1[v] - f[v]) / (atol_of(v) + rtol_of(v) * fabs(y[v])) : (real)0;
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
    OrderRow<BK::kOrderRowInVgpr> nq;
    nq.load(order, lane64);
    real t = (real)0;
    int have_fac = 0;
    ctl rate_prev = 1.0f;
    int rate_age = 0;
    int since_rej = 1000;  // accepted steps since the last rejected one
    int fails = 0;         // attempts in a row that did not end in an accepted step
    float acp[V];          // the correction of the previous step (order-increase estimate), fp32: it only enters a norm
#pragma unroll
    for (int v = 0; v < V; ++v) acp[v] = 0.0f;

    // The next 64 sample times sit in one register pair, lane k holding ts[io0 + k]: a sample time reaches the
    // scalar unit by v_readlane, and the scalar-memory load (with its wait on lgkmcnt, which the LDS traffic of
    // the Newton solver shares) is gone from the dense-output loop.
    int io = 1, io0 = 1;
    double tsv = (io0 + lane64 < nt) ? ts[io0 + lane64] : 1e300;
    real tq_next = (nt > 1) ? (real)lane_value(tsv, 0) : big<real>();

#ifdef MTRPL_STAMPS
    // controller vector work | factorisation | rhs | linear re-solve | reductions | Newton glue | dense output
    unsigned long long tlast_ = __builtin_amdgcn_s_memtime(), ph_ctl = 0, ph_fac = 0, ph_rhs = 0, ph_app = 0,
                       ph_red = 0, ph_glue = 0, ph_dense = 0;
#endif
    // ---- one pass of this loop = one ATTEMPT at a step ----
    while (io < nt) {
        MTRPL_STAMP(ph_ctl);
        if (n_steps >= tune.max_steps || fails > 60 || !(h > (real)1e-13 * fmax(t, (real)1e-3))) {
            st = MTRPL_SOLVE_STOPPED_EARLY;
            break;
        }
        if constexpr (!BK::kOrderRowInVgpr) nq.load(order, lane64);
        // the order / step-size selection runs after this step if it is accepted with n_equal >= order:
        // its two grid sums then travel with the Newton norms instead of costing a reduction of their own
        const bool will_select = n_equal >= order;
        // predictor: z <- A z; z[0] is the predicted state, z[1] the predicted h*y'
        nq_predict<5, V>(z);
        const real el0 = (real)nq.el(0);
        const real c = h * el0;
        // steps that end before early_frac x the first sample time get early_factor x the local tolerance: the
        // norms of such a step are scaled instead of its weights
        // ... and steps taken once the signal has fallen tail_frac (far_frac) below its first sample tail_factor x
        // (far_factor x)
        // ... and the steps of a tauP >> tauN solve amp x tighter while its decay is high-injection
        const ctl nscale = ((tune.early_factor > 1.0 && t + h <= early_until) ? early_mul
                                                                              : (bk.deep() ? deep_mul : (bk.far() ? far_mul : (bk.tail() ? tail_mul : 1.0f)))) *
                           (bk.post() ? 1.0f : amp_mul);
        // error weights of this step, 1/(atol + rtol |y_predicted|), fp32
        float isc[V];
#pragma unroll
        for (int v = 0; v < V; ++v)
            isc[v] = in_norm(v) ? __builtin_amdgcn_rcpf(fmaf(is_E(v) ? rtol_E_f : rtol_f, fabsf((float)z[0][v]), is_E(v) ? atol_E_f : atol_f)) : 0.0f;
        // per-node step: zero on the padded slots (node >= L), whose right-hand side is never looked at
        real hv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) hv[r] = (bk.node(r) < L) ? h : (real)0;
        // goal-oriented error test (DESIGN.md "error control"): error test value / its RMS part
        ctl rho = 1.0f;
        MTRPL_STAMP(ph_ctl);

        // ---- simplified Newton on acor:  (I - h l0 J) dx = h f(y) - z[1] - acor,  y = z[0] + l0 acor ----
        real acor[V];
        float acf[V];  // acor as fp32: what the norms see, and the next step's `acp`
        ctl dsm = 0.0f, ssel0 = 0.0f, ssel1 = 0.0f;
        int converged = 0, n_iter = 0, fresh = 0;
        for (;;) {  // with the kept factors, then once more with fresh ones
#pragma unroll
            for (int v = 0; v < V; ++v) acor[v] = (real)0;
            if (__builtin_expect(!have_fac, 0)) {  // one step in ten: spill code belongs around this block, not around the loop
                // the analytic Jacobian is evaluated inside the factorisation: always at the predictor
                MTRPL_STAMP(ph_glue);
                bk.build(z[0], c);
                MTRPL_STAMP(ph_fac);
                have_fac = 1;
                fresh = 1;
                n_fac++;
                rate_prev = 1.0f;  // new matrix: measure the rate afresh
                rate_age = 0;
            }
            ctl dn_old = 0.0f;
            for (int k = 0; k < 4; ++k) {
                n_iter = k + 1;
                MTRPL_STAMP(ph_glue);
                // the iterate is rebuilt from the correction where it is needed (six FMAs either way) instead of being
                // carried through the re-solve next to it: twelve registers less at the kernel's tightest spot
                real yn[V];
#pragma unroll
                for (int v = 0; v < V; ++v) yn[v] = fma(el0, acor[v], z[0][v]);
                bk.rhs(yn, f);
                MTRPL_STAMP(ph_rhs);
                n_rhs++;
                real b[V], dxv[V];
#pragma unroll
                for (int v = 0; v < V; ++v) b[v] = hv[v % R] * f[v] - z[1][v] - acor[v];
                MTRPL_STAMP(ph_glue);
                bk.apply(b, dxv);
                MTRPL_STAMP(ph_app);
                // the update is applied before the convergence decision (a rejected iterate is simply
                // discarded): the norms of the increment and of the whole correction then share one reduction
                float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    acor[v] += dxv[v];
                    if (!in_norm(v)) { acf[v] = 0.0f; continue; }
                    const float wv = (float)dxv[v] * isc[v];
                    acf[v] = (float)acor[v];
                    const float av = acf[v] * isc[v];
                    s4[0] = fmaf(wv, wv, s4[0]);
                    s4[1] = fmaf(av, av, s4[1]);
                }
                // goal-oriented test: the SIGNED grid sum of the correction of N (its mass component; padded nodes carry
                // an exactly zero correction) and the grid sum of the predicted EXCESS N - n0 -- what every signal is made
                // of --, in the same reduction as the norms
                float sfm[2] = {0.0f, 0.0f};
                if (goal) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        sfm[0] += acf[r];
                        // N - n0 + mass_floor n0, as ONE fp64 FMA per node (N -> n0 in a tail: the difference cannot be
                        // taken in fp32; the floor rides in the same instruction instead of in a register of its own)
                        sfm[1] += (bk.node(r) < L) ? (float)fma((real)(tune.mass_floor - 1.0), p.n0, z[0][r]) : 0.0f;
                    }
                }
                if (will_select) {
                    // stode: ddn from the last Nordsieck row (after the correction), dup from acor - acor_prev
                    const float elK = (float)nq.elK();
                    // (a switch with one loop per case: a select chain over z[1..5][v] is turned into a load with a
                    // run-time row index, and the whole history then lives in memory)
#define MTRPL_DDN_ROW(J) for (int v = 0; v < V; ++v) { if (!in_norm(v)) continue; const float q_ = fmaf(elK, acf[v], (float)z[J][v]) * isc[v]; s4[2] = fmaf(q_, q_, s4[2]); }
                    switch (order) {
                        case 1: MTRPL_DDN_ROW(1) break;
                        case 2: MTRPL_DDN_ROW(2) break;
                        case 3: MTRPL_DDN_ROW(3) break;
                        case 4: MTRPL_DDN_ROW(4) break;
                        default: MTRPL_DDN_ROW(5) break;
                    }
#undef MTRPL_DDN_ROW
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        if (!in_norm(v)) continue;
                        const float a_ = (acf[v] - acp[v]) * isc[v];
                        s4[3] = fmaf(a_, a_, s4[3]);
                    }
                    MTRPL_STAMP(ph_glue);
                    bk.template reduce_f<4>(s4);
                    if (goal) bk.template reduce_f<2>(sfm);
                } else if (goal) {
                    float g4[4] = {s4[0], s4[1], sfm[0], sfm[1]};
                    MTRPL_STAMP(ph_glue);
                    bk.template reduce_f<4>(g4);
                    s4[0] = g4[0];
                    s4[1] = g4[1];
                    sfm[0] = g4[2];
                    sfm[1] = g4[3];
                } else {
                    float s2[2] = {s4[0], s4[1]};
                    MTRPL_STAMP(ph_glue);
                    bk.template reduce_f<2>(s2);
                    s4[0] = s2[0];
                    s4[1] = s2[1];
                }
                MTRPL_STAMP(ph_red);
                // norm of the increment of y = l0 * dx
                const ctl dn = (ctl)el0 * nscale * __builtin_amdgcn_sqrtf(s4[0] * inv_S);
                if (!(dn < 1e30f)) break;  // NaN / inf
                ctl rate = rate_prev;
                const bool negligible = dn < 1e-2f * newton_tol;
                if (k > 0) {
                    rate = dn * __builtin_amdgcn_rcpf(dn_old);
                    if (!negligible) {
                        ctl rp = rate;  // rate^(4-k)
                        for (int q = 1; q < 4 - k; ++q) rp *= rate;
                        if (rate >= 1.0f || rp * dn > newton_tol * (1.0f - rate)) break;
                    }
                    if (rate < 1.0f) {
                        rate_prev = (rate_prev >= 1.0f && tune.rate_first_safety > 0.0)
                                        ? fminf(fmaxf((ctl)tune.rate_first_safety * rate, 1e-4f), 0.9f)
                                        : fmaxf(0.3f * rate_prev, rate);
                        rate_age = 0;
                    }
                }
                const bool rate_ok = (k > 0) || (rate_prev < 0.9f && rate_age < tune.rate_max_age);
                if (negligible || (rate_ok && rate < 1.0f && rate * dn < newton_tol * (1.0f - rate))) {
                    converged = 1;
                    dsm = nscale * __builtin_amdgcn_sqrtf(s4[1] * inv_S) * (ctl)nq.inv_t2();
                    if (goal) {
                        // RMS norm against shape_factor x the tolerance, mass component against the tolerance / fun_weight
                        // (relative to the excess while that is above mass_floor x the equilibrium density, to that level below --
                        // the floor is already in the sum --: a low-injection trace reaches its eighth decade with N - n0 at
                        // a hundredth of n0, held-out trace 17)
                        // (eleven decades down the tolerance is relaxed 30 000-fold, N may sit a thousandth BELOW n0 and the sum
                        // cross zero: nothing is resolved there and there is no mass to test against -- the shape test remains)
                        const ctl dfun = (sfm[1] > 0.0f && !bk.deep()) ? nscale * (ctl)tune.fun_weight * fabsf(sfm[0]) * __builtin_amdgcn_rcpf(rtol_f * sfm[1]) * (ctl)nq.inv_t2() : 0.0f;
                        const ctl dboth = fmaxf(dsm * (ctl)(1.0 / tune.shape_factor), dfun);
                        rho = (dsm > 0.0f) ? dboth * __builtin_amdgcn_rcpf(dsm) : 1.0f;
                        dsm = dboth;
                    }
                    ssel0 = s4[2];
                    ssel1 = s4[3];
                    break;
                }
                dn_old = dn;
            }
            if (converged || fresh) break;
            have_fac = 0;  // refactor with the Jacobian at this predictor, then try once more
        }
        if (converged && n_iter == 1) rate_age++;
        MTRPL_STAMP(ph_glue);
        // 0.9 * (2*4 + 1) / (2*4 + n_iter), n_iter = 1..4
        const ctl safety = (n_iter == 1) ? 0.9f : ((n_iter == 2) ? 0.81f : ((n_iter == 3) ? (ctl)(8.1 / 11.0) : 0.675f));
        // What the step does to the history afterwards: nothing, or z[j] <- kf[j] z[j] + af[j] acor (a change of step
        // size and/or order), preceded by a retraction of the prediction when the attempt is thrown away.  ONE place
        // writes the history back and the loop has one back edge: the register allocation of the 72-register history
        // does not have to be reconciled between five ways round the loop.
        bool rework = false;
        real facz = (real)1;
        int qn = order;
        if (!converged || dsm > 1.0f) {
            // Newton failed with fresh factors (halve the step) or the error test failed (stode's factor)
            TRACE(converged ? 1 : 2, n_iter, dsm);
            // rejected steps in the low half (saturating), Newton failures in the high half
            n_bad += converged ? (((n_bad & 0xffff) != 0xffff) ? 1 : 0) : (1 << 16);
            fails++;
            if (converged) since_rej = 0;
            facz = converged ? (real)fmaxf(0.2f, safety * pow_ctl(dsm, -nq.inv_qp1())) : (real)0.5;
            nq_retract<5, V>(z);
#pragma unroll
            for (int v = 0; v < V; ++v) acor[v] = (real)0;
            rework = true;
        } else {
            // ---- accepted ----
            TRACE(0, n_iter, dsm);
            fails = 0;
            n_steps++;
            n_equal++;
            since_rej++;
            t += h;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const real elj = (real)nq.el(j);  // zero above the order: those rows stay zero
#pragma unroll
                for (int v = 0; v < V; ++v) z[j][v] += elj * acor[v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) acp[v] = acf[v];
            MTRPL_STAMP(ph_ctl);
            // dense output (Taylor polynomial of the Nordsieck array), forward_solver.py:167: only N and P enter the signal
            if (!(tq_next > t * (real)(1.0 + 1e-15))) {
                const real ih = frcp(h);
                // The kind of signal is wave-uniform and fixed for the solve: the sample loop exists once per kind, so that
                // no branch stands between the Horner chains of a sample (written with the kind tested per node the
                // compiler kept four branches per sample, and the four five-deep FMA chains ran one after the other).
                auto samples = [&](auto kind_) {
                    constexpr int KIND = decltype(kind_)::value;
                    do {
                        const real s_ = (tq_next - t) * ih;
                        real nn[R], pp[R], tr[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            nn[r] = z[5][r];
                            pp[r] = z[5][R + r];
                            tr[r] = TRAPS ? z[5][(TRAPS ? 3 : 0) * R + r] : (real)0;
                        }
#pragma unroll
                        for (int j = 4; j >= 0; --j)
#pragma unroll
                            for (int r = 0; r < R; ++r) {
                                nn[r] = nn[r] * s_ + z[j][r];
                                if (KIND == MTRPL_MEAS_TRPL) pp[r] = pp[r] * s_ + z[j][R + r];  // (the photoconductivity is emitted from N alone)
                                if (TRAPS) tr[r] = tr[r] * s_ + z[j][(TRAPS ? 3 : 0) * R + r];
                            }
                        real sp = (real)0;
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const real v = TRAPS ? signal_density(p, KIND, nn[r], pp[r], tr[r]) : signal_density(p, KIND, nn[r], pp[r]);
                            sp += (bk.node(r) < L) ? v : (real)0;
                        }
                        bk.out_sample(io, sp);
                        io++;
                        if (io - io0 >= 64) {
                            io0 = io;
                            tsv = (io0 + lane64 < nt) ? ts[io0 + lane64] : 1e300;
                        }
                        tq_next = (io < nt) ? (real)lane_value(tsv, io - io0) : big<real>();
                    } while (!(tq_next > t * (real)(1.0 + 1e-15)));
                };
                if (meas == MTRPL_MEAS_TRPL) samples(std::integral_constant<int, MTRPL_MEAS_TRPL>{});
                else samples(std::integral_constant<int, MTRPL_MEAS_TRTS>{});
                if constexpr (BK::kParksSamples) bk.out_flush();  // (backends that park the samples of a step: one exchange for all of them)
            }
            MTRPL_STAMP(ph_dense);
            if (will_select) {
                // ---- order / step-size selection (stode) ----
                // The three candidate factors are x^e with their own x and e: evaluated as ONE log2 / multiply / exp2
                // sequence on lanes 0..2 (nothing here is lane-uniform until the three results are read back).
                ctl fm, f0, fp;
                {
                    // (goal-oriented test: the neighbouring orders' estimates are taken to split into shape and mass
                    // like this order's)
                    const ctl ddn = rho * nscale * (ctl)nq.inv_t1() * __builtin_amdgcn_sqrtf(ssel0 * inv_S);
                    const ctl dup = rho * nscale * (ctl)nq.inv_t3() * __builtin_amdgcn_sqrtf(ssel1 * inv_S);
                    ctl x = (lane64 == 0) ? ddn : ((lane64 == 1) ? dsm : dup);
                    const ctl e = (lane64 == 0) ? nq.nik0() : ((lane64 == 1) ? nq.nik1() : nq.nik2());
                    x = fminf(fmaxf(x, 1e-30f), 1e30f);
                    const ctl pw = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * e);
                    fm = (order > 1) ? lane_value(pw, 0) : 0.0f;
                    f0 = lane_value(pw, 1);
                    fp = (order < 5) ? lane_value(pw, 2) : 0.0f;
                }
                // stode's biases: a change of order has to promise more than staying (1.2 / 1.3 / 1.4)
                int dorder = 0;
                ctl fbest = f0 * (ctl)tune.nq_bias_same;
                if (fm * (ctl)tune.nq_bias_down > fbest) { fbest = fm * (ctl)tune.nq_bias_down; dorder = -1; }
                if (fp * (ctl)tune.nq_bias_up > fbest) { fbest = fp * (ctl)tune.nq_bias_up; dorder = 1; }
                ctl facf = fminf(10.0f, (tune.nq_bias_same < 1.0 ? 1.0f : safety) * fbest);
                // a step that was rejected a moment ago says the estimates behind `fac` are optimistic here (stiff
                // surface terms): grow cautiously for a while instead of hunting (reject / shrink / regrow / reject)
                if (since_rej < tune.nq_rej_memory) facf = fminf(facf, (ctl)tune.nq_rej_growth);
                real fac = (real)facf;
                // max_step guards LSODA against stepping over structure; once the signal is down in the rounding noise
                // of N P - n0 p0 (hmax_deep, 11 decades by default) there is none left and the error test alone decides
                if (tune.honor_hmax && h * fac > hmax && !bk.deep()) fac = hmax * frcp(h);
                if (dorder == 0 && fac >= (real)0.9 && fac < (real)tune.nq_deadband_hi) {
                    n_equal = order - 2;  // keep h and the factors; test again three steps on
                } else {
                    facz = fac;
                    qn = order + dorder;
                    rework = true;
                }
            }
        }
        if (rework) {
            // Row j = 1..5 is scaled by fac^j; on an order increase the new last row is acor * l_q / (q+1) (times
            // fac^(q+1)), on an order decrease the dropped row is zeroed: z[j] <- kf[j] z[j] + af[j] acor, wave-uniform
            // kf, af (acor is zero after a discarded attempt).
            const real el_up = (real)nq.el_up();
            real r = (real)1;
#pragma unroll
            for (int j = 1; j < 6; ++j) {
                r *= facz;
                const real kf = (j <= qn && j <= order) ? r : (real)0;
                const real af = (qn > order && j == qn) ? el_up * r : (real)0;
#pragma unroll
                for (int v = 0; v < V; ++v) z[j][v] = kf * z[j][v] + af * acor[v];
            }
            if constexpr (BK::kOrderRowInVgpr) {
                if (qn != order) nq.load(qn, lane64);
            }
            order = qn;
            h *= facz;
            n_equal = 0;
            have_fac = 0;
        }
    }

    if constexpr (BK::kBlocksSamples) {
        if (st == MTRPL_SOLVE_OK) bk.out_finish();
    }
    if (st != MTRPL_SOLVE_OK) {
        // forward_solver.py:168: samples after the last solved time come from a zeroed state
        real zsig = (real)0;
#pragma unroll
        for (int r = 0; r < R; ++r) zsig += (bk.node(r) < L) ? signal_density_literal(p, meas, (real)0, (real)0) : (real)0;
        for (; io < nt; ++io) bk.out_sample(io, zsig);
        if constexpr (BK::kParksSamples) bk.out_flush();
        if constexpr (BK::kBlocksSamples) bk.out_finish();
    } else if (tol_clamped) {
        st = MTRPL_SOLVE_OK_TOL_CLAMPED;
    }
    if (bk.leader()) {
        status[sidx] = st;
        if (stats) {
            int32_t *s = stats + 8 * sidx;
            s[0] = n_steps; s[1] = n_rhs; s[2] = n_fac; s[3] = n_fac + 1;  // a Jacobian per factorisation + the initial one
            s[4] = n_bad & 0xffff; s[5] = n_bad >> 16; s[6] = order;
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
