// irf_kernels.hpp -- instrument-response convolution (laplace.py) and the
// likelihood tail (trial_move_evaluation.py:96-158) on the device.
#pragma once
#include "device_math.hpp"

namespace mtrpl {

constexpr int kMaxIrfTables = 8;
struct IrfTableSet {
    int32_t n;
    int32_t nk[kMaxIrfTables];
    int64_t off[kMaxIrfTables];  // offset (in doubles) of table [nk,3] in the table buffer
    double dt[kMaxIrfTables];    // mean(diff(t_irf)), laplace.py:66
};

// numpy.linspace(a, b, num)[i]
__device__ inline double linspace_at(double a, double b, int num, int i)
{
    const double step = (b - a) / (double)(num - 1);
    return (i == num - 1) ? b : a + (double)i * step;
}

// scipy.integrate.simpson(f, dx=du) for NU equally spaced samples f(0..NU-1): composite Simpson for an odd
// number of samples; for an even number the basic rule on the first NU-1 samples plus the Cartwright
// correction on the last interval (SciPy >= 1.11).
template <class F>
__device__ inline double simpson_uniform(F val, int NU, double du)
{
    double result = 0.0;
    if (NU % 2 == 1) {
        for (int i = 0; i < NU - 2; i += 2) result += val(i) + 4.0 * val(i + 1) + val(i + 2);
        return result * (du / 3.0);
    }
    if (NU == 2) return 0.5 * du * (val(0) + val(1));
    for (int i = 0; i < NU - 3; i += 2) result += val(i) + 4.0 * val(i + 1) + val(i + 2);
    result *= du / 3.0;
    const double h0 = du, h1 = du;
    const double alpha = (2 * h1 * h1 + 3 * h0 * h1) / (6 * (h1 + h0));
    const double beta = (h1 * h1 + 3.0 * h0 * h1) / (6 * h0);
    const double eta = (h1 * h1 * h1) / (6 * h0 * (h0 + h1));
    return result + alpha * val(NU - 1) + beta * val(NU - 2) - eta * val(NU - 3);
}

// laplace.py:129-167 (I_moment): dt * simpson((u - 1/2)^n * linspace(y[m+1-u_lower], y[m+1-u_upper], NU), dx = du),
// u = linspace(u_lower, u_upper, NU).
// ia, ib: the (already wrapped, numpy-style) indices m+1-u_lower, m+1-u_upper
__device__ inline double irf_moment(const double *__restrict__ t, const double *__restrict__ y, int ia, int ib, int n,
                                    int u_lower, int u_upper, int NU)
{
    const double dt = t[1] - t[0];
    const double du = linspace_at((double)u_lower, (double)u_upper, NU, 1) - linspace_at((double)u_lower, (double)u_upper, NU, 0);
    const double ya = y[ia], yb = y[ib];
    auto val = [&](int i) {
        const double u = linspace_at((double)u_lower, (double)u_upper, NU, i);
        const double yi = linspace_at(ya, yb, NU, i);
        double w = 1.0;
        for (int k = 0; k < n; ++k) w *= (u - 0.5);
        return w * yi;
    };
    return dt * simpson_uniform(val, NU, du);
}

// laplace.py:13-41 (make_I_tables): one thread per (lag m, order n in 0..2), u_spacing = 1000; the last
// row stays zero.
__global__ void make_I_table_kernel(const double *__restrict__ t, const double *__restrict__ y, int nk,
                                    double *__restrict__ table)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 3 * nk) return;
    const int m = idx / 3, n = idx % 3;
    table[idx] = (m >= nk - 1) ? 0.0 : irf_moment(t, y, m + 1, m, n, 0, 1, 1000);
}
// one moment with the caller's settings (laplace.I_moment)
__global__ void I_moment_kernel(const double *__restrict__ t, const double *__restrict__ y, int ia, int ib, int n,
                                int u_lower, int u_upper, int NU, double *__restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = irf_moment(t, y, ia, ib, n, u_lower, u_upper, NU);
}
// forward_solver.py:267-274: calculate_RR / calculate_photoc, element-wise
__global__ void density_kernel(const double *__restrict__ N, const double *__restrict__ P, int64_t n, int meas, double ks,
                               double mun, double mup, double n0, double p0, double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = (meas == MTRPL_MEAS_TRPL) ? ks * (N[i] * P[i] - n0 * p0) : kQC * (mun * (N[i] - n0) + mup * (P[i] - p0));
}

// 1-D linear interpolation exactly as scipy.interpolate.griddata does it for 1-D
// points (interp1d linear, NaN outside the data): laplace.py:74, :121.
__device__ inline double interp_linear(const double *x, const double *y, int n, double v)
{
    if (!(v >= x[0] && v <= x[n - 1])) return __builtin_nan("");
    int lo = 0, hi = n;  // np.searchsorted(x, v, side="left")
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x[mid] < v) lo = mid + 1; else hi = mid;
    }
    int idx = lo < 1 ? 1 : lo;
    if (idx > n - 1) idx = n - 1;
    const double slope = (y[idx] - y[idx - 1]) / (x[idx] - x[idx - 1]);
    return slope * (v - x[idx - 1]) + y[idx - 1];
}

__global__ void interp_kernel(const double *__restrict__ x, const double *__restrict__ y, int n,
                              const double *__restrict__ xq, int64_t nq, double *__restrict__ yq)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nq) yq[i] = interp_linear(x, y, n, xq[i]);
}

// resampled_t[i] of laplace.py:68-72
__device__ inline double resampled_time(int64_t i, int64_t n_r, double half_dt, double t_end)
{
    double tq = (double)i * half_dt;
    if (i == n_r - 1 && tq > t_end) tq = t_end;
    return tq;
}

// laplace.py:68-77: resample (t, y) onto the dt_irf/2 grid
__global__ void resample_kernel(const double *__restrict__ t, const double *__restrict__ y, int nt, double half_dt,
                                int64_t n_r, double *__restrict__ ry, int32_t *__restrict__ nan_flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_r) return;
    const double v = interp_linear(t, y, nt, resampled_time(i, n_r, half_dt, t[nt - 1]));
    ry[i] = v;
    if (v != v) atomicOr(nan_flag, 1);
}

// laplace.py:170-212 (convolve): h[0] = 0,
// h[k] = sum_{j=max(0,k-nk)}^{k-1} I0[j] T[k-1-j][0] + I1[j] T[k-1-j][1] + I2[j] T[k-1-j][2].
// 256 outputs per block; the block's window of I0/I1/I2 and the table sit in LDS.
constexpr int kConvBlock = 256;
__global__ __launch_bounds__(kConvBlock) void convolve_kernel(const double *__restrict__ ry, int64_t n,
                                                              const double *__restrict__ table, int nk,
                                                              double *__restrict__ h)
{
    extern __shared__ double smem[];
    double *T = smem;                     // [nk][3]
    double *I0 = T + 3 * nk;              // [kConvBlock + nk]
    double *I1 = I0 + (kConvBlock + nk);
    double *I2 = I1 + (kConvBlock + nk);
    const int64_t nkk = (n - 1) / 2;      // last output index
    const int64_t k0 = (int64_t)blockIdx.x * kConvBlock;
    for (int i = threadIdx.x; i < 3 * nk; i += kConvBlock) T[i] = table[i];
    const int64_t jbase = k0 - nk;        // first j needed by output k0 (may be < 0)
    for (int i = threadIdx.x; i < kConvBlock + nk; i += kConvBlock) {
        const int64_t j = jbase + i;
        double a = 0.0, b = 0.0, c = 0.0;
        if (j >= 0 && j < nkk) {
            const double y0 = ry[2 * j], y1 = ry[2 * j + 1], y2 = ry[2 * j + 2];
            c = 2 * (y2 - 2 * y1 + y0);  // I2, laplace.py:198
            b = (y2 - y0);               // I1, :199
            a = y1;                      // I0, :200
        }
        I0[i] = a; I1[i] = b; I2[i] = c;
    }
    __syncthreads();
    const int64_t k = k0 + threadIdx.x;
    if (k > nkk) return;
    double acc = 0.0;
    const int64_t taps = (k < nk) ? k : nk;
    // j ascending == lag descending (same order as the sequential CPU sum)
    for (int64_t m = taps - 1; m >= 0; --m) {
        const int li = (int)(k - 1 - m - jbase);
        acc += (I0[li] * T[3 * m] + I1[li] * T[3 * m + 1] + I2[li] * T[3 * m + 2]);
    }
    h[k] = acc;
}

// np.argmax ordering: a NaN beats every number, larger beats smaller, ties and
// NaN-vs-NaN go to the smaller index.  (v, i) with i >= n is "empty".
__device__ inline bool argmax_better(double vb, int64_t ib, double va, int64_t ia, int64_t n)
{
    if (ib >= n) return false;
    if (ia >= n) return true;
    const bool an = va != va, bn = vb != vb;
    if (an || bn) return (an && bn) ? (ib < ia) : bn;
    return vb > va || (vb == va && ib < ia);
}

// block-wide argmax of per-thread candidates (256 threads); result in sv[0], si[0]
// first-maximum over the block (any number of full waves): wave-level butterfly, then one thread
// combines the per-wave winners.  sv/si: >= blockDim.x/64 entries of LDS.
__device__ inline void block_argmax(double &best, int64_t &bi, int64_t n, double *sv, int64_t *si)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(best, o, 64);
        const int64_t oi = __shfl_xor((long long)bi, o, 64);
        if (argmax_better(ov, oi, best, bi, n)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double b = sv[0];
        int64_t i = si[0];
        for (int w = 1; w < nw; ++w)
            if (argmax_better(sv[w], si[w], b, i, n)) { b = sv[w]; i = si[w]; }
        sv[0] = b;
        si[0] = i;
    }
    __syncthreads();
    best = sv[0];
    bi = si[0];
    __syncthreads();
}

__global__ __launch_bounds__(256) void argmax_kernel(const double *__restrict__ h, int64_t n, int64_t *__restrict__ out)
{
    __shared__ double sv[4];
    __shared__ int64_t si[4];
    double best = 0.0;
    int64_t bi = n;
    for (int64_t i = threadIdx.x; i < n; i += 256)
        if (argmax_better(h[i], i, best, bi, n)) { best = h[i]; bi = i; }
    block_argmax(best, bi, n, sv, si);
    if (threadIdx.x == 0) out[0] = bi;
}

// conv_t = resampled_t[::2] (- conv_t[argmax] when shifting), laplace.py:80-84
__global__ void conv_time_kernel(int64_t n_conv, int64_t n_r, double half_dt, double t_end, const int64_t *kmax,
                                 int shift, double *__restrict__ conv_t)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_conv) return;
    const double base = shift ? resampled_time(2 * kmax[0], n_r, half_dt, t_end) : 0.0;
    const double v = resampled_time(2 * k, n_r, half_dt, t_end);
    conv_t[k] = shift ? v - base : v;
}

// ---------------------------------------------------------------------------
// Fused likelihood tail, one 256-thread block per (chain, measurement):
// laplace.py:44-126 + trial_move_evaluation.py:96-158.  The convolved trace is
// never written to HBM: pass A finds argmax over all n_conv samples, pass B
// recomputes only the (at most 2 per experimental time) samples that the
// re-interpolation onto the experimental times needs.
//
// The resampled signal of laplace.py:68-74 is a LINEAR interpolant of the nt
// simulated samples, so inside one simulation interval the three Simpson moments
// of laplace.py:198-200 are I2 = 0, I1 = 2*delta*slope, I0 = linear in j, and the
// 189-tap sum of laplace.py:204-210 over a run of such intervals collapses to
// differences of three prefix sums of the moment table:
//   sum_m (a - c1*m) T0[m] + c1*T1[m] = a*dP0 - c1*dQ0 + c1*dP1 .
// A window of 189 taps crosses 3-4 simulation intervals, so one output costs a
// few dozen flops instead of 567 FMAs; intervals that straddle a simulation
// sample (and the snapped last one) are evaluated directly from the definition.
constexpr int kLikeBlock = 768;   // 8 waves per trace: two blocks per CU keep 4 waves per SIMD in flight
constexpr int kLikeWaves = kLikeBlock / 64;
constexpr int kLikeLdsBytes = 159 * 1024;  // dynamic LDS a block may ask for (160 KB per CU minus the kernel's static
                                           // reduction scratch): above this the per-sample arrays move to global memory
constexpr int kLikeGworkBlocks = 1024;     // blocks of the global-work-space variant (4 per CU)

struct ConvCtx {
    const double *st, *sy, *sb;   // LDS: times, values, slopes of the simulation intervals
    const double *T;              // LDS: moment table [nk][3]
    const double *P0, *Q0, *P1;   // LDS: prefix sums [nk+1]
    const int *jlo, *jhi;         // LDS [nt-1]: regular run of interval q = convolution indices jlo[q]..jhi[q]
    int nt, nk;
    int64_t n_r, nkk;
    double half_dt, t_end, inv_2d;
};

// interval index interp_linear would use for time v (v inside [st[0], st[nt-1]])
__device__ inline int conv_segment(const ConvCtx &c, double v)
{
    int lo = 0, hi = c.nt;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.st[mid] < v) lo = mid + 1; else hi = mid;
    }
    int idx = lo < 1 ? 1 : lo;
    if (idx > c.nt - 1) idx = c.nt - 1;
    return idx - 1;
}

__device__ inline double conv_interp(const ConvCtx &c, double v)
{
    if (!(v >= c.st[0] && v <= c.st[c.nt - 1])) return __builtin_nan("");
    const int q = conv_segment(c, v);
    return c.sb[q] * (v - c.st[q]) + c.sy[q];
}

// same interval as conv_segment, found by walking from a nearby guess q (consecutive
// outputs and consecutive runs of one output move forward by at most an interval or two)
__device__ inline int conv_segment_near(const ConvCtx &c, double v, int q)
{
    while (q < c.nt - 2 && c.st[q + 1] < v) ++q;
    while (q > 0 && c.st[q] >= v) --q;
    return q;
}

// h[k] of laplace.py:204-210.  qhint: interval guess for the first tap (in/out).
__device__ inline double conv_fast(const ConvCtx &c, int64_t k, int &qhint)
{
    if (k <= 0) return 0.0;
    const int64_t lo = (k - c.nk > 0) ? k - c.nk : 0, hi = k - 1;
    double acc = 0.0;
    int64_t cur = lo;
    int q = qhint;
    bool first = true;
    while (cur <= hi) {
        const double t0 = resampled_time(2 * cur, c.n_r, c.half_dt, c.t_end);
        const double t1 = resampled_time(2 * cur + 1, c.n_r, c.half_dt, c.t_end);
        const double t2 = resampled_time(2 * cur + 2, c.n_r, c.half_dt, c.t_end);
        q = conv_segment_near(c, t1, q);
        if (first) { qhint = q; first = false; }
        const bool regular = (t0 >= c.st[q]) && (t2 <= c.st[q + 1]) && (cur < c.nkk - 1);
        if (regular) {
            // last interval j of this run: tau_{2j+2} <= st[q+1]
            int64_t je = (int64_t)floor(c.st[q + 1] * c.inv_2d) - 1;
            while (resampled_time(2 * je + 4, c.n_r, c.half_dt, c.t_end) <= c.st[q + 1]) ++je;
            while (resampled_time(2 * je + 2, c.n_r, c.half_dt, c.t_end) > c.st[q + 1]) --je;
            if (je > c.nkk - 2) je = c.nkk - 2;
            if (je < cur) je = cur;
            const int64_t jb = je < hi ? je : hi;
            const int ma = (int)(k - 1 - jb), mb = (int)(k - 1 - cur);
            const double c1 = 2.0 * c.half_dt * c.sb[q];
            const double yc = c.sb[q] * (t1 - c.st[q]) + c.sy[q];  // I0 at j = cur
            // I0[j] = yc + c1*(j - cur) = (yc + c1*mb) - c1*m
            acc += (yc + c1 * (double)mb) * (c.P0[mb + 1] - c.P0[ma]) - c1 * (c.Q0[mb + 1] - c.Q0[ma]) +
                   c1 * (c.P1[mb + 1] - c.P1[ma]);
            cur = jb + 1;
        } else {
            const double y0 = conv_interp(c, t0), y1 = conv_interp(c, t1), y2 = conv_interp(c, t2);
            const double *tt = c.T + 3 * (k - 1 - cur);
            acc += (y1 * tt[0] + (y2 - y0) * tt[1] + (2 * (y2 - 2 * y1 + y0)) * tt[2]);
            ++cur;
        }
    }
    return acc;
}

// Regular run of simulation interval q: the convolution indices j whose three resampled points
// tau_2j, tau_2j+1, tau_2j+2 all lie inside [st[q], st[q+1]] (and j < nkk-1: the last index uses the
// snapped end point).  jhi is non-decreasing in q whether or not the run is empty (jlo > jhi).
__device__ inline void conv_runs(const ConvCtx &c, int q, int &jl_out, int &jh_out)
{
    const double a = c.st[q], b = c.st[q + 1];
    int64_t je = (int64_t)floor(b * c.inv_2d) - 1;
    while (resampled_time(2 * je + 4, c.n_r, c.half_dt, c.t_end) <= b) ++je;
    while (je >= 0 && resampled_time(2 * je + 2, c.n_r, c.half_dt, c.t_end) > b) --je;
    if (je > c.nkk - 2) je = c.nkk - 2;
    int64_t jl = (int64_t)ceil(a * c.inv_2d);
    if (jl < 0) jl = 0;
    while (jl > 0 && resampled_time(2 * (jl - 1), c.n_r, c.half_dt, c.t_end) >= a) --jl;
    while (resampled_time(2 * jl, c.n_r, c.half_dt, c.t_end) < a) ++jl;
    jl_out = (int)jl;
    jh_out = (int)je;
}

__device__ inline double conv_interp_near(const ConvCtx &c, double v, int qhint)
{
    if (!(v >= c.st[0] && v <= c.st[c.nt - 1])) return __builtin_nan("");
    const int q = conv_segment_near(c, v, qhint);
    return c.sb[q] * (v - c.st[q]) + c.sy[q];
}

// h[k] of laplace.py:204-210 with the run table: the same sums, in the same order and with the same
// arithmetic as conv_fast, but the decomposition of the tap window into runs is looked up instead of
// re-derived (floor + searches) for every output.  q (in/out): first interval with jhi[q] >= k - nk;
// monotone in k, so a thread walking consecutive outputs advances it by amortised O(1).
__device__ inline double conv_fast2(const ConvCtx &c, int64_t k64, int &q)
{
    if (k64 <= 0) return 0.0;
    // 32-bit index arithmetic: the caller guarantees n_r < 2^31 (checked once per trace)
    const int k = (int)k64, nr1 = (int)(c.n_r - 1);
    auto tau = [&](int i) {  // resampled_time(i): laplace.py:68-72, last point snapped to t_end
        double tq = (double)i * c.half_dt;
        if (i == nr1 && tq > c.t_end) tq = c.t_end;
        return tq;
    };
    const int lo = (k - c.nk > 0) ? k - c.nk : 0, hi = k - 1;
    const int qlast = c.nt - 2;
    while (q < qlast && c.jhi[q] < lo) ++q;
    int qq = q;
    double acc = 0.0;
    int cur = lo;
    while (cur <= hi) {
        while (qq < qlast && c.jhi[qq] < cur) ++qq;
        const int jl = c.jlo[qq], je = c.jhi[qq];
        if (cur >= jl && cur <= je) {
            const double t1 = tau(2 * cur + 1);
            const int jb = je < hi ? je : hi;
            const int ma = k - 1 - jb, mb = k - 1 - cur;
            const double c1 = 2.0 * c.half_dt * c.sb[qq];
            const double yc = c.sb[qq] * (t1 - c.st[qq]) + c.sy[qq];  // I0 at j = cur
            acc += (yc + c1 * (double)mb) * (c.P0[mb + 1] - c.P0[ma]) - c1 * (c.Q0[mb + 1] - c.Q0[ma]) +
                   c1 * (c.P1[mb + 1] - c.P1[ma]);
            cur = jb + 1;
        } else {
            // a tap that straddles a simulation sample (or the snapped last one): from the definition
            const double t0 = tau(2 * cur), t1 = tau(2 * cur + 1), t2 = tau(2 * cur + 2);
            const double y0 = conv_interp_near(c, t0, qq), y1 = conv_interp_near(c, t1, qq),
                         y2 = conv_interp_near(c, t2, qq);
            const double *tt = c.T + 3 * (k - 1 - cur);
            acc += (y1 * tt[0] + (y2 - y0) * tt[1] + (2 * (y2 - 2 * y1 + y0)) * tt[2]);
            ++cur;
        }
    }
    return acc;
}

// first interval q with jhi[q] >= j (binary search; for random-access callers of conv_fast2)
__device__ inline int conv_run_of(const ConvCtx &c, int64_t j)
{
    int lo = 0, hi = c.nt - 2;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (c.jhi[mid] < j) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// block-wide NaN-propagating maxima of three values (any number of full waves); sv: >= 3*blockDim.x/64 doubles
// of LDS.  Every thread gets the results.
__device__ inline double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }
__device__ inline void block_max3(double &a, double &b, double &c, double *sv)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a = nanmax(a, __shfl_xor(a, o, 64));
        b = nanmax(b, __shfl_xor(b, o, 64));
        c = nanmax(c, __shfl_xor(c, o, 64));
    }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[3 * wave] = a; sv[3 * wave + 1] = b; sv[3 * wave + 2] = c; }
    __syncthreads();
    a = sv[0]; b = sv[1]; c = sv[2];
    for (int w = 1; w < nw; ++w) { a = nanmax(a, sv[3 * w]); b = nanmax(b, sv[3 * w + 1]); c = nanmax(c, sv[3 * w + 2]); }
    __syncthreads();
}
// block-wide sums of three values, same scratch
__device__ inline void block_sum3(double &a, double &b, double &c, double *sv)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); c += __shfl_xor(c, o, 64); }
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[3 * wave] = a; sv[3 * wave + 1] = b; sv[3 * wave + 2] = c; }
    __syncthreads();
    a = sv[0]; b = sv[1]; c = sv[2];
    for (int w = 1; w < nw; ++w) { a += sv[3 * w]; b += sv[3 * w + 1]; c += sv[3 * w + 2]; }
    __syncthreads();
}

// One trace: everything the block needs sits behind the pointers it is given.  `work` holds the
// per-sample arrays (6*max_nt doubles: times, signal, slopes, compared signal, 2u^2, run table), `irfw` the
// IRF work space (moment table + its prefix sums); the kernel below decides which of them are LDS.
template <int BLK>
__device__ inline void loglik_trace(const mtrpl_like_desc ds, int64_t sidx, const double *__restrict__ sig,
                                    const double *__restrict__ tsteps, const double *__restrict__ exp_t,
                                    const double *__restrict__ exp_y, const double *__restrict__ exp_u,
                                    const double *__restrict__ temps, int nT, int max_nt,
                                    const double *__restrict__ tables, const IrfTableSet &set, double neg_frac_tol,
                                    double *__restrict__ ll, double *work, double *irfw, double *red_v,
                                    int64_t *red_i, int *s_cnt, int stage_doubles)
{
    const int nt = ds.nt, n_exp = ds.n_exp, tid = threadIdx.x;
    if (nt > max_nt || n_exp > max_nt || n_exp > nt || nt < 2 || n_exp < 1) {  // caller broke its promise
        for (int j = tid; j < nT; j += BLK) ll[sidx * nT + j] = __builtin_nan("");
        return;
    }
    double *st = work;            // [nt] simulation times
    double *sy = st + nt;         // [nt] simulated signal
    double *sb = sy + nt;         // [nt] interval slopes (laplace.py:74 via interp1d)
    double *sol = sb + nt;        // [n_exp] signal at the experimental times; later err^2
    double *su2 = sol + n_exp;    // [n_exp] 2*unc^2
    double *wk = irfw;            // IRF work space: T[3nk], P0,Q0,P1[nk+1]
    for (int i = tid; i < nt; i += BLK) { st[i] = tsteps[ds.t_off + i]; sy[i] = sig[ds.sig_off + i]; }
    __syncthreads();
    for (int i = tid; i < nt - 1; i += BLK) sb[i] = (sy[i + 1] - sy[i]) / (st[i + 1] - st[i]);
    const double *et = exp_t + ds.exp_off, *ey = exp_y + ds.exp_off, *eu = exp_u + ds.exp_off;
    int n_c = n_exp;  // number of compared samples
    bool failed = false;
    if (ds.irf >= 0) {
        const int nk = set.nk[ds.irf];
        const double *Tg = tables + set.off[ds.irf];
        double *T = wk, *P0 = T + 3 * nk, *Q0 = P0 + (nk + 1), *P1 = Q0 + (nk + 1);
        int *jlo = reinterpret_cast<int *>(su2 + n_exp), *jhi = jlo + nt;  // run table, 2*nt ints
        for (int i = tid; i < 3 * nk; i += BLK) T[i] = Tg[i];
        __syncthreads();
        if (tid < 3) {  // prefix sums of the moment table (sequential: same rounding every time)
            double a = 0.0;
            double *dst = (tid == 0) ? P0 : ((tid == 1) ? Q0 : P1);
            dst[0] = 0.0;
            for (int m = 0; m < nk; ++m) {
                a += (tid == 0) ? T[3 * m] : ((tid == 1) ? (double)m * T[3 * m] : T[3 * m + 1]);
                dst[m + 1] = a;
            }
        }
        __syncthreads();
        ConvCtx c;
        c.st = st; c.sy = sy; c.sb = sb; c.T = T; c.P0 = P0; c.Q0 = Q0; c.P1 = P1;
        c.nt = nt; c.nk = nk;
        c.half_dt = set.dt[ds.irf] / 2;
        c.t_end = st[nt - 1];
        c.inv_2d = 1.0 / (2.0 * c.half_dt);
        // len(np.arange(0, t_end + dt/4, dt/2)), laplace.py:68
        c.n_r = (int64_t)ceil((c.t_end + set.dt[ds.irf] / 4) / c.half_dt);
        c.nkk = (c.n_r - 1) / 2;  // last convolved index
        if (c.n_r >= (int64_t)1 << 30) {  // index arithmetic below is 32-bit (a 2000 ns trace at 0.025 ns has n_r = 160 001)
            for (int j = tid; j < nT; j += BLK) ll[sidx * nT + j] = __builtin_nan("");
            return;
        }
        const int64_t nkk = c.nkk, n_r = c.n_r;
        const double half_dt = c.half_dt, t_end = c.t_end;
        c.jlo = jlo; c.jhi = jhi;
        for (int q = tid; q < nt - 1; q += BLK) conv_runs(c, q, jlo[q], jhi[q]);
        __syncthreads();
        // ---- pass A: argmax of the convolved trace (laplace.py:82) ----
        // The outputs are walked in chunks, and the walk stops as soon as no later output can still win.
        // Every tap of h[k] is y1*T0 + (y2 - y0)*T1 + 2*(y2 - 2*y1 + y0)*T2 with y0, y1, y2 values of the linear
        // interpolant half_dt apart at times >= tau(2*(k - nk)).  With Y = max |sy| and S = max |slope| over the
        // intervals that reach past tau(2*(k0 - nk)):  |y1| <= Y, |y2 - y0| <= 2*half_dt*S,
        // |2*(y2 - 2*y1 + y0)| <= 4*half_dt*S, so for all k >= k0
        //     |h[k]| <= Y*sum|T0| + half_dt*S*(2*sum|T1| + 4*sum|T2|),
        // and once that lies strictly below the maximum found so far, argmax is decided (ties go to the smaller
        // index anyway).  A TRPL decay peaks in its first nanoseconds: most of the 80 000 outputs of a 2000 ns
        // trace are never computed.  A NaN sample keeps the bound NaN and the walk complete; a NaN output found
        // ends it (np.argmax returns the first NaN).
        double w0 = 0.0, w1 = 0.0, w2 = 0.0;
        for (int m = tid; m < nk; m += BLK) { w0 += fabs(T[3 * m]); w1 += fabs(T[3 * m + 1]); w2 += fabs(T[3 * m + 2]); }
        block_sum3(w0, w1, w2, red_v);
        const double wy = w0 * (1.0 + 1e-9), ws = half_dt * (2.0 * w1 + 4.0 * w2) * (1.0 + 1e-9);
        double best = 0.0;
        int64_t bi = nkk + 1;
        {
            // outputs interleaved over the threads: neighbouring lanes work on neighbouring k, so their
            // windows share the runs (LDS broadcasts) and their prefix-sum indices are consecutive
            // (conflict-free); the run pointer still only moves forward per thread
            constexpr int64_t kChunk = 2 * BLK;
            int qh = conv_run_of(c, (tid - nk > 0) ? tid - nk : 0);
            for (int64_t k0 = 0; k0 <= nkk; k0 += kChunk) {
                if (k0 > 0) {
                    const double t_lo = (k0 > nk) ? (double)(2 * (k0 - nk)) * half_dt * (1.0 - 1e-12) : 0.0;
                    double ymax = 0.0, smax = 0.0;
                    for (int i = tid; i < nt; i += BLK) {
                        if (st[i + 1 < nt ? i + 1 : i] >= t_lo) {
                            const double a = fabs(sy[i]);
                            ymax = (a > ymax || a != a) ? a : ymax;
                            if (i < nt - 1) {
                                const double b = fabs(sb[i]);
                                smax = (b > smax || b != b) ? b : smax;
                            }
                        }
                    }
                    double bmax = (best != best) ? __builtin_inf() : best;  // -> the best value of the whole block so far
                    block_max3(bmax, ymax, smax, red_v);
                    if (ymax * wy + smax * ws < bmax) break;                // the same decision in every thread
                }
                const int64_t k_end = (k0 + kChunk <= nkk) ? k0 + kChunk : nkk + 1;
                for (int64_t k = k0 + tid; k < k_end; k += BLK) {
                    const double acc = conv_fast2(c, k, qh);
                    if (argmax_better(acc, k, best, bi, nkk + 1)) { best = acc; bi = k; }
                }
            }
        }
        block_argmax(best, bi, nkk + 1, red_v, red_i);
        const double t_shift = resampled_time(2 * bi, n_r, half_dt, t_end);
        const double ct_first = -t_shift;  // conv_t[0] after the shift
        const double ct_last = resampled_time(2 * nkk, n_r, half_dt, t_end) - t_shift;
        if (ct_last == 0.0 || best != best) failed = true;  // laplace.py:83-84, :76-77
        // ---- post_conv_trim (laplace.py:120-124): exp_t < max(conv_t) ----
        int cnt = 0;
        for (int i = tid; i < n_exp; i += BLK) cnt += (et[i] < ct_last) ? 1 : 0;
        if (tid == 0) s_cnt[0] = 0;
        __syncthreads();
        atomicAdd(&s_cnt[0], cnt);
        __syncthreads();
        // exp_t is increasing: the cut keeps indices 0..last index with exp_t < ct_last
        n_c = s_cnt[0];
        if (n_c == 0) failed = true;  // np.where(...)[0][-1] raises IndexError in the reference
        // ---- pass B: re-interpolate onto the experimental times (laplace.py:121) ----
        for (int i = tid; i < n_c; i += BLK) {
            const double te = et[i];
            double v;
            if (!(te >= ct_first && te <= ct_last)) {
                v = __builtin_nan("");
            } else {
                // searchsorted(conv_t, te, "left") on conv_t[k] = r_t[2k] - t_shift
                int64_t lo = 0, hi = nkk + 1;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (resampled_time(2 * mid, n_r, half_dt, t_end) - t_shift < te) lo = mid + 1; else hi = mid;
                }
                int64_t idx = lo < 1 ? 1 : lo;
                if (idx > nkk) idx = nkk;
                const double x0 = resampled_time(2 * (idx - 1), n_r, half_dt, t_end) - t_shift;
                const double x1 = resampled_time(2 * idx, n_r, half_dt, t_end) - t_shift;
                int qh = conv_run_of(c, (idx - 1 - nk > 0) ? idx - 1 - nk : 0);
                const double h0 = conv_fast2(c, idx - 1, qh), h1 = conv_fast2(c, idx, qh);
                const double slope = (h1 - h0) / (x1 - x0);
                v = slope * (te - x0) + h0;
            }
            sol[i] = v;
        }
    } else {
        // trial_move_evaluation.py:98-101: sol = sol[-len(times_c):]
        for (int i = tid; i < n_exp; i += BLK) sol[i] = sy[nt - n_exp + i];
    }
    __syncthreads();
    // ---- trial_move_evaluation.py:117-130: negative-fraction check, flip ----
    int neg = 0;
    for (int i = tid; i < n_c; i += BLK) neg += (sol[i] < 0) ? 1 : 0;
    if (tid == 0) s_cnt[1] = 0;
    __syncthreads();
    atomicAdd(&s_cnt[1], neg);
    __syncthreads();
    if (!((double)s_cnt[1] < neg_frac_tol * (double)n_c)) failed = true;
    for (int i = tid; i < n_c; i += BLK)
        if (sol[i] < 0) sol[i] *= -1;  // :130
    if (ds.force_min_y) {
        // utils.py:16-31 (set_min_y): min_y = 10 ** min(vals - scale_shift);
        // i_final = np.searchsorted(-sol, -min_y); sol[i_final:] = min_y
        double vmin = __builtin_inf();
        for (int i = tid; i < n_c; i += BLK) vmin = fmin(vmin, ey[i] - ds.scale_shift);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmin = fmin(vmin, __shfl_xor(vmin, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) red_v[tid >> 6] = vmin;
        __syncthreads();
        double vm = red_v[0];
        for (int w = 1; w < (BLK / 64); ++w) vm = fmin(vm, red_v[w]);
        const double min_y = pow(10.0, vm);
        if (tid == 0) {
            int lo = 0, hi = n_c;  // lower_bound of -min_y in -sol, exactly numpy's bisection
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (-sol[mid] < -min_y) lo = mid + 1; else hi = mid;
            }
            s_cnt[0] = lo;
        }
        __syncthreads();
        const int i_final = s_cnt[0];
        __syncthreads();
        for (int i = tid; i < n_c; i += BLK)
            if (i >= i_final) sol[i] = min_y;
        __syncthreads();
    }
    // ---- :147-156: err_sq and ll(T) = -sum err_sq / (sigma^2 T + 2 u^2) ----
    for (int i = tid; i < n_c; i += BLK) {
        const double s = sol[i];
        const double e = (log10(s) + ds.scale_shift - ey[i]);
        sol[i] = e * e;
        su2[i] = 2 * (eu[i] * eu[i]);
    }
    __syncthreads();
    const double sg2 = ds.sigma * ds.sigma;
    if (nT >= 32) {
        // many ladder temperatures: kLadderSplit threads per temperature walk interleaved points, their partial
        // sums are added in a fixed order (the IRF work space is free by now and holds them)
        constexpr int kLadderSplit = 8;
        double *parts = irfw;  // [nT][kLadderSplit] -- when that fits in the 6*nk+3 doubles of the IRF work space
        const bool fits = ds.irf >= 0 && (size_t)nT * kLadderSplit <= 6 * (size_t)set.nk[ds.irf];
        if (fits) {
            for (int u = tid; u < nT * kLadderSplit; u += BLK) {
                const int j = u / kLadderSplit, r = u - j * kLadderSplit;
                const double sT = sg2 * temps[j];
                double tot = 0.0;
                for (int i = r; i < n_c; i += kLadderSplit) tot += sol[i] * frcp1(sT + su2[i]);  // hardware seed + one Newton step: 1e-16, no div ladder
                parts[u] = tot;
            }
            __syncthreads();
            for (int j = tid; j < nT; j += BLK) {
                double tot = 0.0;
                for (int r = 0; r < kLadderSplit; ++r) tot += parts[j * kLadderSplit + r];
                tot = -tot;
                if (failed || tot != tot) tot = -__builtin_inf();  // :70-72,:159-165
                ll[sidx * nT + j] = tot;
            }
        } else {
            const double *e2 = sol, *u2 = su2;
            if (2 * n_c <= stage_doubles) {
                // (one-wave blocks keep their per-sample arrays in device memory: every thread walks all samples here,
                // so they move into the LDS the IRF work space no longer needs)
                for (int i = tid; i < n_c; i += BLK) { irfw[i] = sol[i]; irfw[n_c + i] = su2[i]; }
                __syncthreads();
                e2 = irfw;
                u2 = irfw + n_c;
            }
            // one thread per temperature, its samples in order; a thread that owns several temperatures (a small block
            // under a long ladder) walks the samples once for all of them: independent chains for the reciprocals
            constexpr int KT = (BLK >= 512) ? 1 : ((BLK >= 256) ? 2 : ((BLK >= 128) ? 4 : 8));
            for (int j0 = tid; j0 < nT; j0 += BLK * KT) {
                double sT[KT], tot[KT];
                const int kmax = (nT - (j0 - tid) + BLK - 1) / BLK;  // temperatures left for the block's first thread: uniform
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int j = j0 + k * BLK;
                    sT[k] = sg2 * temps[j < nT ? j : nT - 1];
                    tot[k] = 0.0;
                }
                for (int i = 0; i < n_c; ++i) {
                    const double e = e2[i], u = u2[i];
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        if (k < kmax) tot[k] += e * frcp1(sT[k] + u);
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const int j = j0 + k * BLK;
                    double t = -tot[k];
                    if (failed || t != t) t = -__builtin_inf();
                    if (j < nT) ll[sidx * nT + j] = t;
                }
            }
        }
    } else {
        for (int j = 0; j < nT; ++j) {
            const double sT = sg2 * temps[j];
            double part = 0.0;
            for (int i = tid; i < n_c; i += BLK) part += sol[i] / (sT + su2[i]);
            part = wave_sum(part);
            __syncthreads();
            if ((tid & 63) == 0) red_v[tid >> 6] = part;
            __syncthreads();
            if (tid == 0) {
                double tot = 0.0;
                for (int w = 0; w < (BLK / 64); ++w) tot += red_v[w];
                tot = -tot;
                if (failed || tot != tot) tot = -__builtin_inf();
                ll[sidx * nT + j] = tot;
            }
        }
    }
}

// One BLK-thread block (768: twelve waves) per (chain, measurement) while the per-sample arrays fit in LDS (GWORK = false:
// 48 bytes per sample, 3400 samples in 160 KB next to a 189-tap IRF).  Longer traces -- real TRPL files
// run to tens of thousands of samples, and the reference has no limit (trial_move_evaluation.py:96-158)
// -- keep only the IRF work space in LDS and put the per-sample arrays in a per-block slab of global
// memory (GWORK = true; L2-resident, written and read by this block only), blocks striding over the
// traces.  BLK = 64 / 128 / 256 (GWORK only, nT >= 32 -- there no sum depends on the block size: the same numbers bit for
// bit): one, two or four waves per block and 9 KB of LDS, a block that fits wherever one (two, four) solve waves have left -- for likelihood launches
// that share the device with another stream's queued solves, where a twelve-wave block with 57 KB of LDS waits
// until that queue has drained.
template <bool GWORK, int BLK>
__global__ __launch_bounds__(BLK) void loglik_kernel(const mtrpl_like_desc *__restrict__ descs,
                                                            const double *__restrict__ sig,
                                                            const double *__restrict__ tsteps,
                                                            const double *__restrict__ exp_t,
                                                            const double *__restrict__ exp_y,
                                                            const double *__restrict__ exp_u,
                                                            const double *__restrict__ temps, int nT, int max_nt,
                                                            const double *__restrict__ tables, IrfTableSet set,
                                                            double neg_frac_tol, double *__restrict__ ll, int64_t B,
                                                            double *__restrict__ gwork, int stage_doubles)
{
    extern __shared__ double smem[];
    __shared__ double red_v[3 * (BLK / 64)];
    __shared__ int64_t red_i[(BLK / 64)];
    __shared__ int s_cnt[2];
    double *work = GWORK ? gwork + (size_t)blockIdx.x * 6 * (size_t)max_nt : smem;
    double *irfw = GWORK ? smem : smem + 6 * (size_t)max_nt;
    if (!GWORK) {  // one block per trace
        const int64_t sidx = blockIdx.x;
        if (sidx < B)
            loglik_trace<BLK>(descs[sidx], sidx, sig, tsteps, exp_t, exp_y, exp_u, temps, nT, max_nt, tables, set, neg_frac_tol,
                         ll, work, irfw, red_v, red_i, s_cnt, 0);
        return;
    }
    for (int64_t sidx = blockIdx.x; sidx < B; sidx += gridDim.x) {
        loglik_trace<BLK>(descs[sidx], sidx, sig, tsteps, exp_t, exp_y, exp_u, temps, nT, max_nt, tables, set, neg_frac_tol,
                     ll, work, irfw, red_v, red_i, s_cnt, stage_doubles);
        __syncthreads();
    }
}

}  // namespace mtrpl
