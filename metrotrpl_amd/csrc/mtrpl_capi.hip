// mtrpl_capi.hip -- C ABI (include/metrotrpl_amd.h) + kernel launchers.
// gfx950 only.  There is deliberately no CPU fallback anywhere in this file:
// without a HIP device every entry point fails with MTRPL_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/metrotrpl_amd.h"
#include "irf_kernels.hpp"
#include "backend_wave.hpp"
#include "backend_team.hpp"
#include "backend_spike.hpp"

using namespace mtrpl;

namespace {
// Elimination-block slab of one in-flight nx > 256 launch (4 waves per solve keep their blocks in HBM).
// A slab belongs to exactly one launch at a time: it is handed out again only once the event recorded
// behind that launch has completed, or to a later launch on the SAME stream (stream order serialises).
struct Slab {
    void *p = nullptr;
    size_t bytes = 0;
    hipStream_t stream = nullptr;
    hipEvent_t done = nullptr;
    unsigned long long taken_in = 0;  // the API call that handed it out last (one call may take two slabs)
};
}  // namespace

struct mtrpl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    SolveTuning tune = DefaultTune::make();
    std::string err;
    // IRF tables resident on the device (mtrpl_irf_tables_upload)
    double *d_irf_tab = nullptr;
    std::vector<int32_t> irf_nk;
    std::vector<int64_t> irf_off;
    std::vector<double> irf_dt;
    // (deques: a Slab* handed out stays valid when a later take_slab of the same call appends to the pool)
    std::deque<Slab> slabs;        // elimination-block slabs / likelihood work spaces, sized by the launch
    std::vector<Slab> retired;     // outgrown slabs: freed when the context goes (hipFree would stall every stream)
    // grow-only device scratch of the blocking host-pointer entry points (no hipMalloc/hipFree per call)
    static constexpr int kScratch = 8;
    void *scratch[kScratch] = {};
    size_t scratch_cap[kScratch] = {};
    int kernel_variant = 0;  // see include/metrotrpl_amd.h
    int precision = 64;      // 64: fp64 (the product); 32: the fp32 build of the team kernel (precision sweep only)
    size_t like_lds_attr = 64 * 1024;  // dynamic-LDS limit the likelihood kernel has been raised to
    unsigned long long call_id = 0;
    int loglik_wave_blocks = 0;  // 1 / 2 / 4: likelihood launches with nT >= 32 run that many waves per block (see include/metrotrpl_amd.h)
    int two_per_simd_min_batch = 1025;  // kernel_variant 0: launches of at least this many std / nx <= 128 solves use
                                        // the two-solves-per-SIMD build of the wave kernel (more than one solve per
                                        // SIMD anyway), smaller ones the one-per-SIMD build: same results
    // The pipelined host-pointer solve (mtrpl_solve_batch with more than host_chunk solves): streams the context owns,
    // a ring of pinned staging slots for callers whose output buffer is pageable, one event per slot.
    static constexpr int kPipeMax = 8;
    hipStream_t pipe[kPipeMax] = {};
    int n_pipe = 0;
    hipEvent_t pipe_in = nullptr;
    std::vector<hipEvent_t> pipe_done;
    void *stage = nullptr;
    size_t stage_cap = 0;
    int host_chunk = 2048;   // solves per sub-launch at nx <= 128 (/ 2, / 4 for the two- and four-wave kernels)
    int host_streams = 2;    // sub-launches in flight (measured at B = 8192: 2 x 2048 520 k solves/s, 4 x 1024 500 k, one
                             // launch 495 k; and the fewer streams, the smaller the chance that two of them share a hardware queue)
    std::vector<void *> host_allocs;  // mtrpl_host_alloc
};

// A failed HIP call ends the entry point -- after whatever the blocking host-pointer entry points have already enqueued
// on the context's own stream has drained: their copies go into the CALLER's host buffers and out of device scratch the
// next call reuses, and neither may be touched by work still in flight once the error code is back.
#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            if ((ctx)->stream) (void)hipStreamSynchronize((ctx)->stream);                    \
            return MTRPL_ERR_DEVICE;                                                         \
        }                                                                                    \
    } while (0)

namespace {

// device scratch slot `slot` of the context, at least n bytes; contents undefined
struct DevBuf {
    void *p = nullptr;
    mtrpl_ctx *ctx;
    int slot;
    DevBuf(mtrpl_ctx *c, int s) : ctx(c), slot(s) {}
    hipError_t alloc(size_t n)
    {
        if (n < 8) n = 8;
        if (ctx->scratch_cap[slot] < n) {
            if (ctx->scratch[slot]) (void)hipFree(ctx->scratch[slot]);
            ctx->scratch[slot] = nullptr;
            ctx->scratch_cap[slot] = 0;
            const size_t cap = n + n / 4;
            hipError_t e = hipMalloc(&ctx->scratch[slot], cap);
            if (e != hipSuccess) return e;
            ctx->scratch_cap[slot] = cap;
        }
        p = ctx->scratch[slot];
        return hipSuccess;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};

int bad(mtrpl_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// ---- small kernels for the reference's helper functions ------------------------
// reference layouts: std [N(L),P(L),E(L+1)], traps [N(L),Ntrap(L),P(L),E(L+1)]
template <class BK>
__device__ inline void load_state(const BK &bk, const double *ys, int L, double (&yl)[BK::V], double pad_np)
{
    constexpr int R = BK::R;
    const int oP = (BK::TRAPS ? 2 : 1) * L, oE = (BK::TRAPS ? 3 : 2) * L;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = bk.node(r);
        yl[r] = (i < L) ? ys[i] : pad_np;
        yl[R + r] = (i < L) ? ys[oP + i] : pad_np;
        yl[2 * R + r] = (i < L) ? ys[oE + i] : 0.0;
        if (BK::TRAPS) yl[3 * R + r] = (i < L) ? ys[L + i] : 0.0;
    }
}

// forward_solver.py:332-418 through a backend's own RHS
template <class BK, class... LdsArgs>
__device__ inline void rhs_probe(BK &bk, const double *par, const double *dxs, const double *y, double *dy, int nx,
                                 int64_t s)
{
    constexpr int R = BK::R, V = BK::V;
    const int L = nx, S = (BK::TRAPS ? 4 : 3) * nx + 1;
    const int oP = (BK::TRAPS ? 2 : 1) * L, oE = (BK::TRAPS ? 3 : 2) * L;
    bk.set_problem(make_par<double>(par + 16 * s, dxs[s]), L);
    double yl[V], fl[V];
    load_state(bk, y + s * S, L, yl, 1.0);
    bk.rhs(yl, fl);
    double *fs = dy + s * S;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = bk.node(r);
        if (i < L) {
            fs[i] = fl[r];
            fs[oP + i] = fl[R + r];
            fs[oE + i] = fl[2 * R + r];
            if (BK::TRAPS) fs[L + i] = fl[3 * R + r];
        }
    }
    if (bk.leader()) fs[oE + L] = 0.0;  // dE_L = -(Jn_L + Jp_L)*Lambda = 0 (forward_solver.py:350-352,:363)
}

__global__ __launch_bounds__(64) void rhs_batch_kernel_wave(const double *par, const double *dxs, const double *y,
                                                            double *dy, int nx, int64_t B)
{
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    WaveBackend bk(threadIdx.x, nullptr);
    rhs_probe(bk, par, dxs, y, dy, nx, s);
}
template <int W, bool TRAPS>
__global__ __launch_bounds__(64 * W) void rhs_batch_kernel_team(const double *par, const double *dxs, const double *y,
                                                                double *dy, int nx, int64_t B)
{
    using BK = TeamBackend<double, W, TRAPS, true>;
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    __shared__ double2 xbuf_lds[BK::kXbufRows];
    __shared__ double rbuf_lds[BK::kRbufVals];
    BK bk(threadIdx.x, xbuf_lds, rbuf_lds, nullptr);
    rhs_probe(bk, par, dxs, y, dy, nx, s);
}

// one Newton linear solve (I - c*J(y)) x = b through a backend's factorisation (std layout)
template <class BK>
__device__ inline void newton_probe(BK &bk, const double *par, const double *dxs, const double *cs, const double *y,
                                    const double *b, double *x, int nx, int64_t s)
{
    constexpr int R = BK::R, V = BK::V;
    const int L = nx, S = 3 * nx + 1;
    bk.set_problem(make_par<double>(par + 16 * s, dxs[s]), L);
    double yl[V], bl[V], xl[V];
    load_state(bk, y + s * S, L, yl, 1.0);
    load_state(bk, b + s * S, L, bl, 0.0);
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (bk.node(r) == 0) bl[2 * R + r] = 0.0;  // the E_0 row of I - cJ is an identity row
    bk.build(yl, cs[s]);
    bk.apply(bl, xl);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = bk.node(r);
        if (i < L) {
            x[s * S + i] = xl[r];
            x[s * S + L + i] = xl[R + r];
            x[s * S + 2 * L + i] = (i > 0) ? xl[2 * R + r] : b[s * S + 2 * L];
        }
    }
    if (bk.leader()) x[s * S + 3 * L] = b[s * S + 3 * L];  // E_0 and E_L rows of I - cJ are identity rows
}
template <bool F32BLK>
__global__ __launch_bounds__(64) void newton_probe_kernel_wave(const double *par, const double *dxs, const double *cs,
                                                               const double *y, const double *b, double *x, int nx,
                                                               int64_t B)
{
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    __shared__ double2 fac_lds[(F32BLK ? kFactorLdsBytesF32 : kFactorLdsBytes) / 16];
    WaveBackendT<F32BLK> bk(threadIdx.x, fac_lds);
    newton_probe(bk, par, dxs, cs, y, b, x, nx, s);
}
template <int W>
__global__ __launch_bounds__(64 * W) void rhs_batch_kernel_spike(const double *par, const double *dxs, const double *y,
                                                                double *dy, int nx, int64_t B)
{
    using BK = SpikeBackendT<W>;
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    __shared__ float4 slabs[W * BK::kSlabBytes / 16];
    __shared__ double shared[BK::kSharedDoubles];
    BK bk(threadIdx.x, slabs, shared);
    rhs_probe(bk, par, dxs, y, dy, nx, s);
}
template <int W>
__global__ __launch_bounds__(64 * W) void newton_probe_kernel_spike(const double *par, const double *dxs, const double *cs,
                                                                   const double *y, const double *b, double *x, int nx,
                                                                   int64_t B)
{
    using BK = SpikeBackendT<W>;
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    __shared__ float4 slabs[W * BK::kSlabBytes / 16];
    __shared__ double shared[BK::kSharedDoubles];
    BK bk(threadIdx.x, slabs, shared);
    newton_probe(bk, par, dxs, cs, y, b, x, nx, s);
}
// the controller's merged fp32 wave reductions (device_math.hpp: wave_sums_f) on caller data: in [B][4][64], out [B][7] =
// the sum of row 0 through the 1-sum ladder, of rows 0-1 through the 2-sum ladder, of rows 0-3 through the 4-sum ladder
__global__ __launch_bounds__(64) void reduce_probe_kernel(const float *in, float *out, int64_t B)
{
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    const int lane = threadIdx.x;
    const float *q = in + s * 256;
    float s1[1] = {q[lane]}, s2[2] = {q[lane], q[64 + lane]}, s4[4] = {q[lane], q[64 + lane], q[128 + lane], q[192 + lane]};
    wave_sums_f<1>(s1, lane);
    wave_sums_f<2>(s2, lane);
    wave_sums_f<4>(s4, lane);
    if (lane == 0) {
        float *o = out + s * 7;
        o[0] = s1[0]; o[1] = s2[0]; o[2] = s2[1]; o[3] = s4[0]; o[4] = s4[1]; o[5] = s4[2]; o[6] = s4[3];
    }
}

// forward_solver.py:26-38: one wave per row, any nx (chunked inclusive scan)
__global__ __launch_bounds__(64) void e_field_kernel(const double *N, const double *P, double *E, int nx, int64_t rows,
                                                     double n0, double p0, double pref, double corner_E)
{
    const int64_t row = blockIdx.x;
    if (row >= rows) return;
    const int lane = threadIdx.x;
    const double *n = N + row * nx, *pp = P + row * nx;
    double *e = E + row * (nx + 1);
    if (lane == 0) e[0] = corner_E;
    double carry = 0.0;
    for (int base = 0; base < nx; base += 64) {
        const int j = base + lane;
        double v = (j < nx) ? ((pp[j] - p0) - (n[j] - n0)) : 0.0;
        v = wave_scan_incl(v, lane);
        v += carry;
        if (j < nx) e[j + 1] = corner_E + pref * v;
        carry = shf(v, 63);
    }
}

// forward_solver.py:228-274: one wave per time row
__global__ __launch_bounds__(64) void integral_kernel(const double *N, const double *P, double *out, int nx,
                                                      int64_t rows, double dx, int meas, double ks, double mun,
                                                      double mup, double n0, double p0)
{
    const int64_t row = blockIdx.x;
    if (row >= rows) return;
    const int lane = threadIdx.x;
    const double *n = N + row * nx, *pp = P + row * nx;
    double s = 0.0;
    for (int j = lane; j < nx; j += 64)
        s += (meas == MTRPL_MEAS_TRPL) ? ks * (n[j] * pp[j] - n0 * p0) : kQC * (mun * (n[j] - n0) + mup * (pp[j] - p0));
    s = wave_sum(s);
    if (lane == 0) out[row] = s * dx;  // integrate_1D == dx * sum(y) (end-corrected trapezoid on cell centres)
}

int check_descs(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d, int64_t n_ini, const double *tsteps, int64_t n_t,
                int64_t n_out)
{
    for (int64_t s = 0; s < B; ++s) {
        const mtrpl_solve_desc &q = d[s];
        if (q.nx < 2 || q.nt < 1) return bad(ctx, MTRPL_ERR_BAD_ARG, "nx must be >= 2 and nt >= 1");
        if (q.nx > 512) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 512 not supported by this build");
        if (q.model != MTRPL_MODEL_STD && q.model != MTRPL_MODEL_TRAPS)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "Invalid model");  // forward_solver.py:140
        if (q.model != d[0].model)
            return bad(ctx, MTRPL_ERR_UNSUPPORTED, "one batch must use one model (std or traps)");
        if (q.meas != MTRPL_MEAS_TRPL && q.meas != MTRPL_MEAS_TRTS)
            return bad(ctx, MTRPL_ERR_UNSUPPORTED, "TRTS or TRPL only");  // forward_solver.py:203
        if (q.ini_mode != MTRPL_INI_DENSITY && q.ini_mode != MTRPL_INI_FLUENCE)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "Invalid ini_mode - must be 'density' or 'fluence'");  // :117
        if (q.ini_mode == MTRPL_INI_DENSITY && (q.ini_off < 0 || q.ini_off + q.nx > n_ini))
            return bad(ctx, MTRPL_ERR_BAD_ARG, "initial density array too short");  // :104
        if (q.t_off < 0 || q.t_off + q.nt > n_t || q.out_off < 0 || q.out_off + q.nt > n_out)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "tsteps/out offsets out of range");
        if (tsteps && tsteps[q.t_off] != 0.0)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "Grid error - times must start at t=0");  // sim_utils.py:269-270
        if (!(q.thickness > 0) || !(q.hmax > 0) || !(q.rtol > 0) || !(q.atol >= 0))
            return bad(ctx, MTRPL_ERR_BAD_ARG, "thickness, hmax, rtol must be positive");
    }
    return MTRPL_OK;
}

// a slab of at least `need` bytes that no in-flight launch on another stream is using.  The pool holds at most
// one idle slab that is too small: growing REPLACES the smallest idle one instead of adding to it, so that a
// sequence of launches of varying size does not accumulate stale slabs.  A slab grows GEOMETRICALLY (to at least
// 1.5 x its old size), so a caller whose launches grow step by step retires a block only every few steps and the
// parked memory stays below twice the largest slab; the outgrown block is not freed here -- hipFree synchronises
// the whole device, i.e. stalls every other stream's launches in the middle of a launch -- but parked in
// ctx->retired and freed at a quiet point the CALLER names: mtrpl_ctx_trim, or with the context.
int take_slab(mtrpl_ctx *ctx, std::deque<Slab> &pool, size_t need, hipStream_t s, Slab **out)
{
    Slab *pick = nullptr, *small = nullptr;
    for (Slab &sl : pool) {
        if (sl.taken_in == ctx->call_id) continue;  // handed out earlier in this very call
        bool free_ = sl.stream == s;
        if (!free_) {
            const hipError_t q = hipEventQuery(sl.done);
            if (q == hipSuccess) free_ = true;
            else if (q != hipErrorNotReady) {
                ctx->err = std::string("hipEventQuery(slab): ") + hipGetErrorString(q);
                return MTRPL_ERR_DEVICE;
            }
        }
        if (!free_) continue;
        if (sl.bytes >= need) { if (!pick || sl.bytes < pick->bytes) pick = &sl; }
        else if (!small || sl.bytes < small->bytes) small = &sl;
    }
    if (!pick) {
        void *p = nullptr;
        if (small) need = std::max(need, small->bytes + small->bytes / 2);
        HIP_TRY(ctx, hipMalloc(&p, need));
        if (small) {  // idle (or ordered before us on this stream) and too small
            Slab old;
            old.p = small->p;
            old.bytes = small->bytes;
            ctx->retired.push_back(old);
            small->p = p;
            small->bytes = need;
            pick = small;
        } else {
            Slab sl;
            sl.p = p;
            sl.bytes = need;
            if (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
                (void)hipFree(sl.p);
                return bad(ctx, MTRPL_ERR_DEVICE, "hipEventCreate failed");
            }
            pool.push_back(sl);
            pick = &pool.back();
        }
    }
    pick->stream = s;
    pick->taken_in = ctx->call_id;
    *out = pick;
    return MTRPL_OK;
}
int take_slab(mtrpl_ctx *ctx, size_t need, hipStream_t s, Slab **out) { return take_slab(ctx, ctx->slabs, need, s, out); }
// behind the launch that uses the slab: the event its next user on another stream waits for.  If the launch itself
// failed the slab is handed back at once (stream = nullptr marks "no launch pending": the stale event is complete).
int seal_slab(mtrpl_ctx *ctx, Slab *sl, hipStream_t s, bool launched)
{
    if (!launched) { sl->stream = nullptr; return MTRPL_OK; }
    HIP_TRY(ctx, hipEventRecord(sl->done, s));
    return MTRPL_OK;
}

template <class T, int W, bool TRAPS, bool GBLK, bool BF32 = false>
void launch_team(hipStream_t s, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                 const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats,
                 const SolveTuning &tune, void *gblk)
{
    hipLaunchKernelGGL((solve_kernel_team<T, W, TRAPS, GBLK, BF32>), dim3((unsigned)B), dim3(64 * W), 0, s, d_descs, d_ini,
                       d_tsteps, d_out, d_status, d_stats, B, tune, gblk);
}
template <class T>
int launch_team_any(mtrpl_ctx *ctx, hipStream_t s, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                    const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                    bool traps)
{
    // fp64 solves store their elimination blocks as fp32 (half the LDS: twice the workgroups per CU at W = 2)
    constexpr bool kF32 = sizeof(T) == 8;
#define MTRPL_TEAM(W_, G_, P_)                                                                                       \
    do {                                                                                                             \
        if (traps) launch_team<T, W_, true, G_, kF32>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, P_); \
        else launch_team<T, W_, false, G_, kF32>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, P_);      \
    } while (0)
    if (max_nx <= 128) MTRPL_TEAM(1, false, nullptr);
    else if (max_nx <= 256) MTRPL_TEAM(2, false, nullptr);
    else {
        // 4 waves per solve, 21 elimination blocks per thread.  fp64 solves store them as fp32 (86 KB per solve):
        // that fits in LDS next to the exchange buffers (135 KB of the CU's 160; the register file allows one
        // such workgroup per CU anyway).  The fp32 instantiation (precision sweeps) keeps its blocks in a
        // per-solve slab in HBM, private to this launch.
        if constexpr (kF32) {
            MTRPL_TEAM(4, false, nullptr);
        } else {
            using BK4 = TeamBackend<T, 4, false, true>;
            const size_t need = (size_t)B * BK4::kBlkRows * sizeof(typename BK4::row_t);
            Slab *sl = nullptr;
            const int rc = take_slab(ctx, need, s, &sl);
            if (rc != MTRPL_OK) return rc;
            MTRPL_TEAM(4, true, sl->p);
            const bool launched = hipGetLastError() == hipSuccess;
            const int rs = seal_slab(ctx, sl, s, launched);
            if (!launched) return bad(ctx, MTRPL_ERR_DEVICE, "team kernel launch failed");
            if (rs != MTRPL_OK) return rs;
        }
    }
#undef MTRPL_TEAM
    return MTRPL_OK;
}

}  // namespace

extern "C" {

const char *mtrpl_version(void) { return "metrotrpl_amd 0.3 (gfx950)"; }

// HIP multiplexes streams onto 4 hardware queues by default (GPU_MAX_HW_QUEUES, read when the runtime starts, i.e. at the
// first HIP call of the process); the pipelined host path runs streams of its own next to whatever the caller has, and two
// streams on one queue serialise their launches (measured, B = 8192 host to host: 520 k solves/s, 390-430 k when two of the
// sub-launch streams share a queue -- with 8 queues as soon as the process owned ten streams, not with 16:
// profiles/r05_api_queues.txt).  The library does NOT touch the environment (rounds 4-5 did, from a load-time
// constructor: a process-wide setting changed behind the back of every other HIP user of the process, by a setenv that
// is not safe against a concurrent getenv of a multi-threaded host -- ADVICE r5): a C caller exports
// GPU_MAX_HW_QUEUES=16 itself, the Python package does it at import unless told not to (metrotrpl_amd/_lib.py).

int mtrpl_ctx_create(int device, mtrpl_ctx **out)
{
    if (!out) return MTRPL_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MTRPL_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return MTRPL_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return MTRPL_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MTRPL_ERR_DEVICE;
    mtrpl_ctx *c = new mtrpl_ctx();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return MTRPL_ERR_DEVICE;
    }
    *out = c;
    return MTRPL_OK;
}

int mtrpl_ctx_destroy(mtrpl_ctx *ctx)
{
    if (!ctx) return MTRPL_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    if (ctx->d_irf_tab) (void)hipFree(ctx->d_irf_tab);
    for (Slab &sl : ctx->slabs) {
        if (sl.p) (void)hipFree(sl.p);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    for (Slab &sl : ctx->retired) (void)hipFree(sl.p);
    for (int q = 0; q < ctx->n_pipe; ++q) (void)hipStreamDestroy(ctx->pipe[q]);
    if (ctx->pipe_in) (void)hipEventDestroy(ctx->pipe_in);
    for (hipEvent_t e : ctx->pipe_done) (void)hipEventDestroy(e);
    if (ctx->stage) (void)hipHostFree(ctx->stage);
    for (void *p : ctx->host_allocs) (void)hipHostFree(p);
    for (int k = 0; k < mtrpl_ctx::kScratch; ++k)
        if (ctx->scratch[k]) (void)hipFree(ctx->scratch[k]);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MTRPL_OK;
}

const char *mtrpl_last_error(const mtrpl_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

namespace {
struct OptD { const char *key; double SolveTuning::*m; };
struct OptI { const char *key; int SolveTuning::*m; };
const OptD kOptD[] = {
    {"rtol_scale", &SolveTuning::rtol_scale}, {"rtol_cap", &SolveTuning::rtol_cap},
    {"rtol_floor", &SolveTuning::rtol_floor}, {"rtol_min", &SolveTuning::rtol_min},
    {"rtol_force", &SolveTuning::rtol_force},
    {"atol_rel", &SolveTuning::atol_rel}, {"atol_rel_np", &SolveTuning::atol_rel_np}, {"newton_tol", &SolveTuning::newton_tol},
    {"rate_first_safety", &SolveTuning::rate_first_safety}, {"e_weight", &SolveTuning::e_weight},
    {"trts_rtol_factor", &SolveTuning::trts_rtol_factor}, {"amp_ref", &SolveTuning::amp_ref}, {"amp_post", &SolveTuning::amp_post},
    {"nq_deadband_hi", &SolveTuning::nq_deadband_hi}, {"nq_rej_growth", &SolveTuning::nq_rej_growth},
    {"nq_bias_same", &SolveTuning::nq_bias_same}, {"nq_bias_down", &SolveTuning::nq_bias_down},
    {"nq_bias_up", &SolveTuning::nq_bias_up}, {"early_factor", &SolveTuning::early_factor},
    {"early_frac", &SolveTuning::early_frac}, {"hmax_deep", &SolveTuning::hmax_deep},
    {"hmax_scale", &SolveTuning::hmax_scale}, {"tail_frac", &SolveTuning::tail_frac}, {"tail_factor", &SolveTuning::tail_factor},
    {"far_frac", &SolveTuning::far_frac}, {"far_factor", &SolveTuning::far_factor}, {"deep_factor", &SolveTuning::deep_factor},
    {"shape_factor", &SolveTuning::shape_factor}, {"fun_weight", &SolveTuning::fun_weight},
    {"mass_floor", &SolveTuning::mass_floor},
};
const OptI kOptI[] = {
    {"max_steps", &SolveTuning::max_steps}, {"rate_max_age", &SolveTuning::rate_max_age},
    {"honor_hmax", &SolveTuning::honor_hmax}, {"nq_rej_memory", &SolveTuning::nq_rej_memory},
    {"rtol_mode", &SolveTuning::rtol_mode},
};
// every option still at its shipped value: the kernel instantiation with the options compiled in may run
bool default_tune(const SolveTuning &t)
{
    const SolveTuning d = DefaultTune::make();
    for (const OptD &o : kOptD)
        if (t.*(o.m) != d.*(o.m)) return false;
    for (const OptI &o : kOptI)
        if (t.*(o.m) != d.*(o.m)) return false;
    return true;
}
}  // namespace

int mtrpl_set_option(mtrpl_ctx *ctx, const char *key, double value)
{
    if (!ctx || !key) return MTRPL_ERR_BAD_ARG;
    const std::string k(key);
    for (const OptD &o : kOptD)
        if (k == o.key) { ctx->tune.*(o.m) = value; return MTRPL_OK; }
    for (const OptI &o : kOptI)
        if (k == o.key) { ctx->tune.*(o.m) = (int)value; return MTRPL_OK; }
    if (k == "kernel_variant") {
        const int v = (int)value;
        if (v != 0 && v != 1 && v != 2 && v != 3 && v != 6)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "kernel_variant must be 0 (auto), 1 (team), 2 (wave, fp64 blocks), 3 / 6 (wave, fp32 blocks, two / one per SIMD)");
        ctx->kernel_variant = v;
    } else if (k == "fresh_jac") {
        // removed in round 2 (the Jacobian is always refreshed at a refactorisation): accepted and ignored
    } else if (k == "two_per_simd_min_batch") {
        ctx->two_per_simd_min_batch = (int)value;
    } else if (k == "host_chunk") {
        if (!(value >= 1)) return bad(ctx, MTRPL_ERR_BAD_ARG, "host_chunk must be >= 1");
        ctx->host_chunk = (int)value;
    } else if (k == "host_streams") {
        if (value < 0 || value > mtrpl_ctx::kPipeMax) return bad(ctx, MTRPL_ERR_BAD_ARG, "host_streams must be 0 (one blocking launch) ... 8");
        ctx->host_streams = (int)value;
    } else if (k == "loglik_wave_blocks") {
        if (value != 0 && value != 1 && value != 2 && value != 4)
            return bad(ctx, MTRPL_ERR_BAD_ARG, "loglik_wave_blocks must be 0 (twelve waves per trace), 1, 2 or 4 (waves per block)");
        ctx->loglik_wave_blocks = (int)value;
    } else if (k == "precision") {
        if (value != 32 && value != 64) return bad(ctx, MTRPL_ERR_BAD_ARG, "precision must be 32 or 64");
        ctx->precision = (int)value;
    } else {
        return bad(ctx, MTRPL_ERR_BAD_ARG, "unknown option");
    }
    return MTRPL_OK;
}

int mtrpl_get_option(const mtrpl_ctx *ctx, const char *key, double *value)
{
    if (!ctx || !key || !value) return MTRPL_ERR_BAD_ARG;
    const std::string k(key);
    for (const OptD &o : kOptD)
        if (k == o.key) { *value = ctx->tune.*(o.m); return MTRPL_OK; }
    for (const OptI &o : kOptI)
        if (k == o.key) { *value = ctx->tune.*(o.m); return MTRPL_OK; }
    if (k == "kernel_variant") *value = ctx->kernel_variant;
    else if (k == "two_per_simd_min_batch") *value = ctx->two_per_simd_min_batch;
    else if (k == "loglik_wave_blocks") *value = ctx->loglik_wave_blocks;
    else if (k == "host_chunk") *value = ctx->host_chunk;
    else if (k == "host_streams") *value = ctx->host_streams;
    else if (k == "precision") *value = ctx->precision;
    else if (k == "fresh_jac") *value = 0.0;  // accepted and ignored (set_option)
    else return MTRPL_ERR_BAD_ARG;
    return MTRPL_OK;
}

static int launch_solves(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                         const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                         int32_t model, void *stream, int64_t in_flight, int32_t per_simd = 0);

int mtrpl_solve_batch_dev(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                          const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                          int32_t model, void *stream)
{
    return launch_solves(ctx, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, max_nx, model, stream, B);
}

// ... with the choice between the two builds of the wave kernel made by the CALLER, per launch (0: as above)
int mtrpl_solve_batch_dev2(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                           const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                           int32_t model, void *stream, int32_t solves_per_simd)
{
    if (solves_per_simd < 0 || solves_per_simd > 2)
        return bad(ctx, MTRPL_ERR_BAD_ARG, "solves_per_simd must be 0 (by the size of the launch), 1 or 2");
    return launch_solves(ctx, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, max_nx, model, stream, B, solves_per_simd);
}

// in_flight: the solves the caller has (or is about to have) on the device together with this launch -- what the choice
// between the one- and the two-solves-per-SIMD build of the wave kernel goes by (a sub-launch of the pipelined host path
// says the size of the whole call); per_simd 1 / 2: the caller has chosen (mtrpl_solve_batch_dev2)
static int launch_solves(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                         const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                         int32_t model, void *stream, int64_t in_flight, int32_t per_simd)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!d_descs || !d_tsteps || !d_out || !d_status) return bad(ctx, MTRPL_ERR_BAD_ARG, "null device pointer");
    if (max_nx < 2 || max_nx > 512) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 512 not supported by this build");
    if (model != MTRPL_MODEL_STD && model != MTRPL_MODEL_TRAPS) return bad(ctx, MTRPL_ERR_BAD_ARG, "Invalid model");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->call_id++;
    hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream (NULL = the default stream)
    const bool traps = model == MTRPL_MODEL_TRAPS;
    if (ctx->precision == 32) {
        if (traps) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "the fp32 build covers the std model only");
        const int rc = launch_team_any<float>(ctx, s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, max_nx, false);
        if (rc != MTRPL_OK) return rc;
        HIP_TRY(ctx, hipGetLastError());
        return MTRPL_OK;
    }
    // std model, nx <= 128: the register-resident backends.
    //   wave1 (6): one wave per solve, two nodes per lane; elimination blocks stored as fp32, re-solve in packed
    //              fp32 on an fp64 right-hand side; one solve per SIMD -- the shortest solve
    //   wave2 (3): the same arithmetic on a 256-register budget: two solves per SIMD, 1.3-1.5x the solves per
    //              second once the SIMDs are over-subscribed (the second wave fills the issue slots the first
    //              leaves waiting)
    //   wave  (2): fp64 blocks and an fp64 re-solve: the exact linear algebra, one solve per SIMD
    int variant = ctx->kernel_variant;
    if (variant == 0)
        variant = per_simd ? (per_simd == 2 ? 3 : 6) : ((std::max(B, in_flight) >= ctx->two_per_simd_min_batch) ? 3 : 6);
    const bool ct = default_tune(ctx->tune);
#define MTRPL_WAVE(KERNEL)                                                                                            \
    do {                                                                                                              \
        if (ct) hipLaunchKernelGGL(KERNEL<true>, dim3((unsigned)B), dim3(64), 0, s, d_descs, d_ini, d_tsteps, d_out,  \
                                   d_status, d_stats, B, ctx->tune);                                                  \
        else hipLaunchKernelGGL(KERNEL<false>, dim3((unsigned)B), dim3(64), 0, s, d_descs, d_ini, d_tsteps, d_out,    \
                                d_status, d_stats, B, ctx->tune);                                                     \
    } while (0)
    if (traps || variant == 1) {
        const int rc = launch_team_any<double>(ctx, s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, max_nx, traps);
        if (rc != MTRPL_OK) return rc;
    } else if (max_nx > 128) {
        // std model, 128 < nx <= 512: W = 2 / 4 waves per solve, each with the wave kernel's register-resident linear
        // algebra on its own 128 nodes, coupled by a SPIKE interface solve (backend_spike.hpp)
#define MTRPL_SPIKE(W_)                                                                                               \
    do {                                                                                                              \
        if (ct) hipLaunchKernelGGL((solve_kernel_spike<W_, true>), dim3((unsigned)B), dim3(64 * W_), 0, s, d_descs,   \
                                   d_ini, d_tsteps, d_out, d_status, d_stats, B, ctx->tune);                          \
        else hipLaunchKernelGGL((solve_kernel_spike<W_, false>), dim3((unsigned)B), dim3(64 * W_), 0, s, d_descs,     \
                                d_ini, d_tsteps, d_out, d_status, d_stats, B, ctx->tune);                             \
    } while (0)
        if (max_nx <= 256) MTRPL_SPIKE(2);
        else MTRPL_SPIKE(4);
#undef MTRPL_SPIKE
    } else if (variant == 3) {
        MTRPL_WAVE(solve_kernel_wave2);
    } else if (variant == 6) {
        MTRPL_WAVE(solve_kernel_wave1);
    } else {
        MTRPL_WAVE(solve_kernel_wave);
    }
#undef MTRPL_WAVE
    HIP_TRY(ctx, hipGetLastError());
    return MTRPL_OK;
}

namespace {
// Is p memory the device can copy to on its own (hipHostMalloc / hipHostRegister: mtrpl_host_alloc, torch's pinned
// tensors)?  Pageable memory makes hipMemcpyAsync a blocking, staged copy.
bool host_pinned(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // (older runtimes report pageable memory as an error: not sticky)
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int pipe_setup(mtrpl_ctx *ctx, int ns, int nslots, size_t slot_bytes)
{
    while (ctx->n_pipe < ns) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->pipe[ctx->n_pipe], hipStreamNonBlocking));
        ctx->n_pipe++;
    }
    if (!ctx->pipe_in) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pipe_in, hipEventDisableTiming));
    while ((int)ctx->pipe_done.size() < nslots) {
        hipEvent_t e;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->pipe_done.push_back(e);
    }
    const size_t need = (size_t)nslots * slot_bytes;
    if (need > ctx->stage_cap) {
        if (ctx->stage) (void)hipHostFree(ctx->stage);
        ctx->stage = nullptr;
        ctx->stage_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&ctx->stage, need + need / 4, hipHostMallocDefault));
        ctx->stage_cap = need + need / 4;
    }
    return MTRPL_OK;
}
}  // namespace

// Host pointers in, host pointers out.  Up to host_chunk solves: one launch on the context's stream, as ever.  More: the
// batch is cut into sub-launches of host_chunk solves, issued round-robin on host_streams streams the context owns -- a
// solve holds its SIMD slot for its whole duration and durations spread 1 : 1.5, so the slots the short solves of one
// sub-launch free go to the next one's (1.3x the rate of one big launch) -- and the samples of a
// finished sub-launch travel back while the later ones compute: straight into `out` when that is pinned memory
// (mtrpl_host_alloc), through a ring of pinned staging slots and a host copy otherwise.  Same results as the blocking
// path bit for bit (a solve gives the same samples whatever launch it travels in).
int mtrpl_solve_batch(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *descs, const double *ini, int64_t n_ini,
                      const double *tsteps, int64_t n_t, double *out, int64_t n_out, int32_t *status, int32_t *stats)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!descs || !tsteps || !out || !status) return bad(ctx, MTRPL_ERR_BAD_ARG, "null pointer");
    int rc = check_descs(ctx, B, descs, n_ini, tsteps, n_t, n_out);
    if (rc != MTRPL_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dd(ctx, 0), di(ctx, 1), dt(ctx, 2), dout(ctx, 3), dst(ctx, 4), dstat(ctx, 5);
    HIP_TRY(ctx, dd.alloc(sizeof(mtrpl_solve_desc) * B));
    HIP_TRY(ctx, di.alloc(sizeof(double) * (size_t)n_ini));
    HIP_TRY(ctx, dt.alloc(sizeof(double) * (size_t)n_t));
    HIP_TRY(ctx, dout.alloc(sizeof(double) * (size_t)n_out));
    HIP_TRY(ctx, dst.alloc(sizeof(int32_t) * B));
    HIP_TRY(ctx, dstat.alloc(sizeof(int32_t) * 8 * B));
    int32_t max_nx = 0;
    for (int64_t k = 0; k < B; ++k) max_nx = std::max(max_nx, descs[k].nx);

    // ---- the sub-launches and the stretch of `out` each one completes ----
    const int64_t chunk = std::max<int64_t>(1, ctx->host_chunk / (max_nx > 256 ? 4 : (max_nx > 128 ? 2 : 1)));
    const int64_t nchunks = (B + chunk - 1) / chunk;
    const int ns = (int)std::min<int64_t>(std::max(1, std::min(ctx->host_streams, (int)mtrpl_ctx::kPipeMax)), nchunks);
    std::vector<int64_t> upto;  // sub-launch j completes out[upto[j-1], upto[j]) -- if the descriptors fill `out` in order
    bool pipelined = nchunks > 1 && ctx->host_streams > 0;
    if (pipelined) {
        int64_t prev_end = 0;
        for (int64_t j = 0; j < nchunks && pipelined; ++j) {
            int64_t lo = INT64_MAX, hi = 0;
            for (int64_t k = j * chunk; k < std::min(B, (j + 1) * chunk); ++k) {
                lo = std::min(lo, descs[k].out_off);
                hi = std::max(hi, descs[k].out_off + descs[k].nt);
            }
            if (lo < prev_end) pipelined = false;  // a later sub-launch writes below an earlier one's range: no order to stream in
            upto.push_back(j == nchunks - 1 ? n_out : hi);
            prev_end = hi;
        }
    }
    if (!pipelined) {
        hipStream_t s = ctx->stream;
        HIP_TRY(ctx, hipMemcpyAsync(dd.p, descs, sizeof(mtrpl_solve_desc) * B, hipMemcpyHostToDevice, s));
        if (n_ini > 0 && ini) HIP_TRY(ctx, hipMemcpyAsync(di.p, ini, sizeof(double) * (size_t)n_ini, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemcpyAsync(dt.p, tsteps, sizeof(double) * (size_t)n_t, hipMemcpyHostToDevice, s));
        HIP_TRY(ctx, hipMemsetAsync(dout.p, 0, sizeof(double) * (size_t)n_out, s));
        rc = mtrpl_solve_batch_dev(ctx, B, dd.as<mtrpl_solve_desc>(), di.as<double>(), dt.as<double>(), dout.as<double>(),
                                   dst.as<int32_t>(), dstat.as<int32_t>(), max_nx, descs[0].model, s);
        if (rc != MTRPL_OK) {
            (void)hipStreamSynchronize(s);
            return rc;
        }
        HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipMemcpyAsync(status, dst.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
        if (stats) HIP_TRY(ctx, hipMemcpyAsync(stats, dstat.p, sizeof(int32_t) * 8 * B, hipMemcpyDeviceToHost, s));
        HIP_TRY(ctx, hipStreamSynchronize(s));
        return MTRPL_OK;
    }

    // (pinned from its first to its last byte: a view that starts in a pinned allocation may end outside it)
    const bool direct = host_pinned(out) && host_pinned(out + (n_out - 1));
    const int nslots = direct ? 0 : (int)std::min<int64_t>(nchunks, 2 * ns);
    size_t slot_bytes = 0;
    for (int64_t j = 0; j < nchunks; ++j)
        slot_bytes = std::max(slot_bytes, sizeof(double) * (size_t)(upto[j] - (j ? upto[j - 1] : 0)));
    rc = pipe_setup(ctx, ns, nslots, slot_bytes);
    if (rc != MTRPL_OK) return rc;
    // From the first enqueue on NO failure returns early: the device scratch, the staging ring and the streams are the
    // context's, the next call reuses (or reallocates) them, so whatever this call has put in flight is waited for at
    // the common exit below before any status goes back.
    int fail = MTRPL_OK;
    auto ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        if (fail == MTRPL_OK) {
            ctx->err = std::string("pipelined solve, ") + what + ": " + hipGetErrorString(e);
            fail = MTRPL_ERR_DEVICE;
        }
        return false;
    };
    // inputs on the first stream; the others start behind them
    hipStream_t s0 = ctx->pipe[0];
    ok(hipMemcpyAsync(dd.p, descs, sizeof(mtrpl_solve_desc) * B, hipMemcpyHostToDevice, s0), "descriptor upload") &&
        (!(n_ini > 0 && ini) || ok(hipMemcpyAsync(di.p, ini, sizeof(double) * (size_t)n_ini, hipMemcpyHostToDevice, s0), "initial-condition upload")) &&
        ok(hipMemcpyAsync(dt.p, tsteps, sizeof(double) * (size_t)n_t, hipMemcpyHostToDevice, s0), "time-grid upload") &&
        ok(hipMemsetAsync(dout.p, 0, sizeof(double) * (size_t)n_out, s0), "output clear") &&  // (gaps between traces read as zero, as ever)
        ok(hipEventRecord(ctx->pipe_in, s0), "hipEventRecord");
    for (int q = 1; q < ns && fail == MTRPL_OK; ++q) ok(hipStreamWaitEvent(ctx->pipe[q], ctx->pipe_in, 0), "hipStreamWaitEvent");

    auto drain = [&](int64_t j) {  // sub-launch j's samples from its staging slot into the caller's buffer
        const int slot = (int)(j % nslots);
        if (!ok(hipEventSynchronize(ctx->pipe_done[slot]), "hipEventSynchronize")) return;
        const int64_t a = j ? upto[j - 1] : 0;
        std::memcpy(out + a, static_cast<char *>(ctx->stage) + (size_t)slot * slot_bytes, sizeof(double) * (size_t)(upto[j] - a));
    };
    int64_t drained = 0;
    for (int64_t j = 0; j < nchunks && fail == MTRPL_OK; ++j) {
        if (!direct && j >= nslots) {  // the slot this sub-launch copies into is free once its previous user is out
            drain(j - nslots);
            drained = j - nslots + 1;
            if (fail != MTRPL_OK) break;
        }
        hipStream_t s = ctx->pipe[j % ns];
        const int64_t k0 = j * chunk, nb = std::min(B, k0 + chunk) - k0;
        const int rl = launch_solves(ctx, nb, dd.as<mtrpl_solve_desc>() + k0, di.as<double>(), dt.as<double>(), dout.as<double>(),
                                     dst.as<int32_t>() + k0, dstat.as<int32_t>() + 8 * k0, max_nx, descs[0].model, s, B);
        if (rl != MTRPL_OK) {  // (its message is in ctx->err)
            if (fail == MTRPL_OK) fail = rl;
            break;
        }
        const int64_t a = j ? upto[j - 1] : 0;
        void *dst_host = direct ? static_cast<void *>(out + a)
                                : static_cast<void *>(static_cast<char *>(ctx->stage) + (size_t)(j % nslots) * slot_bytes);
        if (ok(hipMemcpyAsync(dst_host, dout.as<double>() + a, sizeof(double) * (size_t)(upto[j] - a), hipMemcpyDeviceToHost, s),
               "device-to-host copy") && !direct)
            ok(hipEventRecord(ctx->pipe_done[j % nslots], s), "hipEventRecord");
    }
    if (!direct)
        for (int64_t j = drained; j < nchunks && fail == MTRPL_OK; ++j) drain(j);
    // nothing of this call may be in flight when it returns, whatever happened above (the buffers are the context's)
    for (int q = 0; q < ns; ++q) ok(hipStreamSynchronize(ctx->pipe[q]), "hipStreamSynchronize");
    if (fail != MTRPL_OK) return fail;
    HIP_TRY(ctx, hipMemcpy(status, dst.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
    if (stats) HIP_TRY(ctx, hipMemcpy(stats, dstat.p, sizeof(int32_t) * 8 * B, hipMemcpyDeviceToHost));
    return MTRPL_OK;
}

// Pinned host memory from the library (hipHostMalloc): an `out` buffer of mtrpl_solve_batch that comes from here is
// filled by asynchronous device-to-host copies, without the staging pass.  Freed by mtrpl_host_free or with the context.
int mtrpl_host_alloc(mtrpl_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes <= 0) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    void *p = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault));
    ctx->host_allocs.push_back(p);
    *out = p;
    return MTRPL_OK;
}

int mtrpl_host_free(mtrpl_ctx *ctx, void *p)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (!p) return MTRPL_OK;
    auto it = std::find(ctx->host_allocs.begin(), ctx->host_allocs.end(), p);
    if (it == ctx->host_allocs.end()) return bad(ctx, MTRPL_ERR_BAD_ARG, "not a pointer from mtrpl_host_alloc of this context");
    ctx->host_allocs.erase(it);
    HIP_TRY(ctx, hipHostFree(p));
    return MTRPL_OK;
}

// Frees what the context holds but no longer uses -- the device blocks its slab pool has outgrown.  hipFree synchronises
// the device, so this belongs at a quiet point of the caller's choosing (between two runs), never inside a launch path.
int mtrpl_ctx_trim(mtrpl_ctx *ctx)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (Slab &r : ctx->retired) (void)hipFree(r.p);
    ctx->retired.clear();
    return MTRPL_OK;
}

int mtrpl_rhs_batch(mtrpl_ctx *ctx, int64_t B, int32_t nx, int32_t model, const double *par, const double *dx,
                    const double *y, double *dy)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!par || !dx || !y || !dy || nx < 2) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    if (model != MTRPL_MODEL_STD && model != MTRPL_MODEL_TRAPS) return bad(ctx, MTRPL_ERR_BAD_ARG, "Invalid model");
    if (nx > 512) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 512 not supported by this build");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool traps = model == MTRPL_MODEL_TRAPS;
    const size_t S = (traps ? 4 : 3) * (size_t)nx + 1;
    DevBuf dp(ctx, 0), ddx(ctx, 1), dy_in(ctx, 2), dy_out(ctx, 3);
    HIP_TRY(ctx, dp.alloc(sizeof(double) * 16 * B));
    HIP_TRY(ctx, ddx.alloc(sizeof(double) * B));
    HIP_TRY(ctx, dy_in.alloc(sizeof(double) * S * B));
    HIP_TRY(ctx, dy_out.alloc(sizeof(double) * S * B));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dp.p, par, sizeof(double) * 16 * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ddx.p, dx, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dy_in.p, y, sizeof(double) * S * B, hipMemcpyHostToDevice, s));
#define MTRPL_RHS_ARGS dim3((unsigned)B), dim3(64 * W_), 0, s, dp.as<double>(), ddx.as<double>(), dy_in.as<double>(), dy_out.as<double>(), nx, B
#define MTRPL_RHS_TEAM(W__, T_) do { constexpr int W_ = W__; hipLaunchKernelGGL((rhs_batch_kernel_team<W__, T_>), MTRPL_RHS_ARGS); } while (0)
    const int variant = ctx->kernel_variant;
    if (nx <= 128 && !traps && variant != 1) {
        constexpr int W_ = 1;
        hipLaunchKernelGGL(rhs_batch_kernel_wave, MTRPL_RHS_ARGS);
    }
    else if (nx <= 128) { if (traps) MTRPL_RHS_TEAM(1, true); else MTRPL_RHS_TEAM(1, false); }
    else if (!traps && variant != 1 && nx <= 256) { constexpr int W_ = 2; hipLaunchKernelGGL(rhs_batch_kernel_spike<2>, MTRPL_RHS_ARGS); }
    else if (!traps && variant != 1) { constexpr int W_ = 4; hipLaunchKernelGGL(rhs_batch_kernel_spike<4>, MTRPL_RHS_ARGS); }
    else if (nx <= 256) { if (traps) MTRPL_RHS_TEAM(2, true); else MTRPL_RHS_TEAM(2, false); }
    else { if (traps) MTRPL_RHS_TEAM(4, true); else MTRPL_RHS_TEAM(4, false); }
#undef MTRPL_RHS_TEAM
#undef MTRPL_RHS_ARGS
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(dy, dy_out.p, sizeof(double) * S * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

int mtrpl_newton_probe(mtrpl_ctx *ctx, int64_t B, int32_t nx, const double *par, const double *dx, const double *c,
                       const double *y, const double *b, double *x)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!par || !dx || !c || !y || !b || !x || nx < 2) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    if (nx > 512) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 512 not supported by this build");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t S = 3 * (size_t)nx + 1;
    DevBuf dp(ctx, 0), ddx(ctx, 1), dc(ctx, 2), dyy(ctx, 3), db(ctx, 4), dxo(ctx, 5);
    HIP_TRY(ctx, dp.alloc(sizeof(double) * 16 * B));
    HIP_TRY(ctx, ddx.alloc(sizeof(double) * B));
    HIP_TRY(ctx, dc.alloc(sizeof(double) * B));
    HIP_TRY(ctx, dyy.alloc(sizeof(double) * S * B));
    HIP_TRY(ctx, db.alloc(sizeof(double) * S * B));
    HIP_TRY(ctx, dxo.alloc(sizeof(double) * S * B));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dp.p, par, sizeof(double) * 16 * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ddx.p, dx, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dc.p, c, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dyy.p, y, sizeof(double) * S * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(db.p, b, sizeof(double) * S * B, hipMemcpyHostToDevice, s));
#define MTRPL_NP_ARGS 0, s, dp.as<double>(), ddx.as<double>(), dc.as<double>(), dyy.as<double>(), db.as<double>(), dxo.as<double>(), nx, B
    const int variant = ctx->kernel_variant;
    if (nx > 256) {  // the SPIKE backend: four waves per solve
        hipLaunchKernelGGL(newton_probe_kernel_spike<4>, dim3((unsigned)B), dim3(256), MTRPL_NP_ARGS);
    } else if (nx > 128) {
        hipLaunchKernelGGL(newton_probe_kernel_spike<2>, dim3((unsigned)B), dim3(128), MTRPL_NP_ARGS);
    } else if (variant == 3 || variant == 6 || variant == 0) {
        hipLaunchKernelGGL(newton_probe_kernel_wave<true>, dim3((unsigned)B), dim3(64), MTRPL_NP_ARGS);
    } else {
        hipLaunchKernelGGL(newton_probe_kernel_wave<false>, dim3((unsigned)B), dim3(64), MTRPL_NP_ARGS);
    }
#undef MTRPL_NP_ARGS
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(x, dxo.p, sizeof(double) * S * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

int mtrpl_reduce_probe(mtrpl_ctx *ctx, int64_t B, const float *in, float *out)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!in || !out) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf di(ctx, 0), dout(ctx, 1);
    HIP_TRY(ctx, di.alloc(sizeof(float) * 256 * B));
    HIP_TRY(ctx, dout.alloc(sizeof(float) * 7 * B));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(di.p, in, sizeof(float) * 256 * B, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(reduce_probe_kernel, dim3((unsigned)B), dim3(64), 0, s, di.as<float>(), dout.as<float>(), B);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, sizeof(float) * 7 * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

int mtrpl_E_field(mtrpl_ctx *ctx, int64_t rows, int32_t nx, const double *N, const double *P, double n0, double p0,
                  double eps, double dx, double corner_E, double *E)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (rows <= 0) return MTRPL_OK;
    if (!N || !P || !E || nx < 1) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dn(ctx, 0), dpp(ctx, 1), de(ctx, 2);
    const size_t n = (size_t)rows * nx;
    HIP_TRY(ctx, dn.alloc(sizeof(double) * n));
    HIP_TRY(ctx, dpp.alloc(sizeof(double) * n));
    HIP_TRY(ctx, de.alloc(sizeof(double) * (size_t)rows * (nx + 1)));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dn.p, N, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dpp.p, P, sizeof(double) * n, hipMemcpyHostToDevice, s));
    const double pref = kQC / (eps * kEps0) * dx;  // forward_solver.py:28
    hipLaunchKernelGGL(e_field_kernel, dim3((unsigned)rows), dim3(64), 0, s, dn.as<double>(), dpp.as<double>(),
                       de.as<double>(), nx, rows, n0, p0, pref, corner_E);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(E, de.p, sizeof(double) * (size_t)rows * (nx + 1), hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

static int integral_host(mtrpl_ctx *ctx, int64_t rows, int32_t nx, double dx, const double *N, const double *P,
                         int meas, double ks, double mun, double mup, double n0, double p0, double *out)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (rows <= 0) return MTRPL_OK;
    if (!N || !P || !out || nx < 1) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dn(ctx, 0), dpp(ctx, 1), dout(ctx, 2);
    const size_t n = (size_t)rows * nx;
    HIP_TRY(ctx, dn.alloc(sizeof(double) * n));
    HIP_TRY(ctx, dpp.alloc(sizeof(double) * n));
    HIP_TRY(ctx, dout.alloc(sizeof(double) * (size_t)rows));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dn.p, N, sizeof(double) * n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dpp.p, P, sizeof(double) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(integral_kernel, dim3((unsigned)rows), dim3(64), 0, s, dn.as<double>(), dpp.as<double>(),
                       dout.as<double>(), nx, rows, dx, meas, ks, mun, mup, n0, p0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)rows, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

int mtrpl_calculate_PL(mtrpl_ctx *ctx, int64_t rows, int32_t nx, double dx, const double *N, const double *P,
                       double ks, double n0, double p0, double *out)
{
    return integral_host(ctx, rows, nx, dx, N, P, MTRPL_MEAS_TRPL, ks, 0.0, 0.0, n0, p0, out);
}

int mtrpl_calculate_TRTS(mtrpl_ctx *ctx, int64_t rows, int32_t nx, double dx, const double *N, const double *P,
                         double mu_n, double mu_p, double n0, double p0, double *out)
{
    return integral_host(ctx, rows, nx, dx, N, P, MTRPL_MEAS_TRTS, 0.0, mu_n, mu_p, n0, p0, out);
}

static int density_host(mtrpl_ctx *ctx, int64_t n, const double *N, const double *P, int meas, double ks, double mun,
                        double mup, double n0, double p0, double *out)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (n <= 0) return MTRPL_OK;
    if (!N || !P || !out) return bad(ctx, MTRPL_ERR_BAD_ARG, "bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dn(ctx, 0), dpp(ctx, 1), dout(ctx, 2);
    HIP_TRY(ctx, dn.alloc(sizeof(double) * (size_t)n));
    HIP_TRY(ctx, dpp.alloc(sizeof(double) * (size_t)n));
    HIP_TRY(ctx, dout.alloc(sizeof(double) * (size_t)n));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dn.p, N, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dpp.p, P, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(density_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dn.as<double>(), dpp.as<double>(),
                       n, meas, ks, mun, mup, n0, p0, dout.as<double>());
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
    return MTRPL_OK;
}

int mtrpl_calculate_RR(mtrpl_ctx *ctx, int64_t n, const double *N, const double *P, double ks, double n0, double p0,
                       double *out)
{
    return density_host(ctx, n, N, P, MTRPL_MEAS_TRPL, ks, 0.0, 0.0, n0, p0, out);
}

int mtrpl_calculate_photoc(mtrpl_ctx *ctx, int64_t n, const double *N, const double *P, double mu_n, double mu_p,
                           double n0, double p0, double *out)
{
    return density_host(ctx, n, N, P, MTRPL_MEAS_TRTS, 0.0, mu_n, mu_p, n0, p0, out);
}

#include "irf_capi.inc"
#include "pt_capi.inc"

#ifdef MTRPL_TRACE
extern "C" int mtrpl_debug_trace(float *out, int *n)
{
    hipMemcpyFromSymbol(n, HIP_SYMBOL(mtrpl::g_trace_n), sizeof(int));
    hipMemcpyFromSymbol(out, HIP_SYMBOL(mtrpl::g_trace), sizeof(float) * 8 * 8192);
    int zero = 0;
    hipMemcpyToSymbol(HIP_SYMBOL(mtrpl::g_trace_n), &zero, sizeof(int));
    return 0;
}
extern "C" int mtrpl_debug_trace_select(int sidx)
{
    hipMemcpyToSymbol(HIP_SYMBOL(mtrpl::g_trace_sel), &sidx, sizeof(int));
    return 0;
}
#endif

}  // extern "C"
