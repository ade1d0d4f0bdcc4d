// mtrpl_capi.hip -- C ABI (include/metrotrpl_amd.h) + kernel launchers.
// gfx950 only.  There is deliberately no CPU fallback anywhere in this file:
// without a HIP device every entry point fails with MTRPL_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/metrotrpl_amd.h"
#include "irf_kernels.hpp"
#include "solve_kernel_blk.hpp"
#include "solve_kernel_nq.hpp"
#include "solve_kernel_blk_nq.hpp"
#include "mtrpl_fp32_pass.hpp"

using namespace mtrpl;

struct mtrpl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    SolveTuning tune{0.1, 3e-9, 3e-10, 1e-4, 2000000, 0.03, 8, 4.0, 3.0, 1, 0.5, 16.0, 1, 1.2, 80, 1.2, 1.0, 1.0, 1.0, 10.0, 0.5};
    std::string err;
    // IRF tables resident on the device (mtrpl_irf_tables_upload)
    double *d_irf_tab = nullptr;
    std::vector<int32_t> irf_nk;
    std::vector<int64_t> irf_off;
    std::vector<double> irf_dt;
    IrfTableSet *d_irf_set = nullptr;
    // global scratch for the elimination blocks of the 4-wave (nx <= 512) kernel
    double2 *d_gblk = nullptr;
    size_t gblk_bytes = 0;
    int kernel_variant = 0;  // std, nx<=128: 0 wave kernel (Nordsieck BDF); 3 wave kernel (NDF differences); 1 team kernel
    int precision = 64;      // 64: fp64 (the product); 32: the fp32 build of the team kernel (precision sweep only)
};

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
            return MTRPL_ERR_DEVICE;                                                         \
        }                                                                                    \
    } while (0)

namespace {

// RAII device buffer for the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 8); }
    template <class T> T *as() { return static_cast<T *>(p); }
};

int bad(mtrpl_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// ---- small kernels for the reference's helper functions ------------------------
// forward_solver.py:332-372 through the solver's own wave-layout RHS (R=2: nx<=128)
__global__ __launch_bounds__(64) void rhs_batch_kernel_r2(const double *par, const double *dxs, const double *y,
                                                          double *dy, int nx, int64_t B)
{
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    const int lane = threadIdx.x, L = nx, S = 3 * nx + 1;
    const Par p = make_par(par + 16 * s, dxs[s]);
    const double *ys = y + s * S;
    double *fs = dy + s * S;
    double yl[6], fl[6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane * 2 + r;
        yl[r] = (i < L) ? ys[i] : 1.0;
        yl[2 + r] = (i < L) ? ys[L + i] : 1.0;
        yl[4 + r] = (i < L) ? ys[2 * L + i] : 0.0;
    }
    rhs_std<2>(p, yl, fl, lane, L);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane * 2 + r;
        if (i < L) {
            fs[i] = fl[r];
            fs[L + i] = fl[2 + r];
            fs[2 * L + i] = fl[4 + r];
        }
    }
    if (lane == 0) fs[3 * L] = 0.0;  // dE_L = -(Jn_L + Jp_L)*Lambda = 0 (forward_solver.py:350-352,:363)
}

__global__ __launch_bounds__(64) void newton_probe_kernel_r2(const double *par, const double *dxs, const double *cs,
                                                             const double *y, const double *b, double *x, int nx,
                                                             int64_t B)
{
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    const int lane = threadIdx.x, L = nx, S = 3 * nx + 1;
    const Par p = make_par(par + 16 * s, dxs[s]);
    double yl[6], bl[6], xl[6];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane * 2 + r;
        yl[r] = (i < L) ? y[s * S + i] : 1.0;
        yl[2 + r] = (i < L) ? y[s * S + L + i] : 1.0;
        yl[4 + r] = (i < L) ? y[s * S + 2 * L + i] : 0.0;
        bl[r] = (i < L) ? b[s * S + i] : 0.0;
        bl[2 + r] = (i < L) ? b[s * S + L + i] : 0.0;
        bl[4 + r] = (i < L && i > 0) ? b[s * S + 2 * L + i] : 0.0;
    }
    __shared__ double2 fac_lds[kFactorLdsBytes / 16];
    Factors<2> F;
    F.lds = fac_lds;
    build_factors(p, yl, cs[s], lane, L, F);
    apply_factors(F, bl, xl, lane, L);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = lane * 2 + r;
        if (i < L) {
            x[s * S + i] = xl[r];
            x[s * S + L + i] = xl[2 + r];
            x[s * S + 2 * L + i] = (i > 0) ? xl[4 + r] : b[s * S + 2 * L];
        }
    }
    if (lane == 0) x[s * S + 3 * L] = b[s * S + 3 * L];  // E_0 and E_L rows of I - cJ are identity rows
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
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const double t = shf(v, (lane - s) & 63);
            v += (lane >= s) ? t : 0.0;
        }
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

// forward_solver.py:332-418 through the team RHS (any nx <= 128*W, std or traps);
// reference layouts: std [N(L),P(L),E(L+1)], traps [N(L),Ntrap(L),P(L),E(L+1)]
template <int W, bool TRAPS>
__global__ __launch_bounds__(64 * W) void rhs_batch_kernel_blk(const double *par, const double *dxs, const double *y,
                                                               double *dy, int nx, int64_t B)
{
    constexpr int V = TRAPS ? 8 : 6;
    const int64_t s = blockIdx.x;
    if (s >= B) return;
    __shared__ double2 xbuf_lds[team_xbuf_bytes<W>() / 16];
    __shared__ double rbuf_lds[5 * W + 4];
    Team<W> tm;
    tm.tid = threadIdx.x; tm.lane = threadIdx.x & 63; tm.wave = threadIdx.x >> 6;
    tm.xbuf = xbuf_lds; tm.rbuf = rbuf_lds; tm.xpar = 0; tm.rpar = 0;
    const int L = nx, S = (TRAPS ? 4 : 3) * nx + 1;
    const int oP = (TRAPS ? 2 : 1) * L, oE = (TRAPS ? 3 : 2) * L;
    const Par p = make_par(par + 16 * s, dxs[s]);
    const double *ys = y + s * S;
    double *fs = dy + s * S;
    double yl[V], fl[V];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tm.tid * 2 + r;
        yl[r] = (i < L) ? ys[i] : 1.0;
        yl[2 + r] = (i < L) ? ys[oP + i] : 1.0;
        yl[4 + r] = (i < L) ? ys[oE + i] : 0.0;
        if (TRAPS) yl[6 + r] = (i < L) ? ys[L + i] : 0.0;
    }
    team_rhs<W, TRAPS>(tm, p, yl, fl, L);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = tm.tid * 2 + r;
        if (i < L) {
            fs[i] = fl[r];
            fs[oP + i] = fl[2 + r];
            fs[oE + i] = fl[4 + r];
            if (TRAPS) fs[L + i] = fl[6 + r];
        }
    }
    if (tm.tid == 0) fs[oE + L] = 0.0;  // dE_L = 0 (forward_solver.py:350-352,:363)
}

template <int W, bool TRAPS, bool GBLK>
void launch_blk(hipStream_t s, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats,
                const SolveTuning &tune, double2 *gblk, bool ndf_form = false)
{
    if (ndf_form)
        hipLaunchKernelGGL((solve_kernel_blk<W, TRAPS, GBLK>), dim3((unsigned)B), dim3(64 * W), 0, s, d_descs, d_ini,
                           d_tsteps, d_out, d_status, d_stats, B, tune, gblk);
    else
        hipLaunchKernelGGL((solve_kernel_blk_nq<W, TRAPS, GBLK>), dim3((unsigned)B), dim3(64 * W), 0, s, d_descs, d_ini,
                           d_tsteps, d_out, d_status, d_stats, B, tune, gblk);
}

// fp32 build of the team kernel (mtrpl_fp32_pass.hpp); std model only
template <int W, bool GBLK>
void launch_blk32(hipStream_t s, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                  const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats,
                  const SolveTuning &t, double2 *gblk)
{
    const mtrpl32::mtrpl::SolveTuning t32{(float)t.rtol_scale, (float)t.rtol_cap, (float)t.rtol_floor,
                                          (float)t.atol_rel, t.max_steps, (float)t.newton_tol, t.rate_max_age,
                                          (float)t.rate_first_safety, (float)t.e_weight, t.honor_hmax,
                                          (float)t.trts_rtol_factor, (float)t.amp_ref, t.fresh_jac,
                                          (float)t.nq_deadband_hi, t.nq_rej_memory, (float)t.nq_rej_growth, (float)t.nq_bias_same, (float)t.nq_bias_down, (float)t.nq_bias_up,
                                          (float)t.early_factor, (float)t.early_frac};
    hipLaunchKernelGGL((mtrpl32::mtrpl::solve_kernel_blk<W, false, GBLK>), dim3((unsigned)B), dim3(64 * W), 0, s,
                       d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, B, t32, reinterpret_cast<float2 *>(gblk));
}

}  // namespace

extern "C" {

const char *mtrpl_version(void) { return "metrotrpl_amd 0.1 (gfx950)"; }

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
    if (ctx->d_irf_tab) (void)hipFree(ctx->d_irf_tab);
    if (ctx->d_irf_set) (void)hipFree(ctx->d_irf_set);
    if (ctx->d_gblk) (void)hipFree(ctx->d_gblk);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MTRPL_OK;
}

const char *mtrpl_last_error(const mtrpl_ctx *ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int mtrpl_set_option(mtrpl_ctx *ctx, const char *key, double value)
{
    if (!ctx || !key) return MTRPL_ERR_BAD_ARG;
    std::string k(key);
    if (k == "rtol_scale") ctx->tune.rtol_scale = value;
    else if (k == "rtol_cap") ctx->tune.rtol_cap = value;
    else if (k == "rtol_floor") ctx->tune.rtol_floor = value;
    else if (k == "atol_rel") ctx->tune.atol_rel = value;
    else if (k == "max_steps") ctx->tune.max_steps = (int)value;
    else if (k == "rate_max_age") ctx->tune.rate_max_age = (int)value;
    else if (k == "newton_tol") ctx->tune.newton_tol = value;
    else if (k == "rate_first_safety") ctx->tune.rate_first_safety = value;
    else if (k == "e_weight") ctx->tune.e_weight = value;
    else if (k == "honor_hmax") ctx->tune.honor_hmax = (int)value;
    else if (k == "trts_rtol_factor") ctx->tune.trts_rtol_factor = value;
    else if (k == "amp_ref") ctx->tune.amp_ref = value;
    else if (k == "fresh_jac") ctx->tune.fresh_jac = (int)value;
    else if (k == "nq_deadband_hi") ctx->tune.nq_deadband_hi = value;
    else if (k == "nq_rej_memory") ctx->tune.nq_rej_memory = (int)value;
    else if (k == "nq_rej_growth") ctx->tune.nq_rej_growth = value;
    else if (k == "nq_bias_same") ctx->tune.nq_bias_same = value;
    else if (k == "nq_bias_down") ctx->tune.nq_bias_down = value;
    else if (k == "nq_bias_up") ctx->tune.nq_bias_up = value;
    else if (k == "early_factor") ctx->tune.early_factor = value;
    else if (k == "early_frac") ctx->tune.early_frac = value;
    else if (k == "kernel_variant") ctx->kernel_variant = (int)value;
    else if (k == "precision") {
        if (value != 32 && value != 64) return bad(ctx, MTRPL_ERR_BAD_ARG, "precision must be 32 or 64");
        ctx->precision = (int)value;
    }
    else return bad(ctx, MTRPL_ERR_BAD_ARG, "unknown option");
    return MTRPL_OK;
}

int mtrpl_get_option(const mtrpl_ctx *ctx, const char *key, double *value)
{
    if (!ctx || !key || !value) return MTRPL_ERR_BAD_ARG;
    std::string k(key);
    if (k == "rtol_scale") *value = ctx->tune.rtol_scale;
    else if (k == "rtol_cap") *value = ctx->tune.rtol_cap;
    else if (k == "rtol_floor") *value = ctx->tune.rtol_floor;
    else if (k == "atol_rel") *value = ctx->tune.atol_rel;
    else if (k == "max_steps") *value = ctx->tune.max_steps;
    else if (k == "rate_max_age") *value = ctx->tune.rate_max_age;
    else if (k == "newton_tol") *value = ctx->tune.newton_tol;
    else if (k == "rate_first_safety") *value = ctx->tune.rate_first_safety;
    else if (k == "e_weight") *value = ctx->tune.e_weight;
    else if (k == "honor_hmax") *value = ctx->tune.honor_hmax;
    else if (k == "trts_rtol_factor") *value = ctx->tune.trts_rtol_factor;
    else if (k == "amp_ref") *value = ctx->tune.amp_ref;
    else if (k == "fresh_jac") *value = ctx->tune.fresh_jac;
    else if (k == "nq_deadband_hi") *value = ctx->tune.nq_deadband_hi;
    else if (k == "nq_rej_memory") *value = ctx->tune.nq_rej_memory;
    else if (k == "nq_rej_growth") *value = ctx->tune.nq_rej_growth;
    else if (k == "nq_bias_same") *value = ctx->tune.nq_bias_same;
    else if (k == "nq_bias_down") *value = ctx->tune.nq_bias_down;
    else if (k == "nq_bias_up") *value = ctx->tune.nq_bias_up;
    else if (k == "early_factor") *value = ctx->tune.early_factor;
    else if (k == "early_frac") *value = ctx->tune.early_frac;
    else if (k == "kernel_variant") *value = ctx->kernel_variant;
    else if (k == "precision") *value = ctx->precision;
    else return MTRPL_ERR_BAD_ARG;
    return MTRPL_OK;
}

int mtrpl_solve_batch_dev(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs, const double *d_ini,
                          const double *d_tsteps, double *d_out, int32_t *d_status, int32_t *d_stats, int32_t max_nx,
                          int32_t model, void *stream)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!d_descs || !d_tsteps || !d_out || !d_status) return bad(ctx, MTRPL_ERR_BAD_ARG, "null device pointer");
    if (max_nx < 2 || max_nx > 512) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 512 not supported by this build");
    if (model != MTRPL_MODEL_STD && model != MTRPL_MODEL_TRAPS) return bad(ctx, MTRPL_ERR_BAD_ARG, "Invalid model");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t s = (hipStream_t)stream;  // exactly the caller's stream (NULL = the default stream)
    const bool traps = model == MTRPL_MODEL_TRAPS;
    if (ctx->precision == 32) {
        if (traps) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "the fp32 build covers the std model only");
        if (max_nx <= 128) {
            launch_blk32<1, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr);
        } else if (max_nx <= 256) {
            launch_blk32<2, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr);
        } else {
            const size_t need = (size_t)B * team_blk_bytes<4>();
            if (need > ctx->gblk_bytes) {
                if (ctx->d_gblk) (void)hipFree(ctx->d_gblk);
                ctx->d_gblk = nullptr;
                ctx->gblk_bytes = 0;
                HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gblk, need));
                ctx->gblk_bytes = need;
            }
            launch_blk32<4, true>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, ctx->d_gblk);
        }
        HIP_TRY(ctx, hipGetLastError());
        return MTRPL_OK;
    }
    if (max_nx <= 128) {
        if (!traps && ctx->kernel_variant == 0)  // default: the wave kernel with the Nordsieck-form BDF
            hipLaunchKernelGGL(solve_kernel_nq, dim3((unsigned)B), dim3(64), 0, s, d_descs, d_ini, d_tsteps, d_out,
                               d_status, d_stats, B, ctx->tune);
        else if (!traps && ctx->kernel_variant == 3)  // the wave kernel with the NDF difference form (previous default)
            hipLaunchKernelGGL(solve_kernel_r2, dim3((unsigned)B), dim3(64), 0, s, d_descs, d_ini, d_tsteps, d_out,
                               d_status, d_stats, B, ctx->tune);
        else if (!traps)
            launch_blk<1, false, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr, ctx->kernel_variant == 3);
        else
            launch_blk<1, true, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr, ctx->kernel_variant == 3);
    } else if (max_nx <= 256) {
        if (!traps)
            launch_blk<2, false, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr, ctx->kernel_variant == 3);
        else
            launch_blk<2, true, false>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, nullptr, ctx->kernel_variant == 3);
    } else {
        // 4 waves per solve: the 21 elimination blocks per thread (172 KB per solve) do not fit
        // in LDS next to the exchange buffers -> per-solve slab in HBM (L2/MALL-resident)
        const size_t need = (size_t)B * team_blk_bytes<4>();
        if (need > ctx->gblk_bytes) {
            if (ctx->d_gblk) (void)hipFree(ctx->d_gblk);
            ctx->d_gblk = nullptr;
            ctx->gblk_bytes = 0;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gblk, need));
            ctx->gblk_bytes = need;
        }
        if (!traps)
            launch_blk<4, false, true>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, ctx->d_gblk, ctx->kernel_variant == 3);
        else
            launch_blk<4, true, true>(s, B, d_descs, d_ini, d_tsteps, d_out, d_status, d_stats, ctx->tune, ctx->d_gblk, ctx->kernel_variant == 3);
    }
    HIP_TRY(ctx, hipGetLastError());
    return MTRPL_OK;
}

int mtrpl_solve_batch(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *descs, const double *ini, int64_t n_ini,
                      const double *tsteps, int64_t n_t, double *out, int64_t n_out, int32_t *status, int32_t *stats)
{
    if (!ctx) return MTRPL_ERR_BAD_ARG;
    if (B <= 0) return MTRPL_OK;
    if (!descs || !tsteps || !out || !status) return bad(ctx, MTRPL_ERR_BAD_ARG, "null pointer");
    int rc = check_descs(ctx, B, descs, n_ini, tsteps, n_t, n_out);
    if (rc != MTRPL_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf dd, di, dt, dout, dst, dstat;
    HIP_TRY(ctx, dd.alloc(sizeof(mtrpl_solve_desc) * B));
    HIP_TRY(ctx, di.alloc(sizeof(double) * (size_t)n_ini));
    HIP_TRY(ctx, dt.alloc(sizeof(double) * (size_t)n_t));
    HIP_TRY(ctx, dout.alloc(sizeof(double) * (size_t)n_out));
    HIP_TRY(ctx, dst.alloc(sizeof(int32_t) * B));
    HIP_TRY(ctx, dstat.alloc(sizeof(int32_t) * 8 * B));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dd.p, descs, sizeof(mtrpl_solve_desc) * B, hipMemcpyHostToDevice, s));
    if (n_ini > 0 && ini) HIP_TRY(ctx, hipMemcpyAsync(di.p, ini, sizeof(double) * (size_t)n_ini, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dt.p, tsteps, sizeof(double) * (size_t)n_t, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemsetAsync(dout.p, 0, sizeof(double) * (size_t)n_out, s));
    int32_t max_nx = 0;
    for (int64_t k = 0; k < B; ++k) max_nx = std::max(max_nx, descs[k].nx);
    rc = mtrpl_solve_batch_dev(ctx, B, dd.as<mtrpl_solve_desc>(), di.as<double>(), dt.as<double>(), dout.as<double>(),
                               dst.as<int32_t>(), dstat.as<int32_t>(), max_nx, descs[0].model, s);
    if (rc != MTRPL_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, dout.p, sizeof(double) * (size_t)n_out, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipMemcpyAsync(status, dst.p, sizeof(int32_t) * B, hipMemcpyDeviceToHost, s));
    if (stats) HIP_TRY(ctx, hipMemcpyAsync(stats, dstat.p, sizeof(int32_t) * 8 * B, hipMemcpyDeviceToHost, s));
    HIP_TRY(ctx, hipStreamSynchronize(s));
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
    DevBuf dp, ddx, dy_in, dy_out;
    HIP_TRY(ctx, dp.alloc(sizeof(double) * 16 * B));
    HIP_TRY(ctx, ddx.alloc(sizeof(double) * B));
    HIP_TRY(ctx, dy_in.alloc(sizeof(double) * S * B));
    HIP_TRY(ctx, dy_out.alloc(sizeof(double) * S * B));
    hipStream_t s = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dp.p, par, sizeof(double) * 16 * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(ddx.p, dx, sizeof(double) * B, hipMemcpyHostToDevice, s));
    HIP_TRY(ctx, hipMemcpyAsync(dy_in.p, y, sizeof(double) * S * B, hipMemcpyHostToDevice, s));
#define MTRPL_RHS_LAUNCH(W_, T_)                                                                                  \
    hipLaunchKernelGGL((rhs_batch_kernel_blk<W_, T_>), dim3((unsigned)B), dim3(64 * W_), 0, s, dp.as<double>(), \
                       ddx.as<double>(), dy_in.as<double>(), dy_out.as<double>(), nx, B)
    if (nx <= 128 && !traps && ctx->kernel_variant != 1)
        hipLaunchKernelGGL(rhs_batch_kernel_r2, dim3((unsigned)B), dim3(64), 0, s, dp.as<double>(), ddx.as<double>(),
                           dy_in.as<double>(), dy_out.as<double>(), nx, B);
    else if (nx <= 128) { if (traps) MTRPL_RHS_LAUNCH(1, true); else MTRPL_RHS_LAUNCH(1, false); }
    else if (nx <= 256) { if (traps) MTRPL_RHS_LAUNCH(2, true); else MTRPL_RHS_LAUNCH(2, false); }
    else { if (traps) MTRPL_RHS_LAUNCH(4, true); else MTRPL_RHS_LAUNCH(4, false); }
#undef MTRPL_RHS_LAUNCH
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
    if (nx > 128) return bad(ctx, MTRPL_ERR_UNSUPPORTED, "nx > 128 not supported by this build");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t S = 3 * (size_t)nx + 1;
    DevBuf dp, ddx, dc, dyy, db, dxo;
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
    hipLaunchKernelGGL(newton_probe_kernel_r2, dim3((unsigned)B), dim3(64), 0, s, dp.as<double>(), ddx.as<double>(),
                       dc.as<double>(), dyy.as<double>(), db.as<double>(), dxo.as<double>(), nx, B);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(x, dxo.p, sizeof(double) * S * B, hipMemcpyDeviceToHost, s));
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
    DevBuf dn, dpp, de;
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
    DevBuf dn, dpp, dout;
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

#include "irf_capi.inc"

}  // extern "C"
