/*
 * metrotrpl_amd.h -- C ABI of the MI355X forward-solver engine for MetroTRPL's
 * likelihood hot path.
 *
 * The reference has no FFI: its boundary is the Python import
 *   from forward_solver import solve            (trial_move_evaluation.py:4)
 *   from laplace import make_I_tables, do_irf_convolution, post_conv_trim
 *                                                (trial_move_evaluation.py:6, metropolis.py:28)
 *   from sim_utils import Grid                   (trial_move_evaluation.py:7)
 * metrotrpl_amd/{forward_solver,laplace,sim_utils}.py keep those names and
 * bind the entry points below through ctypes.
 *
 * Conventions
 *  - plain pointers + sizes, no C++/torch types; the library never keeps a
 *    caller pointer after a call returns;
 *  - every function returns an int status (MTRPL_OK = 0); nothing throws
 *    across the boundary; mtrpl_last_error() gives the message;
 *  - "host" entry points take host pointers, do the H2D/D2H themselves through
 *    device scratch the context keeps (no allocation per call once warm) and
 *    block until the result is in the caller's buffers;
 *  - "_dev" entry points take device pointers + a hipStream_t (as void*) and
 *    are asynchronous on that stream (inputs resident in HBM).  Launches on
 *    DIFFERENT streams of one context may be in flight together: every piece
 *    of device memory a launch needs beyond the caller's buffers (the
 *    elimination-block slab of nx > 256 solves, the work space of long traces
 *    in the likelihood kernel) is private to that launch -- handed out from a
 *    per-context pool and only re-used once the launch has completed or by a
 *    later launch on the same stream.  The context itself is not thread-safe:
 *    one host thread issues the calls;
 *  - state, residual, right-hand side, Jacobian, output samples and every helper function are IEEE fp64.
 *    Two things inside the solve kernels are not: the LINEAR ALGEBRA of the simplified
 *    Newton iteration is factorised in fp64 but its elimination blocks are stored rounded to fp32 and re-applied in
 *    fp32 -- an approximate inverse on an fp64 residual, so the corrector converges to the same fp64 solution
 *    (kernel_variant 2 keeps fp64 blocks and an fp64 re-solve) -- and the error norms / step-size controller, which
 *    only steer the integration, are evaluated in fp32.
 */
#ifndef METROTRPL_AMD_H
#define METROTRPL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes -> Python exceptions raised by the shim (forward_solver.py:104,
 * :115,:117,:140 ValueError; :37,:203,:225 NotImplementedError) */
#define MTRPL_OK 0
#define MTRPL_ERR_BAD_ARG 1     /* ValueError */
#define MTRPL_ERR_UNSUPPORTED 2 /* NotImplementedError */
#define MTRPL_ERR_DEVICE 3      /* RuntimeError: HIP failure */
#define MTRPL_ERR_NO_DEVICE 4   /* RuntimeError: no gfx950 device / wrong arch */

/* per-solve status written by the solve kernels */
#define MTRPL_SOLVE_OK 0
#define MTRPL_SOLVE_STOPPED_EARLY 1 /* integrator gave up: later samples computed
                                       from a zeroed state, like forward_solver.py:168 */
#define MTRPL_SOLVE_OK_TOL_CLAMPED 2 /* solved; the requested RTOL was below the fp64 noise floor of
                                        this system (option rtol_min) and was raised to it */

enum { MTRPL_MEAS_TRPL = 0, MTRPL_MEAS_TRTS = 1 };    /* forward_solver.py:186-203 */
enum { MTRPL_MODEL_STD = 0, MTRPL_MODEL_TRAPS = 1 };  /* forward_solver.py:420-423 */
enum { MTRPL_INI_DENSITY = 0, MTRPL_INI_FLUENCE = 1 }; /* forward_solver.py:100-117 */

/* parameter slots of mtrpl_solve_desc.par, SOLVER units (= state * units,
 * forward_solver.py:119,128-138) */
enum {
    MTRPL_P_N0 = 0, MTRPL_P_P0, MTRPL_P_MU_N, MTRPL_P_MU_P, MTRPL_P_KS, MTRPL_P_CN, MTRPL_P_CP,
    MTRPL_P_SF, MTRPL_P_SB, MTRPL_P_TAUN, MTRPL_P_TAUP, MTRPL_P_EPS, MTRPL_P_TM,
    MTRPL_P_KC, MTRPL_P_NT, MTRPL_P_TAUE, MTRPL_NPAR
};

/* One forward simulation = one call of forward_solver.solve (forward_solver.py:41-203)
 * together with its Grid (sim_utils.py:246-277). */
typedef struct mtrpl_solve_desc {
    double par[16];   /* MTRPL_P_* slots */
    double thickness; /* Grid.thickness [nm] */
    double hmax;      /* Grid.hmax: the reference's max integrator step [ns]; caps the step only with the option
                         honor_hmax = 1 (default 0: the engine's own error control does not need the guard) */
    double rtol;      /* RTOL of the call (default 1e-7, :18).  The engine's contract is the CONVERGED
                         solution (1e-6 gate), so RTOL acts one way: a looser value than the default
                         is not a licence to be less accurate (internal rtol = min(0.1*RTOL, rtol_cap),
                         then the policy factors of DESIGN.md section 3), a tighter one tightens the
                         integration below the policy's own floor, down to the fp64 noise floor of this
                         system (option rtol_min); a request below that is raised to it and the solve
                         reports MTRPL_SOLVE_OK_TOL_CLAMPED.  A caller who WANTS the reference's trade of accuracy
                         for speed opts in with the context option rtol_mode = 1: internal rtol = 0.1*RTOL
                         whatever RTOL is (up to 1e-3), every policy factor scaling with it -- RTOL = 1e-5 then
                         takes a third fewer steps than the default and is still closer to the converged trace
                         than the reference is at its own default tolerances (tests/test_gpu_tolerances.py). */
    double atol;      /* ATOL of the call (default 1e-10, :19), solver units (nm^-3 for the densities):
                         an upper bound of the absolute tolerance of the densities, which is otherwise
                         rtol_int * atol_rel_np * max(N0, P0) (about 3e-21 nm^-3).  The default changes
                         nothing (the rule is far tighter); a smaller ATOL resolves a still deeper tail.
                         The field keeps its own absolute floor (option atol_rel). */
    double min_y;     /* Grid.min_y clamp (:190-192), output units */
    double ini[3];    /* fluence mode: [fluence cm^-2, alpha cm^-1, direction] (:106-113) */
    int32_t nx;       /* Grid.nx */
    int32_t nt;       /* len(Grid.tSteps) */
    int32_t meas;     /* MTRPL_MEAS_* */
    int32_t model;    /* MTRPL_MODEL_* */
    int32_t ini_mode; /* MTRPL_INI_* */
    int32_t reserved;
    int64_t ini_off;  /* density mode: first of nx values [cm^-3] in the ini array */
    int64_t t_off;    /* first of nt values in the tsteps array (tsteps[t_off] must be 0) */
    int64_t out_off;  /* first of nt values in the out array */
} mtrpl_solve_desc;

typedef struct mtrpl_ctx mtrpl_ctx;

/* ---- context ---------------------------------------------------------- */
/* (Hardware queues.  The HIP runtime multiplexes streams onto 4 hardware queues unless GPU_MAX_HW_QUEUES says otherwise
 * when the runtime starts, i.e. at the first HIP call of the process; two streams on one queue serialise their launches.
 * The streams the pipelined mtrpl_solve_batch runs, next to the caller's own, want 16 (measured: 390-430 k solves/s with
 * 8 queues in a process that owns ten streams, 485-528 k with 16).  The LIBRARY NEVER TOUCHES THE ENVIRONMENT: a C caller
 * exports GPU_MAX_HW_QUEUES=16 before its first HIP call; the Python package does so when it is imported unless
 * MTRPL_KEEP_HW_QUEUES=1 is set (metrotrpl_amd.configure_runtime).  Without it every call is still correct, only
 * slower -- host_streams = 0 then gives the one-launch rate back.) */
int mtrpl_ctx_create(int device, mtrpl_ctx **ctx);
int mtrpl_ctx_destroy(mtrpl_ctx *ctx);
/* Frees the device memory the context holds but no longer uses (blocks its per-launch slab pool has outgrown: a slab
 * grows geometrically, so they sum to less than twice the largest slab).  hipFree synchronises the device: call it at a
 * quiet point (between two runs), never while launches are in flight.  Nothing inside a launch path frees. */
int mtrpl_ctx_trim(mtrpl_ctx *ctx);
/* Pinned host memory (hipHostMalloc).  An `out` buffer of mtrpl_solve_batch that comes from here (or from any other
 * pinned allocation, e.g. a torch tensor with pin_memory=True) is filled by asynchronous device-to-host copies while
 * later sub-launches compute; a pageable one goes through a pinned staging ring and a host copy.  Freed by
 * mtrpl_host_free, or with the context. */
int mtrpl_host_alloc(mtrpl_ctx *ctx, int64_t bytes, void **out);
int mtrpl_host_free(mtrpl_ctx *ctx, void *p);
const char *mtrpl_last_error(const mtrpl_ctx *ctx);
const char *mtrpl_version(void);
/* Integrator options.  Unknown key -> MTRPL_ERR_BAD_ARG.
 *   rtol_scale, rtol_cap              internal rtol = min(rtol_scale * desc.rtol, rtol_cap) ...
 *   trts_rtol_factor                  ... x this for TRTS solves ...
 *   amp_ref, amp_post                 ... / max(1, (1 + tauP/tauN) / amp_ref) (0 = off; default 10) while the excess density is
 *                                     above amp_post x p0 (default 0.03; judged by the emitted signal: until the high- to
 *                                     low-injection transition is behind the solve) ...
 *   tail_frac, tail_factor            ... x tail_factor once an emitted sample has fallen below tail_frac (1e-4) x the
 *                                     first one (round 3: 0.4; 1 = off since the mass test below does that job) ...
 *   far_frac, far_factor              ... x far_factor (30; round 3: 3) instead below far_frac (1e-9): nothing measured reaches there ...
 *   rtol_mode                         0 (default): as above, the converged trace is the contract; 1: the caller's RTOL
 *                                     decides in both directions (forward_solver.py:163-165): internal rtol =
 *                                     rtol_scale * desc.rtol, capped at 1e-3 only
 *   rtol_floor                        ... these two policy factors never take it below rtol_floor;
 *   rtol_min                          a caller's own RTOL does go below rtol_floor, down to rtol_min
 *                                     (then MTRPL_SOLVE_OK_TOL_CLAMPED)
 *   rtol_force                        > 0: internal rtol = exactly this for every solve (tolerance sweeps)
 *   atol_rel                          absolute tolerance of the field components = rtol_int * atol_rel * max(N0, P0)
 *   atol_rel_np                       ... of the densities = min(rtol_int * atol_rel_np * max(N0, P0), desc.atol):
 *                                     default 1e-9 resolves a trace over 8 decades of decay inside the 1e-6 gate
 *                                     (1e-4: the first 4, 6 % fewer steps)
 *   shape_factor, fun_weight          the goal-oriented error test: the RMS norm of the local error
 *                                     estimate may reach shape_factor (30) x the tolerance as long as its MASS component
 *                                     -- the signed grid sum of the estimate for N against the grid sum of the excess
 *                                     N - n0, x fun_weight (3) -- stays inside the tolerance itself; shape_factor <= 1: the
 *                                     plain RMS test
 *   mass_floor                        ... the excess never counted smaller than mass_floor (1e-3) x n0 per node
 *   e_weight                          relative error weight of the field components (x rtol_int), TRPL solves only
 *   newton_tol, rate_max_age, rate_first_safety   simplified-Newton convergence control
 *   nq_deadband_hi                    a proposed step factor in [0.9, this) keeps the step and the factors (1.2)
 *   nq_rej_memory, nq_rej_growth      step growth capped at nq_rej_growth for nq_rej_memory accepted steps
 *                                     after a rejected one; nq_bias_same/_down/_up: multipliers of the step factors
 *   early_factor, early_frac          local tolerance x early_factor for steps that end before
 *                                     early_frac * (first requested sample time)
 *   honor_hmax                        1: desc.hmax is the maximum step (the reference's max_step); 0 (default): it is not --
 *                                     max_step guards LSODA's error control, this engine's own does not need it (no
 *                                     sample of any accuracy fixture moves, 9 % fewer steps) ...
 *   hmax_scale                        ... x hmax_scale (default 1) ...
 *   hmax_deep                         ... while there is a signal: once an emitted sample has fallen below hmax_deep x
 *                                     the first sample (default 1e-11: the state sits at equilibrium to rounding) the
 *                                     error test alone sets the step for the rest of the solve (the flag latches and is
 *                                     evaluated at output samples); 0: desc.hmax always holds
 *   deep_factor                       ... and from that sample on the local tolerance is multiplied by deep_factor (default 1000) on top
 *                                     of far_factor: eleven decades down no sample is resolved, and none is gated
 *   fresh_jac                         accepted and ignored (removed in round 2: the Jacobian is always refreshed at a
 *                                     refactorisation)
 *   max_steps                         a solve that needs more stops early (status 1, tail zero-filled)
 *   kernel_variant                    std model, nx <= 128: 0 automatic -- the wave kernel (one wavefront per solve)
 *                                     with the elimination blocks stored as fp32 and the re-solve in packed fp32 on
 *                                     an fp64 right-hand side (an approximate inverse applied to the fp64 residual
 *                                     of the simplified Newton iteration: same converged corrector, same accuracy
 *                                     gates), built twice: one solve per SIMD (6: the shortest solve) for launches
 *                                     of fewer than two_per_simd_min_batch solves, two per SIMD (3: 1.3-1.5x the
 *                                     solves per second when more solves than SIMDs are in flight) from there on
 *                                     (default 1025 = more solves than SIMDs on this device; a caller with several
 *                                     smaller launches in flight on different streams says so per launch,
 *                                     mtrpl_solve_batch_dev2, or sets the option to 1); a solve gives the same
 *                                     result on either build.  2: the wave kernel with fp64 blocks and an fp64
 *                                     re-solve (the exact linear algebra; agrees with the default inside the
 *                                     accuracy gate, not bitwise); 1 team kernel (the one traps and nx > 128 use).
 *                                     (4 / 5, round 2's wave-pair kernels, and pair_max_batch are gone: -> MTRPL_ERR_BAD_ARG)
 *   loglik_wave_blocks                0 (default): mtrpl_loglik_batch_dev runs twelve wavefronts and up to 159 KB of
 *                                     LDS per trace.  1 (2, 4): launches with nT >= 32 run ONE (two, four) wavefront(s) and the IRF work space
 *                                     (9 KB at 189 taps) per block, per-sample arrays in device memory -- a block that
 *                                     fits wherever a single solve wavefront has left, for a likelihood launch that
 *                                     shares the device with another stream's queued solves (a twelve-wave block waits
 *                                     until that queue has drained); the same table bit for bit (with 32 and more
 *                                     temperatures no sum depends on the block size).  Measured on three chain groups of a
 *                                     512-chain ensemble: 6.7 / 6.9 / 8.5 ms per iteration with 1 / 2 / 4 waves per block,
 *                                     8.5 with twelve
 *   host_chunk, host_streams          mtrpl_solve_batch (host pointers) with more than host_chunk (2048; / 2, / 4 for the
 *                                     two- and four-wave kernels of nx > 128) solves runs as sub-launches of that size,
 *                                     round-robin on host_streams (2; 0 = always one blocking launch, at most 8) streams
 *                                     the context owns, each sub-launch's samples copied back while the later ones
 *                                     compute.  Same results bit for bit
 *   precision                         64 (product) | 32 (fp32 instantiation of the team kernel: precision sweep only) */
int mtrpl_set_option(mtrpl_ctx *ctx, const char *key, double value);
int mtrpl_get_option(const mtrpl_ctx *ctx, const char *key, double *value);

/* ---- a1/a6/a7/a8/a9: batched forward solve ----------------------------- */
/* replaces forward_solver.solve (forward_solver.py:41-203) for B calls at once.
 * stats (may be NULL): 8 int32 per solve = steps, rhs evals, factorisations,
 * jacobian refreshes, rejected steps, newton failures, final order, residency of the solve on
 * its SIMD in units of 16 shader-clock ticks.
 * Blocking; a big batch is pipelined INSIDE the call (options host_chunk / host_streams above): sub-launches on the
 * context's own streams so that the SIMD slots the short solves of one free go to the next, device-to-host copies of
 * finished sub-launches under the later ones' compute.  The caller needs no streams and sets nothing.  (Needs the
 * descriptors to fill `out` in ascending order -- out_off non-decreasing from one sub-launch to the next, as every
 * packer of this repo lays them out; otherwise the call falls back to one launch.) */
int mtrpl_solve_batch(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *descs,
                      const double *ini, int64_t n_ini, const double *tsteps, int64_t n_t,
                      double *out, int64_t n_out, int32_t *status, int32_t *stats);
/* Device-resident variant.  The descriptors live in HBM, so the caller states what
 * the kernel selection needs: max_nx >= every desc's nx (<= 512) and the model all
 * B descriptors share. */
int mtrpl_solve_batch_dev(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs,
                          const double *d_ini, const double *d_tsteps, double *d_out,
                          int32_t *d_status, int32_t *d_stats, int32_t max_nx, int32_t model,
                          void *stream);
/* The same launch with the build of the wave kernel chosen by the caller, for THIS launch only (std model, nx <= 128,
 * kernel_variant 0; ignored elsewhere): solves_per_simd = 1 -- one solve per SIMD, the shortest solve --, 2 -- two per
 * SIMD, 1.3-1.5x the solves per second once more solves than SIMDs are in flight --, 0 -- as mtrpl_solve_batch_dev
 * chooses, by B against the option two_per_simd_min_batch.  For callers with several launches in flight on different
 * streams (chain groups of a parallel-tempering ensemble, lanes of the dense sampler): what is on the device together
 * is their knowledge, not a property of the context -- nothing has to flip a context option around a launch.
 * A solve gives the same samples on either build. */
int mtrpl_solve_batch_dev2(mtrpl_ctx *ctx, int64_t B, const mtrpl_solve_desc *d_descs,
                           const double *d_ini, const double *d_tsteps, double *d_out,
                           int32_t *d_status, int32_t *d_stats, int32_t max_nx, int32_t model,
                           void *stream, int32_t solves_per_simd);

/* ---- a4/a5: RHS of the ODE system (forward_solver.py:332-418) ---------- */
/* y, dy: [B, S] with S = 3nx+1 (std: N,P,E) or 4nx+1 (traps: N,Ntrap,P,E);
 * par: [B,16] MTRPL_P_* slots; dx: [B]. Host pointers. */
int mtrpl_rhs_batch(mtrpl_ctx *ctx, int64_t B, int32_t nx, int32_t model, const double *par,
                    const double *dx, const double *y, double *dy);

/* ---- a3: E_field (forward_solver.py:26-38), rows x nx -> rows x (nx+1) - */
int mtrpl_E_field(mtrpl_ctx *ctx, int64_t rows, int32_t nx, const double *N, const double *P,
                  double n0, double p0, double eps, double dx, double corner_E, double *E);

/* ---- a7/a8: spatial integrals (forward_solver.py:228-274) -------------- */
int mtrpl_calculate_PL(mtrpl_ctx *ctx, int64_t rows, int32_t nx, double dx, const double *N,
                       const double *P, double ks, double n0, double p0, double *out);
int mtrpl_calculate_TRTS(mtrpl_ctx *ctx, int64_t rows, int32_t nx, double dx, const double *N,
                         const double *P, double mu_n, double mu_p, double n0, double p0,
                         double *out);

/* forward_solver.py:267-274: calculate_RR = ks (N P - n0 p0) and calculate_photoc =
 * q_C (mu_n (N - n0) + mu_p (P - p0)), element-wise over n values. */
int mtrpl_calculate_RR(mtrpl_ctx *ctx, int64_t n, const double *N, const double *P, double ks,
                       double n0, double p0, double *out);
int mtrpl_calculate_photoc(mtrpl_ctx *ctx, int64_t n, const double *N, const double *P, double mu_n,
                           double mu_p, double n0, double p0, double *out);

/* ---- test entry: one Newton linear solve (I - c*J(y)) x = b ------------- */
/* y, b, x: [B, 3nx+1] std layout. Exposes the kernel's E-elimination +
 * block cyclic-reduction solver for the parity tests. */
int mtrpl_newton_probe(mtrpl_ctx *ctx, int64_t B, int32_t nx, const double *par, const double *dx,
                       const double *c, const double *y, const double *b, double *x);

/* ---- test entry: the controller's merged fp32 wave reductions -------------- */
/* in: [B][4][64] floats (4 per-lane quantities of one wavefront); out: [B][7] = sum of row 0 through the one-sum
 * ladder, of rows 0,1 through the two-sum ladder, of rows 0..3 through the four-sum ladder (DPP quad / row
 * steps + the two permlane swaps of gfx950).  The error norms of the integrator go through these. */
int mtrpl_reduce_probe(mtrpl_ctx *ctx, int64_t B, const float *in, float *out);

/* ---- a10: IRF moment tables (laplace.py:13-41, :129-167) ---------------- */
/* irf_t, irf_y: [nk]; table: [nk,3] (last row zero). */
int mtrpl_make_I_table(mtrpl_ctx *ctx, int32_t nk, const double *irf_t, const double *irf_y,
                       double *table);

/* laplace.py:129-167 (I_moment) with the caller's settings: one moment I_m^n of the IRF. */
int mtrpl_I_moment(mtrpl_ctx *ctx, int32_t nk, const double *irf_t, const double *irf_y, int32_t m,
                   int32_t n, int32_t u_lower, int32_t u_upper, int32_t u_spacing, double *out);

/* ---- a11/a12: IRF convolution (laplace.py:44-85, :170-212) -------------- */
/* Number of convolved samples do_irf_convolution would return for a trace
 * ending at t_end with IRF spacing dt_irf: len(arange(0,t_end+dt/4,dt/2)[::2]). */
int64_t mtrpl_irf_conv_len(double t_end, double dt_irf);
/* One trace (host pointers): t,y [nt] -> conv_t, conv_y [n_conv]; success flag
 * as laplace.py:76-84. */
int mtrpl_irf_convolve(mtrpl_ctx *ctx, int32_t nt, const double *t, const double *y, int32_t nk,
                       const double *table, double dt_irf, int32_t time_max_shift, int64_t n_conv,
                       double *conv_t, double *conv_y, int32_t *success);
/* laplace.py:170-212 alone: resampled_y [n] -> h_k [(n-1)/2+1] */
int mtrpl_convolve(mtrpl_ctx *ctx, int64_t n, const double *resampled_y, int32_t nk,
                   const double *table, double *h_k);

/* ---- a13: the 1-D linear interpolation of post_conv_trim (laplace.py:121) and of
 * the resampling step (laplace.py:74): scipy griddata semantics, NaN outside. */
int mtrpl_interp_linear(mtrpl_ctx *ctx, int64_t n, const double *x, const double *y, int64_t nq,
                        const double *xq, double *yq);

/* ---- a11-a14 fused, batched, device-resident: solve output -> log-likelihood
 * (trial_move_evaluation.py:75-158, laplace.py:44-126). See DESIGN.md. ---- */
typedef struct mtrpl_like_desc {
    int64_t sig_off;   /* first of nt simulated samples (output of the solve) */
    int64_t t_off;     /* simulation times (nt) */
    int64_t exp_off;   /* first of n_exp experimental times/vals/uncs */
    int32_t nt;
    int32_t n_exp;
    int32_t irf;       /* index of the IRF table to apply, -1 = none */
    int32_t force_min_y; /* utils.set_min_y (utils.py:16-31, trial_move_evaluation.py:136-140) */
    double scale_shift; /* log10(scale factor), trial_move_evaluation.py:58 */
    double sigma;       /* model_uncertainty[meas_type] */
} mtrpl_like_desc;

int mtrpl_irf_tables_upload(mtrpl_ctx *ctx, int32_t n_tables, const int32_t *nk,
                            const double *tables /* concatenated [nk,3] */, const double *dt_irf);
/* ll[B, nT]: -sum err_sq / (sigma^2 T + 2 u^2) per measurement and temperature;
 * -inf where the reference would return -inf.  max_nt >= every desc's nt and
 * n_exp: it sizes the staging of the per-sample arrays, in LDS while 48 bytes per
 * sample fit next to the IRF tables (about 3400 samples), in a per-launch slab of
 * device memory beyond that -- no upper limit on the trace length. */
int mtrpl_loglik_batch_dev(mtrpl_ctx *ctx, int64_t B, const mtrpl_like_desc *d_descs,
                           const double *d_sig, const double *d_tsteps, const double *d_exp_t,
                           const double *d_exp_y, const double *d_exp_u, const double *d_T,
                           int32_t nT, int32_t max_nt, double *d_ll, void *stream);

/* ---- parallel tempering: a launch's decisions taken where its log-likelihoods are -------------------------------
 * (the Metropolis rule of metropolis.py:42-63 for a chain block.  A launch evaluates, per chain, n_props proposals:
 * 1 -- y1 for iteration k+1 -- or, prefetching, 3 -- y1, y2r for k+2 from the current state, y2a for k+2 from y1.
 * The caller -- metrotrpl_amd/pt.py:_DecidePipe -- prepares the next launch's descriptors for every outcome (two /
 * four) while the launch runs, so that no launch waits for the host between two replica exchanges.)
 *
 * mtrpl_pt_decide_dev, for chain i of c, own ladder slot col[i], row row[i] of the block's current table:
 *     L_g = tab[g, i, col[i]]               tab [n_props, c, nT]: the proposals' rows, summed over the measurements
 *     a1  = u[0, i] < exp(L_0 - cur_ll[i])  (a NaN ratio rejects, metropolis.py:49-51)
 *     a2  = n_props == 3 && u[1, i] < exp((a1 ? L_2 : L_1) - (a1 ? L_0 : cur_ll[i]))
 *     cur_ll[i], cur_tab[row[i], 0..nT) <- the log-likelihood / the row of the state the decisions leave
 *     outcome[i] = a1 + 2 a2;  result [4, c] = outcome, L_0, L_1, L_2 (what the host needs to follow the chain)
 * mtrpl_pt_select_dev: next [n_rows, bytes_per_row] <- cand [n_cand, n_rows, bytes_per_row] at
 *     outcome[chain_of_row[r]] (n_cand = 1: the first launch after an exchange, nothing else read; else the
 *     outcomes in the order s, y1[, y2r, y2a]); a row is one mtrpl_solve_desc, in whatever order the launch
 *     issues them.
 * Everything is device memory; both only enqueue on `stream`. */
int mtrpl_pt_decide_dev(mtrpl_ctx *ctx, int32_t c, int32_t nT, int32_t n_props, const double *d_tab,
                        const int32_t *d_col, const int32_t *d_row, const double *d_u, double *d_cur_ll,
                        double *d_cur_tab, int32_t *d_outcome, double *d_result, void *stream);
int mtrpl_pt_select_dev(mtrpl_ctx *ctx, int64_t n_rows, int32_t n_cand, int64_t bytes_per_row, const void *d_cand,
                        const int32_t *d_chain_of_row, const int32_t *d_outcome, void *d_next, void *stream);

#ifdef __cplusplus
}
#endif
#endif
