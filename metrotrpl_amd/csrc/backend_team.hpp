// backend_team.hpp -- the general backend: W wavefronts (64*W threads, one workgroup) cooperate on one
// forward solve, two nodes per thread, nx <= 128*W, std or traps model (forward_solver.py:332-418).
// Covers what the register-resident backends do not: the traps model and nx up to 512
// (BASELINE.json configs[4]).  Every cross-thread exchange goes through a ping-pong LDS buffer as
// 16-byte rows (one ds_write_b128 + one ds_read_b128 per pair of values), which works across waves
// with one s_barrier per exchange.  For the traps model the trapped density N_trap couples only to the
// N of its own node, so it is eliminated per node exactly like E is per edge and the Newton matrix
// stays 2x2 block tridiagonal.
//
// The scalar type is a template parameter: `double` is the product; `float` exists for the precision
// sweep of configs[4] only (fp32 cannot hold this system: I - cJ has condition numbers of 1e6-1e7).
#pragma once
#include "solve_core.hpp"

namespace mtrpl {

// BF32 (fp64 solves only): the elimination blocks are stored rounded to fp32, one 16-byte row per block and
// thread -- half the bytes of the slab the W = 4 kernel streams on every re-solve.  As in the wave kernels they
// only define the approximate inverse the simplified Newton iteration applies to an fp64 residual.
template <class T_, int W, bool TRAPS_, bool GBLK, bool BF32 = false>
struct TeamBackend {
    using real = T_;
    using row_t = vec2_t<real>;  // one 2-value row (16 B in fp64)
    using M2r = M2T<real>;
    using V2r = V2T<real>;
    static constexpr int R = 2, V = TRAPS_ ? 8 : 6, T = 64 * W;
    static constexpr bool TRAPS = TRAPS_;
    static constexpr bool kFieldInNorms = false;  // the field components in the controller's norms (solve_core.hpp)
    static constexpr bool kOrderRowInVgpr = false;
    static constexpr bool kBlocksSamples = false;
    static constexpr bool kScalarPolicy = false;  // policy constants of a solve held in scalar registers (solve_core.hpp)
    static constexpr bool kParksSamples = false;
    static constexpr int LEVELS = 6 + (W >= 2 ? 1 : 0) + (W >= 4 ? 1 : 0);  // PCR levels over T rows
    static constexpr int NBLK = 5 + 2 * LEVELS;                              // elimination blocks per thread
    static constexpr int KMAX = 6;                                           // rows per exchange
    static constexpr int kXbufRows = 2 * KMAX * T;
    static constexpr int kBlkRows = (BF32 ? NBLK : NBLK * 2) * T;  // in units of 16 bytes (fp64 solves)
    static constexpr int kRbufVals = 2 * 4 * W + W + 4;
    enum { BINV_E = 0, BIA_E, BIC_E, CR_AL, CR_GA, PAL0 = 5 };

    ParT<real> p;
    int L;
    int tid, lane, wave;
    Emitter em;
    row_t *xbuf;  // [2][KMAX][T]
    real *rbuf;   // [2][4][W] + scratch
    int xpar, rpar;
    // factors
    real w[2][2], eco[2][4], minv_[2], cd;
    real tq[2], tca[2];  // traps: 1/(1+c b_i), c a_i   (dNt_i = tq*(bNt + tca*dN_i))
    real tcb[2];         // traps: c b_i * tq          (rN_i += tcb * bNt_i)
    M2r Bfin;
    row_t *blk;  // [NBLK][2][T] rows (BF32: [NBLK][T] float4): elimination blocks of this thread at [.][.][tid]

    __device__ inline TeamBackend(int tid_, row_t *xbuf_, real *rbuf_, row_t *blk_) : xbuf(xbuf_), rbuf(rbuf_), blk(blk_)
    {
        tid = tid_;
        lane = tid_ & 63;
        wave = tid_ >> 6;
        xpar = 0;
        rpar = 0;
    }
    __device__ inline void set_problem(const ParT<real> &p_, int L_) { p = p_; L = L_; }
    __device__ inline int node(int r) const { return tid * R + r; }
    __device__ inline bool leader() const { return tid == 0; }
    __device__ inline void sync() const
    {
        if (W > 1) __syncthreads();
    }
    static __device__ inline row_t d2(real x, real y) { return row_t{x, y}; }
    static __device__ inline void m2rows(const M2r &m, row_t &r0, row_t &r1) { r0 = d2(m.a, m.b); r1 = d2(m.c, m.d); }
    static __device__ inline M2r rows2m(const row_t &r0, const row_t &r1) { return M2r{r0.x, r0.y, r1.x, r1.y}; }

    // write K rows, read them back from thread `src`
    template <int K>
    __device__ inline void xget(const row_t (&in)[K], int src, row_t (&out)[K])
    {
        row_t *b = xbuf + xpar * (KMAX * T);
#pragma unroll
        for (int k = 0; k < K; ++k) b[k * T + tid] = in[k];
        sync();
#pragma unroll
        for (int k = 0; k < K; ++k) out[k] = b[k * T + src];
        xpar ^= 1;
    }
    // ... and from two threads at once (the i-s / i+s pair of a PCR level)
    template <int K>
    __device__ inline void xget2(const row_t (&in)[K], int sa, int sb, row_t (&oa)[K], row_t (&ob)[K])
    {
        row_t *b = xbuf + xpar * (KMAX * T);
#pragma unroll
        for (int k = 0; k < K; ++k) b[k * T + tid] = in[k];
        sync();
#pragma unroll
        for (int k = 0; k < K; ++k) {
            oa[k] = b[k * T + sa];
            ob[k] = b[k * T + sb];
        }
        xpar ^= 1;
    }

    // ---- whole-grid primitives ----
    // fp32 sums over all threads of the team (they only steer the controller); every thread gets bitwise the
    // same result
    template <int NS>
    __device__ inline void reduce_f(float (&s)[NS])
    {
        static_assert(NS <= 4, "rbuf holds 4 sums per wave");
        wave_sums_f<NS>(s, lane);
        if (W == 1) return;
        float *r = reinterpret_cast<float *>(rbuf + rpar * (4 * W));
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) r[k * W + wave] = s[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            float a = 0.0f;
#pragma unroll
            for (int w_ = 0; w_ < W; ++w_) a += r[k * W + w_];
            s[k] = a;
        }
        rpar ^= 1;
    }
    __device__ inline real scan_excl(real run)
    {
        const real incl = wave_scan_incl(run, lane);
        real excl = incl - run;
        if (W > 1) {  // add the totals of the waves to the left
            real *r_ = rbuf + 2 * 4 * W;
            if (lane == 63) r_[wave] = incl;
            __syncthreads();
            for (int w_ = 0; w_ < W; ++w_)
                if (w_ < wave) excl += r_[w_];
            __syncthreads();
        }
        return excl;
    }
    __device__ inline real max_all(real v)
    {
        v = wave_max(v);
        if (W > 1) {
            real *r_ = rbuf + 2 * 4 * W;
            if (lane == 0) r_[wave] = v;
            __syncthreads();
            for (int w_ = 0; w_ < W; ++w_) v = fmax(v, r_[w_]);
            __syncthreads();
        }
        return v;
    }
    __device__ inline void out_begin(double *o, double min_y, double scale, double deep, double tail, double far, double post) { em.begin(o, min_y, scale, deep, tail, far, post); }
    __device__ inline bool post() const { return em.post(); }
    __device__ inline bool tail() const { return em.tail(); }
    __device__ inline bool far() const { return em.far(); }
    __device__ inline bool deep() const { return em.deep(); }
    __device__ inline real team_sum(real partial)
    {
        // full-precision sum (reduce_f() is the controller-grade one)
        real s = wave_sum(partial);
        if (W > 1) {
            real *r = rbuf + rpar * (4 * W);
            if (lane == 0) r[wave] = s;
            __syncthreads();
            s = (real)0;
#pragma unroll
            for (int w_ = 0; w_ < W; ++w_) s += r[w_];
            rpar ^= 1;
        }
        return s;
    }
    __device__ inline void out_first(real partial) { em.first((double)team_sum(partial), (double)p.dx, tid == 0); }
    __device__ inline void out_sample(int k, real partial) { em.put(k, (double)team_sum(partial), (double)p.dx, tid == 0); }
    __device__ inline void out_flush() {}

    __device__ inline void put(int b, const M2r &m) const
    {
        if constexpr (BF32) {
            reinterpret_cast<float4 *>(blk)[b * T + tid] = float4{(float)m.a, (float)m.b, (float)m.c, (float)m.d};
        } else {
            blk[(b * 2 + 0) * T + tid] = row_t{m.a, m.b};
            blk[(b * 2 + 1) * T + tid] = row_t{m.c, m.d};
        }
    }
    __device__ inline M2r get(int b) const
    {
        if constexpr (BF32) {
            const float4 q = reinterpret_cast<const float4 *>(blk)[b * T + tid];
            return M2r{(real)q.x, (real)q.y, (real)q.z, (real)q.w};
        } else {
            const row_t u = blk[(b * 2 + 0) * T + tid], v = blk[(b * 2 + 1) * T + tid];
            return M2r{u.x, u.y, v.x, v.y};
        }
    }

    // ---- RHS ----
    // y = [N0,N1,P0,P1,E0,E1(,Nt0,Nt1)], E on the left face of the node; node i = 2*tid + r.
    __device__ inline void rhs(const real (&y)[V], real (&f)[V])
    {
        row_t o1[1], in1[1] = {d2(y[1], y[3])};
        xget<1>(in1, (tid + T - 1) & (T - 1), o1);  // N, P of the node left of this thread's first node
        real Jn[3], Jp[3];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = 2 * tid + r;
            const real N = y[r], P = y[2 + r], E = y[4 + r];
            const real Nm = (r == 0) ? o1[0].x : y[0];
            const real Pm = (r == 0) ? o1[0].y : y[2];
            real jn, jp;
            edge_flux(p, Nm, N, Pm, P, E, jn, jp);  // forward_solver.py:356-359
            if (i == 0) {  // :346-351
                const real Sft = surface_term(p, p.Sf, N, P);
                jn = Sft;
                jp = -Sft;
            }
            if (i >= L) { jn = (real)0; jp = (real)0; }
            Jn[r] = jn;
            Jp[r] = jp;
        }
        row_t o2[1], in2[1] = {d2(Jn[0], Jp[0])};
        xget<1>(in2, (tid + 1) & (T - 1), o2);  // fluxes on the right face of this thread's last node
        Jn[2] = o2[0].x;
        Jp[2] = o2[0].y;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = 2 * tid + r;
            const real N = y[r], P = y[2 + r];
            const real NP = N * P - p.n0p0;
            real jnr = Jn[r + 1], jpr = Jp[r + 1];
            if (i == L - 1) {  // :347,:350,:352
                const real Sbt = p.Sb * NP * frcp1(N + P);
                jnr = -Sbt;
                jpr = Sbt;
            }
            const real recomb = bulk_recomb(p, N, P, NP);  // :366
            const bool real_ = i < L;
            real fn = (jnr - Jn[r]) * p.idx - recomb;         // :369
            const real fp = -(jpr - Jp[r]) * p.idx - recomb;  // :370
            if (TRAPS) {  // :410-416
                const real Nt = y[6 + r];
                const real trap = p.kC * N * (p.Nt - Nt), detrap = Nt / p.tE;
                fn = fn + detrap - trap;
                f[6 + r] = real_ ? (trap - detrap) : (real)0;
            }
            f[r] = real_ ? fn : (real)0;
            f[2 + r] = real_ ? fp : (real)0;
            f[4 + r] = (real_ && i > 0) ? (-(Jn[r] + Jp[r]) * p.Lam) : (real)0;  // :363 (edge 0: exactly 0)
        }
    }

    // ---- Newton matrix factors ----
    __device__ inline void build(const real (&y_in)[V], real c)
    {
        real y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = pinned(y_in[v]);  // keep the Jacobian inside the branch it is called from
        const real cL = c * p.Lam;
        cd = c * p.idx;
        const real dnx = p.dnx, dpx = p.dpx;
        row_t o1[1], in1[1] = {d2(y[1], y[3])};
        xget<1>(in1, (tid + T - 1) & (T - 1), o1);
        M2r Lm[2], Rm[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = 2 * tid + r;
            const real N = y[r], P = y[2 + r], E = y[4 + r];
            const real Nm = (r == 0) ? o1[0].x : y[0];
            const real Pm = (r == 0) ? o1[0].y : y[2];
            EdgeJac<real> ej = edge_jac_interior(p, cL, dnx, dpx, Nm, N, Pm, P, E);
            if (i == 0) ej = edge_jac_front(p, N, P);
            if (i >= L) ej = edge_jac_none<real>();
            Lm[r] = ej.lm;
            Rm[r] = ej.rm;
            w[r][0] = ej.w0; w[r][1] = ej.w1;
            eco[r][0] = ej.e0; eco[r][1] = ej.e1; eco[r][2] = ej.e2; eco[r][3] = ej.e3;
            minv_[r] = ej.mv;
        }
        // edge matrices of the next thread's first edge
        row_t nin[4], nout[4];
        m2rows(Lm[0], nin[0], nin[1]);
        m2rows(Rm[0], nin[2], nin[3]);
        xget<4>(nin, (tid + 1) & (T - 1), nout);
        const M2r LmNext = rows2m(nout[0], nout[1]), RmNext = rows2m(nout[2], nout[3]);
        M2r A[2], B[2], C[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = 2 * tid + r;
            const real N = y[r], P = y[2 + r];
            M2r lmn = (r == 1) ? LmNext : Lm[1];
            M2r rmn = (r == 1) ? RmNext : Rm[1];
            if (i == L - 1) {  // back surface: J_L = (-SL, SL)
                real sLn, sLp;
                surface_jac(p, p.Sb, N, P, sLn, sLp);
                lmn = M2r{-sLn, -sLp, sLn, sLp};
                rmn = mzero<real>();
            }
            real RN, RP;
            bulk_jac(p, N, P, RN, RP);
            node_row(c, cd, Lm[r], Rm[r], lmn, rmn, RN, RP, A[r], B[r], C[r]);
            if (TRAPS) {
                // dN row: +c*a dN - c*b dNt ; dNt row: (1 + c b) dNt - c a dN = bNt   (a, b below)
                const real Nt = y[6 + r];
                const real ta = p.kC * (p.Nt - Nt), tb = p.kC * N + (real)1 / p.tE;
                const real q = frcp((real)1 + c * tb);
                tq[r] = q;
                tca[r] = c * ta;
                tcb[r] = c * tb * q;
                B[r].a += c * ta * q;
            }
            if (i >= L) {
                A[r] = mzero<real>();
                C[r] = mzero<real>();
                B[r] = mident<real>();
                if (TRAPS) { tq[r] = (real)1; tca[r] = (real)0; tcb[r] = (real)0; }
            }
        }
        // in-thread cyclic reduction: row e = 2*tid eliminated, row k = 2*tid+1 kept
        const M2r Binv_e = minv(B[0]);
        put(BINV_E, Binv_e);
        put(BIA_E, mmul(Binv_e, A[0]));
        put(BIC_E, mmul(Binv_e, C[0]));
        row_t uin[6], uo[6];
        m2rows(Binv_e, uin[0], uin[1]);
        m2rows(A[0], uin[2], uin[3]);
        m2rows(C[0], uin[4], uin[5]);
        xget<6>(uin, (tid + 1) & (T - 1), uo);
        const M2r Binv_up = rows2m(uo[0], uo[1]), A_up = rows2m(uo[2], uo[3]), C_up = rows2m(uo[4], uo[5]);
        const M2r cr_al = mmul(A[1], Binv_e);
        const M2r cr_ga = mmul(C[1], Binv_up);
        put(CR_AL, cr_al);
        put(CR_GA, cr_ga);
        M2r Ak = mneg(mmul(cr_al, A[0]));
        M2r Ck = mneg(mmul(cr_ga, C_up));
        M2r Bk = msub(msub(B[1], mmul(cr_al, C[0])), mmul(cr_ga, A_up));
        // parallel cyclic reduction over the T kept rows
#pragma unroll
        for (int lv = 0; lv < LEVELS; ++lv) {
            const int s = 1 << lv;
            const M2r Bi = minv(Bk);
            row_t pin[6], pm[6], pp[6];
            m2rows(Bi, pin[0], pin[1]);
            m2rows(Ak, pin[2], pin[3]);
            m2rows(Ck, pin[4], pin[5]);
            xget2<6>(pin, (tid - s) & (T - 1), (tid + s) & (T - 1), pm, pp);
            const M2r Bim = rows2m(pm[0], pm[1]), Akm = rows2m(pm[2], pm[3]), Ckm = rows2m(pm[4], pm[5]);
            const M2r Bip = rows2m(pp[0], pp[1]), Akp = rows2m(pp[2], pp[3]), Ckp = rows2m(pp[4], pp[5]);
            const M2r al = mmul(Ak, Bim), ga = mmul(Ck, Bip);
            put(PAL0 + 2 * lv, al);
            put(PAL0 + 2 * lv + 1, ga);
            Bk = msub(msub(Bk, mmul(al, Ckm)), mmul(ga, Akp));
            Ak = mneg(mmul(al, Akm));
            Ck = mneg(mmul(ga, Ckp));
        }
        Bfin = minv(Bk);
    }

    // x = M^{-1} b, layout as y
    __device__ inline void apply(const real (&b)[V], real (&x)[V])
    {
        const int tnext = (tid + 1) & (T - 1), tprev = (tid + T - 1) & (T - 1);
        real bn[2] = {b[0], b[1]};
        if (TRAPS) {
            bn[0] += tcb[0] * b[6];
            bn[1] += tcb[1] * b[7];
        }
        V2r wb[3];
        wb[0] = V2r{w[0][0] * b[4], w[0][1] * b[4]};
        wb[1] = V2r{w[1][0] * b[5], w[1][1] * b[5]};
        row_t i1[1] = {d2(wb[0].x, wb[0].y)}, o1[1];
        xget<1>(i1, tnext, o1);
        wb[2] = V2r{o1[0].x, o1[0].y};
        V2r r[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
            r[q] = V2r{bn[q] + cd * (wb[q + 1].x - wb[q].x), b[2 + q] - cd * (wb[q + 1].y - wb[q].y)};
        row_t i2[1] = {d2(r[0].x, r[0].y)}, o2[1];
        xget<1>(i2, tnext, o2);
        const V2r re_up{o2[0].x, o2[0].y};
        const V2r t1 = mvec(get(CR_AL), r[0]), t2 = mvec(get(CR_GA), re_up);
        V2r rk{r[1].x - t1.x - t2.x, r[1].y - t1.y - t2.y};
#pragma unroll
        for (int lv = 0; lv < LEVELS; ++lv) {
            const int s = 1 << lv;
            row_t pi[1] = {d2(rk.x, rk.y)}, pm[1], pp[1];
            xget2<1>(pi, (tid - s) & (T - 1), (tid + s) & (T - 1), pm, pp);
            const V2r u1 = mvec(get(PAL0 + 2 * lv), V2r{pm[0].x, pm[0].y});
            const V2r u2 = mvec(get(PAL0 + 2 * lv + 1), V2r{pp[0].x, pp[0].y});
            rk = V2r{rk.x - u1.x - u2.x, rk.y - u1.y - u2.y};
        }
        const V2r xk = mvec(Bfin, rk);
        row_t i3[1] = {d2(xk.x, xk.y)}, o3[1];
        xget<1>(i3, tprev, o3);
        const V2r xk_prev{o3[0].x, o3[0].y};
        const V2r s0 = mvec(get(BINV_E), r[0]), s1 = mvec(get(BIA_E), xk_prev), s2 = mvec(get(BIC_E), xk);
        const V2r xe{s0.x - s1.x - s2.x, s0.y - s1.y - s2.y};
        x[0] = xe.x; x[1] = xk.x; x[2] = xe.y; x[3] = xk.y;
        x[4] = b[4] * minv_[0] - (eco[0][0] * xk_prev.x + eco[0][1] * xe.x + eco[0][2] * xk_prev.y + eco[0][3] * xe.y);
        x[5] = b[5] * minv_[1] - (eco[1][0] * xe.x + eco[1][1] * xk.x + eco[1][2] * xe.y + eco[1][3] * xk.y);
        if (TRAPS) {
            x[6] = tq[0] * (b[6] + tca[0] * x[0]);
            x[7] = tq[1] * (b[7] + tca[1] * x[1]);
        }
    }
};

// GBLK: the elimination blocks live in a per-solve slab in HBM (W = 4: 172 KB per solve do not fit LDS
// next to the exchange buffers; the slab is L2/MALL-resident and read as coalesced 16-byte rows).
// `gblk` is private to this launch (mtrpl_capi.hip takes it from a per-launch pool).
template <class T_, int W, bool TRAPS, bool GBLK, bool BF32 = false>
__global__ __launch_bounds__(64 * W) void solve_kernel_team(const mtrpl_solve_desc *__restrict__ descs,
                                                             const double *__restrict__ ini,
                                                             const double *__restrict__ tsteps,
                                                             double *__restrict__ out, int32_t *__restrict__ status,
                                                             int32_t *__restrict__ stats, int64_t B, SolveTuning tune,
                                                             void *__restrict__ gblk)
{
    using BK = TeamBackend<T_, W, TRAPS, GBLK, BF32>;
    using row_t = typename BK::row_t;
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if ((int64_t)blockIdx.x >= B) return;
    const int64_t sidx = blockIdx.x;
    __shared__ row_t xbuf_lds[BK::kXbufRows];
    __shared__ T_ rbuf_lds[BK::kRbufVals];
    __shared__ row_t blk_lds[GBLK ? 1 : BK::kBlkRows];
    row_t *blkp = GBLK ? (static_cast<row_t *>(gblk) + (size_t)blockIdx.x * BK::kBlkRows) : blk_lds;
    BK bk(threadIdx.x, xbuf_lds, rbuf_lds, blkp);
    integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

}  // namespace mtrpl
