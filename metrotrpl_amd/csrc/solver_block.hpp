// solver_block.hpp -- generalised device code: W wavefronts (64*W threads, one
// workgroup) cooperate on one forward solve, two nodes per thread, nx <= 128*W,
// std or traps model (forward_solver.py:332-418).
//
// Same numerical method as solver_device.hpp (NDF + analytic Jacobian +
// E-elimination + cyclic reduction), but every cross-thread exchange goes through
// a ping-pong LDS buffer as 16-byte rows (one ds_write_b128 + one ds_read_b128 per
// pair of doubles instead of four ds_bpermute_b32), which works across waves with
// one s_barrier per exchange.  For the traps model the trapped density N_trap
// couples only to the N of its own node, so it is eliminated per node exactly like
// E is per edge and the Newton matrix stays 2x2 block tridiagonal.
// (no #pragma once: mtrpl_fp32_pass.hpp compiles this file a second time with `double` read as `float`)
#if (!defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVER_BLOCK_HPP_F64)) || (defined(MTRPL_FP32_PASS) && !defined(MTRPL_SOLVER_BLOCK_HPP_F32))
#ifdef MTRPL_FP32_PASS
#define MTRPL_SOLVER_BLOCK_HPP_F32
#else
#define MTRPL_SOLVER_BLOCK_HPP_F64
#endif
#include "solver_device.hpp"

namespace mtrpl {

template <int W>
struct Team {
    static constexpr int T = 64 * W;                                    // threads per solve
    static constexpr int LEVELS = 6 + (W >= 2 ? 1 : 0) + (W >= 4 ? 1 : 0);  // PCR levels over T rows
    static constexpr int NBLK = 5 + 2 * LEVELS;                          // elimination blocks per thread
    static constexpr int KMAX = 6;                                       // double2 rows per exchange
    int tid, lane, wave;
    double2 *xbuf;  // [2][KMAX][T]
    double *rbuf;   // [2][2][W]
    int xpar, rpar;
    __device__ inline void sync() const
    {
        if (W > 1) __syncthreads();
    }
};

template <int W>
constexpr int team_xbuf_bytes() { return 2 * Team<W>::KMAX * Team<W>::T * 16; }

// write K rows, read them back from thread `src`
template <int W, int K>
__device__ inline void xget(Team<W> &tm, const double2 (&in)[K], int src, double2 (&out)[K])
{
    constexpr int T = Team<W>::T;
    double2 *b = tm.xbuf + tm.xpar * (Team<W>::KMAX * T);
#pragma unroll
    for (int k = 0; k < K; ++k) b[k * T + tm.tid] = in[k];
    tm.sync();
#pragma unroll
    for (int k = 0; k < K; ++k) out[k] = b[k * T + src];
    tm.xpar ^= 1;
}
// ... and from two threads at once (the i-s / i+s pair of a PCR level)
template <int W, int K>
__device__ inline void xget2(Team<W> &tm, const double2 (&in)[K], int sa, int sb, double2 (&oa)[K], double2 (&ob)[K])
{
    constexpr int T = Team<W>::T;
    double2 *b = tm.xbuf + tm.xpar * (Team<W>::KMAX * T);
#pragma unroll
    for (int k = 0; k < K; ++k) b[k * T + tm.tid] = in[k];
    tm.sync();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        oa[k] = b[k * T + sa];
        ob[k] = b[k * T + sb];
    }
    tm.xpar ^= 1;
}

// sums over all threads of the team; every thread gets bitwise the same result
template <int W>
__device__ inline void team_sum2(Team<W> &tm, double &a, double &b)
{
    a = wave_sum(a);
    b = wave_sum(b);
    if (W == 1) return;
    double *r = tm.rbuf + tm.rpar * (2 * W);
    if (tm.lane == 0) {
        r[tm.wave] = a;
        r[W + tm.wave] = b;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        sa += r[w];
        sb += r[W + w];
    }
    a = sa;
    b = sb;
    tm.rpar ^= 1;
}
template <int W>
__device__ inline double team_sum(Team<W> &tm, double a)
{
    double b = 0.0;
    if (W == 1) return wave_sum(a);
    team_sum2<W>(tm, a, b);
    return a;
}

__device__ inline double2 d2(double x, double y) { return double2{x, y}; }

// ---- RHS ---------------------------------------------------------------------
// y = [N0,N1,P0,P1,E0,E1(,Nt0,Nt1)], E on the left face of the node; node i = 2*tid + r.
template <int W, bool TRAPS>
__device__ inline void team_rhs(Team<W> &tm, const Par &p, const double *y, double *f, int L)
{
    constexpr int T = Team<W>::T;
    const int tid = tm.tid;
    double2 o1[1], in1[1] = {d2(y[1], y[3])};
    xget<W, 1>(tm, in1, (tid + T - 1) & (T - 1), o1);  // N, P of the node left of this thread's first node
    double Jn[3], Jp[3];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = 2 * tid + r;
        const double N = y[r], P = y[2 + r], E = y[4 + r];
        const double Nm = (r == 0) ? o1[0].x : y[0];
        const double Pm = (r == 0) ? o1[0].y : y[2];
        double jn = p.mun * ((Nm + N) / 2 * E) + p.DnT * ((N - Nm) * p.idx);  // forward_solver.py:356-359
        double jp = p.mup * ((Pm + P) / 2 * E) - p.DpT * ((P - Pm) * p.idx);
        if (i == 0) {  // :346-351
            const double Sft = p.Sf * (N * P - p.n0p0) * frcp(N + P);
            jn = Sft;
            jp = -Sft;
        }
        if (i >= L) { jn = 0.0; jp = 0.0; }
        Jn[r] = jn;
        Jp[r] = jp;
    }
    double2 o2[1], in2[1] = {d2(Jn[0], Jp[0])};
    xget<W, 1>(tm, in2, (tid + 1) & (T - 1), o2);  // fluxes on the right face of this thread's last node
    Jn[2] = o2[0].x;
    Jp[2] = o2[0].y;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = 2 * tid + r;
        const double N = y[r], P = y[2 + r];
        const double NP = N * P - p.n0p0;
        double jnr = Jn[r + 1], jpr = Jp[r + 1];
        if (i == L - 1) {  // :347,:350,:352
            const double Sbt = p.Sb * NP * frcp(N + P);
            jnr = -Sbt;
            jpr = Sbt;
        }
        const double recomb = ((p.Cn * N + p.Cp * P) + p.ks + frcp((p.tn * P) + (p.tp * N))) * NP;  // :366
        const bool real = i < L;
        double fn = (jnr - Jn[r]) * p.idx - recomb;      // :369
        const double fp = -(jpr - Jp[r]) * p.idx - recomb;  // :370
        if (TRAPS) {  // :410-416
            const double Nt = y[6 + r];
            const double trap = p.kC * N * (p.Nt - Nt), detrap = Nt / p.tE;
            fn = fn + detrap - trap;
            f[6 + r] = real ? (trap - detrap) : 0.0;
        }
        f[r] = real ? fn : 0.0;
        f[2 + r] = real ? fp : 0.0;
        f[4 + r] = (real && i > 0) ? (-(Jn[r] + Jp[r]) * p.Lam) : 0.0;  // :363 (edge 0: exactly 0)
    }
}

// ---- Newton matrix factors ----------------------------------------------------
template <int W, bool TRAPS>
struct TeamFactors {
    double w[2][2], eco[2][4], minv[2], cd;
    double tq[2], tca[2];  // traps: 1/(1+c b_i), c a_i   (dNt_i = tq*(bNt + tca*dN_i))
    double tcb[2];         // traps: c b_i * tq          (rN_i += tcb * bNt_i)
    M2 Bfin;
    double2 *blk;  // [NBLK][2][T] rows of 16 B: elimination blocks of this thread at [.][.][tid]
    enum { BINV_E = 0, BIA_E, BIC_E, CR_AL, CR_GA, PAL0 = 5 };
    __device__ inline void put(int b, int tid, const M2 &m) const
    {
        constexpr int T = Team<W>::T;
        blk[(b * 2 + 0) * T + tid] = double2{m.a, m.b};
        blk[(b * 2 + 1) * T + tid] = double2{m.c, m.d};
    }
    __device__ inline M2 get(int b, int tid) const
    {
        constexpr int T = Team<W>::T;
        const double2 u = blk[(b * 2 + 0) * T + tid], v = blk[(b * 2 + 1) * T + tid];
        return M2{u.x, u.y, v.x, v.y};
    }
};
template <int W>
constexpr int team_blk_bytes() { return Team<W>::NBLK * 2 * Team<W>::T * 16; }

__device__ inline void m2rows(const M2 &m, double2 &r0, double2 &r1) { r0 = d2(m.a, m.b); r1 = d2(m.c, m.d); }
__device__ inline M2 rows2m(const double2 &r0, const double2 &r1) { return M2{r0.x, r0.y, r1.x, r1.y}; }

template <int W, bool TRAPS>
__device__ inline void team_build(Team<W> &tm, const Par &p, const double *y, double c, int L, TeamFactors<W, TRAPS> &F)
{
    constexpr int T = Team<W>::T;
    const int tid = tm.tid;
    const double cL = c * p.Lam, cd = c * p.idx;
    const double dnx = p.DnT * p.idx, dpx = p.DpT * p.idx;
    F.cd = cd;
    double2 o1[1], in1[1] = {d2(y[1], y[3])};
    xget<W, 1>(tm, in1, (tid + T - 1) & (T - 1), o1);
    M2 Lm[2], Rm[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = 2 * tid + r;
        const double N = y[r], P = y[2 + r], E = y[4 + r];
        const double Nm = (r == 0) ? o1[0].x : y[0];
        const double Pm = (r == 0) ? o1[0].y : y[2];
        const double an = 0.5 * p.mun * E, ap = 0.5 * p.mup * E;
        const double jnm = an - dnx, jnp = an + dnx, jpm = ap + dpx, jpp = ap - dpx;
        const double jne = p.mun * 0.5 * (Nm + N), jpe = p.mup * 0.5 * (Pm + P);
        const double mi = frcp(1.0 + cL * (jne + jpe));
        const double g = jne * cL * mi, gp = jpe * cL * mi;
        M2 lm{(1.0 - g) * jnm, -g * jpm, -gp * jnm, (1.0 - gp) * jpm};
        M2 rm{(1.0 - g) * jnp, -g * jpp, -gp * jnp, (1.0 - gp) * jpp};
        double w0 = jne * mi, w1 = jpe * mi, mv = mi;
        double e0 = cL * mi * jnm, e1 = cL * mi * jnp, e2 = cL * mi * jpm, e3 = cL * mi * jpp;
        if (i == 0) {  // front surface: J_0 = (S0, -S0), no field
            const double NP = N * P - p.n0p0, s = N + P, is2 = frcp(s * s);
            const double s0n = p.Sf * (P * s - NP) * is2, s0p = p.Sf * (N * s - NP) * is2;
            lm = mzero();
            rm = M2{s0n, s0p, -s0n, -s0p};
            w0 = w1 = e0 = e1 = e2 = e3 = 0.0;
            mv = 1.0;
        }
        if (i >= L) {
            lm = mzero();
            rm = mzero();
            w0 = w1 = e0 = e1 = e2 = e3 = 0.0;
            mv = 1.0;
        }
        Lm[r] = lm;
        Rm[r] = rm;
        F.w[r][0] = w0; F.w[r][1] = w1;
        F.eco[r][0] = e0; F.eco[r][1] = e1; F.eco[r][2] = e2; F.eco[r][3] = e3;
        F.minv[r] = mv;
    }
    // edge matrices of the next thread's first edge
    double2 nin[4], nout[4];
    m2rows(Lm[0], nin[0], nin[1]);
    m2rows(Rm[0], nin[2], nin[3]);
    xget<W, 4>(tm, nin, (tid + 1) & (T - 1), nout);
    const M2 LmNext = rows2m(nout[0], nout[1]), RmNext = rows2m(nout[2], nout[3]);
    M2 A[2], B[2], C[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int i = 2 * tid + r;
        const double N = y[r], P = y[2 + r];
        M2 lmn = (r == 1) ? LmNext : Lm[1];
        M2 rmn = (r == 1) ? RmNext : Rm[1];
        const double NP = N * P - p.n0p0;
        if (i == L - 1) {  // back surface: J_L = (-SL, SL)
            const double s = N + P, is2 = frcp(s * s);
            const double sLn = p.Sb * (P * s - NP) * is2, sLp = p.Sb * (N * s - NP) * is2;
            lmn = M2{-sLn, -sLp, sLn, sLp};
            rmn = mzero();
        }
        const double den = p.tn * P + p.tp * N, iden = frcp(den);
        const double kk = (p.Cn * N + p.Cp * P) + p.ks + iden;
        const double RN = (p.Cn - p.tp * iden * iden) * NP + kk * P;
        const double RP = (p.Cp - p.tn * iden * iden) * NP + kk * N;
        M2 a{cd * Lm[r].a, cd * Lm[r].b, -cd * Lm[r].c, -cd * Lm[r].d};
        M2 cc{-cd * rmn.a, -cd * rmn.b, cd * rmn.c, cd * rmn.d};
        M2 b{1.0 + cd * (Rm[r].a - lmn.a) + c * RN, cd * (Rm[r].b - lmn.b) + c * RP,
             -cd * (Rm[r].c - lmn.c) + c * RN, 1.0 - cd * (Rm[r].d - lmn.d) + c * RP};
        if (TRAPS) {
            // dN row: +c*a dN - c*b dNt ; dNt row: (1 + c b) dNt - c a dN = bNt   (a, b below)
            const double Nt = y[6 + r];
            const double ta = p.kC * (p.Nt - Nt), tb = p.kC * N + 1.0 / p.tE;
            const double q = frcp(1.0 + c * tb);
            F.tq[r] = q;
            F.tca[r] = c * ta;
            F.tcb[r] = c * tb * q;
            b.a += c * ta * q;
        }
        if (i >= L) {
            a = mzero();
            cc = mzero();
            b = M2{1.0, 0.0, 0.0, 1.0};
            if (TRAPS) { F.tq[r] = 1.0; F.tca[r] = 0.0; F.tcb[r] = 0.0; }
        }
        A[r] = a;
        B[r] = b;
        C[r] = cc;
    }
    // in-thread cyclic reduction: row e = 2*tid eliminated, row k = 2*tid+1 kept
    using TF = TeamFactors<W, TRAPS>;
    const M2 Binv_e = minv(B[0]);
    F.put(TF::BINV_E, tid, Binv_e);
    F.put(TF::BIA_E, tid, mmul(Binv_e, A[0]));
    F.put(TF::BIC_E, tid, mmul(Binv_e, C[0]));
    double2 uin[6], uo[6];
    m2rows(Binv_e, uin[0], uin[1]);
    m2rows(A[0], uin[2], uin[3]);
    m2rows(C[0], uin[4], uin[5]);
    xget<W, 6>(tm, uin, (tid + 1) & (T - 1), uo);
    const M2 Binv_up = rows2m(uo[0], uo[1]), A_up = rows2m(uo[2], uo[3]), C_up = rows2m(uo[4], uo[5]);
    const M2 cr_al = mmul(A[1], Binv_e);
    const M2 cr_ga = mmul(C[1], Binv_up);
    F.put(TF::CR_AL, tid, cr_al);
    F.put(TF::CR_GA, tid, cr_ga);
    M2 Ak = mneg(mmul(cr_al, A[0]));
    M2 Ck = mneg(mmul(cr_ga, C_up));
    M2 Bk = msub(msub(B[1], mmul(cr_al, C[0])), mmul(cr_ga, A_up));
    // parallel cyclic reduction over the T kept rows
#pragma unroll
    for (int lv = 0; lv < Team<W>::LEVELS; ++lv) {
        const int s = 1 << lv;
        const M2 Bi = minv(Bk);
        double2 pin[6], pm[6], pp[6];
        m2rows(Bi, pin[0], pin[1]);
        m2rows(Ak, pin[2], pin[3]);
        m2rows(Ck, pin[4], pin[5]);
        xget2<W, 6>(tm, pin, (tid - s) & (T - 1), (tid + s) & (T - 1), pm, pp);
        const M2 Bim = rows2m(pm[0], pm[1]), Akm = rows2m(pm[2], pm[3]), Ckm = rows2m(pm[4], pm[5]);
        const M2 Bip = rows2m(pp[0], pp[1]), Akp = rows2m(pp[2], pp[3]), Ckp = rows2m(pp[4], pp[5]);
        const M2 al = mmul(Ak, Bim), ga = mmul(Ck, Bip);
        F.put(TF::PAL0 + 2 * lv, tid, al);
        F.put(TF::PAL0 + 2 * lv + 1, tid, ga);
        Bk = msub(msub(Bk, mmul(al, Ckm)), mmul(ga, Akp));
        Ak = mneg(mmul(al, Akm));
        Ck = mneg(mmul(ga, Ckp));
    }
    F.Bfin = minv(Bk);
}

// x = M^{-1} b, layout as y
template <int W, bool TRAPS>
__device__ inline void team_apply(Team<W> &tm, const TeamFactors<W, TRAPS> &F, const double *b, double *x)
{
    constexpr int T = Team<W>::T;
    using TF = TeamFactors<W, TRAPS>;
    const int tid = tm.tid;
    const int tnext = (tid + 1) & (T - 1), tprev = (tid + T - 1) & (T - 1);
    double bn[2] = {b[0], b[1]};
    if (TRAPS) {
        bn[0] += F.tcb[0] * b[6];
        bn[1] += F.tcb[1] * b[7];
    }
    V2 wb[3];
    wb[0] = V2{F.w[0][0] * b[4], F.w[0][1] * b[4]};
    wb[1] = V2{F.w[1][0] * b[5], F.w[1][1] * b[5]};
    double2 i1[1] = {d2(wb[0].x, wb[0].y)}, o1[1];
    xget<W, 1>(tm, i1, tnext, o1);
    wb[2] = V2{o1[0].x, o1[0].y};
    V2 r[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
        r[q] = V2{bn[q] + F.cd * (wb[q + 1].x - wb[q].x), b[2 + q] - F.cd * (wb[q + 1].y - wb[q].y)};
    double2 i2[1] = {d2(r[0].x, r[0].y)}, o2[1];
    xget<W, 1>(tm, i2, tnext, o2);
    const V2 re_up{o2[0].x, o2[0].y};
    const V2 t1 = mvec(F.get(TF::CR_AL, tid), r[0]), t2 = mvec(F.get(TF::CR_GA, tid), re_up);
    V2 rk{r[1].x - t1.x - t2.x, r[1].y - t1.y - t2.y};
#pragma unroll
    for (int lv = 0; lv < Team<W>::LEVELS; ++lv) {
        const int s = 1 << lv;
        double2 pi[1] = {d2(rk.x, rk.y)}, pm[1], pp[1];
        xget2<W, 1>(tm, pi, (tid - s) & (T - 1), (tid + s) & (T - 1), pm, pp);
        const V2 u1 = mvec(F.get(TF::PAL0 + 2 * lv, tid), V2{pm[0].x, pm[0].y});
        const V2 u2 = mvec(F.get(TF::PAL0 + 2 * lv + 1, tid), V2{pp[0].x, pp[0].y});
        rk = V2{rk.x - u1.x - u2.x, rk.y - u1.y - u2.y};
    }
    const V2 xk = mvec(F.Bfin, rk);
    double2 i3[1] = {d2(xk.x, xk.y)}, o3[1];
    xget<W, 1>(tm, i3, tprev, o3);
    const V2 xk_prev{o3[0].x, o3[0].y};
    const V2 s0 = mvec(F.get(TF::BINV_E, tid), r[0]), s1 = mvec(F.get(TF::BIA_E, tid), xk_prev),
             s2 = mvec(F.get(TF::BIC_E, tid), xk);
    const V2 xe{s0.x - s1.x - s2.x, s0.y - s1.y - s2.y};
    x[0] = xe.x; x[1] = xk.x; x[2] = xe.y; x[3] = xk.y;
    x[4] = b[4] * F.minv[0] - (F.eco[0][0] * xk_prev.x + F.eco[0][1] * xe.x + F.eco[0][2] * xk_prev.y + F.eco[0][3] * xe.y);
    x[5] = b[5] * F.minv[1] - (F.eco[1][0] * xe.x + F.eco[1][1] * xk.x + F.eco[1][2] * xe.y + F.eco[1][3] * xk.y);
    if (TRAPS) {
        x[6] = F.tq[0] * (b[6] + F.tca[0] * x[0]);
        x[7] = F.tq[1] * (b[7] + F.tca[1] * x[1]);
    }
}

}  // namespace mtrpl

#endif  // MTRPL_SOLVER_BLOCK_HPP
