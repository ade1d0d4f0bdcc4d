// backend_wave.hpp -- one 64-lane wavefront integrates one forward simulation: lane l owns the two
// grid nodes 2l, 2l+1 (N_i, P_i and the edge field E_i on the LEFT face of each), so nx <= 128.  The
// whole solver state lives on the CU (Nordsieck history and Newton vectors in this wave's VGPRs/AGPRs,
// Newton matrix factors in its LDS slab); HBM is touched for the per-solve descriptor, the output
// sample times and the nt output samples only.
//
// Linear algebra of I - c*J (block tridiagonal, 2x2 blocks in (dN_i, dP_i) after the per-edge
// elimination of the field, model_physics.hpp): one in-lane cyclic-reduction level followed by 6
// parallel-cyclic-reduction levels over the 64 lanes (nearest-neighbour exchange by DPP wave shifts,
// stride 32 by v_permlane32_swap, the strides between by ds_bpermute; one wave per workgroup, so no
// barriers); all 17 elimination blocks per lane are kept in this wave's LDS slab as 16-byte rows, so each
// further Newton iteration is a cheap re-solve.
//
// Kernels (bottom of the file): solve_kernel_wave1 / solve_kernel_wave2 -- the default: fp64 factorisation,
// blocks stored as fp32 (20 KB of LDS per wave), re-solve in packed fp32 on an fp64 right-hand side; built
// with the whole register file (one solve per SIMD: the shortest solve) and on a 256-register budget (two
// solves per SIMD: the second wave fills the issue slots the first leaves waiting), same arithmetic, same
// results.  solve_kernel_wave -- fp64 blocks (34 KB) and an fp64 re-solve: the exact linear algebra.
#pragma once
#include "solve_core.hpp"

namespace mtrpl {

constexpr int kFactorLdsBytes = 17 * 2 * 64 * 16;  // 34816 B per wave
constexpr int kFactorLdsBytesF32 = 20 * 64 * 16;     // 20480 B per wave with blocks stored as 4 floats (17 + Bfin + 2 rows of eco):
                                                     // 8 waves per CU fill the 160 KB exactly

// F32BLK: the elimination blocks are STORED rounded to fp32 (one 16-byte row per block and lane) and the
// re-solve exchanges and multiplies in packed fp32 (apply_f32); factorisation, residual, state and the running
// right-hand side of the re-solve stay fp64.  The blocks only define the approximate inverse the simplified
// Newton iteration applies to an fp64 residual: the iteration converges to the same corrector solution
// (measured: same step counts, +0.7 % Newton iterations, per-decade accuracy table unchanged).
template <bool F32BLK, bool ROWV = false>
struct WaveBackendT {
    using real = double;
    static constexpr int R = 2, V = 6;
    static constexpr bool TRAPS = false;
    static constexpr bool kFieldInNorms = false;  // the field components in the controller's norms (solve_core.hpp)
    static constexpr bool kParksSamples = false;  // out_sample() parks, out_flush() emits
    static constexpr bool kOrderRowInVgpr = ROWV;  // the order table's row by v_readlane (solve_core.hpp: OrderRow): the one-per-SIMD kernel
    static constexpr bool kBlocksSamples = true;  // the samples leave in blocks of 64: integrate() calls out_finish() at the end
    static constexpr bool kScalarPolicy = false;  // policy constants of a solve held in scalar registers (solve_core.hpp)

    Par p;
    int L, lane;
    Emitter em;
    // Output samples wait in a register -- lane j holds sample kb + j -- and leave 64 at a time, one coalesced 512-byte
    // store per block (and the partial block at the end): a store per sample kept a vector-memory operation in flight in
    // front of every `s_waitcnt vmcnt(0)` of the sample loop, which waited for its acknowledgement -- 3 % of a solve on an
    // otherwise idle SIMD (measured by leaving the store out).
    double vout;
    int kb, nb;  // first sample of the block being filled, samples in it
    // factors
    double w[2][2];    // dJ/dbE weights per own edge
    double eco[2][4];  // dE_i = bE_i*minv - eco.(dN_{i-1}, dN_i, dP_{i-1}, dP_i)   (fp64 blocks only; else LDS rows ECO0/1)
    double mvv[2];
    double cd;         // c/dx
    M2 Bfin;           // inverse of the last diagonal block of the PCR                  (fp64 blocks only; else LDS row BFIN)
    // The 17 elimination blocks (in-lane CR: Binv_e, BiA_e, BiC_e, cr_al, cr_ga; PCR: al[6], ga[6]) live
    // in this wave's LDS slab: [17 blocks][2 halves][64 lanes] x 16 B, i.e. every access is a
    // conflict-free ds_read/write_b128 (lane stride 16 B).
    void *lds;
    enum { BINV_E = 0, BIA_E, BIC_E, CR_AL, CR_GA, PAL0, PGA0 = PAL0 + 6, NBLK = PGA0 + 6,
           BFIN = NBLK, ECO0, ECO1 };  // fp32-stored blocks only: the last diagonal inverse and the field back-substitution rows

    int lds_off;  // lane * 16: byte offset of this lane's 16-byte row inside a block row
    __device__ inline WaveBackendT(int lane_, void *lds_) : lane(lane_), lds(lds_), lds_off(lane_ * 16) {}
    __device__ inline void set_problem(const ParT<real> &p_, int L_) { p = p_; L = L_; }
    __device__ inline int node(int r) const { return lane * R + r; }
    __device__ inline bool leader() const { return lane == 0; }

    __device__ inline void put(int blk, const M2 &m) const
    {
        if (F32BLK) {
            // stored by columns (a, c | b, d): a block times a vector is then two packed fp32 FMAs
            static_cast<float4 *>(lds)[blk * 64 + lane] = float4{(float)m.a, (float)m.c, (float)m.b, (float)m.d};
        } else {
            static_cast<double2 *>(lds)[(blk * 2 + 0) * 64 + lane] = double2{m.a, m.b};
            static_cast<double2 *>(lds)[(blk * 2 + 1) * 64 + lane] = double2{m.c, m.d};
        }
    }
    __device__ inline M2 get(int blk)
    {
        // Opaque row offset: keeps the block loads inside the Newton loop (16-B rows re-read from LDS
        // every iteration) instead of letting the compiler hoist 64+ VGPRs of loop-invariant blocks and
        // push the history out to AGPRs (measured: AGPRs 256 -> 116, 12.8 -> 12.3 ms per 1024-solve batch).
        // The offset register itself is declared modified, so hiding it costs no instruction.
        asm volatile("" : "+v"(lds_off));
        const char *base = static_cast<const char *>(lds) + lds_off;
        if (F32BLK) {
            const float4 q = *reinterpret_cast<const float4 *>(base + blk * 1024);
            return M2{(double)q.x, (double)q.z, (double)q.y, (double)q.w};
        }
        const double2 u = *reinterpret_cast<const double2 *>(base + (blk * 2 + 0) * 1024);
        const double2 v = *reinterpret_cast<const double2 *>(base + (blk * 2 + 1) * 1024);
        return M2{u.x, u.y, v.x, v.y};
    }

    // ---- whole-grid primitives: everything is inside one wave ----
    template <int NS>
    __device__ inline void reduce_f(float (&s)[NS]) { wave_sums_f<NS>(s, lane); }
    __device__ inline double scan_excl(double run) const { return wave_scan_incl(run, lane) - run; }
    __device__ inline double max_all(double v) const { return wave_max(v); }
    __device__ inline void out_begin(double *o, double min_y, double scale, double deep, double tail, double far, double post)
    {
        em.begin(o, min_y, scale, deep, tail, far, post);
        vout = 0.0;
        kb = 1;
        nb = 0;
    }
    __device__ inline bool post() const { return em.post(); }
    __device__ inline bool tail() const { return em.tail(); }
    __device__ inline bool far() const { return em.far(); }
    __device__ inline bool deep() const { return em.deep(); }
    __device__ inline void out_first(double partial) { em.first(uniform(wave_sum(partial)), p.dx, lane == 0); }
    __device__ inline void out_sample(int k, double partial)
    {
        const double v = em.value(uniform(wave_sum(partial)), p.dx);
        vout = (lane == nb) ? v : vout;  // (samples arrive in order: k = kb + nb)
        if (++nb == 64) out_finish();
    }
    __device__ inline void out_finish()
    {
        if (lane < nb) em.o[kb + lane] = vout;
        kb += nb;
        nb = 0;
    }
    __device__ inline void out_flush() {}

    // ---- RHS: forward_solver.py:332-372 (dydt_numba) ----
    // Padded nodes (i >= L) carry N=P=1, E=0 for ever (integrate() gives them a zero step): an edge between two
    // of them has exactly zero flux, the edge between the last real node and the first padded one is the back
    // surface for the real node and garbage for the padded one, whose f is never looked at.  So nothing is masked
    // here but the two surfaces.
    __device__ inline void rhs(const double (&y)[V], double (&f)[V]) const
    {
        double Jn[R + 1], Jp[R + 1];
        const double Nl = from_prev(y[R - 1]);
        const double Pl = from_prev(y[2 * R - 1]);
        // The two surface terms S (N P - n0 p0) / (N + P) (:346-347) live in two different lanes -- the front's in lane 0
        // (node 0), the back's in the lane of node L - 1, slot (L - 1) & 1 -- so ONE evaluation per lane serves both: each
        // lane forms the term of the node it might own a surface of.  (Only a two-node grid has both in one lane.)
        const bool back1 = ((L - 1) & 1) != 0;  // wave-uniform: the back surface sits in slot 1 of its lane
        const bool s1 = back1 && lane != 0;
        const double Ns = s1 ? y[1] : y[0], Ps = s1 ? y[R + 1] : y[R];
        const double St = ((lane == 0) ? p.Sf : p.Sb) * (Ns * Ps - p.n0p0) * frcp1(Ns + Ps);
        double Sbt = St;
        if (__builtin_expect(L == 2, 0)) Sbt = p.Sb * (y[1] * y[R + 1] - p.n0p0) * frcp1(y[1] + y[R + 1]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            const double N = y[r], P = y[R + r], E = y[2 * R + r];
            const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
            const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
            double jn, jp;
            edge_flux(p, Nm, N, Pm, P, E, jn, jp);  // :356-359
            if (r == 0) {  // :346-351 (lane 0 only)
                jn = (i == 0) ? St : jn;
                jp = (i == 0) ? -St : jp;
            }
            Jn[r] = jn;
            Jp[r] = jp;
        }
        Jn[R] = from_next(Jn[0]);
        Jp[R] = from_next(Jp[0]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            const double N = y[r], P = y[R + r];
            const double NP = N * P - p.n0p0;
            double jnr = Jn[r + 1], jpr = Jp[r + 1];
            if (i == L - 1) {  // :347,:350,:352
                jnr = -Sbt;
                jpr = Sbt;
            }
            const double recomb = bulk_recomb(p, N, P, NP);  // :366
            f[r] = (jnr - Jn[r]) * p.idx - recomb;       // :369
            f[R + r] = -(jpr - Jp[r]) * p.idx - recomb;  // :370
            f[2 * R + r] = -(Jn[r] + Jp[r]) * p.Lam;     // :363 (edge 0: Sft - Sft, exactly 0)
        }
    }

    // ---- factorisation of M = I - c*J(yJ) ----
    __device__ inline void build(const double (&y_in)[V], double c)
    {
        double y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = pinned(y_in[v]);  // keep the Jacobian inside the branch it is called from
        M2 Lm[R], Rm[R], A[R], B[R], C[R];
        const double cL = c * p.Lam;
        const double dnx = p.dnx, dpx = p.dpx;
        const double Nl = from_prev(y[R - 1]);
        const double Pl = from_prev(y[2 * R - 1]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            const double N = y[r], P = y[R + r], E = y[2 * R + r];
            const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
            const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
            EdgeJac<double> ej = edge_jac_interior(p, cL, dnx, dpx, Nm, N, Pm, P, E);
            if (i == 0) ej = edge_jac_front(p, N, P);
            if (i >= L) ej = edge_jac_none<double>();
            Lm[r] = ej.lm;
            Rm[r] = ej.rm;
            w[r][0] = ej.w0; w[r][1] = ej.w1;
            if (F32BLK) {  // (e0, e2 | e1, e3): the row times (dN_a, dP_a), (dN_b, dP_b) is one packed multiply and one packed FMA
                static_cast<float4 *>(lds)[(ECO0 + r) * 64 + lane] = float4{(float)ej.e0, (float)ej.e2, (float)ej.e1, (float)ej.e3};
            } else {
                eco[r][0] = ej.e0; eco[r][1] = ej.e1; eco[r][2] = ej.e2; eco[r][3] = ej.e3;
            }
            mvv[r] = ej.mv;
        }
        cd = c * p.idx;
        // lane 63 wraps to lane 0's edge 0: harmless, its last node is the back surface or padding
        const M2 LmNext = mnext(Lm[0]);
        const M2 RmNext = mnext(Rm[0]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = lane * R + r;
            const double N = y[r], P = y[R + r];
            M2 lmn = (r == R - 1) ? LmNext : Lm[(r == R - 1) ? r : r + 1];
            M2 rmn = (r == R - 1) ? RmNext : Rm[(r == R - 1) ? r : r + 1];
            if (i == L - 1) {  // back surface: J_L = (-SL, SL)
                double sLn, sLp;
                surface_jac(p, p.Sb, N, P, sLn, sLp);
                lmn = M2{-sLn, -sLp, sLn, sLp};
                rmn = mzero<double>();
            }
            double RN, RP;
            bulk_jac(p, N, P, RN, RP);
            node_row(c, cd, Lm[r], Rm[r], lmn, rmn, RN, RP, A[r], B[r], C[r]);
            if (i >= L) {
                A[r] = mzero<double>();
                C[r] = mzero<double>();
                B[r] = mident<double>();
            }
        }
        // in-lane cyclic reduction: row e = 2*lane (r=0) eliminated, row k = 2*lane+1 (r=1) kept
        const M2 Binv_e = minv(B[0]);
        put(BINV_E, Binv_e);
        put(BIA_E, mmul(Binv_e, A[0]));
        put(BIC_E, mmul(Binv_e, C[0]));
        const M2 Binv_up = mnext(Binv_e);
        const M2 A_up = mnext(A[0]);
        const M2 C_up = mnext(C[0]);
        const M2 cr_al = mmul(A[1], Binv_e);
        const M2 cr_ga = mmul(C[1], Binv_up);
        put(CR_AL, cr_al);
        put(CR_GA, cr_ga);
        M2 Ak = mneg(mmul(cr_al, A[0]));
        M2 Ck = mneg(mmul(cr_ga, C_up));
        M2 Bk = msub(msub(B[1], mmul(cr_al, C[0])), mmul(cr_ga, A_up));
        // parallel cyclic reduction over the 64 kept rows
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
            const int s = 1 << lv;
            const int lm_ = (lane - s) & 63, lp_ = (lane + s) & 63;
            const M2 Bi = minv(Bk);
            M2 Bim, Bip, Akm, Akp, Ckm, Ckp;
            if (lv == 0) {
                Bim = mprev(Bi); Bip = mnext(Bi); Akm = mprev(Ak); Akp = mnext(Ak); Ckm = mprev(Ck); Ckp = mnext(Ck);
            } else if (lv == 5) {
                mswap32(Bi, Bim, Bip); mswap32(Ak, Akm, Akp); mswap32(Ck, Ckm, Ckp);
            } else {
                Bim = mshf(Bi, lm_); Bip = mshf(Bi, lp_); Akm = mshf(Ak, lm_); Akp = mshf(Ak, lp_);
                Ckm = mshf(Ck, lm_); Ckp = mshf(Ck, lp_);
            }
            const M2 al = mmul(Ak, Bim), ga = mmul(Ck, Bip);
            put(PAL0 + lv, al);
            put(PGA0 + lv, ga);
            Bk = msub(msub(Bk, mmul(al, Ckm)), mmul(ga, Akp));
            Ak = mneg(mmul(al, Akm));
            Ck = mneg(mmul(ga, Ckp));
        }
        if (F32BLK)
            put(BFIN, minv(Bk));
        else
            Bfin = minv(Bk);
    }

    // ---- the re-solve with fp32-stored blocks: exchanged vectors and block products in packed fp32 ----
    // Everything the blocks touch is rounded to fp32 anyway, so the vectors travel as fp32 (half the dwords per
    // exchange) and a block times a vector is two v_pk_fma_f32; the running right-hand side, the E-elimination
    // and the field back-substitution stay fp64.  Still an approximate inverse applied to an fp64 residual.
    typedef float v2f __attribute__((ext_vector_type(2)));
    struct PB { v2f ac, bd; };
    __device__ inline PB get_pb(int blk)
    {
        asm volatile("" : "+v"(lds_off));
        const float4 q = *reinterpret_cast<const float4 *>(static_cast<const char *>(lds) + lds_off + blk * 1024);
        return PB{v2f{q.x, q.y}, v2f{q.z, q.w}};
    }
    static __device__ inline v2f pb_acc(const PB &B, v2f r, v2f acc) { return B.bd * v2f(r.y) + (B.ac * v2f(r.x) + acc); }
    static __device__ inline v2f narrow(const V2 &v) { return v2f{(float)v.x, (float)v.y}; }

    __device__ inline void apply_f32(const double (&b)[V], double (&x)[V])
    {
        const PB cr_al = get_pb(CR_AL), cr_ga = get_pb(CR_GA);
        PB al = get_pb(PAL0), ga = get_pb(PGA0);
        V2 wb[3];
        wb[0] = V2{w[0][0] * b[4], w[0][1] * b[4]};
        wb[1] = V2{w[1][0] * b[5], w[1][1] * b[5]};
        wb[2] = vnext(wb[0]);
        V2 r[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const V2 nxt = wb[q + 1];
            r[q] = V2{b[q] + cd * (nxt.x - wb[q].x), b[2 + q] - cd * (nxt.y - wb[q].y)};
        }
        const v2f r0 = narrow(r[0]);
        const v2f r0_up{from_next(r0.x), from_next(r0.y)};
        const v2f t = pb_acc(cr_ga, r0_up, pb_acc(cr_al, r0, v2f(0.0f)));
        // the reduced right-hand side goes through the six PCR levels as fp32: every product it meets is fp32 already
        v2f rk32 = narrow(V2{r[1].x - (double)t.x, r[1].y - (double)t.y});
        PB b_inv = al, b_ia = al, b_ic = al, b_fin = al, e0 = al, e1 = al;
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
            const int s = 1 << lv;
            PB al_n = al, ga_n = ga;
            if (lv < 5) {
                al_n = get_pb(PAL0 + lv + 1);
                ga_n = get_pb(PGA0 + lv + 1);
            }
            if (lv == 4) {
                b_fin = get_pb(BFIN);
                b_inv = get_pb(BINV_E);
            } else if (lv == 5) {
                b_ia = get_pb(BIA_E);
                b_ic = get_pb(BIC_E);
                e0 = get_pb(ECO0);
                e1 = get_pb(ECO1);
            }
            const v2f rf = rk32;
            v2f rm, rp;
            if (lv == 0) {
                rm = v2f{from_prev(rf.x), from_prev(rf.y)};
                rp = v2f{from_next(rf.x), from_next(rf.y)};
            } else if (lv == 5) {
                float mx, px, my, py;
                swap32(rf.x, mx, px);
                swap32(rf.y, my, py);
                rm = v2f{mx, my};
                rp = v2f{px, py};
            } else {
                rm = v2f{shf(rf.x, (lane - s) & 63), shf(rf.y, (lane - s) & 63)};
                rp = v2f{shf(rf.x, (lane + s) & 63), shf(rf.y, (lane + s) & 63)};
            }
            rk32 = rk32 - (ga.bd * v2f(rp.y) + (ga.ac * v2f(rp.x) + (al.bd * v2f(rm.y) + al.ac * v2f(rm.x))));
            al = al_n;
            ga = ga_n;
        }
        const v2f xk32 = pb_acc(b_fin, rk32, v2f(0.0f));
        const v2f xkp32{from_prev(xk32.x), from_prev(xk32.y)};
        const v2f se = pb_acc(b_inv, r0, v2f(0.0f)) - pb_acc(b_ic, xk32, pb_acc(b_ia, xkp32, v2f(0.0f)));
        x[0] = (double)se.x; x[1] = (double)xk32.x; x[2] = (double)se.y; x[3] = (double)xk32.y;
        // field back-substitution: the corrections of the densities are fp32 values here, so are the products
        const v2f d0 = e0.ac * xkp32 + e0.bd * se, d1 = e1.ac * se + e1.bd * xk32;
        x[4] = b[4] * mvv[0] - (double)(d0.x + d0.y);
        x[5] = b[5] * mvv[1] - (double)(d1.x + d1.y);
    }

    // x = M^{-1} b; b, x in lane layout [N0,N1,P0,P1,E0,E1]
    __device__ inline void apply(const double (&b)[V], double (&x)[V])
    {
        if constexpr (F32BLK) {
            apply_f32(b, x);
        } else {
            apply_f64(b, x);
        }
    }
    __device__ inline void apply_f64(const double (&b)[V], double (&x)[V])
    {
        // E-elimination right-hand side
        V2 wb[3];
        wb[0] = V2{w[0][0] * b[4], w[0][1] * b[4]};
        wb[1] = V2{w[1][0] * b[5], w[1][1] * b[5]};
        wb[2] = vnext(wb[0]);  // lane 63 gets 0: its last node is the back surface or padding
        V2 r[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const V2 nxt = wb[q + 1];  // edge L and padded edges carry zero weights
            r[q] = V2{b[q] + cd * (nxt.x - wb[q].x), b[2 + q] - cd * (nxt.y - wb[q].y)};
        }
        // cyclic reduction of the right-hand side
        const V2 re_up = vnext(r[0]);
        const V2 t1 = mvec(get(CR_AL), r[0]), t2 = mvec(get(CR_GA), re_up);
        V2 rk{r[1].x - t1.x - t2.x, r[1].y - t1.y - t2.y};
        // The elimination blocks of level lv+1 are requested from LDS before the exchange of level lv is waited
        // for: their latency hides behind it instead of following it (one wave per SIMD: nothing else would hide
        // it).  One LDS round trip per level instead of two: 6.08 -> 5.74 ms per 1024-solve launch.
        M2 al = get(PAL0), ga = get(PGA0);
#pragma unroll
        for (int lv = 0; lv < 6; ++lv) {
            const int s = 1 << lv;
            M2 al_n = al, ga_n = ga;
            if (lv < 5) { al_n = get(PAL0 + lv + 1); ga_n = get(PGA0 + lv + 1); }
            V2 rm, rp;
            if (lv == 0) {          // nearest neighbours: DPP wave shifts
                rm = vprev(rk);
                rp = vnext(rk);
            } else if (lv == 5) {   // stride 32: v_permlane32_swap, no LDS round trip
                vswap32(rk, rm, rp);
            } else {
                rm = vshf(rk, (lane - s) & 63);
                rp = vshf(rk, (lane + s) & 63);
            }
            const V2 u1 = mvec(al, rm), u2 = mvec(ga, rp);
            rk = V2{rk.x - u1.x - u2.x, rk.y - u1.y - u2.y};
            al = al_n;
            ga = ga_n;
        }
        const V2 xk = mvec(Bfin, rk);
        const V2 xk_prev = vprev(xk);
        const V2 s0 = mvec(get(BINV_E), r[0]), s1 = mvec(get(BIA_E), xk_prev), s2 = mvec(get(BIC_E), xk);
        const V2 xe{s0.x - s1.x - s2.x, s0.y - s1.y - s2.y};
        x[0] = xe.x; x[1] = xk.x; x[2] = xe.y; x[3] = xk.y;
        // back-substitute the eliminated field unknowns
        x[4] = b[4] * mvv[0] - (eco[0][0] * xk_prev.x + eco[0][1] * xe.x + eco[0][2] * xk_prev.y + eco[0][3] * xe.y);
        x[5] = b[5] * mvv[1] - (eco[1][0] * xe.x + eco[1][1] * xk.x + eco[1][2] * xe.y + eco[1][3] * xk.y);
    }
};

using WaveBackend = WaveBackendT<false>;

// CT: the integrator options are the shipped ones and compiled in (the host checks) -- none of them sits in
// (spilled) scalar registers.
template <bool CT>
__global__ __launch_bounds__(64, 1) void solve_kernel_wave(const mtrpl_solve_desc *__restrict__ descs,
                                                            const double *__restrict__ ini,
                                                            const double *__restrict__ tsteps,
                                                            double *__restrict__ out, int32_t *__restrict__ status,
                                                            int32_t *__restrict__ stats, int64_t B, SolveTuning tune)
{
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if ((int64_t)blockIdx.x >= B) return;
    const int64_t sidx = blockIdx.x;
    __shared__ double2 fac_lds[kFactorLdsBytes / 16];
    WaveBackend bk(threadIdx.x & 63, fac_lds);
    if (CT)
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, DefaultTune{}, clk0_);
    else
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

// Two solves per SIMD: fp32-stored blocks (20 KB of LDS per wave, 8 waves per CU = all of its 160 KB) and a
// 256-register budget.
template <bool CT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void solve_kernel_wave2(
    const mtrpl_solve_desc *__restrict__ descs, const double *__restrict__ ini, const double *__restrict__ tsteps,
    double *__restrict__ out, int32_t *__restrict__ status, int32_t *__restrict__ stats, int64_t B, SolveTuning tune)
{
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if ((int64_t)blockIdx.x >= B) return;
    const int64_t sidx = blockIdx.x;
    __shared__ float4 fac_lds[kFactorLdsBytesF32 / 16];
    WaveBackendT<true> bk(threadIdx.x & 63, fac_lds);
    if (CT)
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, DefaultTune{}, clk0_);
    else
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

// The same backend with the whole register file: one solve per SIMD, the shortest solve (the packed-fp32 re-solve
// makes it faster than the fp64-block kernel).  Same arithmetic as solve_kernel_wave2: a solve gives the same
// result on either.
template <bool CT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1))) void solve_kernel_wave1(
    const mtrpl_solve_desc *__restrict__ descs, const double *__restrict__ ini, const double *__restrict__ tsteps,
    double *__restrict__ out, int32_t *__restrict__ status, int32_t *__restrict__ stats, int64_t B, SolveTuning tune)
{
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if ((int64_t)blockIdx.x >= B) return;
    const int64_t sidx = blockIdx.x;
    __shared__ float4 fac_lds[kFactorLdsBytesF32 / 16];
    WaveBackendT<true, true> bk(threadIdx.x & 63, fac_lds);
    if (CT)
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, DefaultTune{}, clk0_);
    else
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

}  // namespace mtrpl
