// backend_spike.hpp -- W wavefronts (one workgroup) integrate one forward simulation of the std model with
// nx <= 128*W (BASELINE.json configs[4]: nx = 512 -> W = 4; nx = 256 -> W = 2), each wave running the
// register / DPP linear algebra of backend_wave.hpp on ITS OWN 128 nodes, the waves coupled by a SPIKE interface
// solve (Polizzi & Sameh) instead of parallel cyclic reduction across the whole workgroup.
//
// Why.  The general team backend (backend_team.hpp) runs every exchange of its 8 PCR levels (W = 4) through LDS
// with an s_barrier each: 11 barriers per re-solve, 14 per Newton iteration -- measured 36 % of its wave cycles
// parked on s_waitcnt, VALU issuing in 34 % (profiles/r03_z_pmc_team4.json).  Here a Newton iteration has FOUR:
//
//   rhs      1   halo: the (N, P, E) of a wave's first node and the (N, P) of its last go to the neighbours
//   apply    2   halo of the E-elimination right-hand side, then the interface solve:
//                  x_k = g_k - V_k x_last(w-1) - W_k x_first(w+1),   g = D_w^-1 r (wave-local, no barrier),
//                V = D_w^-1 (e_first A_cut), W = D_w^-1 (e_last C_cut) the spikes of the cut couplings, kept per lane;
//                the 2(W-1) interface blocks solve a small system whose inverse was tabulated by the factorisation:
//                every wave publishes g_first / g_last (one barrier) and lanes 0-3 of each wave evaluate the four
//                numbers that wave needs as 4(W-1)-term dot products
//   norms    1   cross-wave sum of the controller's fp32 sums
//
// The reduced matrix (kept rows after the in-lane cyclic reduction: one 2x2 block row per thread, T = 64 W rows):
//   row k:  Ak x_{k-1} + Bk x_k + Ck x_{k+1} = r_k;   wave w owns rows 64w .. 64w+63; D_w = its rows with the
//   couplings across the wave boundary (Ak of its first row, Ck of its last) cut out.
// Interface unknowns, in this order: l_0, f_1, l_1, f_2, ... (l_j = x_last of wave j, f_j = x_first of wave j):
//   l_j     + V_last(j)    l_{j-1} + W_last(j)    f_{j+1} = g_last(j)
//   f_{j+1} + V_first(j+1) l_j     + W_first(j+1) f_{j+2} = g_first(j+1)          j = 0 .. W-2
// block tridiagonal in the pairs u_j = (l_j, f_{j+1}); its inverse is computed once per factorisation by every wave
// for itself (block Thomas on wave-uniform 2x2 blocks, lane m carrying the unit right-hand side e_m), and the four
// rows a wave needs -- x_last(w-1), x_first(w+1) -- are kept in LDS.
//
// Arithmetic as solve_kernel_wave1: fp64 factorisation, blocks stored rounded to fp32, re-solve in packed fp32 on an
// fp64 right-hand side (an approximate inverse applied to the fp64 residual of the simplified Newton iteration).
#pragma once
#include "backend_wave.hpp"

namespace mtrpl {

template <int W>
struct SpikeBackendT {
    using real = double;
    static constexpr int R = 2, V = 6, T = 64 * W;
    static constexpr bool TRAPS = false;
    static constexpr bool kFieldInNorms = false;  // the field components in the controller's norms (solve_core.hpp)
    static constexpr bool kOrderRowInVgpr = false;
    static constexpr bool kBlocksSamples = false;
    static constexpr bool kScalarPolicy = true;  // policy constants of a solve held in scalar registers (solve_core.hpp)
    static constexpr bool kParksSamples = true;  // out_sample() parks, out_flush() emits
    static constexpr int NI = W - 1;       // interfaces
    static constexpr int NIF = 4 * NI;     // scalar interface unknowns
    // This wave's block slab, 19 rows of 16 bytes per lane (with the workgroup's shared 4 KB: 80 KB at W = 4, two
    // workgroups fill a CU's 160 KB exactly):
    //   the in-lane cyclic reduction (5), al / ga of PCR levels 0-4 (10), ONE row for level 5 -- at stride 32 the lower
    //   half of the wave has no row below (al = 0) and the upper half none above (ga = 0) --, the last diagonal inverse,
    //   and the two spikes as V and Z = I + V + W (see build()).  The field back-substitution needs no row: its four
    //   coefficients per edge are c Lam mv (mu E / 2 -+ D/dx), rebuilt from qE = c Lam mv E (a register).  (The spikes
    //   as eight bf16 in one row: the weighted norms punish an absolute error of 4e-3 |V xl| in the components a
    //   Beer-Lambert profile keeps five decades below the surface -- Newton failed on every thin film.)
    enum { BINV_E = 0, BIA_E, BIC_E, CR_AL, CR_GA, PAL0, PGA0 = PAL0 + 5, L5ROW = PGA0 + 5, BFIN, VSPK, ZSPK, NROWS };
    static_assert(NROWS == 19, "19 rows per wave");
    static constexpr int kSlabBytes = NROWS * 64 * 16;  // per wave
    // shared by the workgroup: ping-pong exchange buffer [2][W][2][KX] doubles (what a wave's first / last lane
    // publish), the coefficient tables [W][4][NIF]
    static constexpr int KX = 12;
    static constexpr int kXchgDoubles = 2 * W * 2 * KX;
    static constexpr int kTabDoubles = W * 4 * NIF;
    static constexpr int PMAX = 16;  // parked output samples per flush
    static constexpr int kPendDoubles = 2 * PMAX * W;
    static constexpr int kSharedDoubles = kXchgDoubles + kTabDoubles + kPendDoubles;

    Par p;
    int L, lane, wave, tid;
    Emitter em;
    double w[2][2], mvv[2], cd;
    float kdn, kdp;  // c Lam D_n / dx, c Lam D_p / dx (wave-uniform)
    float qE[2];     // c Lam mv E of this lane's two edges (0 on edge 0 and on padded edges)
    void *lds;      // this wave's slab
    double *xb;     // exchange buffer
    double *tab;    // this wave's coefficient table [4][NIF]
    double *pend;   // parked partial sums of output samples [2][PMAX][W]
    int lds_off;
    int xpar, ppar, npend, k0pend;

    __device__ inline SpikeBackendT(int tid_, void *slabs, double *shared)
    {
        tid = tid_;
        lane = tid_ & 63;
        wave = uniform(tid_ >> 6);
        lds = static_cast<char *>(slabs) + wave * kSlabBytes;
        lds_off = lane * 16;
        xb = shared;
        tab = shared + kXchgDoubles + wave * 4 * NIF;
        pend = shared + kXchgDoubles + kTabDoubles;
        xpar = 0;
        ppar = 0;
        npend = 0;
        k0pend = 0;
    }
    __device__ inline void set_problem(const ParT<real> &p_, int L_) { p = p_; L = L_; }
    __device__ inline int node(int r) const { return tid * R + r; }
    __device__ inline bool leader() const { return tid == 0; }

    // ---- cross-wave exchange: the first lane of every wave publishes f[], the last lane l[]; after ONE barrier every
    // thread may read what any wave published.  Two buffers in turn: a wave that is one exchange ahead writes the
    // other buffer, and it cannot be two ahead (the barrier between).
    __device__ inline double *slot(int wv, int which) const { return xb + ((xpar * W + wv) * 2 + which) * KX; }
    template <int KF, int KL>
    __device__ inline void publish(const double (&f)[KF], const double (&l)[KL])
    {
        static_assert(KF <= KX && KL <= KX, "exchange slot too small");
        if (lane == 0) {
            double *q = slot(wave, 0);
#pragma unroll
            for (int k = 0; k < KF; ++k) q[k] = f[k];
        }
        if (lane == 63) {
            double *q = slot(wave, 1);
#pragma unroll
            for (int k = 0; k < KL; ++k) q[k] = l[k];
        }
        wg_barrier();
    }
    // LDS-only workgroup barrier: this wave's LDS writes are done (lgkmcnt), then s_barrier.  (__syncthreads() also
    // waits for vmcnt(0), i.e. for the global store of the last output sample to reach memory -- once per sample, on the
    // critical path of all four waves; nothing here communicates through global memory.)
    static __device__ inline void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    // what the previous wave's last lane / the next wave's first lane published (wave-uniform; the outermost waves
    // read their own slot: finite numbers that only ever meet zero couplings)
    __device__ inline const double *prev_last() const { return slot(wave > 0 ? wave - 1 : 0, 1); }
    __device__ inline const double *next_first() const { return slot(wave < W - 1 ? wave + 1 : W - 1, 0); }
    __device__ inline void done() { xpar ^= 1; }

    __device__ inline void put(int blk, const M2 &m) const
    {
        static_cast<float4 *>(lds)[blk * 64 + lane] = float4{(float)m.a, (float)m.c, (float)m.b, (float)m.d};
    }
    __device__ inline M2 getm(int blk)  // a stored block (fp32, by columns) as fp64
    {
        asm volatile("" : "+v"(lds_off));
        const float4 q = *reinterpret_cast<const float4 *>(static_cast<const char *>(lds) + lds_off + blk * 1024);
        return M2{(double)q.x, (double)q.z, (double)q.y, (double)q.w};
    }
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

    // ---- whole-grid primitives ----
    template <int NS>
    __device__ inline void reduce_f(float (&s)[NS])
    {
        static_assert(NS <= 4, "");
        wave_sums_f<NS>(s, lane);
        // the (wave-uniform) sums travel as ONE 16-byte row of floats per wave
        float4 q{s[0], NS > 1 ? s[1] : 0.0f, NS > 2 ? s[2] : 0.0f, NS > 3 ? s[3] : 0.0f};
        if (lane == 0) *reinterpret_cast<float4 *>(slot(wave, 0)) = q;
        wg_barrier();
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int wv = 0; wv < W; ++wv) {
            const float4 r = *reinterpret_cast<const float4 *>(slot(wv, 0));
            a[0] += r.x; a[1] += r.y; a[2] += r.z; a[3] += r.w;
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = a[k];
        done();
    }
    __device__ inline double team_sum(double partial)
    {
        double f[1] = {uniform(wave_sum(partial))}, l[1] = {0.0};
        publish(f, l);
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < W; ++wv) s += slot(wv, 0)[0];
        done();
        return s;
    }
    __device__ inline double scan_excl(double run)
    {
        const double incl = wave_scan_incl(run, lane);
        double f[1] = {0.0}, l[1] = {incl};
        publish(f, l);
        double excl = incl - run;
        for (int wv = 0; wv < W; ++wv)
            if (wv < wave) excl += slot(wv, 1)[0];
        done();
        return excl;
    }
    __device__ inline double max_all(double v)
    {
        double f[1] = {wave_max(v)}, l[1] = {0.0};
        publish(f, l);
        double m = slot(0, 0)[0];
#pragma unroll
        for (int wv = 1; wv < W; ++wv) m = fmax(m, slot(wv, 0)[0]);
        done();
        return m;
    }
    __device__ inline void out_begin(double *o, double min_y, double scale, double deep, double tail, double far, double post) { em.begin(o, min_y, scale, deep, tail, far, post); }
    // (rmin and post_frac are non-negative floats in scalar registers, neither ever a NaN: compared as integers, on the scalar
    // unit.  A vector compare takes ONE scalar operand, and the copy of the other was a register held for the whole solve.)
    __device__ inline bool post() const { return !(__float_as_int(em.rmin) >= __float_as_int(em.post_frac)); }
    __device__ inline bool tail() const { return em.tail(); }
    __device__ inline bool far() const { return em.far(); }
    __device__ inline bool deep() const { return em.deep(); }
    __device__ inline void out_first(double partial)
    {
        em.first(uniform(team_sum(partial)), p.dx, tid == 0);
        em.inv_first = uniform(em.inv_first);  // (wave-uniform results of vector arithmetic: to scalar registers, kScalarPolicy)
        em.post_frac = to_scalar(em.post_frac);
    }
    // Output samples: the waves' partial sums of the samples one step crosses are parked in LDS and added up behind ONE
    // barrier when the step is through with them (out_flush) -- a step of the cruise phase crosses 2 to 25 samples.
    __device__ inline void out_sample(int k, double partial)
    {
        const double s_ = wave_sum(partial);
        if (lane == 0) pend[(ppar * PMAX + npend) * W + wave] = s_;
        if (npend == 0) k0pend = k;
        if (++npend == PMAX) out_flush();
    }
    __device__ inline void out_flush()
    {
        if (npend == 0) return;
        wg_barrier();
        for (int n = 0; n < npend; ++n) {
            const double *q = pend + (ppar * PMAX + n) * W;
            double tot = 0.0;
#pragma unroll
            for (int wv = 0; wv < W; ++wv) tot += q[wv];
            em.put(k0pend + n, tot, p.dx, tid == 0);
        }
        em.rmin = uniform(em.rmin);
        em.clamped = uniform(em.clamped);
        npend = 0;
        ppar ^= 1;  // (the other buffer until the next flush: a wave cannot be two flushes ahead)
    }

    // ---- RHS: forward_solver.py:332-372, as WaveBackendT::rhs with the two neighbours across a wave boundary from
    // the halo ----
    __device__ inline void rhs(const double (&y)[V], double (&f)[V])
    {
        {
            const double hf[3] = {y[0], y[R], y[2 * R]}, hl[2] = {y[R - 1], y[2 * R - 1]};
            publish(hf, hl);
        }
        const double *pl = prev_last(), *nf = next_first();
        const double plN = pl[0], plP = pl[1], nfN = nf[0], nfP = nf[1], nfE = nf[2];
        done();
        double Jn[R + 1], Jp[R + 1];
        const double Nlw = pinned(from_prev(y[R - 1])), Plw = pinned(from_prev(y[2 * R - 1]));
        const double Nl = (lane == 0) ? plN : Nlw;
        const double Pl = (lane == 0) ? plP : Plw;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            const double N = y[r], P = y[R + r], E = y[2 * R + r];
            const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
            const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
            double jn, jp;
            edge_flux(p, Nm, N, Pm, P, E, jn, jp);  // :356-359
            if (r == 0) {  // :346-351 (the very first node only)
                const double Sft = surface_term(p, p.Sf, N, P);
                jn = (i == 0) ? Sft : jn;
                jp = (i == 0) ? -Sft : jp;
            }
            Jn[r] = jn;
            Jp[r] = jp;
        }
        // fluxes on the right face of this thread's last node: the next thread's first edge -- for the last lane of a
        // wave that edge belongs to the next wave and is evaluated here from the halo (the same expression, the same
        // operands: both sides get the same bits)
        double jnh, jph;
        edge_flux(p, y[R - 1], nfN, y[2 * R - 1], nfP, nfE, jnh, jph);
        const double jnw = pinned(from_next(Jn[0])), jpw = pinned(from_next(Jp[0]));
        Jn[R] = (lane == 63) ? jnh : jnw;
        Jp[R] = (lane == 63) ? jph : jpw;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            const double N = y[r], P = y[R + r];
            const double NP = N * P - p.n0p0;
            double jnr = Jn[r + 1], jpr = Jp[r + 1];
            if (i == L - 1) {  // :347,:350,:352
                const double Sbt = p.Sb * NP * frcp1(N + P);
                jnr = -Sbt;
                jpr = Sbt;
            }
            const double recomb = bulk_recomb(p, N, P, NP);  // :366
            f[r] = (jnr - Jn[r]) * p.idx - recomb;       // :369
            f[R + r] = -(jpr - Jp[r]) * p.idx - recomb;  // :370
            f[2 * R + r] = -(Jn[r] + Jp[r]) * p.Lam;     // :363 (edge 0: Sft - Sft, exactly 0)
        }
    }

    // A cross-lane move whose result feeds a select on the lane id must be EXECUTED BY EVERY LANE: written as
    // `lane == 0 ? halo : from_prev(x)` the compiler moves the DPP instruction under the exec mask of the lanes that
    // use its result, and lane 1 then reads a disabled lane 0 (measured: the right-hand side of nodes 1, 2 and 125 of
    // every wave wrong).  pinned() -- an empty volatile asm on the value -- keeps the move where it is written.
    static __device__ inline M2 mpin(const M2 &m) { return M2{pinned(m.a), pinned(m.b), pinned(m.c), pinned(m.d)}; }
    static __device__ inline M2 m2_of(const double *q) { return M2{q[0], q[1], q[2], q[3]}; }
    static __device__ inline void m2_to(const M2 &m, double *q) { q[0] = m.a; q[1] = m.b; q[2] = m.c; q[3] = m.d; }

    // ---- factorisation of M = I - c*J(y) ----
    __device__ inline void build(const double (&y_in)[V], double c)
    {
        double y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = pinned(y_in[v]);
        {
            const double hf[3] = {y[0], y[R], y[2 * R]}, hl[2] = {y[R - 1], y[2 * R - 1]};
            publish(hf, hl);
        }
        const double *pl = prev_last(), *nf = next_first();
        const double plN = pl[0], plP = pl[1], nfN = nf[0], nfP = nf[1], nfE = nf[2];
        done();
        M2 Lm[R], Rm[R], A[R], B[R], C[R];
        const double cL = c * p.Lam;
        const double dnx = p.dnx, dpx = p.dpx;
        const double Nlw = pinned(from_prev(y[R - 1])), Plw = pinned(from_prev(y[2 * R - 1]));
        const double Nl = (lane == 0) ? plN : Nlw;
        const double Pl = (lane == 0) ? plP : Plw;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
            const double N = y[r], P = y[R + r], E = y[2 * R + r];
            const double Nm = (r == 0) ? Nl : y[(r == 0) ? 0 : r - 1];
            const double Pm = (r == 0) ? Pl : y[R + ((r == 0) ? 0 : r - 1)];
            EdgeJac<double> ej = edge_jac_interior(p, cL, dnx, dpx, Nm, N, Pm, P, E);
            if (i == 0) ej = edge_jac_front(p, N, P);
            if (i >= L) ej = edge_jac_none<double>();
            Lm[r] = ej.lm;
            Rm[r] = ej.rm;
            w[r][0] = ej.w0; w[r][1] = ej.w1;
            // the four coefficients of the field back-substitution, e0, e1 = c Lam mv (mu_n E / 2 -+ D_n/dx) and
            // e2, e3 = c Lam mv (mu_p E / 2 +- D_p/dx), are rebuilt from ONE number per edge where they are used
            qE[r] = (i > 0 && i < L) ? (float)(cL * ej.mv * E) : 0.0f;
            mvv[r] = ej.mv;
        }
        kdn = (float)(cL * dnx);
        kdp = (float)(cL * dpx);
        cd = c * p.idx;
        // edge matrices of the next thread's first edge: evaluated HERE from that node's (N, P, E) -- three doubles by a
        // wave shift, or from the halo across a wave boundary -- instead of shifting the neighbour's two finished 2x2
        // blocks (and holding a second pair of them for the halo): the same expression on the same operands as the
        // owner's, fewer live registers in the factorisation
        const double Nnw = pinned(from_next(y[0])), Pnw = pinned(from_next(y[R])), Enw = pinned(from_next(y[2 * R]));
        const double Nn = (lane == 63) ? nfN : Nnw, Pn = (lane == 63) ? nfP : Pnw, En = (lane == 63) ? nfE : Enw;
        EdgeJac<double> ejn = edge_jac_interior(p, cL, dnx, dpx, y[R - 1], Nn, y[2 * R - 1], Pn, En);
        if (tid * R + R >= L) ejn = edge_jac_none<double>();
        const M2 LmNext = ejn.lm, RmNext = ejn.rm;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = tid * R + r;
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
        // in-lane cyclic reduction: row e = 2*tid (r=0) eliminated, row k = 2*tid+1 (r=1) kept.  (Written so that each of
        // the next thread's three blocks is consumed where it arrives.)
        const M2 Binv_e = minv(B[0]);
        put(BINV_E, Binv_e);
        put(BIA_E, mmul(Binv_e, A[0]));
        put(BIC_E, mmul(Binv_e, C[0]));
        if (lane == 0) {
            double *q = slot(wave, 0);
            m2_to(Binv_e, q); m2_to(A[0], q + 4); m2_to(C[0], q + 8);
        }
        wg_barrier();
        const double *nq = next_first();
        M2 Ak, Bk, Ck;
        {
            const M2 Binv_up = msel(lane == 63, m2_of(nq), mpin(mnext(Binv_e)));
            const M2 cr_ga = mmul(C[1], Binv_up);  // (the very last thread: C[1] = 0, whatever Binv_up is)
            put(CR_GA, cr_ga);
            const M2 C_up = msel(lane == 63, m2_of(nq + 8), mpin(mnext(C[0])));
            Ck = mneg(mmul(cr_ga, C_up));
            const M2 A_up = msel(lane == 63, m2_of(nq + 4), mpin(mnext(A[0])));
            Bk = msub(B[1], mmul(cr_ga, A_up));
        }
        done();
        {
            const M2 cr_al = mmul(A[1], Binv_e);
            put(CR_AL, cr_al);
            Ak = mneg(mmul(cr_al, A[0]));
            Bk = msub(Bk, mmul(cr_al, C[0]));
        }
        // Cut the couplings across the wave boundaries: they are the right-hand sides of the two spikes.  The spikes ride
        // through the factorisation in fp64 next to Bk, Ak, Ck: the interface system they define is nearly singular in a
        // stiff step (I + V + W -> 0 for the smooth mode), so its entries need the exact factors -- a second pass through
        // the fp32-STORED factors (tried: 23 spilled registers instead of 200) is off by 6e-4 at the wave's ends, 60 times
        // the margin, and Newton crawls.  What keeps the registers down instead: levels 1-4 are a REAL loop (unrolled, the
        // scheduler hoists the exchanges of all levels and both spike columns: 200 spilled registers on this budget).
        M2 Vr = msel(lane == 0 && wave > 0, Ak, mzero<double>());
        M2 Wr = msel(lane == 63 && wave < W - 1, Ck, mzero<double>());
        Ak = msel(lane == 0, mzero<double>(), Ak);
        Ck = msel(lane == 63, mzero<double>(), Ck);
        // One PCR level, its ten neighbour blocks fetched in FIVE groups of two, each group requested only when the one
        // before has been consumed: the source of a group's exchange is pinned behind the previous group's result (volatile
        // asms keep their order), so the scheduler cannot hoist all ten exchanges -- 80 registers -- to the top of the
        // level.  `xm` / `xp` fetch a block from the row below / above at this level's stride.
        auto level = [&](int lv, auto &&xm, auto &&xp) {
            const M2 Bi = minv(Bk);
            const M2 al = mmul(Ak, xm(Bi)), ga = mmul(Ck, xp(Bi));
            if (lv < 5) {
                static_cast<float4 *>(lds)[(PAL0 + lv) * 64 + lane] = float4{(float)al.a, (float)al.c, (float)al.b, (float)al.d};
                static_cast<float4 *>(lds)[(PGA0 + lv) * 64 + lane] = float4{(float)ga.a, (float)ga.c, (float)ga.b, (float)ga.d};
            } else {
                put(L5ROW, msel(lane < 32, ga, al));  // (al = 0 where lane < 32, ga = 0 where lane >= 32)
            }
            const M2 alp = mpin(al), gap = mpin(ga);
            const M2 Ak0 = mpin(Ak), Ck0 = mpin(Ck);
            Bk = mpin(msub(msub(Bk, mmul(alp, xm(Ck0))), mmul(gap, xp(Ak0))));
            const M2 Ak1 = mpin(Ak0), Ck1 = mpin(Ck0);
            Ak = mpin(mneg(mmul(alp, xm(Ak1))));
            Ck = mpin(mneg(mmul(gap, xp(Ck1))));
            const M2 Vr0 = mpin(Vr);
            Vr = mpin(msub(msub(Vr0, mmul(alp, xm(Vr0))), mmul(gap, xp(Vr0))));
            const M2 Wr0 = mpin(Wr);
            Wr = mpin(msub(msub(Wr0, mmul(alp, xm(Wr0))), mmul(gap, xp(Wr0))));
        };
        // level 0: nearest neighbours by DPP wave shifts
        level(0, [&](const M2 &m) { return mprev(m); }, [&](const M2 &m) { return mnext(m); });
#pragma unroll 1
        for (int lv = 1; lv < 5; ++lv) {
            const int lm_ = (lane - (1 << lv)) & 63, lp_ = (lane + (1 << lv)) & 63;
            level(lv, [&](const M2 &m) { return mshf(m, lm_); }, [&](const M2 &m) { return mshf(m, lp_); });
        }
        // level 5: stride 32 by v_permlane32_swap
        level(5, [&](const M2 &m) { M2 dn, up; mswap32(m, dn, up); return dn; }, [&](const M2 &m) { M2 dn, up; mswap32(m, dn, up); return up; });
        const M2 Bfin = minv(Bk);
        put(BFIN, Bfin);
        const M2 Vk = mmul(Bfin, Vr), Wk = mmul(Bfin, Wr);
        {
            // NOT (V, W) as bf16 in one row, and stored as (V, Z = I + V + W): in a stiff step I + V + W is tiny (the
            // smooth mode of a wave's 128 nodes is nearly harmonic: what it is at the two ends, it is inside);
            // x = g + xf - Z xf - V (xl - xf) keeps the two products small where the cancellation is.
            const M2 Zk{1.0 + Vk.a + Wk.a, Vk.b + Wk.b, Vk.c + Wk.c, 1.0 + Vk.d + Wk.d};
            put(VSPK, Vk);
            put(ZSPK, Zk);
        }
        // ---- the reduced system: every wave publishes its spikes at its first and last row ----
        {
            double hf[8], hl[8];
            m2_to(Vk, hf); m2_to(Wk, hf + 4);
            m2_to(Vk, hl); m2_to(Wk, hl + 4);
            publish(hf, hl);
        }
        // block Thomas on the pairs u_j = (l_j, f_{j+1}), j = 0 .. NI-1, with
        //   D_j = [[I, Wl_j], [Vf_{j+1}, I]],  L_j = [[Vl_j, 0], [0, 0]] (on u_{j-1}),  U_j = [[0, 0], [0, Wf_{j+1}]] (on u_{j+1});
        // elimination keeps the shape: D'_j = [[I, Q_j], [Vf_{j+1}, I]], Q_j = Wl_j - Vl_j (E_{j-1})_lf Wf_j,
        // E_j = D'_j^-1 = [[I + Q S Vf, -Q S], [-S Vf, S]], S = (I - Vf Q)^-1.  Lane m (< NIF) carries the unit
        // right-hand side e_m, so it ends up with column m of the inverse; lanes >= NIF compute a zero column.
        // The published spikes are read where they are needed (that buffer stays put until every wave has passed the
        // next barrier), Q_j and S_j wait for the back substitution in this wave's own slots of the other buffer.
        const int pb = xpar;
        done();
        double *qs_ = slot(wave, 0);  // 8 NI <= 24 doubles
        auto pub = [&](int wv, int which) { return xb + ((pb * W + wv) * 2 + which) * KX; };
        // E_j h'_j of the last pair stays in registers, the earlier ones wait for the back substitution in this wave's own
        // table (free until its last lines here fill it): lane m carries the unit right-hand side e_m, whose E_j h'_j is zero
        // for every pair before its own (j < m / 4) -- row j has 4 (j + 1) lanes that are not, four doubles each, and the
        // NI - 1 rows are 8 NI (NI - 1) doubles = the table's 4 NIF at NI = 3.  (An array indexed by the counter of a loop
        // that is not unrolled lives in scratch memory; unrolled, these two loops spill 45 registers more.)
        static_assert(8 * NI * (NI - 1) <= 4 * NIF, "parked rows fit this wave's table");
        auto parked = [&](int j) { return tab + 8 * j * (j + 1) + 4 * lane; };
        V2 el_last{0.0, 0.0}, ef_last{0.0, 0.0};
        {
            M2 Qp = mzero<double>(), Sp = mident<double>(), Wfp = mzero<double>();
#pragma unroll 1
            for (int j = 0; j < NI; ++j) {  // (a real loop, like the spike columns above: one interface at a time)
                const double *a = pub(j, 1), *b = pub(j + 1, 0);
                const M2 Vl = m2_of(a);      // V_last(j): acts on l_{j-1}
                const M2 Wl = m2_of(a + 4);  // W_last(j): acts on f_{j+1}
                const M2 Vf = m2_of(b);      // V_first(j+1): acts on l_j
                M2 Q = Wl;
                if (j > 0) Q = msub(Wl, mneg(mmul(mmul(Vl, mmul(Qp, Sp)), Wfp)));  // Q_j = Wl_j + Vl_j Q_{j-1} S_{j-1} Wf_{j-1}
                const M2 S = minv(msub(mident<double>(), mmul(Vf, Q)));
                if (lane == 0) {
                    m2_to(Q, qs_ + 8 * j);
                    m2_to(S, qs_ + 8 * j + 4);
                }
                V2 hl_{(lane == 4 * j) ? 1.0 : 0.0, (lane == 4 * j + 1) ? 1.0 : 0.0};
                const V2 hf_{(lane == 4 * j + 2) ? 1.0 : 0.0, (lane == 4 * j + 3) ? 1.0 : 0.0};
                if (j > 0) {  // h'_j = h_j - L_j E_{j-1} h'_{j-1}: only the l-part changes, by Vl_j (E_{j-1} h'_{j-1})_l
                    const V2 t = mvec(Vl, el_last);  // (still the previous pair's)
                    hl_ = V2{hl_.x - t.x, hl_.y - t.y};
                }
                // E h = [hl + Q S (Vf hl - hf) ; S (hf - Vf hl)]
                const V2 vh = mvec(Vf, hl_);
                const V2 sf = mvec(S, V2{hf_.x - vh.x, hf_.y - vh.y});
                const V2 qsv = mvec(Q, sf);
                el_last = V2{hl_.x - qsv.x, hl_.y - qsv.y};
                ef_last = sf;
                if (j < NI - 1 && lane < 4 * (j + 1)) {
                    double *q = parked(j);
                    q[0] = el_last.x; q[1] = el_last.y; q[2] = ef_last.x; q[3] = ef_last.y;
                }
                Qp = Q;
                Sp = S;
                Wfp = m2_of(b + 4);  // W_first(j+1): acts on f_{j+2}
            }
        }
        // back substitution: u_j = E_j h'_j - E_j U_j u_{j+1};  E_j U_j u = [-Q S Wf f_{j+2} ; S Wf f_{j+2}];
        // rows this wave needs: x_last(wave-1) = l_{wave-1}, x_first(wave+1) = f_{wave+1} = f-part of pair `wave`
        V2 xl{0.0, 0.0}, xf{0.0, 0.0}, ufn{0.0, 0.0};
#pragma unroll 1
        for (int j = NI - 1; j >= 0; --j) {
            V2 ul = el_last, uf = ef_last;
            if (j < NI - 1) {
                const bool mine = lane < 4 * (j + 1);
                const double *q = mine ? parked(j) : tab;  // (a valid address in every lane)
                ul = V2{mine ? q[0] : 0.0, mine ? q[1] : 0.0};
                uf = V2{mine ? q[2] : 0.0, mine ? q[3] : 0.0};
            }
            if (j < NI - 1) {
                const M2 Q = m2_of(qs_ + 8 * j), S = m2_of(qs_ + 8 * j + 4), Wf = m2_of(pub(j + 1, 0) + 4);
                const V2 sw = mvec(S, mvec(Wf, ufn));
                const V2 qsw = mvec(Q, sw);
                ul = V2{ul.x + qsw.x, ul.y + qsw.y};
                uf = V2{uf.x - sw.x, uf.y - sw.y};
            }
            if (j == wave - 1) xl = ul;
            if (j == wave) xf = uf;
            ufn = uf;
        }
        if (lane < NIF) {
            tab[0 * NIF + lane] = xl.x;
            tab[1 * NIF + lane] = xl.y;
            tab[2 * NIF + lane] = xf.x;
            tab[3 * NIF + lane] = xf.y;
        }
        // (read back by this wave's own lanes only, behind the barriers of the next exchanges)
    }

    // x = M^{-1} b; b, x in lane layout [N0,N1,P0,P1,E0,E1]
    __device__ inline void apply(const double (&b)[V], double (&x)[V])
    {
        const PB cr_al = get_pb(CR_AL), cr_ga = get_pb(CR_GA);
        PB al = get_pb(PAL0), ga = get_pb(PGA0);
        V2 wb[3];
        wb[0] = V2{w[0][0] * b[4], w[0][1] * b[4]};
        wb[1] = V2{w[1][0] * b[5], w[1][1] * b[5]};
        V2 r[2];
        r[0] = V2{b[0] + cd * (wb[1].x - wb[0].x), b[2] - cd * (wb[1].y - wb[0].y)};
        {
            const double hf[4] = {wb[0].x, wb[0].y, r[0].x, r[0].y}, hl[1] = {0.0};
            publish(hf, hl);
        }
        const double *nq = next_first();
        // the last wave's last lane: its last node is the back surface or padding, the weights beyond are zero
        const bool edge_ = lane == 63;
        const bool none_ = edge_ && wave == W - 1;
        const V2 wbn0 = vnext(wb[0]);
        const V2 wbn{pinned(wbn0.x), pinned(wbn0.y)};
        wb[2] = V2{none_ ? 0.0 : (edge_ ? nq[0] : wbn.x), none_ ? 0.0 : (edge_ ? nq[1] : wbn.y)};
        const double r0ux = nq[2], r0uy = nq[3];
        done();
        r[1] = V2{b[1] + cd * (wb[2].x - wb[1].x), b[3] - cd * (wb[2].y - wb[1].y)};
        const v2f r0 = narrow(r[0]);
        const float r0wx = pinned(from_next(r0.x)), r0wy = pinned(from_next(r0.y));
        const v2f r0_up{edge_ ? (float)r0ux : r0wx, edge_ ? (float)r0uy : r0wy};
        const v2f t = pb_acc(cr_ga, r0_up, pb_acc(cr_al, r0, v2f(0.0f)));
        v2f rk32 = narrow(V2{r[1].x - (double)t.x, r[1].y - (double)t.y});
        PB b_inv = al, b_ia = al, b_ic = al, b_fin = al, b5 = al;
        PB vs_ = al, zs_ = al;
#pragma unroll
        for (int lv = 0; lv < 5; ++lv) {
            const int s = 1 << lv;
            PB al_n = al, ga_n = ga;
            if (lv < 4) {
                al_n = get_pb(PAL0 + lv + 1);
                ga_n = get_pb(PGA0 + lv + 1);
            } else {
                b5 = get_pb(L5ROW);
                b_fin = get_pb(BFIN);
            }
            if (lv == 2) {
                vs_ = get_pb(VSPK);
                zs_ = get_pb(ZSPK);
            } else if (lv == 3) {
                b_inv = get_pb(BINV_E);
            }
            const v2f rf = rk32;
            v2f rm, rp;
            if (lv == 0) {
                rm = v2f{from_prev(rf.x), from_prev(rf.y)};
                rp = v2f{from_next(rf.x), from_next(rf.y)};
            } else {
                rm = v2f{shf(rf.x, (lane - s) & 63), shf(rf.y, (lane - s) & 63)};
                rp = v2f{shf(rf.x, (lane + s) & 63), shf(rf.y, (lane + s) & 63)};
            }
            rk32 = rk32 - (ga.bd * v2f(rp.y) + (ga.ac * v2f(rp.x) + (al.bd * v2f(rm.y) + al.ac * v2f(rm.x))));
            al = al_n;
            ga = ga_n;
        }
        {  // stride 32: one block per lane, the partner 32 lanes above (lower half) or below (upper half)
            b_ia = get_pb(BIA_E);
            b_ic = get_pb(BIC_E);
            float mx, px, my, py;
            swap32(rk32.x, mx, px);
            swap32(rk32.y, my, py);
            const v2f r5{lane < 32 ? px : mx, lane < 32 ? py : my};
            rk32 = rk32 - (b5.bd * v2f(r5.y) + b5.ac * v2f(r5.x));
        }
        const PB vs = vs_, zs = zs_;
        const v2f g32 = pb_acc(b_fin, rk32, v2f(0.0f));  // this wave's solution with the couplings cut
        // ---- interface solve ----
        {
            const double hf[2] = {(double)g32.x, (double)g32.y};
            publish(hf, hf);
        }
        // lanes 0..3 each evaluate one of the four numbers this wave needs: sum_m tab[row][m] * h[m],
        // h = (g_last(0), g_first(1), g_last(1), g_first(2), ...)
        const int row = lane & 3;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const double *gl = slot(j, 1), *gf = slot(j + 1, 0);
            const double *tr_ = tab + row * NIF + 4 * j;
            acc = fma(tr_[0], gl[0], acc);
            acc = fma(tr_[1], gl[1], acc);
            acc = fma(tr_[2], gf[0], acc);
            acc = fma(tr_[3], gf[1], acc);
        }
        done();
        const v2f xl{(float)lane_value(acc, 0), (float)lane_value(acc, 1)};
        const v2f xf{(float)lane_value(acc, 2), (float)lane_value(acc, 3)};
        const v2f xk32 = (g32 + xf) - pb_acc(zs, xf, pb_acc(vs, xl - xf, v2f(0.0f)));
        const float xpwx = pinned(from_prev(xk32.x)), xpwy = pinned(from_prev(xk32.y));
        // lane 0: the previous wave's last row (wave 0: meets a zero block)
        const v2f xkp32{lane == 0 ? xl.x : xpwx, lane == 0 ? xl.y : xpwy};
        const v2f se = pb_acc(b_inv, r0, v2f(0.0f)) - pb_acc(b_ic, xk32, pb_acc(b_ia, xkp32, v2f(0.0f)));
        x[0] = (double)se.x; x[1] = (double)xk32.x; x[2] = (double)se.y; x[3] = (double)xk32.y;
        // field back-substitution: (e0, e1) = ea -+ c Lam mv D_n/dx, (e2, e3) = ep +- c Lam mv D_p/dx; edge 0 and padded
        // edges have none
        const bool v0 = tid * R > 0 && tid * R < L, v1 = tid * R + 1 < L;
        const float kn0 = v0 ? kdn * (float)mvv[0] : 0.0f, kp0 = v0 ? kdp * (float)mvv[0] : 0.0f;
        const float kn1 = v1 ? kdn * (float)mvv[1] : 0.0f, kp1 = v1 ? kdp * (float)mvv[1] : 0.0f;
        const float hn = 0.5f * (float)p.mun, hp = 0.5f * (float)p.mup;
        const float a0 = hn * qE[0], p0_ = hp * qE[0], a1 = hn * qE[1], p1_ = hp * qE[1];
        const PB e0{v2f{a0 - kn0, p0_ + kp0}, v2f{a0 + kn0, p0_ - kp0}}, e1{v2f{a1 - kn1, p1_ + kp1}, v2f{a1 + kn1, p1_ - kp1}};
        const v2f d0 = e0.ac * xkp32 + e0.bd * se, d1 = e1.ac * se + e1.bd * xk32;
        x[4] = b[4] * mvv[0] - (double)(d0.x + d0.y);
        x[5] = b[5] * mvv[1] - (double)(d1.x + d1.y);
    }
};

// One workgroup of W waves per solve, on a 256-register budget: two waves per SIMD, i.e. two workgroups per CU at W = 4
// (2 x 80 KB of LDS) and four at W = 2 -- while one workgroup waits at an exchange the other issues (+31 ... +38 % solves per
// second over one workgroup per CU, measured with 200 registers of the factorisation still spilled).
template <int W, bool CT>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2, 2))) void solve_kernel_spike(
    const mtrpl_solve_desc *__restrict__ descs, const double *__restrict__ ini, const double *__restrict__ tsteps,
    double *__restrict__ out, int32_t *__restrict__ status, int32_t *__restrict__ stats, int64_t B, SolveTuning tune)
{
    using BK = SpikeBackendT<W>;
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if ((int64_t)blockIdx.x >= B) return;
    const int64_t sidx = blockIdx.x;
    __shared__ float4 slabs[W * BK::kSlabBytes / 16];
    __shared__ double shared[BK::kSharedDoubles];
    BK bk(threadIdx.x, slabs, shared);
    if (CT)
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, DefaultTune{}, clk0_);
    else
        integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

}  // namespace mtrpl
