// backend_pair.hpp -- the low-latency backend: W in {1, 2} wavefronts (one workgroup) integrate one
// forward simulation with ONE grid node per lane, wave w owning the contiguous nodes 64w .. 64w+63
// (nx <= 64 W; std model).  Per lane everything is half of the single-wave backend's work (3 values
// instead of 6 in every Nordsieck row, Newton vector and right-hand side), so a solve that has the
// GPU's SIMDs to itself -- an under-filled launch: parallel-tempering shards of 64 chains per GPU --
// finishes sooner, and the halved register footprint lets two such waves share a SIMD in a full one.
//
// Every exchange INSIDE a wave stays in registers: nearest neighbours by DPP wave shifts, the longer
// strides of the parallel cyclic reduction by ds_bpermute.  LDS is touched for exactly two things:
//
//  (1) the one coupling between the two waves.  The block-tridiagonal Newton matrix is split at the wave
//      boundary (rows 63 | 64) in the manner of the SPIKE algorithm (Polizzi & Sameh 2006): each wave
//      runs a 6-level PCR on its own 64 rows with the coupling block cut, the factorisation also solves
//      for the spike S_w = M_w^-1 (e_b x cut block), and a Newton re-solve needs one 32-byte exchange:
//      the local solutions g_63, g_64 of the two boundary rows.  Both waves then solve the 2-block
//      interface system  x_63 = K (g_63 - S0_63 g_64),  x_64 = g_64 - S1_64 x_63,  K = (I - S0_63 S1_64)^-1,
//      redundantly and correct their rows, x = g - S x_other.
//  (2) the grid sums (error norms, output samples) and the boundary node's state, which travel together:
//      one 64-byte record per wave per exchange, one s_barrier.  Per step that is one exchange for the
//      predicted boundary state (carrying the pending output samples) and two per Newton iteration.
//
// Both waves execute the same scalar control flow on bitwise identical reduced values (a + b == b + a),
// so every barrier is reached by both.
#pragma once
#include "solve_core.hpp"

namespace mtrpl {

// NREG: PCR levels (of 6) whose two elimination blocks per lane are held in registers; the others
// live in LDS as conflict-free 16-byte rows.
template <int W, int NREG>
struct PairBackend {
    using real = double;
    static constexpr int R = 1, V = 3, T = 64 * W;
    static constexpr bool TRAPS = false;
    static constexpr int NLDS = 6 - NREG;                      // levels kept in LDS
    static constexpr int kBlkBytes = NLDS * 2 * 2 * T * 16;    // [level][al|ga][row][T] x 16 B
    static constexpr int kXchBytes = 2 * 2 * 8 * 8;            // [parity][wave][8 doubles]

    Par p;
    int L;
    int lane, wave, nd;  // nd = global node of this lane
    int bl;              // boundary lane of this wave: 63 (wave 0), 0 (wave 1)
    Emitter em;
    // factors of the current matrix
    double w0, w1, e0, e1, e2, e3, mv, cd, hcur;
    M2 alr[NREG > 0 ? NREG : 1], gar[NREG > 0 ? NREG : 1];
    M2 Bfin, S;          // S: spike of this lane's row
    M2 K, S0b, S1b;      // interface blocks (uniform)
    double wn0, wn1;     // wave 0, lane 63: weights of edge 64
    // boundary node of the other wave as last published: N, P, E and z1_E + acor_E
    double nbN, nbP, nbE, nbZ;
    double fE_nb;        // wave 0, lane 63: dE/dt of edge 64 from the last rhs()
    // LDS
    double2 *blk;
    double *xs;
    int par;
    // output samples waiting for the next exchange (wave sums of this wave)
    double pend[4];
    int n_pend, io_first;

    __device__ inline PairBackend(int tid, double2 *blk_, double *xs_) : blk(blk_), xs(xs_)
    {
        lane = tid & 63;
        wave = uniform(tid >> 6);
        nd = tid;
        bl = (wave == 0) ? 63 : 0;
        par = 0;
        n_pend = 0;
        io_first = 0;
        nbN = nbP = 1.0;
        nbE = nbZ = fE_nb = wn0 = wn1 = 0.0;
        hcur = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) pend[k] = 0.0;
    }
    // One node per lane leaves VGPRs to spare while the scalar file is over-subscribed (the controller's
    // state, the tuning constants, the pointers): the model parameters live in VGPRs here -- a VALU operand
    // either way, and no v_readlane to fetch a spilled SGPR back.
    static __device__ inline double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }
    __device__ inline void set_problem(const Par &q, int L_)
    {
        L = L_;
        p.n0 = in_vgpr(q.n0); p.p0 = in_vgpr(q.p0); p.mun = in_vgpr(q.mun); p.mup = in_vgpr(q.mup);
        p.ks = in_vgpr(q.ks); p.Cn = in_vgpr(q.Cn); p.Cp = in_vgpr(q.Cp); p.Sf = in_vgpr(q.Sf); p.Sb = in_vgpr(q.Sb);
        p.tn = in_vgpr(q.tn); p.tp = in_vgpr(q.tp); p.Lam = in_vgpr(q.Lam); p.Tm = q.Tm;
        p.kC = q.kC; p.Nt = q.Nt; p.tE = q.tE;
        p.dx = in_vgpr(q.dx); p.idx = in_vgpr(q.idx); p.DnT = in_vgpr(q.DnT); p.DpT = in_vgpr(q.DpT);
        p.n0p0 = in_vgpr(q.n0p0);
    }
    __device__ inline int node(int) const { return nd; }
    __device__ inline bool leader() const { return nd == 0; }

    // ---- the exchange: boundary lane posts K doubles, everybody reads the other wave's ----
    template <int KD>
    __device__ inline void xchg(const double (&mine)[KD], double (&theirs)[KD])
    {
        static_assert(KD <= 8 && KD % 2 == 0, "record is 8 doubles, moved as 16-byte rows");
        double2 *slot = reinterpret_cast<double2 *>(xs) + par * 8;  // [wave][4 rows]
        if (lane == bl) {
#pragma unroll
            for (int k = 0; k < KD / 2; ++k) slot[wave * 4 + k] = double2{mine[2 * k], mine[2 * k + 1]};
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KD / 2; ++k) {
            const double2 v = slot[(1 - wave) * 4 + k];
            theirs[2 * k] = v.x;
            theirs[2 * k + 1] = v.y;
        }
        par ^= 1;
    }
    __device__ inline void emit_pending(double t0, double t1, double t2, double t3)
    {
        // totals are own wave sum + the other wave's (a + b == b + a: both waves see the same values and
        // keep the same emitter state); lane 0 of wave 0 stores
        // (static register indices only: a run-time index would send the record to scratch memory)
        const bool writer = (wave == 0) && (lane == 0);
        if (n_pend > 0) em.put(io_first, pend[0] + t0, p.dx, writer);
        if (n_pend > 1) em.put(io_first + 1, pend[1] + t1, p.dx, writer);
        if (n_pend > 2) em.put(io_first + 2, pend[2] + t2, p.dx, writer);
        if (n_pend > 3) em.put(io_first + 3, pend[3] + t3, p.dx, writer);
        n_pend = 0;
    }

    __device__ inline void publish(const double (&y)[V], const double (&zE)[R])
    {
        if (W == 1) return;
        const double mine[8] = {y[0], y[1], y[2], zE[0], pend[0], pend[1], pend[2], pend[3]};
        double th[8];
        xchg<8>(mine, th);
        nbN = th[0]; nbP = th[1]; nbE = th[2]; nbZ = th[3];
        emit_pending(th[4], th[5], th[6], th[7]);
    }
    template <int NS>
    __device__ inline void reduce(double (&s)[NS], const double (&y)[V], const double (&zE)[R])
    {
        static_assert(NS <= 4, "record holds 4 sums + the boundary state");
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = uniform(wave_sum_ctl(s[k]));
        if (W == 1) return;
        constexpr int KD = (NS + 4 + 1) & ~1;
        double mine[KD], th[KD];
#pragma unroll
        for (int k = 0; k < KD; ++k) mine[k] = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) mine[k] = s[k];
        mine[NS] = y[0]; mine[NS + 1] = y[1]; mine[NS + 2] = y[2]; mine[NS + 3] = zE[0];
        xchg<KD>(mine, th);
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] = uniform(mine[k] + th[k]);
        nbN = th[NS]; nbP = th[NS + 1]; nbE = th[NS + 2]; nbZ = th[NS + 3];
    }
    __device__ inline double scan_excl(double run)
    {
        const double incl = wave_scan_incl(run, lane);
        double excl = incl - run;
        if (W == 2) {
            const double tot = lane_value(incl, 63);
            const double mine[2] = {tot, 0.0};
            double th[2];
            xchg<2>(mine, th);
            if (wave == 1) excl += th[0];
        }
        return excl;
    }
    __device__ inline double max_all(double v)
    {
        v = wave_max(v);
        if (W == 2) {
            const double mine[2] = {v, 0.0};
            double th[2];
            xchg<2>(mine, th);
            v = fmax(v, th[0]);
        }
        return v;
    }
    __device__ inline void out_begin(double *o, double min_y, double scale, double deep) { em.begin(o, min_y, scale, deep); }
    __device__ inline bool deep() const { return em.deep; }
    __device__ inline void flush()
    {
        const double mine[4] = {pend[0], pend[1], pend[2], pend[3]};
        double th[4];
        xchg<4>(mine, th);
        emit_pending(th[0], th[1], th[2], th[3]);
    }
    __device__ inline void out_sample(int k, double partial)
    {
        const double s = uniform(wave_sum(partial));
        if (W == 1) {
            em.put(k, s, p.dx, lane == 0);
            return;
        }
        if (n_pend == 4) flush();
        if (n_pend == 0) io_first = k;
        pend[0] = (n_pend == 0) ? s : pend[0];
        pend[1] = (n_pend == 1) ? s : pend[1];
        pend[2] = (n_pend == 2) ? s : pend[2];
        pend[3] = (n_pend == 3) ? s : pend[3];
        n_pend++;
    }
    __device__ inline void out_finish()
    {
        if (W == 2 && n_pend > 0) flush();
    }

    // ---- elimination blocks: registers for the first NREG levels, LDS rows for the rest ----
    __device__ inline void put_lds(int slot, const M2 &m) const
    {
        const int t = wave * 64 + lane;
        blk[(slot * 2 + 0) * T + t] = double2{m.a, m.b};
        blk[(slot * 2 + 1) * T + t] = double2{m.c, m.d};
    }
    __device__ inline M2 get_lds(int slot) const
    {
        int t = wave * 64 + lane;
        asm volatile("" : "+v"(t));  // keep the loads inside the Newton loop (see backend_wave.hpp)
        const double2 u = blk[(slot * 2 + 0) * T + t], v = blk[(slot * 2 + 1) * T + t];
        return M2{u.x, u.y, v.x, v.y};
    }
    // neighbour at distance s inside this wave (wrapped values only ever meet zero blocks)
    template <int S_> __device__ inline M2 mdn(const M2 &m) const { return (S_ == 1) ? mprev(m) : mshf(m, (lane - S_) & 63); }
    template <int S_> __device__ inline M2 mup(const M2 &m) const { return (S_ == 1) ? mnext(m) : mshf(m, (lane + S_) & 63); }
    template <int S_> __device__ inline V2 vdn(const V2 &v) const { return (S_ == 1) ? vprev(v) : vshf(v, (lane - S_) & 63); }
    template <int S_> __device__ inline V2 vup(const V2 &v) const { return (S_ == 1) ? vnext(v) : vshf(v, (lane + S_) & 63); }

    // ---- RHS: forward_solver.py:332-372 ----
    __device__ inline void rhs(const double (&y)[V], double (&f)[V])
    {
        const double N = y[0], P = y[1], E = y[2];
        double Nm = from_prev(N), Pm = from_prev(P);
        if (W == 2) {
            const bool first = (wave == 1) && (lane == 0);
            Nm = first ? nbN : Nm;
            Pm = first ? nbP : Pm;
        }
        double jn, jp;
        edge_flux(p, Nm, N, Pm, P, E, jn, jp);  // :356-359
        if (nd == 0) {  // :346-351
            const double Sft = surface_term(p, p.Sf, N, P);
            jn = Sft;
            jp = -Sft;
        }
        if (nd >= L) { jn = 0.0; jp = 0.0; }
        double jnr = from_next(jn), jpr = from_next(jp);
        if (W == 2 && wave == 0) {
            // edge 64 belongs to the other wave: lane 63 evaluates it from the published boundary node
            double jb, pb;
            edge_flux(p, N, nbN, P, nbP, nbE, jb, pb);
            if (64 >= L) { jb = 0.0; pb = 0.0; }
            const bool last = lane == 63;
            jnr = last ? jb : jnr;
            jpr = last ? pb : jpr;
            fE_nb = -(jb + pb) * p.Lam;
        }
        const double NP = N * P - p.n0p0;
        if (nd == L - 1) {  // :347,:350,:352
            const double Sbt = p.Sb * NP * frcp(N + P);
            jnr = -Sbt;
            jpr = Sbt;
        }
        const double recomb = bulk_recomb(p, N, P, NP);  // :366
        const bool real_ = nd < L;
        f[0] = real_ ? ((jnr - jn) * p.idx - recomb) : 0.0;       // :369
        f[1] = real_ ? (-(jpr - jp) * p.idx - recomb) : 0.0;      // :370
        f[2] = (real_ && nd > 0) ? (-(jn + jp) * p.Lam) : 0.0;   // :363 (edge 0: exactly 0)
    }

    // ---- factorisation of M = I - c*J(y); h = the step behind c (needed for the edge-64 entry) ----
    __device__ inline void build(const double (&y_in)[V], double c, double h)
    {
        double y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = pinned(y_in[v]);  // keep the Jacobian inside the branch it is called from
        hcur = h;
        const double cL = c * p.Lam;
        const double dnx = p.DnT * p.idx, dpx = p.DpT * p.idx;
        cd = c * p.idx;
        const double N = y[0], P = y[1], E = y[2];
        double Nm = from_prev(N), Pm = from_prev(P);
        if (W == 2) {
            const bool first = (wave == 1) && (lane == 0);
            Nm = first ? nbN : Nm;
            Pm = first ? nbP : Pm;
        }
        EdgeJac<double> ej = edge_jac_interior(p, cL, dnx, dpx, Nm, N, Pm, P, E);
        ej = ejsel(nd == 0, edge_jac_front(p, N, P), ej);
        ej = ejsel(nd >= L, edge_jac_none<double>(), ej);
        w0 = ej.w0; w1 = ej.w1; mv = ej.mv;
        e0 = ej.e0; e1 = ej.e1; e2 = ej.e2; e3 = ej.e3;
        M2 lmn = mnext(ej.lm), rmn = mnext(ej.rm);
        if (W == 2 && wave == 0) {
            EdgeJac<double> eb = edge_jac_interior(p, cL, dnx, dpx, N, nbN, P, nbP, nbE);
            eb = ejsel(64 >= L, edge_jac_none<double>(), eb);
            const bool last = lane == 63;
            lmn = msel(last, eb.lm, lmn);
            rmn = msel(last, eb.rm, rmn);
            wn0 = eb.w0;
            wn1 = eb.w1;
        }
        {  // back surface: J_L = (-SL, SL)
            double sLn, sLp;
            surface_jac(p, p.Sb, N, P, sLn, sLp);
            const bool back = nd == L - 1;
            lmn = msel(back, M2{-sLn, -sLp, sLn, sLp}, lmn);
            rmn = msel(back, mzero<double>(), rmn);
        }
        double RN, RP;
        bulk_jac(p, N, P, RN, RP);
        M2 A, B, C;
        node_row(c, cd, ej.lm, ej.rm, lmn, rmn, RN, RP, A, B, C);
        {
            const bool pad = nd >= L;
            A = msel(pad, mzero<double>(), A);
            C = msel(pad, mzero<double>(), C);
            B = msel(pad, mident<double>(), B);
        }
        // cut the coupling across the wave boundary; it comes back through the spike
        M2 cut = mzero<double>();
        if (W == 2) {
            if (wave == 0) {
                const bool last = lane == 63;
                cut = msel(last, C, cut);
                C = msel(last, mzero<double>(), C);
            } else {
                const bool first = lane == 0;
                cut = msel(first, A, cut);
                A = msel(first, mzero<double>(), A);
            }
        }
        M2 Ak = A, Bk = B, Ck = C;
        pcr_factor<0>(Ak, Bk, Ck);
        Bfin = minv(Bk);
        if (W == 2) {
            M2 rk = cut;
            pcr_solve_m<0>(rk);
            S = mmul(Bfin, rk);
            const M2 Sb = M2{lane_value(S.a, bl), lane_value(S.b, bl), lane_value(S.c, bl), lane_value(S.d, bl)};
            const double mine[4] = {Sb.a, Sb.b, Sb.c, Sb.d};
            double th[4];
            xchg<4>(mine, th);
            const M2 So = M2{th[0], th[1], th[2], th[3]};
            S0b = msel(wave == 0, Sb, So);
            S1b = msel(wave == 0, So, Sb);
            K = minv(msub(mident<double>(), mmul(S0b, S1b)));
        }
    }
    template <int LV>
    __device__ inline void pcr_factor(M2 &Ak, M2 &Bk, M2 &Ck)
    {
        if constexpr (LV < 6) {
            constexpr int s = 1 << LV;
            const M2 Bi = minv(Bk);
            const M2 Bim = mdn<s>(Bi), Bip = mup<s>(Bi);
            const M2 Akm = mdn<s>(Ak), Akp = mup<s>(Ak);
            const M2 Ckm = mdn<s>(Ck), Ckp = mup<s>(Ck);
            const M2 al = mmul(Ak, Bim), ga = mmul(Ck, Bip);
            if constexpr (LV < NREG) {
                alr[LV] = al;
                gar[LV] = ga;
            } else {
                put_lds(2 * (LV - NREG), al);
                put_lds(2 * (LV - NREG) + 1, ga);
            }
            Bk = msub(msub(Bk, mmul(al, Ckm)), mmul(ga, Akp));
            Ak = mneg(mmul(al, Akm));
            Ck = mneg(mmul(ga, Ckp));
            pcr_factor<LV + 1>(Ak, Bk, Ck);
        }
    }
    template <int LV> __device__ inline M2 al_of() const
    {
        if constexpr (LV < NREG) return alr[LV]; else return get_lds(2 * (LV - NREG));
    }
    template <int LV> __device__ inline M2 ga_of() const
    {
        if constexpr (LV < NREG) return gar[LV]; else return get_lds(2 * (LV - NREG) + 1);
    }
    template <int LV>
    __device__ inline void pcr_solve_m(M2 &rk) const
    {
        if constexpr (LV < 6) {
            constexpr int s = 1 << LV;
            const M2 rm = mdn<s>(rk), rp = mup<s>(rk);
            rk = msub(msub(rk, mmul(al_of<LV>(), rm)), mmul(ga_of<LV>(), rp));
            pcr_solve_m<LV + 1>(rk);
        }
    }
    template <int LV>
    __device__ inline void pcr_solve_v(V2 &rk) const
    {
        if constexpr (LV < 6) {
            constexpr int s = 1 << LV;
            const V2 rm = vdn<s>(rk), rp = vup<s>(rk);
            const V2 u1 = mvec(al_of<LV>(), rm), u2 = mvec(ga_of<LV>(), rp);
            rk = V2{rk.x - u1.x - u2.x, rk.y - u1.y - u2.y};
            pcr_solve_v<LV + 1>(rk);
        }
    }

    // x = M^{-1} b; b, x = (N, P, E) of this lane's node
    __device__ inline void apply(const double (&b)[V], double (&x)[V])
    {
        const double bE = b[2];
        const V2 wb{w0 * bE, w1 * bE};
        V2 wbn = vnext(wb);  // lane 63 gets 0 ...
        if (W == 2 && wave == 0) {
            // ... except in wave 0, whose last node borders edge 64: bE_64 = h f_E64 - (z1_E64 + acor_E64)
            const double bEn = hcur * fE_nb - nbZ;
            const bool last = lane == 63;
            wbn.x = last ? wn0 * bEn : wbn.x;
            wbn.y = last ? wn1 * bEn : wbn.y;
        }
        V2 rk{b[0] + cd * (wbn.x - wb.x), b[1] - cd * (wbn.y - wb.y)};
        pcr_solve_v<0>(rk);
        V2 xv = mvec(Bfin, rk);
        V2 x63{0.0, 0.0};
        if (W == 2) {
            const double mine[2] = {xv.x, xv.y};
            double th[2];
            xchg<2>(mine, th);
            const V2 gown{lane_value(xv.x, bl), lane_value(xv.y, bl)}, goth{th[0], th[1]};
            const V2 g63 = vsel(wave == 0, gown, goth), g64 = vsel(wave == 0, goth, gown);
            const V2 t_ = mvec(S0b, g64);
            x63 = mvec(K, V2{g63.x - t_.x, g63.y - t_.y});
            const V2 u_ = mvec(S1b, x63);
            const V2 x64{g64.x - u_.x, g64.y - u_.y};
            const V2 xo = vsel(wave == 0, x64, x63);
            const V2 c_ = mvec(S, xo);
            xv = V2{xv.x - c_.x, xv.y - c_.y};
        }
        V2 xp = vprev(xv);
        if (W == 2) {
            const bool first = (wave == 1) && (lane == 0);
            xp.x = first ? x63.x : xp.x;
            xp.y = first ? x63.y : xp.y;
        }
        x[0] = xv.x;
        x[1] = xv.y;
        // back-substitute the eliminated field unknown of this lane's left edge
        x[2] = bE * mv - (e0 * xp.x + e1 * xv.x + e2 * xp.y + e3 * xv.y);
    }
};

// MINW: waves per SIMD the register budget must allow (1: latency build, blocks in registers;
// 2: two solves' waves share a SIMD in a full launch)
template <int W, int NREG, int MINW>
__global__ __launch_bounds__(64 * W, MINW) void solve_kernel_pair(const mtrpl_solve_desc *__restrict__ descs,
                                                                   const double *__restrict__ ini,
                                                                   const double *__restrict__ tsteps,
                                                                   double *__restrict__ out,
                                                                   int32_t *__restrict__ status,
                                                                   int32_t *__restrict__ stats, int64_t B,
                                                                   SolveTuning tune)
{
    using BK = PairBackend<W, NREG>;
    const int64_t sidx = blockIdx.x;
    const unsigned long long clk0_ = __builtin_amdgcn_s_memtime();
    if (sidx >= B) return;
    __shared__ double2 blk_lds[BK::kBlkBytes > 0 ? BK::kBlkBytes / 16 : 1];
    __shared__ double2 xs_lds[BK::kXchBytes / 16];
    BK bk(threadIdx.x, blk_lds, reinterpret_cast<double *>(xs_lds));
    integrate(bk, descs[sidx], ini, tsteps, out, status, stats, sidx, tune, clk0_);
}

}  // namespace mtrpl
