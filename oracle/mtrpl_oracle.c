/*
 * mtrpl_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, no FMA contraction) of the arithmetic on
 * MetroTRPL's likelihood hot path.  It is the CHECKER for the HIP kernels in
 * metrotrpl_amd/csrc/; nothing in the shipped product path may include, link,
 * dlopen or call anything in this file.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it.
 *
 * Every function cites the reference lines it follows (paths relative to the
 * reference checkout).  Operation ORDER follows the reference expressions so
 * that results agree with the reference's numpy/Numba arithmetic to the last
 * bit wherever numpy itself is sequential (compile with -ffp-contract=off).
 *
 * The time integrator of the reference is NOT in the reference tree: it is
 * SciPy's `solve_ivp(method="LSODA")` / `odeint` (ODEPACK), pinned
 * scipy==1.12.0 in Requirements/requirements_win_basic.txt:14.  The oracle
 * drives the *same* SciPy LSODA (oracle/oracle.py) with the RHS functions
 * below through a ctypes trampoline, i.e. exactly the structure of
 * forward_solver.py:142-165 with Numba's JIT replaced by gcc.
 *
 * Parity pin: tests/test_oracle_golden.py checks these functions against
 * vectors produced by importing the reference itself (tests/golden/).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* forward_solver.py:21-24 */
static const double eps0 = 8.854 * 1e-12 * 1e-9;
static const double q_C = 1.602e-19;
static const double kB = 8.61773e-5;

double mtrpl_oracle_eps0(void) { return eps0; }
double mtrpl_oracle_q_C(void) { return q_C; }
double mtrpl_oracle_kB(void) { return kB; }

/* forward_solver.py:332-372 (dydt_numba); numpy twin :277-329.
 * y = [N(L), P(L), E(L+1)], dy same layout. */
void mtrpl_oracle_dydt_std(double t, const double *y, double *dy, int L,
                           double dx, double N0, double P0, double mu_n,
                           double mu_p, double r_rad, double CN, double CP,
                           double sr0, double srL, double tauN, double tauP,
                           double Lambda, double Tm)
{
    (void)t;
    const double *N = y, *P = y + L, *Ef = y + 2 * L;
    double *Jn = (double *)calloc((size_t)(L + 1), sizeof(double));
    double *Jp = (double *)calloc((size_t)(L + 1), sizeof(double));
    int i;

    /* :344-352 */
    double NP0 = N[0] * P[0] - N0 * P0;
    double NPL = N[L - 1] * P[L - 1] - N0 * P0;
    double Sft = sr0 * NP0 / (N[0] + P[0]);
    double Sbt = srL * NPL / (N[L - 1] + P[L - 1]);
    Jn[0] = Sft;
    Jn[L] = -Sbt;
    Jp[0] = -Sft;
    Jp[L] = Sbt;

    /* :355-359 */
    for (i = 1; i < L; i++) {
        Jn[i] = mu_n * ((N[i - 1] + N[i]) / 2 * Ef[i]) +
                mu_n * kB * Tm * ((N[i] - N[i - 1]) / dx);
        Jp[i] = mu_p * ((P[i - 1] + P[i]) / 2 * Ef[i]) -
                mu_p * kB * Tm * ((P[i] - P[i - 1]) / dx);
    }
    /* :362-363 */
    for (i = 0; i < L + 1; i++)
        dy[2 * L + i] = -(Jn[i] + Jp[i]) * Lambda;

    /* :366-370 */
    for (i = 0; i < L; i++) {
        double NP = N[i] * P[i] - N0 * P0;
        double recomb =
            ((CN * N[i] + CP * P[i]) + r_rad + 1 / ((tauN * P[i]) + (tauP * N[i]))) * NP;
        dy[i] = ((Jn[i + 1] - Jn[i]) / dx - recomb);
        dy[L + i] = (-(Jp[i + 1] - Jp[i]) / dx - recomb);
    }
    free(Jn);
    free(Jp);
}

/* forward_solver.py:374-418 (dydt_numba_traps).
 * y = [N(L), N_trap(L), P(L), E(L+1)]. */
void mtrpl_oracle_dydt_traps(double t, const double *y, double *dy, int L,
                             double dx, double N0, double P0, double mu_n,
                             double mu_p, double r_rad, double CN, double CP,
                             double sr0, double srL, double tauN, double tauP,
                             double Lambda, double Tm, double kC, double Nt,
                             double tauE)
{
    (void)t;
    const double *N = y, *Ntr = y + L, *P = y + 2 * L, *Ef = y + 3 * L;
    double *Jn = (double *)calloc((size_t)(L + 1), sizeof(double));
    double *Jp = (double *)calloc((size_t)(L + 1), sizeof(double));
    int i;

    double NP0 = N[0] * P[0] - N0 * P0;
    double NPL = N[L - 1] * P[L - 1] - N0 * P0;
    double Sft = sr0 * NP0 / (N[0] + P[0]);
    double Sbt = srL * NPL / (N[L - 1] + P[L - 1]);
    Jn[0] = Sft;
    Jn[L] = -Sbt;
    Jp[0] = -Sft;
    Jp[L] = Sbt;

    for (i = 1; i < L; i++) {
        Jn[i] = mu_n * ((N[i - 1] + N[i]) / 2 * Ef[i]) +
                mu_n * kB * Tm * ((N[i] - N[i - 1]) / dx);
        Jp[i] = mu_p * ((P[i - 1] + P[i]) / 2 * Ef[i]) -
                mu_p * kB * Tm * ((P[i] - P[i - 1]) / dx);
    }
    for (i = 0; i < L + 1; i++)
        dy[3 * L + i] = -(Jn[i] + Jp[i]) * Lambda;

    for (i = 0; i < L; i++) {
        double NP = N[i] * P[i] - N0 * P0;
        double recomb =
            ((CN * N[i] + CP * P[i]) + r_rad + 1 / ((tauN * P[i]) + (tauP * N[i]))) * NP;
        double trap = kC * N[i] * (Nt - Ntr[i]);
        double detrap = Ntr[i] / tauE;
        dy[i] = ((Jn[i + 1] - Jn[i]) / dx - recomb + detrap - trap);
        dy[L + i] = trap - detrap;
        dy[2 * L + i] = (-(Jp[i + 1] - Jp[i]) / dx - recomb);
    }
    free(Jn);
    free(Jp);
}

/* forward_solver.py:26-38 (E_field), one row. E has nx+1 entries. */
void mtrpl_oracle_E_field(const double *N, const double *P, int nx, double n0,
                          double p0, double eps, double dx, double corner_E,
                          double *E)
{
    double pref = q_C / (eps * eps0) * dx;
    double cs = 0.0;
    int j;
    E[0] = corner_E;
    for (j = 0; j < nx; j++) {
        double term = ((P[j] - p0) - (N[j] - n0));
        cs = (j == 0) ? term : cs + term; /* np.cumsum: sequential adds */
        E[j + 1] = corner_E + pref * cs;
    }
}

/* forward_solver.py:258-264 (integrate_1D) -- same summation order. */
double mtrpl_oracle_integrate_1D(double dx, const double *y, int n)
{
    double y_int = y[0] * dx / 2;
    int i;
    for (i = 1; i < n; i++)
        y_int += dx * (y[i] + y[i - 1]) / 2;
    y_int += y[n - 1] * dx / 2;
    return y_int;
}

/* forward_solver.py:228-236, :267-269 (calculate_PL / calculate_RR),
 * N,P row-major [nt, nx]. */
void mtrpl_oracle_calculate_PL(double dx, const double *N, const double *P,
                               int nt, int nx, double ks, double n0, double p0,
                               double *out)
{
    double *rr = (double *)malloc(sizeof(double) * (size_t)nx);
    int k, i;
    for (k = 0; k < nt; k++) {
        for (i = 0; i < nx; i++)
            rr[i] = ks * (N[(size_t)k * nx + i] * P[(size_t)k * nx + i] - n0 * p0);
        out[k] = mtrpl_oracle_integrate_1D(dx, rr, nx);
    }
    free(rr);
}

/* forward_solver.py:239-247, :272-274 (calculate_TRTS / calculate_photoc). */
void mtrpl_oracle_calculate_TRTS(double dx, const double *N, const double *P,
                                 int nt, int nx, double mu_n, double mu_p,
                                 double n0, double p0, double *out)
{
    double *pc = (double *)malloc(sizeof(double) * (size_t)nx);
    int k, i;
    for (k = 0; k < nt; k++) {
        for (i = 0; i < nx; i++)
            pc[i] = q_C * (mu_n * (N[(size_t)k * nx + i] - n0) +
                           mu_p * (P[(size_t)k * nx + i] - p0));
        out[k] = mtrpl_oracle_integrate_1D(dx, pc, nx);
    }
    free(pc);
}

/* forward_solver.py:189-192 / :197-200: unit scale then min_y clamp, in place. */
void mtrpl_oracle_scale_clamp(double *sig, int nt, double scale, double min_y)
{
    int k, i_final = 0;
    for (k = 0; k < nt; k++)
        sig[k] *= scale;
    /* np.argmax(sig < min_y): first True, or 0 when none */
    for (k = 0; k < nt; k++)
        if (sig[k] < min_y) {
            i_final = k;
            break;
        }
    if (sig[i_final] < min_y)
        for (k = i_final; k < nt; k++)
            sig[k] = min_y;
}

/* laplace.py:170-212 (convolve); n = len(resampled_y), I_table [nk_irf,3],
 * h_k has (n-1)/2 + 1 entries.  Summation is sequential over the window
 * (numpy's np.sum is pairwise, so agreement is to rounding, not to the bit). */
void mtrpl_oracle_convolve(const double *ry, int n, const double *I_table,
                           int nk_irf, double *h_k)
{
    int nk = (n - 1) / 2;
    int k, j;
    h_k[0] = 0.0;
    for (k = 1; k <= nk; k++) {
        int lo = k - nk_irf;
        double acc = 0.0;
        if (lo < 0)
            lo = 0;
        for (j = lo; j < k; j++) {
            double I2 = 2 * (ry[2 * j + 2] - 2 * ry[2 * j + 1] + ry[2 * j]);
            double I1 = (ry[2 * j + 2] - ry[2 * j]);
            double I0 = ry[2 * j + 1];
            const double *T = I_table + 3 * (size_t)(k - 1 - j);
            acc += (I0 * T[0] + I1 * T[1] + I2 * T[2]);
        }
        h_k[k] = acc;
    }
}

/* scipy.integrate.simpson (SciPy 1.12-1.15, _quadrature.py) for uniformly
 * spaced samples, dx given.  Even sample counts use the Cartwright
 * last-interval correction.  Used by I_moment (laplace.py:166). */
static double simpson_uniform(const double *y, int N, double dx)
{
    double result = 0.0;
    int i;
    if (N % 2 == 0) {
        if (N == 2)
            return 0.5 * dx * (y[1] + y[0]);
        for (i = 0; i < N - 3; i += 2)
            result += y[i] + 4.0 * y[i + 1] + y[i + 2];
        result *= dx / 3.0;
        {
            double h0 = dx, h1 = dx;
            double alpha = (2 * h1 * h1 + 3 * h0 * h1) / (6 * (h1 + h0));
            double beta = (h1 * h1 + 3.0 * h0 * h1) / (6 * h0);
            double eta = (1 * h1 * h1 * h1) / (6 * h0 * (h0 + h1));
            result += alpha * y[N - 1] + beta * y[N - 2] - eta * y[N - 3];
        }
        return result;
    }
    for (i = 0; i < N - 2; i += 2)
        result += y[i] + 4.0 * y[i + 1] + y[i + 2];
    result *= dx / 3.0;
    return result;
}

/* numpy.linspace(a, b, num)[i] */
static double linspace_at(double a, double b, int num, int i)
{
    double step = (b - a) / (double)(num - 1);
    if (i == num - 1)
        return b;
    return a + (double)i * step;
}

/* laplace.py:129-167 (I_moment) with u_lower=0, u_upper=1. */
double mtrpl_oracle_I_moment(const double *t, const double *y, int m, int n,
                             int u_spacing)
{
    double dt = t[1] - t[0];
    double *v = (double *)malloc(sizeof(double) * (size_t)u_spacing);
    double du = linspace_at(0.0, 1.0, u_spacing, 1) - linspace_at(0.0, 1.0, u_spacing, 0);
    double integral;
    int i;
    for (i = 0; i < u_spacing; i++) {
        double u = linspace_at(0.0, 1.0, u_spacing, i);
        double yi = linspace_at(y[m + 1], y[m], u_spacing, i);
        double w = 1.0;
        int p;
        for (p = 0; p < n; p++)
            w *= (u - 0.5);
        if (n == 0)
            w = 1.0;
        v[i] = w * yi;
    }
    integral = dt * simpson_uniform(v, u_spacing, du);
    free(v);
    return integral;
}

/* laplace.py:13-41 (make_I_tables) for one IRF: table [nk,3], last row 0. */
void mtrpl_oracle_make_I_table(const double *t, const double *y, int nk,
                               double *table)
{
    int m, n;
    memset(table, 0, sizeof(double) * 3 * (size_t)nk);
    for (m = 0; m < nk - 1; m++)
        for (n = 0; n < 3; n++)
            table[3 * m + n] = mtrpl_oracle_I_moment(t, y, m, n, 1000);
}

/* 1-D linear interpolation as scipy.interpolate.griddata(points 1-D) does it:
 * interp1d(kind="linear", bounds_error=False, fill_value=nan) on sorted x.
 * (laplace.py:74 and :121).  x strictly increasing. */
void mtrpl_oracle_interp_linear(const double *x, const double *y, int n,
                                const double *xq, int nq, double *yq)
{
    int q;
    for (q = 0; q < nq; q++) {
        double v = xq[q];
        int lo = 0, hi = n, idx;
        double slope;
        if (!(v >= x[0] && v <= x[n - 1])) { /* also catches NaN */
            yq[q] = NAN;
            continue;
        }
        /* np.searchsorted(x, v, side="left") */
        while (lo < hi) {
            int mid = (lo + hi) / 2;
            if (x[mid] < v)
                lo = mid + 1;
            else
                hi = mid;
        }
        idx = lo;
        if (idx < 1)
            idx = 1;
        if (idx > n - 1)
            idx = n - 1;
        slope = (y[idx] - y[idx - 1]) / (x[idx] - x[idx - 1]);
        yq[q] = slope * (v - x[idx - 1]) + y[idx - 1];
    }
}

/* trial_move_evaluation.py:117-130, :147-156: negative-fraction check, sign
 * flip, err_sq and ll(T) for one measurement.  sol is modified in place
 * (negatives flipped).  Returns 0 ok, 1 too many negatives.
 * ll_out[j] = -sum(err_sq / (sigma^2 * T[j] + 2 u^2)). */
int mtrpl_oracle_loglik(double *sol, const double *vals, const double *uncs,
                        int n, double scale_shift, double sigma,
                        const double *T, int nT, double neg_frac_tol,
                        double *ll_out)
{
    int i, j, n_fails = 0;
    for (i = 0; i < n; i++)
        if (sol[i] < 0)
            n_fails++;
    if (!((double)n_fails < neg_frac_tol * (double)n))
        return 1;
    for (i = 0; i < n; i++)
        if (sol[i] < 0)
            sol[i] *= -1;
    for (j = 0; j < nT; j++) {
        double acc = 0.0;
        for (i = 0; i < n; i++) {
            double e = (log10(sol[i]) + scale_shift - vals[i]);
            double err_sq = e * e;
            acc += err_sq / (sigma * sigma * T[j] + 2 * (uncs[i] * uncs[i]));
        }
        ll_out[j] = -acc;
    }
    return 0;
}
