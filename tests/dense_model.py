"""Dense numpy restatement of the std model for the linear-algebra tests: right-hand side (forward_solver.py:332-372),
its analytic Jacobian as a dense matrix, initial condition (forward_solver.py:100-122) and the PL signal.  Test
infrastructure only: the kernels' block-tridiagonal Newton solver is checked against np.linalg.solve on this."""
import numpy as np

kB = 8.61773e-5                 # forward_solver.py:21-24
q_C = 1.602e-19
eps0 = 8.854 * 1e-12 * 1e-9


class Problem:
    def __init__(self, state_user, units, ix, iniPar, thickness, nx):
        s = np.asarray(state_user, float) * units
        self.nx = nx
        self.dx = thickness / nx
        g = lambda n: s[ix[n]]
        self.n0, self.p0, self.mun, self.mup = g("n0"), g("p0"), g("mu_n"), g("mu_p")
        self.ks, self.Cn, self.Cp, self.Sf, self.Sb = g("ks"), g("Cn"), g("Cp"), g("Sf"), g("Sb")
        self.tn, self.tp, self.Tm = g("tauN"), g("tauP"), g("Tm")
        self.Lam = q_C / (g("eps") * eps0)
        x = np.linspace(self.dx / 2, thickness - self.dx / 2, nx)
        if len(iniPar) <= 3:
            fl, al = iniPar[0] * 1e-14, iniPar[1] * 1e-7
            dN = fl * al * np.exp(-al * x)
            if len(iniPar) > 2 and int(iniPar[2]) < 0:
                dN = dN[::-1]
        else:
            dN = np.asarray(iniPar) * 1e-21
        N = dN + self.n0
        P = dN + self.p0
        E = np.concatenate([[0.0], self.Lam * self.dx * np.cumsum((P - self.p0) - (N - self.n0))])
        self.y0 = np.concatenate([N, P, E])

    def f(self, y):
        L, dx = self.nx, self.dx
        N, P, E = y[:L], y[L:2 * L], y[2 * L:]
        NP = N * P - self.n0 * self.p0
        Jn = np.zeros(L + 1)
        Jp = np.zeros(L + 1)
        Sft = self.Sf * NP[0] / (N[0] + P[0])
        Sbt = self.Sb * NP[-1] / (N[-1] + P[-1])
        Jn[0], Jn[L], Jp[0], Jp[L] = Sft, -Sbt, -Sft, Sbt
        Dn, Dp = self.mun * kB * self.Tm, self.mup * kB * self.Tm
        Jn[1:L] = self.mun * 0.5 * (N[:-1] + N[1:]) * E[1:L] + Dn * (N[1:] - N[:-1]) / dx
        Jp[1:L] = self.mup * 0.5 * (P[:-1] + P[1:]) * E[1:L] - Dp * (P[1:] - P[:-1]) / dx
        R = ((self.Cn * N + self.Cp * P) + self.ks + 1 / (self.tn * P + self.tp * N)) * NP
        dN = (Jn[1:] - Jn[:-1]) / dx - R
        dP = -(Jp[1:] - Jp[:-1]) / dx - R
        dE = -(Jn + Jp) * self.Lam
        return np.concatenate([dN, dP, dE])

    def jac(self, y):
        L, dx = self.nx, self.dx
        S = 3 * L + 1
        N, P, E = y[:L], y[L:2 * L], y[2 * L:]
        J = np.zeros((S, S))
        iN = lambda i: i
        iP = lambda i: L + i
        iE = lambda i: 2 * L + i
        Dn, Dp = self.mun * kB * self.Tm, self.mup * kB * self.Tm
        # edge fluxes i=1..L-1
        dJn = {}
        dJp = {}
        for i in range(1, L):
            dJn[i] = {iN(i - 1): 0.5 * self.mun * E[i] - Dn / dx, iN(i): 0.5 * self.mun * E[i] + Dn / dx,
                      iE(i): self.mun * 0.5 * (N[i - 1] + N[i])}
            dJp[i] = {iP(i - 1): 0.5 * self.mup * E[i] + Dp / dx, iP(i): 0.5 * self.mup * E[i] - Dp / dx,
                      iE(i): self.mup * 0.5 * (P[i - 1] + P[i])}
        NP0 = N[0] * P[0] - self.n0 * self.p0
        s0 = N[0] + P[0]
        dS0 = {iN(0): self.Sf * (P[0] * s0 - NP0) / s0 ** 2, iP(0): self.Sf * (N[0] * s0 - NP0) / s0 ** 2}
        NPL = N[-1] * P[-1] - self.n0 * self.p0
        sL = N[-1] + P[-1]
        dSL = {iN(L - 1): self.Sb * (P[-1] * sL - NPL) / sL ** 2, iP(L - 1): self.Sb * (N[-1] * sL - NPL) / sL ** 2}
        dJn[0] = dS0
        dJp[0] = {k: -v for k, v in dS0.items()}
        dJn[L] = {k: -v for k, v in dSL.items()}
        dJp[L] = dSL
        for i in range(L):
            for k, v in dJn[i + 1].items():
                J[iN(i), k] += v / dx
            for k, v in dJn[i].items():
                J[iN(i), k] -= v / dx
            for k, v in dJp[i + 1].items():
                J[iP(i), k] -= v / dx
            for k, v in dJp[i].items():
                J[iP(i), k] += v / dx
            NP = N[i] * P[i] - self.n0 * self.p0
            den = self.tn * P[i] + self.tp * N[i]
            k_ = (self.Cn * N[i] + self.Cp * P[i]) + self.ks + 1 / den
            dRdN = (self.Cn - self.tp / den ** 2) * NP + k_ * P[i]
            dRdP = (self.Cp - self.tn / den ** 2) * NP + k_ * N[i]
            for r in (iN(i), iP(i)):
                J[r, iN(i)] -= dRdN
                J[r, iP(i)] -= dRdP
        for i in range(1, L):
            for k, v in dJn[i].items():
                J[iE(i), k] -= v * self.Lam
            for k, v in dJp[i].items():
                J[iE(i), k] -= v * self.Lam
        return J

    def signal(self, y):
        L = self.nx
        return self.dx * np.sum(self.ks * (y[:L] * y[L:2 * L] - self.n0 * self.p0)) * 1e23
