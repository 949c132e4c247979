"""Thin (range-only) evaluation of the m-space ELBO + gradient for numerically rank-deficient Gram matrices (RBF):
only the r leading eigenpairs of G_d are used, the null block is treated as exactly zero.  Compares with oracle/kron.py's
full finish() at the headline configuration."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense as D, kron as Kr

n, m, kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 128, "rbf"
theta = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
X, y, x1, x2 = D.gen_grid(n, n)
Y = y.reshape(n, n)
g = np.linspace(0, 1, m)
f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
ref = Kr.elbo_step(Y, f1, f2, theta)
print("full elbo", ref.elbo, "grad", ref.grad)
ell1, ell2, s1, s2, v = theta
d1, d2 = ref.d1, ref.d2
N = n * n
pay = Kr.local_partials(Y, d1, d2)

def thin(cut, pad):
    out = {}
    Qs, ls, rs = [], [], []
    for d in (d1, d2):
        lam, Q = np.linalg.eigh(d.G)
        lam, Q = lam[::-1], Q[:, ::-1]
        r = int((lam > cut * lam[0]).sum()) + pad
        Qs.append(Q[:, :r]); ls.append(lam[:r]); rs.append(r)
    Q1, Q2 = Qs; l1, l2 = ls
    P = Q1.T @ pay["C"] @ Q2; P1 = Q1.T @ pay["C1"] @ Q2; P2 = Q1.T @ pay["C2"] @ Q2
    a = np.outer(l1, l2) / v; Dm = 1 + a; beta = P / Dm; invD = 1 / Dm
    tr1, tr2 = np.trace(d1.G), np.trace(d2.G)             # = sum of ALL eigenvalues
    yy = pay["yy"]
    elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + np.log(Dm).sum() + yy / v - (P * beta).sum() / v ** 2)
            - (N * s1 * s2 - tr1 * tr2) / (2 * v))
    def ell_grad(dd, Q, Pd, tr_other, rsum_m, rlam_m, Xm, Xlam, lam_self):
        # rsum_m = sum_i2 (1/D - 1) ; rlam_m = sum_i2 lam2 (1/D - 1): both vanish on null rows
        E = Q.T @ dd.Mk @ Q
        F = Q.T @ (dd.H + dd.H.T) @ Q
        e, f = np.diag(E), np.diag(F)
        trF = 2 * np.trace(dd.H)
        quad = 2 * (beta * Pd).sum() - (E * Xm).sum() - (F * Xlam).sum() / v
        return (-0.5 * ((e * rsum_m).sum() + ((f * rlam_m).sum() + tr_other * trF) / v - quad / v ** 2)
                + tr_other / (2 * v) * (trF - (e * lam_self).sum()))
    g1 = ell_grad(d1, Q1, P1, tr2, (invD - 1).sum(1), ((invD - 1) * l2[None, :]).sum(1), beta @ beta.T, (beta * l2[None, :]) @ beta.T, l1)
    g2 = ell_grad(d2, Q2, P2, tr1, (invD - 1).sum(0), ((invD - 1) * l1[:, None]).sum(0), beta.T @ beta, (beta * l1[:, None]).T @ beta, l2)
    saD = (a * invD).sum(); sb2 = (beta * beta).sum()
    gs1 = -0.5 * (saD - sb2 / v ** 2) / s1 + tr1 * tr2 / (2 * v * s1) - N * s2 / (2 * v)
    gs2 = -0.5 * (saD - sb2 / v ** 2) / s2 + tr1 * tr2 / (2 * v * s2) - N * s1 / (2 * v)
    gv = (-0.5 * (N / v - saD / v - yy / v ** 2 + (beta * beta * (2 + a)).sum() / v ** 3) + (N * s1 * s2 - tr1 * tr2) / (2 * v ** 2))
    return rs, elbo, np.array([g1, g2, gs1, gs2, gv])

for cut, pad in ((1e-14, 0), (1e-14, 4), (1e-13, 0), (1e-12, 0), (1e-10, 0)):
    rs, e, gr = thin(cut, pad)
    print(f"cut {cut:g} pad {pad}: ranks {rs}  elbo rel err {abs(e - ref.elbo) / abs(ref.elbo):.2e}  grad rel err {np.abs(gr - ref.grad).max() / np.abs(ref.grad).max():.2e}",
          "per-comp", np.abs(gr - ref.grad) / np.abs(ref.grad))
lam = np.linalg.eigvalsh(d1.G)[::-1]
print("lam1/lmax:", (lam / lam[0])[:40])
