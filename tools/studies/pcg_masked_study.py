"""Prototype of the iterative masked step: PCG with matricised Kronecker MVMs, Kronecker-eigen preconditioner P = I + rho p G1 (x) G2,
stochastic Lanczos quadrature for log det and probe estimators with control variates for the derivative traces.
Compares with the dense oracle (Kr.elbo_step_masked)."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.linalg as sla
from oracle import dense as D, kron as Kr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 24
nz = int(sys.argv[3]) if len(sys.argv) > 3 else 16
kind = sys.argv[4] if len(sys.argv) > 4 else "matern12"
basis = "b0" if kind == "matern12" else "points"
maxit = 60
theta = np.array([0.2, 0.3, 1.0, 0.8, 0.01])
X, y, x1, x2 = D.gen_grid(n, n)
Y = y.reshape(n, n)
W = (np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(float)
g = np.linspace(0, 1, m + 1 if basis == "b0" else m)
f1, f2 = Kr.Factor(basis, kind, g, x1), Kr.Factor(basis, kind, g, x2)
t0 = time.time(); ref = Kr.elbo_step_masked(Y, W, f1, f2, theta); print("dense", time.time() - t0, "s  elbo", ref.elbo)

ell1, ell2, s1, s2, v = theta
d1, d2 = Kr.dim_prepare(f1, ell1, 1.0), Kr.dim_prepare(f2, ell2, 1.0)
B1, V1, B2, V2 = d1.B, d1.V, d2.B, d2.V
m1, m2 = B1.shape[0], B2.shape[0]; M = m1 * m2
Ym = Y * W; N = int(W.sum()); yy = float((Ym * Ym).sum())
rho = s1 * s2 / v
Wt = W.T                                  # [i, j]
p = N / (n * n)

def field(L, V, R):                       # F[i, j] = l_i^T V r_j for every grid point: (n1 x n2); V batched [..., m1, m2]
    return np.einsum("ai,...ab,bj->...ij", L, V, R, optimize=True)
def back(L, F, R):                        # sum_ij F[i,j] l_i r_j^T
    return np.einsum("ai,...ij,bj->...ab", L, F, R, optimize=True)
def Aop(V):                               # Sigma~ V
    return V + rho * back(B1, Wt * field(B1, V, B2), B2)

lam1, Q1 = np.linalg.eigh(B1 @ B1.T); lam2, Q2 = np.linalg.eigh(B2 @ B2.T)
dP = 1.0 + rho * p * np.outer(lam1, lam2)
def Pinv(V):  return Q1 @ ((Q1.T @ V @ Q2) / dP) @ Q2.T
def Phalf(V): return Q1 @ ((Q1.T @ V @ Q2) * np.sqrt(dP)) @ Q2.T
def Pinvhalf(V): return Q1 @ ((Q1.T @ V @ Q2) / np.sqrt(dP)) @ Q2.T

rng = np.random.default_rng(12345)
Z0 = rng.choice([-1.0, 1.0], size=(nz, m1, m2))
Zs = Phalf(Z0)                             # z ~ (0, P)
c0 = B1 @ Ym.T @ B2.T
RHS = np.concatenate([c0[None], Zs])       # block of nz + 1 right-hand sides

def dots(A, B): return (A * B).sum(axis=(1, 2))
# block PCG with Lanczos coefficients
Xs = np.zeros_like(RHS); R = RHS.copy(); Zp = Pinv(R); Pd = Zp.copy(); rz = dots(R, Zp)
alphas, betas = [], []
r0 = np.sqrt(dots(R, R))
for it in range(maxit):
    AP = Aop(Pd)
    al = rz / dots(Pd, AP)
    Xs += al[:, None, None] * Pd
    R -= al[:, None, None] * AP
    Zp = Pinv(R)
    rz_new = dots(R, Zp)
    be = rz_new / rz
    alphas.append(al); betas.append(be)
    Pd = Zp + be[:, None, None] * Pd
    rz = rz_new
    rel = np.sqrt(dots(R, R)) / r0
    if rel.max() < 1e-10: break
print("PCG iterations", it + 1, "max rel resid", rel.max())
al = np.array(alphas); be = np.array(betas); k = al.shape[0]
# Lanczos tridiagonals of P^-1/2 Sigma~ P^-1/2 started at z0 / |z0|
ld_slq = 0.0
for zi in range(nz):
    a, b = al[:, 1 + zi], be[:, 1 + zi]
    T = np.zeros((k, k))
    for j in range(k):
        T[j, j] = 1 / a[j] + (b[j - 1] / a[j - 1] if j > 0 else 0.0)
        if j + 1 < k: T[j, j + 1] = T[j + 1, j] = math.sqrt(b[j]) / a[j]
    w, U = np.linalg.eigh(T)
    ld_slq += (Z0[zi] ** 2).sum() * float((U[0] ** 2) @ np.log(w))
ld_slq /= nz
logdet = float(np.log(dP).sum()) + ld_slq
a0 = Xs[0]; q = float((c0 * a0).sum())
# exact pieces
nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)
trPhi = float(nb1 @ Wt @ nb2)
elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + logdet + yy / v - (s1 * s2 / v ** 2) * q) - (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v))
print("elbo iter", elbo, "rel err", abs(elbo - ref.elbo) / abs(ref.elbo), " logdet err abs", logdet - (2 * np.log(np.diag(np.linalg.cholesky(np.eye(M) + rho * Kr._assemble(B1, B1, B2, B2, W)))).sum()))

# ---- trace terms: tr(Sigma~^-1 D) = tr(P^-1 D) [exact] + E[(u - w)^T D w],  u = Sigma~^-1 z, w = P^-1 z
U = Xs[1:]; Wz = Pinv(Zs); dU = U - Wz
R1, R2 = Q1.T @ B1, Q2.T @ B2; RV1, RV2 = Q1.T @ V1, Q2.T @ V2
iD = 1.0 / dP
def tr_Pinv_phi(Ra, Rb, Sa, Sb):           # tr(P^-1 assemble(P1a, P1b, P2a, P2b)) with rotated factors
    return float((iD * ((Ra * Rb) @ Wt @ (Sa * Sb).T)).sum())
def est(La, Lb, Ra, Rb):                   # E[(u-w)^T Phi(La,Lb;Ra,Rb) w], Phi V = La (Wt o (Lb^T V Rb)) Ra^T
    Fu = field(La, dU, Ra); Fw = field(Lb, Wz, Rb)
    return float((Wt * Fu * Fw).sum()) / nz
trSP = tr_Pinv_phi(R1, R1, R2, R2) + est(B1, B1, B2, B2)
# symmetric derivative: Phi' + Phi'^T
trS1 = 2 * tr_Pinv_phi(R1, RV1, R2, R2) + est(B1, V1, B2, B2) + est(V1, B1, B2, B2)
trS2 = 2 * tr_Pinv_phi(R1, R1, R2, RV2) + est(B1, B1, B2, V2) + est(B1, B1, V2, B2)
# Mk terms: tr(Sigma~^-1 (Mk1 (x) I)) = tr(P^-1 (Mk1 x I)) + E[(u-w)^T (Mk1 W)]
def tr_Mk(Mk, dim):
    if dim == 1:
        ex = float((iD * np.diag(Q1.T @ Mk @ Q1)[:, None]).sum())
        st = float(sum((dU[z] * (Mk @ Wz[z])).sum() for z in range(nz))) / nz
    else:
        ex = float((iD * np.diag(Q2.T @ Mk @ Q2)[None, :]).sum())
        st = float(sum((dU[z] * (Wz[z] @ Mk.T)).sum() for z in range(nz))) / nz
    return ex + st
A0 = a0
aPa = (q - float((a0 * a0).sum())) / rho
common = -0.5 * (rho * trSP - (s1 * s2 / v ** 2) * q + (s1 * s2 / v ** 2) * rho * aPa)
g_s1 = common / s1 - (N * s2 - s2 * trPhi) / (2 * v)
g_s2 = common / s2 - (N * s1 - s1 * trPhi) / (2 * v)
g_v = (-0.5 * (N / v - (rho / v) * trSP - yy / v ** 2 + 2 * s1 * s2 * q / v ** 3 - (s1 * s2 * rho / v ** 3) * aPa) + (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v ** 2))
def ell_grad(dim):
    if dim == 1:
        Mk, m_other, trS = d1.Mk, m2, trS1
        C1 = V1 @ Ym.T @ B2.T
        quadMk = np.einsum("ik,ij,kj->", Mk, A0, A0)
        Z = float((W * ((B2.T @ A0.T @ V1) * (B2.T @ A0.T @ B1))).sum())
        hv = (V1 * B1).sum(0); tr1 = float(hv @ (W.T @ nb2)); PT = (B1 * (W.T @ nb2)[None, :]) @ B1.T
    else:
        Mk, m_other, trS = d2.Mk, m1, trS2
        C1 = B1 @ Ym.T @ V2.T
        quadMk = np.einsum("ik,ji,jk->", Mk, A0, A0)
        Z = float((W * ((V2.T @ A0.T @ B1) * (B2.T @ A0.T @ B1))).sum())
        hv = (V2 * B2).sum(0); tr1 = float(hv @ (W @ nb1)); PT = (B2 * (W @ nb1)[None, :]) @ B2.T
    ld = tr_Mk(Mk, dim) - m_other * np.trace(Mk) + rho * trS
    quad = 2 * float((a0 * C1).sum()) - quadMk - 2 * rho * Z
    return -0.5 * (ld - (s1 * s2 / v ** 2) * quad) + (s1 * s2 / (2 * v)) * (2 * tr1 - float((Mk * PT.T).sum()))
grad = np.array([ell_grad(1), ell_grad(2), g_s1, g_s2, g_v])
print("grad iter ", grad); print("grad dense", ref.grad)
print("grad rel err (vs max)", np.abs(grad - ref.grad).max() / np.abs(ref.grad).max(), " per comp", np.abs(grad - ref.grad) / np.abs(ref.grad))
