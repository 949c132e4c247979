"""Structured (Kronecker) CPU twin of the HIP engine (TEST INFRASTRUCTURE / cpu_baseline "port").

Computes the SAME numbers as oracle/dense.py -- the collapsed ELBO of
kronecker_structure.py:249-278, q(v) of :825-849 and the posterior of :199-230 --
from the per-dimension factors only, never forming anything of size N x N or
M x N.  numpy float64 + LAPACK (scipy); the hyper-parameter
gradient is analytic (no autograd).  The HIP engine implements exactly this
algorithm; this file is its readable specification and the timed CPU baseline.

Layout: observations Y[j, i] = y(x1[i], x2[j]) stored [n2, n1] row-major (x1
fastest: src/utils/datagenerators.py:70-72).  Inducing index u = i1*m2 + i2
(kronecker_structure.py:805, :822).  theta = [ell1, ell2, s1, s2, sigma2].

Jitter policy: K_d = s_d * (kappa_d + eps_d I) with eps_d the first of
(0, 1e-8, 1e-7, 1e-6) for which the Cholesky of the unit-outputscale factor
succeeds (so K_d scales exactly with s_d; eps_d = 0 for the reference's
Matern-1/2 models).  oracle/dense.py applies the same policy.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional, Tuple

import numpy as np
import scipy.linalg as sla

JITTERS = (0.0, 1e-8, 1e-7, 1e-6)
NOISE_LOWER = 1e-4
SQ3, SQ5 = math.sqrt(3.0), math.sqrt(5.0)


# ----------------------------------------------------------------------------
# factor builders: value and d/d ell (unit outputscale; callers multiply by s)
# ----------------------------------------------------------------------------
def kappa_and_dell(kind: str, dist: np.ndarray, ell: float):
    r = dist / ell
    if kind == "matern12":
        e = np.exp(-r)
        return e, e * r / ell
    if kind == "matern32":
        a = SQ3 * r
        e = np.exp(-a)
        return (1 + a) * e, a * a * e / ell
    if kind == "matern52":
        a = SQ5 * r
        e = np.exp(-a)
        return (1 + a + a * a / 3) * e, (a * a / 3) * (1 + a) * e / ell
    if kind == "rbf":
        e = np.exp(-0.5 * r * r)
        return e, e * r * r / ell
    raise ValueError(kind)


def points_factor(kind, z, x, ell):
    """A0[k,p] = kappa(|z_k - x_p|/ell), dA0/dell  (kronecker_structure.py:318-319, :336-337)."""
    return kappa_and_dell(kind, np.abs(z[:, None] - x[None, :]), ell)


def b0_K(m: int, delta: float, ell: float):
    """Unit-outputscale Kuu_d and d/dell (kronecker_structure.py:723-739), stable form:
    r_k = exp(-k t) * 4 sinh^2(t/2), r_0 = 2 (expm1(-t) + t), t = delta/ell."""
    t = delta / ell
    k = np.arange(m, dtype=np.float64)
    r = np.exp(-k * t) * (4.0 * math.sinh(0.5 * t) ** 2)
    r[0] = 2.0 * (math.expm1(-t) + t)
    # dr_k/dell = (delta/ell^2) [(k-1)e^{-(k-1)t} + (k+1)e^{-(k+1)t} - 2k e^{-kt}]
    #           = (t/ell) e^{-kt} [k*4sinh^2(t/2) - 2 sinh(t)]
    dr = (t / ell) * np.exp(-k * t) * (k * 4.0 * math.sinh(0.5 * t) ** 2 - 2.0 * math.sinh(t))
    dr[0] = (2.0 * t / ell) * math.expm1(-t)
    idx = np.abs(np.arange(m)[:, None] - np.arange(m)[None, :])
    T, dT = r[idx], dr[idx]
    return ell * ell * T, 2.0 * ell * T + ell * ell * dT


def b0_K_f32(m: int, delta: float, ell: float):
    """Reference-literal variant: mesh/delta are float32 there, so c*delta is rounded to float32 before the
    float64 division (kronecker_structure.py:731-733 under torch's type promotion); three-term form."""
    df = np.float32(delta)
    k = np.arange(m)
    a0 = ((k - 1).astype(np.float32) * df).astype(np.float64)
    a1 = ((k + 1).astype(np.float32) * df).astype(np.float64)
    a2 = (k.astype(np.float32) * df).astype(np.float64)
    e0, e1, e2 = np.exp(-a0 / ell), np.exp(-a1 / ell), np.exp(-a2 / ell)
    r = e0 + e1 - 2 * e2
    dr = (e0 * a0 + e1 * a1 - 2 * e2 * a2) / ell ** 2
    t = float(df) / ell
    r[0] = 2 * (math.exp(-t) + t - 1)
    dr[0] = 2 * (math.exp(-t) * t / ell - t / ell)
    idx = np.abs(np.arange(m)[:, None] - np.arange(m)[None, :])
    T, dT = r[idx], dr[idx]
    return ell * ell * T, 2.0 * ell * T + ell * ell * dT


def b0_A(mesh: np.ndarray, x: np.ndarray, ell: float):
    """Unit-outputscale Kuf_d and d/dell (kronecker_structure.py:768-790).  Cell k is
    (mesh[k], mesh[k+1]] (searchsorted right=False), x <= mesh[0] counts as below."""
    a, b = mesh[:-1, None], mesh[1:, None]
    xx = x[None, :]
    ua, ub = np.abs(xx - a), np.abs(xx - b)
    ea, eb = np.exp(-ua / ell), np.exp(-ub / ell)
    E1, E2 = ell * ea, ell * eb
    dE1, dE2 = ea * (1 + ua / ell), eb * (1 + ub / ell)
    inside = (xx > a) & (xx <= b)
    sign = np.where(xx <= a, 1.0, -1.0)
    A = np.where(inside, 2 * ell - (E1 + E2), sign * (E1 - E2))
    dA = np.where(inside, 2 - (dE1 + dE2), sign * (dE1 - dE2))
    return A, dA


@dataclass
class Factor:
    """One dimension's inducing-feature description."""
    basis: str                  # "b0" | "points" | "vff" | "b1" | "one" (trivial factor for 1-D models)
    kind: str                   # matern12 | matern32 | matern52 | rbf
    grid: np.ndarray            # mesh (m+1 knots) for b0, inducing coords (m) for points, knots (m) for b1,
                                # [a, b, omega_0 .. omega_M] for vff (m = 2M + 1)
    x: np.ndarray               # the n_d unique observation coordinates along this dim
    f32_kdelta: bool = False    # reproduce the reference's float32 (k*delta) rounding (float32 mesh)

    @property
    def m(self) -> int:
        if self.basis == "one":
            return 1
        if self.basis == "vff":
            return 2 * (len(self.grid) - 3) + 1
        return len(self.grid) - 1 if self.basis == "b0" else len(self.grid)

    @property
    def inverse(self) -> bool:
        """Inter-domain bases whose Kuu_d scales with 1/s_d and whose Kuf_d carries no s_d (vff, b1): B = L^{-1} Kuf is
        the same sqrt(s) L0^{-1} A0 as for the kernel-evaluated bases, so ELBO, gradient and posterior are unchanged;
        only u-space quantities (q(v)) see L = L0 / sqrt(s) instead of sqrt(s) L0."""
        return self.basis in ("vff", "b1")

    def build(self, ell: float, x: Optional[np.ndarray] = None):
        """-> (K0, dK0, A0, dA0) at unit outputscale."""
        x = self.x if x is None else x
        if self.basis == "one":
            o = np.ones((1, len(x)))
            return np.ones((1, 1)), np.zeros((1, 1)), o, np.zeros_like(o)
        g = np.asarray(self.grid, dtype=np.float64)
        if self.basis == "b0":
            K, dK = (b0_K_f32 if self.f32_kdelta else b0_K)(len(g) - 1, float(g[1] - g[0]), ell)
            A, dA = b0_A(g, x, ell)
        elif self.basis == "vff":
            K, dK = vff_K(g[0], g[1], g[2:], ell)
            A, dA = vff_A(g[0], g[1], g[2:], x, ell)
        elif self.basis == "b1":
            K, dK = b1_K(g, ell)
            A, dA = b1_A(g, x)
        else:
            K, dK = points_factor(self.kind, g, g, ell)
            A, dA = points_factor(self.kind, g, x, ell)
        return K, dK, A, dA


def vff_K(a: float, b: float, om: np.ndarray, ell: float):
    """Unit-scale VFF Kuu factor (K_d = K0 / s) and dK0/dell: diag(alpha0) + beta0 beta0^T,
    alpha0 = (b-a)/4 * c * (1/ell + w^2 ell), c = 2 for w = 0 (kronecker_structure.py:400-462)."""
    M = len(om) - 1
    w = np.concatenate([om, om[1:]])
    c = np.ones(2 * M + 1)
    c[0] = 2.0
    alpha = (b - a) / 4.0 * c * (1.0 / ell + w * w * ell)
    dalpha = (b - a) / 4.0 * c * (-1.0 / (ell * ell) + w * w)
    beta = np.concatenate([np.ones(M + 1), np.zeros(M)])
    return np.diag(alpha) + np.outer(beta, beta), np.diag(dalpha)


def vff_A(a: float, b: float, om: np.ndarray, x: np.ndarray, ell: float):
    """VFF Kuf factor (no outputscale) and its ell-derivative (non-zero only outside [a, b))."""
    x = np.asarray(x, float)
    inside = (x >= a) & (x < b)
    xa = x - a
    r = np.minimum(np.abs(x - a), np.abs(x - b))
    e = np.exp(-r / ell)
    real = np.where(inside[None, :], np.cos(om[:, None] * xa[None, :]), e[None, :] * np.ones((len(om), 1)))
    imag = np.where(inside[None, :], np.sin(om[1:, None] * xa[None, :]), 0.0)
    dreal = np.where(inside[None, :], 0.0, (r / (ell * ell) * e)[None, :] * np.ones((len(om), 1)))
    return np.vstack([real, imag]), np.vstack([dreal, np.zeros_like(imag)])


def b1_K(mesh: np.ndarray, ell: float):
    """Unit-scale B1-spline Kuu factor (K_d = K0 / s): (A ell + B / ell + BC) / 2 (kronecker_structure.py:560-614)."""
    m = len(mesh)
    d = float(mesh[1] - mesh[0])
    off = (np.abs(np.arange(m)[:, None] - np.arange(m)[None, :]) == 1).astype(float)
    ends = np.zeros(m)
    ends[0] = ends[-1] = 1.0
    A = (2 / 3) * d * np.eye(m) + (1 / 6) * d * off - np.diag(ends) * (d / 3)
    B = (2 / d) * np.eye(m) - off / d - np.diag(ends) / d
    return (A * ell + B / ell + np.diag(ends)) / 2.0, (A - B / (ell * ell)) / 2.0


def b1_A(mesh: np.ndarray, x: np.ndarray):
    """Hat functions at x (bspline.py:24-112); no hyper-parameter."""
    x = np.asarray(x, float)
    v = np.asarray(mesh, float)
    d = v[1] - v[0]
    A = np.maximum(0.0, 1.0 - np.abs(x[None, :] - v[:, None]) / d)
    A[:, (x < v[0]) | (x > v[-1])] = 0.0
    # interval conventions of the reference: the left half hat excludes x = v1 (it is 0 there anyway), nothing else differs
    return A, np.zeros_like(A)


def chol_jitter(K0: np.ndarray) -> Tuple[np.ndarray, float]:
    for jit in JITTERS:
        try:
            L = np.linalg.cholesky(K0 + jit * np.eye(K0.shape[0]))
            if np.isfinite(L).all():
                return L, jit
        except np.linalg.LinAlgError:
            pass
    raise np.linalg.LinAlgError("factor not positive definite after jitter 1e-6")


# ----------------------------------------------------------------------------
# per-dimension "B-basis" quantities (everything that touches n_d-sized data)
# ----------------------------------------------------------------------------
@dataclass
class DimState:
    L: np.ndarray       # chol(s*(K0+eps I))
    jit: float
    B: np.ndarray       # L^{-1} A            (m x n)
    V: np.ndarray       # L^{-1} dA/dell      (m x n)
    Mk: np.ndarray      # L^{-1} dK/dell L^{-T}  (m x m)
    G: Optional[np.ndarray] = None    # B B^T   (summed over ranks)
    H: Optional[np.ndarray] = None    # V B^T
    lam: Optional[np.ndarray] = None
    Q: Optional[np.ndarray] = None


def dim_prepare(f: Factor, ell: float, s: float, cols: Optional[slice] = None) -> DimState:
    K0, dK0, A0, dA0 = f.build(ell)
    L0, jit = chol_jitter(K0)
    if f.inverse:                    # K_d = K0 / s, Kuf_d = A0: the same B, V, Mk, but L = L0 / sqrt(s)
        L = L0 / math.sqrt(s)
        A, dA, dK = A0, dA0, dK0 / s
    else:
        L = math.sqrt(s) * L0
        A, dA, dK = s * A0, s * dA0, s * dK0
    if cols is not None:
        A, dA = A[:, cols], dA[:, cols]
    B = sla.solve_triangular(L, A, lower=True)
    V = sla.solve_triangular(L, dA, lower=True)
    X = sla.solve_triangular(L, dK, lower=True)
    Mk = sla.solve_triangular(L, X.T, lower=True).T
    return DimState(L=L, jit=jit, B=B, V=V, Mk=Mk)


@dataclass
class StepState:
    theta: np.ndarray
    d1: DimState
    d2: DimState
    P: np.ndarray
    D: np.ndarray
    beta: np.ndarray
    N: int
    yy: float
    elbo: float = 0.0
    grad: np.ndarray = field(default_factory=lambda: np.zeros(5))


def local_partials(Y: np.ndarray, d1: DimState, d2: DimState):
    """What one rank contributes before the single all-reduce: the packed payload
    {G2, H2, C, C1, C2, yy} for the rows of Y it owns (dim-2 shard), plus the
    replicated dim-1 Gram pair.  Y: [n2_local, n1]."""
    S = Y.T @ np.vstack([d2.B, d2.V]).T          # (n1, 2 m2) -- the only pass over Y
    m2 = d2.B.shape[0]
    C = d1.B @ S[:, :m2]
    C1 = d1.V @ S[:, :m2]
    C2 = d1.B @ S[:, m2:]
    G2 = d2.B @ d2.B.T
    H2 = d2.V @ d2.B.T
    yy = float((Y * Y).sum())
    return dict(G2=G2, H2=H2, C=C, C1=C1, C2=C2, yy=yy)


def finish(theta, d1: DimState, d2: DimState, pay: dict, N: int) -> StepState:
    """m-space part: eig, rotations, ELBO and its analytic gradient (replicated on every rank)."""
    ell1, ell2, s1, s2, v = [float(t) for t in theta]
    d1.G, d1.H = d1.B @ d1.B.T, d1.V @ d1.B.T
    d2.G, d2.H = pay["G2"], pay["H2"]
    for d in (d1, d2):
        d.lam, d.Q = np.linalg.eigh(d.G)
    Q1, Q2, l1, l2 = d1.Q, d2.Q, d1.lam, d2.lam
    m1, m2 = len(l1), len(l2)
    P = Q1.T @ pay["C"] @ Q2
    P1 = Q1.T @ pay["C1"] @ Q2
    P2 = Q1.T @ pay["C2"] @ Q2
    a = np.outer(l1, l2) / v
    D = 1.0 + a
    beta = P / D
    yy = pay["yy"]
    sl1, sl2 = l1.sum(), l2.sum()
    elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + np.log(D).sum()
                    + yy / v - (P * beta).sum() / v ** 2)
            - (N * s1 * s2 - sl1 * sl2) / (2 * v))

    def ell_grad(dd: DimState, Pd, lam_other_sum, m_other, r, rlam, X, Xlam, lam_self):
        E = dd.Q.T @ dd.Mk @ dd.Q
        F = dd.Q.T @ (dd.H + dd.H.T) @ dd.Q
        e, f = np.diag(E), np.diag(F)
        quad = 2 * (beta * Pd).sum() - (E * X).sum() - (F * Xlam).sum() / v
        return (-0.5 * ((e * r).sum() + (f * rlam).sum() / v - m_other * e.sum() - quad / v ** 2)
                + lam_other_sum / (2 * v) * (f.sum() - (e * lam_self).sum()))

    invD = 1.0 / D
    g_ell1 = ell_grad(d1, P1, sl2, m2, invD.sum(1), (invD * l2[None, :]).sum(1),
                      beta @ beta.T, (beta * l2[None, :]) @ beta.T, l1)
    g_ell2 = ell_grad(d2, P2, sl1, m1, invD.sum(0), (invD * l1[:, None]).sum(0),
                      beta.T @ beta, (beta * l1[:, None]).T @ beta, l2)
    saD = (a * invD).sum()
    sb2 = (beta * beta).sum()
    g_s1 = -0.5 * (saD - sb2 / v ** 2) / s1 + sl1 * sl2 / (2 * v * s1) - N * s2 / (2 * v)
    g_s2 = -0.5 * (saD - sb2 / v ** 2) / s2 + sl1 * sl2 / (2 * v * s2) - N * s1 / (2 * v)
    g_v = (-0.5 * (N / v - saD / v - yy / v ** 2 + (beta * beta * (2 + a)).sum() / v ** 3)
           + (N * s1 * s2 - sl1 * sl2) / (2 * v ** 2))
    st = StepState(theta=np.asarray(theta, float), d1=d1, d2=d2, P=P, D=D, beta=beta, N=N, yy=yy)
    st.elbo = float(elbo)
    st.grad = np.array([g_ell1, g_ell2, g_s1, g_s2, g_v])
    return st


def elbo_step(Y: np.ndarray, f1: Factor, f2: Factor, theta, n_ranks: int = 1) -> StepState:
    """Full-grid ELBO + d/dtheta.  n_ranks>1 emulates the dim-2 (row) shard + one all-reduce."""
    ell1, ell2, s1, s2, v = [float(t) for t in theta]
    n2, n1 = Y.shape
    d1 = dim_prepare(f1, ell1, s1)
    pays = []
    bounds = np.linspace(0, n2, n_ranks + 1).astype(int)
    d2_full = None
    for r in range(n_ranks):
        sl = slice(bounds[r], bounds[r + 1])
        d2 = dim_prepare(f2, ell2, s2, cols=sl)
        pays.append(local_partials(Y[sl], d1, d2))
        d2_full = d2
    pay = {k: sum(p[k] for p in pays) for k in pays[0]}      # the all-reduce
    if n_ranks > 1:
        d2_full = dim_prepare(f2, ell2, s2)
    return finish(theta, d1, d2_full, pay, N=n1 * n2)


def grad_raw(grad_theta: np.ndarray, raw: np.ndarray) -> np.ndarray:
    """Chain rule through softplus (gpytorch Positive / GreaterThan(1e-4))."""
    return grad_theta / (1.0 + np.exp(-np.asarray(raw, float)))


def theta_from_raw(raw) -> np.ndarray:
    raw = np.asarray(raw, float)
    th = np.logaddexp(0.0, raw)
    th[-1] += NOISE_LOWER
    return th


# ----------------------------------------------------------------------------
# q(v) and posterior from a finished step
# ----------------------------------------------------------------------------
def q_v(st: StepState):
    """mean (m1, m2) [flat index u = i1*m2+i2] and diag of the covariance (m1, m2);
    kronecker_structure.py:846, :848."""
    v = st.theta[4]
    R1, R2 = st.d1.L @ st.d1.Q, st.d2.L @ st.d2.Q
    mean = R1 @ (st.beta / v) @ R2.T
    var = (R1 * R1) @ (1.0 / st.D) @ (R2 * R2).T
    return mean, var


def q_v_cov(st: StepState) -> np.ndarray:
    """Dense M x M covariance Kuu Sigma^{-1} Kuu (small M only)."""
    R = np.kron(st.d1.L @ st.d1.Q, st.d2.L @ st.d2.Q)
    return (R / st.D.reshape(-1)[None, :]) @ R.T


def posterior(st: StepState, f1: Factor, f2: Factor, x_star: np.ndarray):
    """Point-wise posterior mean and variance at x_star (N*, 2); kronecker_structure.py:222-227."""
    ell1, ell2, s1, s2, v = st.theta
    Ts = []
    for f, d, ell, s, col in ((f1, st.d1, ell1, s1, 0), (f2, st.d2, ell2, s2, 1)):
        _, _, A0, _ = f.build(ell, x=np.asarray(x_star[:, col], float))
        Ts.append(d.Q.T @ sla.solve_triangular(d.L, A0 if f.inverse else s * A0, lower=True))
    T1, T2 = Ts
    mean = np.einsum("ip,ij,jp->p", T1, st.beta / v, T2)
    var = s1 * s2 + np.einsum("ip,ij,jp->p", T1 * T1, 1.0 / st.D - 1.0, T2 * T2)
    return mean, var


# ----------------------------------------------------------------------------
# Kron solve  X = K1^{-1} Y K2^{-T}  from Cholesky factors (BASELINE metric ii)
# ----------------------------------------------------------------------------
def kron_solve(L1: np.ndarray, L2: np.ndarray, Y: np.ndarray) -> np.ndarray:
    """(K1 (x) K2)^{-1} vec(Y) matricised: Y is (n1, n2), K_d = L_d L_d^T."""
    T = sla.cho_solve((L1, True), Y)
    return sla.cho_solve((L2, True), T.T).T


# ----------------------------------------------------------------------------
# masked / partially observed grids (BASELINE config 5): Phi = Kuf W Kuf^T is no longer a Kronecker
# product, so Sigma~ = I + rho Phi~0 is assembled in M-space (M = m1 m2) from the per-dimension factors
# and factored densely.  Everything stays at unit outputscale; rho = s1 s2 / sigma^2.
# ----------------------------------------------------------------------------
@dataclass
class MaskedState:
    theta: np.ndarray
    d1: DimState
    d2: DimState
    Sinv: np.ndarray          # Sigma~^{-1}  (M x M)
    A0: np.ndarray            # mat(Sigma~^{-1} c~0)  (m1 x m2)
    N: int
    elbo: float = 0.0
    grad: np.ndarray = field(default_factory=lambda: np.zeros(5))


def _assemble(P1a, P1b, P2a, P2b, W):
    """sum over observed (i, j) of (P1a[:,i] (x) P2a[:,j]) (P1b[:,i] (x) P2b[:,j])^T  -> (M x M).
    W[j, i] in {0, 1}.  Done as T[i, (a,b)] = sum_j W[j,i] P2a[a,j] P2b[b,j]  (one GEMM), then
    R[(i1,k1), (a,b)] = sum_i P1a[i1,i] P1b[k1,i] T[i,(a,b)]  (one GEMM), then a permutation."""
    m1, n1 = P1a.shape
    m2, n2 = P2a.shape
    PP2 = (P2a[:, None, :] * P2b[None, :, :]).reshape(m2 * m2, n2)
    T = W.T @ PP2.T                                        # (n1, m2^2)
    PP1 = (P1a[:, None, :] * P1b[None, :, :]).reshape(m1 * m1, n1)
    R = (PP1 @ T).reshape(m1, m1, m2, m2)                  # [i1, k1, i2, k2]
    return R.transpose(0, 2, 1, 3).reshape(m1 * m2, m1 * m2)


def elbo_step_masked(Y: np.ndarray, W: np.ndarray, f1: Factor, f2: Factor, theta) -> MaskedState:
    """Collapsed ELBO (kronecker_structure.py:249-278) and its gradient when only the grid points with
    W[j, i] = 1 are observed (the reference simply gets the scattered subset as X, y)."""
    ell1, ell2, s1, s2, v = [float(t) for t in theta]
    W = np.asarray(W, dtype=np.float64)
    Ym = Y * W
    N = int(W.sum())
    yy = float((Ym * Ym).sum())
    d1, d2 = dim_prepare(f1, ell1, 1.0), dim_prepare(f2, ell2, 1.0)      # unit outputscale
    B1, V1, B2, V2 = d1.B, d1.V, d2.B, d2.V
    m1, m2 = B1.shape[0], B2.shape[0]
    M = m1 * m2
    rho = s1 * s2 / v
    Phi = _assemble(B1, B1, B2, B2, W)
    Sig = np.eye(M) + rho * Phi
    Lc = np.linalg.cholesky(Sig)
    Sinv = sla.cho_solve((Lc, True), np.eye(M))
    logdet = 2.0 * np.log(np.diag(Lc)).sum()
    c0 = (B1 @ Ym.T @ B2.T).reshape(-1)
    a0 = Sinv @ c0
    q = float(c0 @ a0)
    A0 = a0.reshape(m1, m2)
    trPhi = float(np.trace(Phi))
    elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + logdet + yy / v - (s1 * s2 / v ** 2) * q)
            - (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v))
    # --- outputscales and noise: closed forms through rho
    trSP = (M - np.trace(Sinv)) / rho                        # tr(Sigma~^{-1} Phi~0)
    aPa = (q - float(a0 @ a0)) / rho                         # a0^T Phi~0 a0
    common = -0.5 * (rho * trSP - (s1 * s2 / v ** 2) * q + (s1 * s2 / v ** 2) * rho * aPa)
    g_s1 = common / s1 - (N * s2 - s2 * trPhi) / (2 * v)
    g_s2 = common / s2 - (N * s1 - s1 * trPhi) / (2 * v)
    g_v = (-0.5 * (N / v - (rho / v) * trSP - yy / v ** 2 + 2 * s1 * s2 * q / v ** 3 - (s1 * s2 * rho / v ** 3) * aPa)
           + (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v ** 2))
    # --- lengthscales
    nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)             # ||b_i||^2
    S4 = Sinv.reshape(m1, m2, m1, m2)

    def ell_grad(dim):
        if dim == 1:
            Phip = _assemble(B1, V1, B2, B2, W)
            Mk, m_other = d1.Mk, m2
            PTS = np.einsum("ajbj->ab", S4)                               # partial trace of Sigma~^{-1} over index 2
            C1 = (V1 @ Ym.T @ B2.T).reshape(-1)
            quadMk = np.einsum("ik,ij,kj->", Mk, A0, A0)
            Z = float((W * ((B2.T @ A0.T @ V1) * (B2.T @ A0.T @ B1))).sum())   # [j, i] layout
            hv = (V1 * B1).sum(0)
            tr1 = float(hv @ (W.T @ nb2))
            PT = (B1 * (W.T @ nb2)[None, :]) @ B1.T
        else:
            Phip = _assemble(B1, B1, B2, V2, W)
            Mk, m_other = d2.Mk, m1
            PTS = np.einsum("iaib->ab", S4)
            C1 = (B1 @ Ym.T @ V2.T).reshape(-1)
            quadMk = np.einsum("ik,ji,jk->", Mk, A0, A0)
            Z = float((W * ((V2.T @ A0.T @ B1) * (B2.T @ A0.T @ B1))).sum())
            hv = (V2 * B2).sum(0)
            tr1 = float(hv @ (W @ nb1))
            PT = (B2 * (W @ nb1)[None, :]) @ B2.T
        ld = float((Mk * PTS.T).sum()) - m_other * np.trace(Mk) + 2 * rho * float((Sinv * Phip).sum())
        quad = 2 * float(a0 @ C1) - quadMk - 2 * rho * Z
        return -0.5 * (ld - (s1 * s2 / v ** 2) * quad) + (s1 * s2 / (2 * v)) * (2 * tr1 - float((Mk * PT.T).sum()))

    st = MaskedState(theta=np.asarray(theta, float), d1=d1, d2=d2, Sinv=Sinv, A0=A0, N=N)
    st.elbo = float(elbo)
    st.grad = np.array([ell_grad(1), ell_grad(2), g_s1, g_s2, g_v])
    return st


@dataclass
class MaskedIterState:
    theta: np.ndarray
    A0: np.ndarray            # mat(Sigma~^{-1} c~0) (m1 x m2), from the PCG solve
    N: int
    iters: int = 0
    elbo: float = 0.0
    grad: np.ndarray = field(default_factory=lambda: np.zeros(5))


def elbo_step_masked_iter(Y: np.ndarray, W: np.ndarray, f1: Factor, f2: Factor, theta, nprobe: int = 16, tol: float = 1e-10,
                          maxit: int = 100, seed: int = 0) -> MaskedIterState:
    """The masked step WITHOUT the M x M matrices (what gpytorch does for the reference above max_cholesky_size = 800:
    CG + stochastic Lanczos, kronecker_structure.py:269, :273 -- here preconditioned, with control variates and fixed probes):

      * Sigma~ V = V + rho B1 (W^T o (B1^T V B2)) B2^T   -- matricised Kronecker MVM, V an m1 x m2 matrix;
      * preconditioner P = I + rho p (G1 (x) G2), p = observed fraction, G_d = B_d B_d^T = Q_d diag(lam_d) Q_d^T:
        E[Phi~] = p Phi~_full for an unstructured mask, and P is diagonal in the Kronecker eigenbasis;
      * a0 = Sigma~^-1 c0 by PCG; log|Sigma~| = log|P| + tr log(P^-1/2 Sigma~ P^-1/2), the second term by stochastic Lanczos
        quadrature from the PCG coefficients of probe right-hand sides z = P^1/2 z0 (z0 Rademacher, FIXED by `seed`);
      * every derivative trace tr(Sigma~^-1 D) = tr(P^-1 D) [closed form] + mean_z (Sigma~^-1 z - P^-1 z)^T D P^-1 z
        (control variate: the stochastic part only carries the difference between Sigma~^-1 and P^-1).

    Agreement with elbo_step_masked (tools/studies/pcg_masked_study.py, 16 probes): ELBO 3e-6 relative, gradient 2e-6 of its
    largest component at M = 576 .. 1024 -- the stated tolerances of the HIP entry vggp_elbo_step_masked_iter are 1e-5 / 1e-4."""
    ell1, ell2, s1, s2, v = [float(t) for t in theta]
    W = np.asarray(W, dtype=np.float64)
    Ym = Y * W
    N = int(W.sum())
    yy = float((Ym * Ym).sum())
    d1, d2 = dim_prepare(f1, ell1, 1.0), dim_prepare(f2, ell2, 1.0)
    B1, V1, B2, V2 = d1.B, d1.V, d2.B, d2.V
    m1, m2 = B1.shape[0], B2.shape[0]
    M = m1 * m2
    rho = s1 * s2 / v
    Wt = W.T
    p = N / float(W.size)

    def fld(L, V, R):                  # F[..., i, j] = l_i^T V r_j
        return np.einsum("ai,...ab,bj->...ij", L, V, R, optimize=True)

    def back(L, F, R):                 # sum_ij F[i, j] l_i r_j^T
        return np.einsum("ai,...ij,bj->...ab", L, F, R, optimize=True)

    def Aop(V):
        return V + rho * back(B1, Wt * fld(B1, V, B2), B2)

    lam1, Q1 = np.linalg.eigh(B1 @ B1.T)
    lam2, Q2 = np.linalg.eigh(B2 @ B2.T)
    dP = 1.0 + rho * p * np.outer(np.maximum(lam1, 0.0), np.maximum(lam2, 0.0))

    def rot(V, w):
        return Q1 @ ((Q1.T @ V @ Q2) * w) @ Q2.T

    Z0 = np.random.default_rng(seed).choice([-1.0, 1.0], size=(nprobe, m1, m2))
    Zs, Wz = rot(Z0, np.sqrt(dP)), rot(Z0, 1.0 / np.sqrt(dP))          # z ~ (0, P),  w = P^-1 z
    c0 = B1 @ Ym.T @ B2.T
    RHS = np.concatenate([c0[None], Zs])

    def dots(A, B):
        return (A * B).sum(axis=(1, 2))

    X = np.zeros_like(RHS)
    R = RHS.copy()
    Zp = rot(R, 1.0 / dP)
    Pd = Zp.copy()
    rz = dots(R, Zp)
    r0 = np.sqrt(dots(R, R))
    al_h, be_h = [], []
    active = np.ones(len(RHS), bool)
    kcol = np.zeros(len(RHS), int)
    for it in range(maxit):
        AP = Aop(Pd)
        pAp = dots(Pd, AP)
        al = np.where(active, rz / np.where(pAp > 0, pAp, 1.0), 0.0)
        X += al[:, None, None] * Pd
        R -= al[:, None, None] * AP
        Zp = rot(R, 1.0 / dP)
        rz_new = dots(R, Zp)
        be = np.where(active, rz_new / np.where(rz > 0, rz, 1.0), 0.0)
        al_h.append(al)
        be_h.append(be)
        kcol += active
        Pd = Zp + be[:, None, None] * Pd
        rz = rz_new
        active &= np.sqrt(dots(R, R)) > tol * r0
        if not active.any():
            break
    al_h, be_h = np.array(al_h), np.array(be_h)
    ld = 0.0
    for zi in range(nprobe):                                           # Gauss quadrature of log on the Lanczos tridiagonals
        k = kcol[1 + zi]
        a, b = al_h[:k, 1 + zi], be_h[:k, 1 + zi]
        T = np.zeros((k, k))
        for j in range(k):
            T[j, j] = 1.0 / a[j] + (b[j - 1] / a[j - 1] if j > 0 else 0.0)
            if j + 1 < k:
                T[j, j + 1] = T[j + 1, j] = math.sqrt(b[j]) / a[j]
        w, U = np.linalg.eigh(T)
        ld += M * float((U[0] ** 2) @ np.log(w))                       # |z0|^2 = M for Rademacher probes
    logdet = float(np.log(dP).sum()) + ld / nprobe
    a0 = X[0]
    q = float((c0 * a0).sum())
    nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)
    trPhi = float(nb1 @ Wt @ nb2)
    elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + logdet + yy / v - (s1 * s2 / v ** 2) * q)
            - (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v))
    dU = X[1:] - Wz
    R1, R2, RV1, RV2 = Q1.T @ B1, Q2.T @ B2, Q1.T @ V1, Q2.T @ V2
    iD = 1.0 / dP

    def tr_exact(Ra, Rb, Sa, Sb):       # tr(P^-1 assemble(.)) with the factors rotated into the eigenbasis of P
        return float((iD * ((Ra * Rb) @ Wt @ (Sa * Sb).T)).sum())

    def est(La, Lb, Ra, Rb):            # mean_z (u - w)^T Phi w,  Phi V = La (W^T o (Lb^T V Rb)) Ra^T
        return float((Wt * fld(La, dU, Ra) * fld(Lb, Wz, Rb)).sum()) / nprobe

    trSP = tr_exact(R1, R1, R2, R2) + est(B1, B1, B2, B2)
    trS = {1: 2 * tr_exact(R1, RV1, R2, R2) + est(B1, V1, B2, B2) + est(V1, B1, B2, B2),
           2: 2 * tr_exact(R1, R1, R2, RV2) + est(B1, B1, B2, V2) + est(B1, B1, V2, B2)}

    def tr_Mk(Mk, dim):                 # tr(Sigma~^-1 (Mk (x) I)) resp. (I (x) Mk)
        if dim == 1:
            return float((iD * np.diag(Q1.T @ Mk @ Q1)[:, None]).sum()) + float((dU * (Mk @ Wz)).sum()) / nprobe
        return float((iD * np.diag(Q2.T @ Mk @ Q2)[None, :]).sum()) + float((dU * (Wz @ Mk.T)).sum()) / nprobe

    aPa = (q - float((a0 * a0).sum())) / rho
    common = -0.5 * (rho * trSP - (s1 * s2 / v ** 2) * q + (s1 * s2 / v ** 2) * rho * aPa)
    g_s1 = common / s1 - (N * s2 - s2 * trPhi) / (2 * v)
    g_s2 = common / s2 - (N * s1 - s1 * trPhi) / (2 * v)
    g_v = (-0.5 * (N / v - (rho / v) * trSP - yy / v ** 2 + 2 * s1 * s2 * q / v ** 3 - (s1 * s2 * rho / v ** 3) * aPa)
           + (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v ** 2))

    def ell_grad(dim):
        if dim == 1:
            Mk, m_other = d1.Mk, m2
            C1 = V1 @ Ym.T @ B2.T
            quadMk = np.einsum("ik,ij,kj->", Mk, a0, a0)
            Z = float((W * ((B2.T @ a0.T @ V1) * (B2.T @ a0.T @ B1))).sum())
            tr1 = float((V1 * B1).sum(0) @ (W.T @ nb2))
            PT = (B1 * (W.T @ nb2)[None, :]) @ B1.T
        else:
            Mk, m_other = d2.Mk, m1
            C1 = B1 @ Ym.T @ V2.T
            quadMk = np.einsum("ik,ji,jk->", Mk, a0, a0)
            Z = float((W * ((V2.T @ a0.T @ B1) * (B2.T @ a0.T @ B1))).sum())
            tr1 = float((V2 * B2).sum(0) @ (W @ nb1))
            PT = (B2 * (W @ nb1)[None, :]) @ B2.T
        ldd = tr_Mk(Mk, dim) - m_other * np.trace(Mk) + rho * trS[dim]
        quad = 2 * float((a0 * C1).sum()) - quadMk - 2 * rho * Z
        return -0.5 * (ldd - (s1 * s2 / v ** 2) * quad) + (s1 * s2 / (2 * v)) * (2 * tr1 - float((Mk * PT.T).sum()))

    st = MaskedIterState(theta=np.asarray(theta, float), A0=a0, N=N, iters=int(kcol.max()))
    st.elbo = float(elbo)
    st.grad = np.array([ell_grad(1), ell_grad(2), g_s1, g_s2, g_v])
    return st


def q_v_masked(st: MaskedState, f1: Optional[Factor] = None, f2: Optional[Factor] = None):
    """q(v) mean and covariance diagonal, (m1, m2): mu = Kuu Sigma^{-1} c / sigma^2, S = Kuu Sigma^{-1} Kuu.
    f1, f2 only matter for the inter-domain bases (Kuu_d = K0 / s_d): L_d = s_d^(e_d/2) L0_d with e_d = -1 there."""
    _, _, s1, s2, v = st.theta
    L1, L2 = st.d1.L, st.d2.L                                  # unit-outputscale Cholesky factors
    e1 = -1 if (f1 is not None and f1.inverse) else 1
    e2 = -1 if (f2 is not None and f2.inverse) else 1
    mean = (s1 ** ((1 + e1) / 2) * s2 ** ((1 + e2) / 2) / v) * (L1 @ st.A0 @ L2.T)
    Lk = np.kron(L1, L2)
    var = (s1 ** e1) * (s2 ** e2) * np.einsum("ab,bc,ac->a", Lk, st.Sinv, Lk)
    return mean, var.reshape(mean.shape)


def posterior_masked(st: MaskedState, f1: Factor, f2: Factor, x_star: np.ndarray):
    """Point-wise posterior mean / variance at x_star (N*, 2) in masked mode (kronecker_structure.py:222-227 on the
    observed subset): t = (L1^{-1} a1*) (x) (L2^{-1} a2*), mean = rho t^T a0, var = s1 s2 (1 - |t|^2 + t^T Sigma~^{-1} t)."""
    ell1, ell2, s1, s2, v = st.theta
    ts = []
    for f, d, ell, col in ((f1, st.d1, ell1, 0), (f2, st.d2, ell2, 1)):
        _, _, A0, _ = f.build(ell, x=np.asarray(x_star[:, col], float))
        ts.append(sla.solve_triangular(d.L, A0, lower=True))
    T = np.einsum("ip,jp->ijp", ts[0], ts[1]).reshape(-1, x_star.shape[0])
    mean = (s1 * s2 / v) * (T.T @ st.A0.reshape(-1))
    var = s1 * s2 * (1.0 - (T * T).sum(0) + np.einsum("up,uw,wp->p", T, st.Sinv, T))
    return mean, var


# ----------------------------------------------------------------------------
# gridded read-out q_u -> p(v|u) -> q_v of B0 cell features (gridded_kronecker_structure.py:396-438, :613-654), Kronecker
# in the per-dimension cross-covariances
# ----------------------------------------------------------------------------
def cross_b0(f: Factor, mesh: np.ndarray, ell: float):
    """Unit-outputscale Cov(v, u) along one dimension (mv x m) for B0 cells on `mesh`, and the unit diagonal of Kvv."""
    mesh = np.asarray(mesh, float)
    g = np.asarray(f.grid, float)
    if f.basis == "points":
        if f.kind != "matern12":
            raise ValueError("the B0 cross-covariance closed forms are Matern-1/2")
        C = b0_A(mesh, g, ell)[0]
    elif f.basis == "vff":
        a, om = g[0], g[2:]
        d = mesh[1] - mesh[0]
        k0 = np.full((len(mesh) - 1, 1), d)
        kc = (np.sin(om[1:] * (mesh[1:] - a)[:, None]) - np.sin(om[1:] * (mesh[:-1] - a)[:, None])) / om[1:]
        ks = -(np.cos(om[1:] * (mesh[1:] - a)[:, None]) - np.cos(om[1:] * (mesh[:-1] - a)[:, None])) / om[1:]
        C = np.hstack([k0, kc, ks])
    elif f.basis == "b1":
        # GriddedMatern12ASVGP._Kvu_along_dim (gridded_kronecker_structure.py:831-838): delta at the two knots of each cell
        ns, K = len(mesh) - 1, len(g)
        padding = (K - (ns + 1)) // 2
        C = np.zeros((ns, K))
        for i in range(ns):
            C[i, padding + i] = C[i, padding + i + 1] = g[1] - g[0]
    else:
        raise NotImplementedError(f.basis)
    kd = np.full(len(mesh) - 1, b0_K(len(mesh) - 1, float(mesh[1] - mesh[0]), ell)[0][0, 0])
    return C, kd


def readout(st: StepState, f1: Factor, f2: Factor, C1, C2, kd1, kd2, literal: bool = True):
    """mean (mv1, mv2) and variance of q(v): t_d = Q_d^T L_d^{-1} Kuv_d, mean = T1^T (beta / v) T2,
    var = Kvv_aa Kvv_bb + sum_ij T1[i,a]^2 W_ij T2[j,b]^2 with W = D - 1 (literal reference) or 1/D - 1."""
    _, _, s1, s2, v = st.theta
    Ts = []
    for f, d, s, C in ((f1, st.d1, s1, C1), (f2, st.d2, s2, C2)):
        Kuv = C.T if f.inverse else s * C.T
        Ts.append(d.Q.T @ sla.solve_triangular(d.L, Kuv, lower=True))
    T1, T2 = Ts
    W = st.D - 1.0 if literal else 1.0 / st.D - 1.0
    mean = T1.T @ (st.beta / v) @ T2
    var = s1 * s2 * np.outer(kd1, kd2) + (T1 * T1).T @ W @ (T2 * T2)
    return mean, var


# ----------------------------------------------------------------------------
# gradient with respect to the inducing-point coordinates ("points" basis: SVGP's trainable Z,
# kronecker_structure.py:303-304 registers Z as a Parameter; autograd differentiates through kernel(Z) and kernel(Z, x))
# ----------------------------------------------------------------------------
def z_grad(st: StepState, f1: Factor, f2: Factor, Y: np.ndarray):
    """dELBO/dz for the inducing coordinates of both dimensions -> (g1 [m1], g2 [m2]).

    The analytic lengthscale gradient of finish() is LINEAR in the perturbation (dK, dA) it is fed:
    dELBO = <W_M, L^-1 dK L^-T> + <W_V, L^-1 dA>, with weights that only depend on the step's state.  Hence the sensitivities
    Kbar = L^-T W_M L^-1 and Abar = L^-T W_V, and for a stationary kernel d kappa(z_i, x)/d z_i = -(d kappa/d ell) ell/(z_i - x):
    g_i = -ell [ sum_k Abar_ik dA_ik / (z_i - x_k) + sum_{j != i} (Kbar_ij + Kbar_ji) dK_ij / (z_i - z_j) ]."""
    ell1, ell2, s1, s2, v = [float(t) for t in st.theta]
    d1, d2, beta, D = st.d1, st.d2, st.beta, st.D
    invD = 1.0 / D
    l1, l2 = d1.lam, d2.lam
    m1, m2 = len(l1), len(l2)
    WP = beta / v ** 2

    def weights(r, rlam, X, Xlam, lam_other_sum, m_other, lam_self):
        WE = -0.5 * np.diag(r) + 0.5 * m_other * np.eye(len(r)) - X / (2 * v ** 2) - lam_other_sum / (2 * v) * np.diag(lam_self)
        WF = -0.5 * np.diag(rlam) / v - Xlam / (2 * v ** 3) + lam_other_sum / (2 * v) * np.eye(len(r))
        return WE, WF

    WE1, WF1 = weights(invD.sum(1), (invD * l2[None, :]).sum(1), beta @ beta.T, (beta * l2[None, :]) @ beta.T, l2.sum(), m2, l1)
    WE2, WF2 = weights(invD.sum(0), (invD * l1[:, None]).sum(0), beta.T @ beta, (beta * l1[:, None]).T @ beta, l1.sum(), m1, l2)
    out = []
    for dd, f, ell, s, WE, WF, proj in (
            (d1, f1, ell1, s1, WE1, WF1, d1.Q @ WP @ d2.Q.T @ (d2.B @ Y)),            # (m1 x n1)
            (d2, f2, ell2, s2, WE2, WF2, d2.Q @ WP.T @ d1.Q.T @ (d1.B @ Y.T))):       # (m2 x n2)
        if f.basis != "points":
            out.append(np.zeros(f.m))
            continue
        Q = dd.Q
        WM = Q @ WE @ Q.T
        WV = Q @ (WF + WF.T) @ Q.T @ dd.B + proj
        Abar = s * sla.solve_triangular(dd.L, WV, lower=True, trans="T")              # w.r.t. the unit-scale A0
        Z1 = sla.solve_triangular(dd.L, WM, lower=True, trans="T")
        Kbar = s * sla.solve_triangular(dd.L, Z1.T, lower=True, trans="T").T          # L^-T W_M L^-1, w.r.t. K0
        K0, dK0, A0, dA0 = f.build(ell)
        z = np.asarray(f.grid, float)
        dzx = z[:, None] - f.x[None, :]
        dzz = z[:, None] - z[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            tA = np.where(dzx != 0.0, dA0 / dzx, 0.0)
            tK = np.where(dzz != 0.0, dK0 / dzz, 0.0)
        out.append(-ell * ((Abar * tA).sum(1) + ((Kbar + Kbar.T) * tK).sum(1)))
    return out[0], out[1]


# ----------------------------------------------------------------------------
# scattered observations (along-track points: notebooks 6 / 61 / 7 feed arbitrary (x1, x2) pairs to the same _elbo):
# Kuf[:, k] = a1(x1_k) (x) a2(x2_k) is the Khatri-Rao product of the per-dimension factors evaluated at the POINTS, Phi = Kuf Kuf^T
# is assembled in M-space exactly as in the masked case -- the sums over observed grid points become sums over the points.
# ----------------------------------------------------------------------------
def elbo_step_scattered(X: np.ndarray, y: np.ndarray, f1: Factor, f2: Factor, theta) -> MaskedState:
    """Collapsed ELBO (kronecker_structure.py:249-278) and its gradient for N scattered points X (N, 2), y (N).
    f1.x / f2.x are ignored: the factors are evaluated at X[:, 0] and X[:, 1]."""
    ell1, ell2, s1, s2, v = [float(t) for t in theta]
    X = np.asarray(X, float)
    y = np.asarray(y, float).reshape(-1)
    N = len(y)
    yy = float(y @ y)
    g1 = Factor(f1.basis, f1.kind, f1.grid, X[:, 0].copy(), f1.f32_kdelta)
    g2 = Factor(f2.basis, f2.kind, f2.grid, X[:, 1].copy(), f2.f32_kdelta)
    d1, d2 = dim_prepare(g1, ell1, 1.0), dim_prepare(g2, ell2, 1.0)      # unit outputscale, columns = points
    B1, V1, B2, V2 = d1.B, d1.V, d2.B, d2.V
    m1, m2 = B1.shape[0], B2.shape[0]
    M = m1 * m2
    rho = s1 * s2 / v
    kr = lambda P1, P2: (P1[:, None, :] * P2[None, :, :]).reshape(M, N)   # Khatri-Rao: row (i1, i2), column k
    Zt = kr(B1, B2)
    Phi = Zt @ Zt.T
    Sig = np.eye(M) + rho * Phi
    Lc = np.linalg.cholesky(Sig)
    Sinv = sla.cho_solve((Lc, True), np.eye(M))
    logdet = 2.0 * np.log(np.diag(Lc)).sum()
    c0 = Zt @ y
    a0 = Sinv @ c0
    q = float(c0 @ a0)
    A0 = a0.reshape(m1, m2)
    nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)
    trPhi = float(nb1 @ nb2)
    elbo = (-0.5 * (N * math.log(2 * math.pi) + N * math.log(v) + logdet + yy / v - (s1 * s2 / v ** 2) * q)
            - (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v))
    trSP = (M - np.trace(Sinv)) / rho
    aPa = (q - float(a0 @ a0)) / rho
    common = -0.5 * (rho * trSP - (s1 * s2 / v ** 2) * q + (s1 * s2 / v ** 2) * rho * aPa)
    g_s1 = common / s1 - (N * s2 - s2 * trPhi) / (2 * v)
    g_s2 = common / s2 - (N * s1 - s1 * trPhi) / (2 * v)
    g_v = (-0.5 * (N / v - (rho / v) * trSP - yy / v ** 2 + 2 * s1 * s2 * q / v ** 3 - (s1 * s2 * rho / v ** 3) * aPa)
           + (N * s1 * s2 - s1 * s2 * trPhi) / (2 * v ** 2))
    S4 = Sinv.reshape(m1, m2, m1, m2)
    zb = np.einsum("ik,ij,jk->k", B1, A0, B2)                          # b1_k^T A0 b2_k

    def ell_grad(dim):
        if dim == 1:
            Phip = Zt @ kr(V1, B2).T                                     # sum_k (b1 (x) b2)(v1 (x) b2)^T
            Mk, m_other = d1.Mk, m2
            PTS = np.einsum("ajbj->ab", S4)
            C1 = kr(V1, B2) @ y
            quadMk = np.einsum("ik,ij,kj->", Mk, A0, A0)
            Z = float(np.einsum("ik,ij,jk->k", V1, A0, B2) @ zb)
            tr1 = float(((V1 * B1).sum(0)) @ nb2)
            PT = (B1 * nb2[None, :]) @ B1.T
        else:
            Phip = Zt @ kr(B1, V2).T
            Mk, m_other = d2.Mk, m1
            PTS = np.einsum("iaib->ab", S4)
            C1 = kr(B1, V2) @ y
            quadMk = np.einsum("ik,ji,jk->", Mk, A0, A0)
            Z = float(np.einsum("ik,ij,jk->k", B1, A0, V2) @ zb)
            tr1 = float(((V2 * B2).sum(0)) @ nb1)
            PT = (B2 * nb1[None, :]) @ B2.T
        ld = float((Mk * PTS.T).sum()) - m_other * np.trace(Mk) + 2 * rho * float((Sinv * Phip).sum())
        quad = 2 * float(a0 @ C1) - quadMk - 2 * rho * Z
        return -0.5 * (ld - (s1 * s2 / v ** 2) * quad) + (s1 * s2 / (2 * v)) * (2 * tr1 - float((Mk * PT.T).sum()))

    st = MaskedState(theta=np.asarray(theta, float), d1=d1, d2=d2, Sinv=Sinv, A0=A0, N=N)
    st.elbo = float(elbo)
    st.grad = np.array([ell_grad(1), ell_grad(2), g_s1, g_s2, g_v])
    return st


def z_grad_scattered(st: MaskedState, X: np.ndarray, y: np.ndarray, f1: Factor, f2: Factor):
    """dELBO/dz of elbo_step_scattered for the inducing coordinates of both dimensions -> (g1 [m1], g2 [m2]) (the spec of
    vggp_zgrad_scattered; the reference gets it from autograd through _elbo(), kronecker_structure.py:249-278, Z being a
    Parameter of the SVGP classes :303-304).

    With the unit-scale whitened factors B_d = L0_d^-1 A0_d at the points, the ELBO depends on the inducing coordinates only
    through z_k = b1_k (x) b2_k:  G_B1[:, k] = -rho mat(Sigma~^-1 z_k) b2_k + (s1 s2 / v^2)(y_k - rho b1_k^T A0 b2_k) A0 b2_k
    + (s1 s2 / v) |b2_k|^2 b1_k  (and the mirror image for dimension 2).  The ELBO is a function of A0^T K0^-1 A0 alone, so
    Abar = L0^-T G_B and Kbar = -1/2 L0^-T (G_B B^T) L0^-1, contracted with d kappa / d z as in z_grad."""
    ell1, ell2, s1, s2, v = [float(t) for t in st.theta]
    X = np.asarray(X, float)
    y = np.asarray(y, float).reshape(-1)
    d1, d2 = st.d1, st.d2
    B1, B2 = d1.B, d2.B
    m1, m2, N = B1.shape[0], B2.shape[0], len(y)
    rho = s1 * s2 / v
    Zt = (B1[:, None, :] * B2[None, :, :]).reshape(m1 * m2, N)
    U = (st.Sinv @ Zt).reshape(m1, m2, N)
    u1 = np.einsum("ijk,jk->ik", U, B2)
    u2 = np.einsum("ijk,ik->jk", U, B1)
    UB1, UB2 = st.A0 @ B2, st.A0.T @ B1                                  # A0 b2_k (m1 x N), A0^T b1_k (m2 x N)
    zb = (B1 * UB1).sum(0)
    w = (s1 * s2 / v ** 2) * (y - rho * zb)
    nb1, nb2 = (B1 * B1).sum(0), (B2 * B2).sum(0)
    G1 = -rho * u1 + w[None, :] * UB1 + (s1 * s2 / v) * nb2[None, :] * B1
    G2 = -rho * u2 + w[None, :] * UB2 + (s1 * s2 / v) * nb1[None, :] * B2
    out = []
    for dd, f, ell, G, x in ((d1, f1, ell1, G1, X[:, 0]), (d2, f2, ell2, G2, X[:, 1])):
        if f.basis != "points":
            out.append(np.zeros(f.m))
            continue
        Abar = sla.solve_triangular(dd.L, G, lower=True, trans="T")
        Z1 = sla.solve_triangular(dd.L, -0.5 * (G @ dd.B.T), lower=True, trans="T")
        Kbar = sla.solve_triangular(dd.L, Z1.T, lower=True, trans="T").T
        K0, dK0, A0, dA0 = Factor(f.basis, f.kind, f.grid, x.copy(), f.f32_kdelta).build(ell)
        z = np.asarray(f.grid, float)
        dzx = z[:, None] - x[None, :]
        dzz = z[:, None] - z[None, :]
        with np.errstate(divide="ignore", invalid="ignore"):
            tA = np.where(dzx != 0.0, dA0 / dzx, 0.0)
            tK = np.where(dzz != 0.0, dK0 / dzz, 0.0)
        out.append(-ell * ((Abar * tA).sum(1) + ((Kbar + Kbar.T) * tK).sum(1)))
    return out[0], out[1]
