"""Dense float64 restatement of the reference's sparse-GP algebra (TEST INFRASTRUCTURE).

Follows, line for line, the reference files below (paths relative to the
reference checkout; nothing is imported from it):

  src/models/sparse/kronecker_structure.py
      :134-150  _sigma          -> DenseKron._sigma
      :199-230  posterior       -> DenseKron.posterior
      :249-278  _elbo           -> DenseKron._elbo
      :306-338  Matern12SVGP    -> basis="points" factor builders
      :702-739  _Kuu_along_dim  -> b0_Kuu_along_dim
      :741-790  _Kuf_along_dim  -> b0_Kuf_along_dim
      :792-823  _Kuu / _Kuf     -> DenseKron._Kuu / _Kuf (kron + row-wise Khatri-Rao)
      :825-849  q_v             -> DenseKron.q_v
  src/models/sparse/gridded_kronecker_structure.py :1255-1433 (Matern12GriddedGP:
      identical maths to kronecker_structure.py :671-849)
  src/models/sparse/univariate_structure.py
      :104-120, :184-263, :693-717, :740-825 -> Dense1D (one factor)

The six third-party calls on the path are replaced by their documented
definitions (gpytorch / linear_operator are not installed; PARITY UNPINNED at
that boundary, see oracle/__init__.py):

  MaternKernel(nu)(a,b)            exp(-r) | (1+sqrt3 r)exp(-sqrt3 r) |
                                   (1+sqrt5 r+5r^2/3)exp(-sqrt5 r),  r=|a-b|/ell
  (new) RBF                        exp(-r^2/2)
  ScaleKernel                      multiply by outputscale
  GaussianLikelihood.noise         softplus(raw)+1e-4 ; lengthscale/outputscale softplus(raw)
  lazify(A).inv_matmul(B)          A^{-1}B by Cholesky (psd_safe_cholesky jitter policy)
  MultivariateNormal(0,C).log_prob -1/2 (y^T C^{-1} y + log|C| + N log 2 pi), Cholesky
  ToeplitzLinearOperator(r).to_dense()[i,j] = r[|i-j|]

Jitter policy (shared with the HIP engine and oracle/kron.py): each per-dimension
factor is Cholesky-factored at unit outputscale; on failure jitter eps = 1e-8,
1e-7, 1e-6 (linear_operator's psd_safe_cholesky schedule for float64) is added
to the unit-outputscale diagonal, i.e. Kuu_d = s_d (kappa_d + eps_d I), so the
factor scales exactly with s_d.  The dense Kuu is kron of the jittered factors.
For the reference's Matern-1/2 models the factors are well conditioned and no
jitter is ever added, so this equals the literal reference.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch

DT = torch.float64
KINDS = ("matern12", "matern32", "matern52", "rbf")
NOISE_LOWER = 1e-4          # gpytorch GaussianLikelihood default GreaterThan(1e-4)
JITTERS = (0.0, 1e-8, 1e-7, 1e-6)


# ----------------------------------------------------------------------------
# parameter transforms (gpytorch Positive()/GreaterThan() constraints)
# ----------------------------------------------------------------------------
def softplus(x: torch.Tensor) -> torch.Tensor:
    return torch.nn.functional.softplus(x)


def inv_softplus(y: torch.Tensor) -> torch.Tensor:
    y = torch.as_tensor(y, dtype=DT)
    return y + torch.log(-torch.expm1(-y))


def constrained_from_raw(raw: torch.Tensor) -> torch.Tensor:
    """raw = [raw_ell1, raw_ell2, raw_s1, raw_s2, raw_noise] -> theta (same order)."""
    th = softplus(raw)
    return torch.cat([th[:4], th[4:5] + NOISE_LOWER])


def raw_from_constrained(theta: Sequence[float]) -> torch.Tensor:
    th = torch.as_tensor(theta, dtype=DT).clone()
    th[4] = th[4] - NOISE_LOWER
    return inv_softplus(th)


# ----------------------------------------------------------------------------
# stationary kernels (kronecker_structure.py:30-32 uses nu=1/2 only; the other
# three are the build's extensions in the same factor-builder shape)
# ----------------------------------------------------------------------------
def kappa(kind: str, r: torch.Tensor) -> torch.Tensor:
    if kind == "matern12":
        return torch.exp(-r)
    if kind == "matern32":
        a = math.sqrt(3.0) * r
        return (1.0 + a) * torch.exp(-a)
    if kind == "matern52":
        a = math.sqrt(5.0) * r
        return (1.0 + a + a * a / 3.0) * torch.exp(-a)
    if kind == "rbf":
        return torch.exp(-0.5 * r * r)
    raise ValueError(kind)


def pairwise(kind: str, a: torch.Tensor, b: torch.Tensor, ell, s) -> torch.Tensor:
    """ScaleKernel(Matern/RBF)(a, b).evaluate() for 1-D inputs a (m,), b (n,) -> (m, n)."""
    r = torch.abs(a[:, None] - b[None, :]) / ell
    return s * kappa(kind, r)


# ----------------------------------------------------------------------------
# B0-spline (gridded) Matern-1/2 factor builders
# ----------------------------------------------------------------------------
def b0_Kuu_along_dim(m: int, delta, ell, s) -> torch.Tensor:
    """kronecker_structure.py:723-739 (== gridded_kronecker_structure.py:1307-1323)."""
    # NB: the reference's mesh / delta are float32 (torch.linspace default dtype, the
    # model's .to(float64) does not touch plain attributes), so (k*delta) is rounded
    # to float32 there before the float64 division.  torch's promotion rules reproduce
    # that here when a float32 mesh is passed; a float64 mesh gives the exact maths.
    k = torch.arange(m)
    first_row = (torch.exp((-(k - 1) * delta) / ell)
                 + torch.exp((-(k + 1) * delta) / ell)
                 - 2 * torch.exp((-k * delta) / ell))
    diag0 = 2 * (torch.exp(-delta / ell) + (delta / ell) - 1)
    first_row = torch.cat([diag0.reshape(1), first_row[1:]])
    idx = torch.abs(torch.arange(m)[:, None] - torch.arange(m)[None, :])
    Kuu = first_row[idx]                       # ToeplitzLinearOperator(first_row).to_dense()
    return Kuu * (ell ** 2 * s)


def b0_Kuf_along_dim(mesh: torch.Tensor, ell, s, x: torch.Tensor) -> torch.Tensor:
    """kronecker_structure.py:768-790 (== gridded_kronecker_structure.py:1352-1374)."""
    m = mesh.shape[0] - 1
    k = torch.arange(m)
    indicator = -torch.sign(torch.searchsorted(mesh, x.contiguous(), right=False)[None, :]
                            - k[:, None] - 1).to(DT)
    exp_1 = ell * torch.exp(-torch.abs(x[None, :] - mesh[:-1, None]) / ell)
    exp_2 = ell * torch.exp(-torch.abs(x[None, :] - mesh[1:, None]) / ell)
    outside = indicator * (exp_1 - exp_2)
    inside = 2 * ell - (exp_1 + exp_2)
    Kuf = torch.where(indicator == 0, inside, outside)
    return Kuf * s


# ----------------------------------------------------------------------------
# linear-algebra definitions of the linear_operator / gpytorch calls
# ----------------------------------------------------------------------------
# ----------------------------------------------------------------------------
# Variational Fourier features (Matern-1/2) -- Matern12VFFGP, kronecker_structure.py:346-515, basis/fourier.py
# ----------------------------------------------------------------------------
def vff_omegas(M: int, a: float, b: float) -> torch.Tensor:
    """fourier.py:13 -- float32 in the reference (python float * int64 arange / python float)."""
    return (2 * torch.pi) * torch.arange(M + 1) / (b - a)


def vff_Kuu_along_dim(a: float, b: float, omegas: torch.Tensor, ell, s) -> torch.Tensor:
    """kronecker_structure.py:376-462: Kuu_d = diag(alpha) + beta beta^T with the Matern-1/2 spectral density
    S(w) = 2 s lam / (lam^2 + w^2), lam = 1/ell.  NB it scales with 1/s, and Kuf_d carries no s at all.
    The reference evaluates S in float32 (float32 omegas against a 0-dim float64 lam) and casts at the end; here the
    float32-rounded omegas are used in float64 arithmetic."""
    om = omegas.to(DT)
    lam = 1.0 / ell.reshape(())
    S_inv = (lam ** 2 + om ** 2) / (2 * s * lam)
    alpha = ((b - a) / 2) * torch.cat([2 * S_inv[0][None], S_inv[1:], S_inv[1:]])
    beta = torch.cat((torch.ones(len(om), dtype=DT) / torch.sqrt(s), torch.zeros(len(om) - 1, dtype=DT)))
    return torch.diag(alpha) + beta[:, None] * beta[None, :]


def vff_Kuf_along_dim(a: float, b: float, omegas: torch.Tensor, ell, x: torch.Tensor) -> torch.Tensor:
    """kronecker_structure.py:464-480 + fourier.py:14-88: inside [a, b): cos(w (x-a)) then sin(w[1:] (x-a)); outside:
    exp(-lam r) on the cosine rows (r = distance to the nearer boundary), 0 on the sine rows.  (2M+1) x n."""
    om = omegas.to(DT)
    lam = 1.0 / ell.reshape(())
    inside = torch.logical_and(x >= a, x < b)
    xa = x - a
    cosr = torch.cos(om[:, None] * xa[None, :])
    sinr = torch.sin(om[1:, None] * xa[None, :])
    r = torch.minimum(torch.abs(x - a), torch.abs(x - b))
    out_real = torch.exp(-lam * r)[None, :] * torch.ones(len(om), 1, dtype=DT)
    real = torch.where(inside[None, :], cosr, out_real)
    imag = torch.where(inside[None, :], sinr, torch.zeros_like(sinr))
    return torch.cat([real, imag], dim=0)


# ----------------------------------------------------------------------------
# B1-spline inducing features (Matern-1/2) -- Matern12B1SplineASVGP, kronecker_structure.py:524-660, basis/bspline.py
# ----------------------------------------------------------------------------
def b1_Kuu_along_dim(mesh: torch.Tensor, ell, s) -> torch.Tensor:
    """kronecker_structure.py:560-614, literally: (A * ell + B / ell + BC) / (2 s) with A the L2 Gram matrix of the hat
    functions (tridiagonal 2d/3, d/6; d/3 at the two ends), B their gradient Gram matrix (2/d, -1/d; 1/d at the ends) and
    BC = diag(1, 0, ..., 0, 1).  (The reference builds the matrices in float32; float64 here.)"""
    m = mesh.shape[0]
    d = (mesh[1] - mesh[0]).to(DT)
    i = torch.arange(m)
    off = (torch.abs(i[:, None] - i[None, :]) == 1).to(DT)
    eye = torch.eye(m, dtype=DT)
    ends = torch.zeros(m, dtype=DT)
    ends[0] = ends[-1] = 1.0
    A = (2 / 3) * d * eye + (1 / 6) * d * off - torch.diag(ends) * (d / 3)
    B = (2 / d) * eye - (1 / d) * off - torch.diag(ends) / d
    BC = torch.diag(ends)
    ell0 = ell.reshape(())
    return (A * ell0 + B / ell0 + BC) / (2 * s)


def b1_Kuf_along_dim(mesh: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """bspline.py:24-112: left half hat on [v0, v1), interior hats (rising on [v_k-1, v_k], falling on (v_k, v_k+1]),
    right half hat on [v_K-2, v_K-1]; no hyper-parameter enters.  K x n."""
    v = mesh.to(DT)
    K = v.shape[0]
    rows = []
    B1 = torch.logical_and(x >= v[0], x < v[1]).to(DT)
    rows.append(((v[1] - x) / (v[1] - v[0])) * B1)
    for mm in range(K - 2):
        vm, vm1, vm2 = v[mm], v[mm + 1], v[mm + 2]
        r0 = torch.logical_and(x >= vm, x <= vm1).to(DT)
        r1 = torch.logical_and(x > vm1, x <= vm2).to(DT)
        rows.append(((x - vm) / (vm1 - vm)) * r0 + ((vm2 - x) / (vm2 - vm1)) * r1)
    r0 = torch.logical_and(x >= v[-2], x <= v[-1]).to(DT)
    rows.append(((x - v[-2]) / (v[-1] - v[-2])) * r0)
    return torch.vstack(rows)


def psd_safe_cholesky(A: torch.Tensor) -> Tuple[torch.Tensor, float]:
    """linear_operator.utils.cholesky.psd_safe_cholesky, float64 schedule."""
    for jit in JITTERS:
        Aj = A if jit == 0.0 else A + jit * torch.eye(A.shape[0], dtype=A.dtype)
        L, info = torch.linalg.cholesky_ex(Aj)
        if int(info) == 0 and bool(torch.isfinite(L).all()):
            return L, jit
    raise torch.linalg.LinAlgError("matrix not positive definite after jitter 1e-6")


def factor_jitter(K: torch.Tensor) -> float:
    return psd_safe_cholesky(K.detach())[1]


def inv_matmul(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    L, _ = psd_safe_cholesky(A)
    return torch.cholesky_solve(B if B.dim() == 2 else B[:, None], L).reshape(B.shape)


def mvn_log_prob(cov: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    L, _ = psd_safe_cholesky(cov)
    z = torch.linalg.solve_triangular(L, y[:, None], upper=False)[:, 0]
    return -0.5 * ((z * z).sum() + 2.0 * torch.log(torch.diagonal(L)).sum()
                   + y.shape[0] * math.log(2.0 * math.pi))


class MVN:
    """Minimal stand-in for the returned gpytorch MultivariateNormal (mean + covariance)."""

    def __init__(self, mean, cov):
        self.mean = mean
        self.covariance_matrix = cov

    @property
    def variance(self):
        return torch.diagonal(self.covariance_matrix)


# ----------------------------------------------------------------------------
# 2-D Kronecker models
# ----------------------------------------------------------------------------
class DenseKron:
    """KroneckerStructure (kronecker_structure.py:15-278) + its factor builders.

    basis="b0"     Matern12B0SplineGriddedGP / Matern12GriddedGP (kind must be matern12):
                   grid_1, grid_2 are the knot meshes (nknots each).
    basis="points" Matern12SVGP-shaped pairwise factors (kind in KINDS): grid_1,
                   grid_2 are the per-dimension inducing coordinates Z[:, d].
    basis="vff"    Matern12VFFGP (kronecker_structure.py:346-515): grid_d = (a_d, b_d, nfrequencies).
    basis="b1"     Matern12B1SplineASVGP (:524-660): grid_d = knot mesh (nknots).
    raw: 5 raw parameters [ell1, ell2, s1, s2, noise] (gpytorch init = zeros).
    mask: optional bool (N,) -- observed points; masked-out rows of X, y are dropped
          (the reference simply never sees them).
    """

    def __init__(self, X, y, basis: str, kind: str, grid_1, grid_2, raw=None, mask=None):
        X = torch.as_tensor(X, dtype=DT)
        y = torch.as_tensor(y, dtype=DT)
        if mask is not None:
            mask = torch.as_tensor(mask, dtype=torch.bool)
            X, y = X[mask], y[mask]
        self.train_inputs = (X,)
        self.train_targets = y
        self.basis, self.kind = basis, kind
        if basis in ("b0", "vff", "b1") and kind != "matern12":
            raise ValueError("the B0 / VFF / B1 closed forms exist for Matern-1/2 only")
        if basis == "vff":
            # grid_d = (a, b, M): domain limits and number of frequencies; omegas as the reference builds them (float32)
            self.grid_1 = (float(grid_1[0]), float(grid_1[1]), vff_omegas(int(grid_1[2]), float(grid_1[0]), float(grid_1[1])))
            self.grid_2 = (float(grid_2[0]), float(grid_2[1]), vff_omegas(int(grid_2[2]), float(grid_2[0]), float(grid_2[1])))
        else:
            # b0 / b1: keep the mesh dtype as given (float32 in the reference, see b0_Kuu_along_dim)
            gdt = None if basis in ("b0", "b1") else DT
            self.grid_1 = torch.as_tensor(grid_1, dtype=gdt)
            self.grid_2 = torch.as_tensor(grid_2, dtype=gdt)
        self.raw = (torch.zeros(5, dtype=DT) if raw is None else torch.as_tensor(raw, dtype=DT)).clone()
        self.raw.requires_grad_(True)
        self._jit = None

    # -- hyper-parameters ------------------------------------------------------
    def theta(self) -> torch.Tensor:
        return constrained_from_raw(self.raw)

    # -- per-dimension factors ---------------------------------------------------
    def _Kuu_d(self, d: int) -> torch.Tensor:
        th = self.theta()
        # `lengthscale[0]` in the reference is a (1,)-shaped float64 tensor (gpytorch lengthscale is (1,1)),
        # `outputscale` is 0-dim: keep those shapes, they decide torch's type promotion against the float32 mesh
        ell, s = th[d:d + 1], th[2 + d]
        g = self.grid_1 if d == 0 else self.grid_2
        if self.basis == "b0":
            return b0_Kuu_along_dim(g.shape[0] - 1, g[1] - g[0], ell, s)
        if self.basis == "vff":
            a, b, om = g
            return vff_Kuu_along_dim(a, b, om, ell, s)
        if self.basis == "b1":
            return b1_Kuu_along_dim(g, ell, s)
        return pairwise(self.kind, g, g, ell, s)

    def _Kuf_d(self, d: int, x: torch.Tensor) -> torch.Tensor:
        th = self.theta()
        ell, s = th[d:d + 1], th[2 + d]
        g = self.grid_1 if d == 0 else self.grid_2
        if self.basis == "b0":
            return b0_Kuf_along_dim(g, ell, s, x)
        if self.basis == "vff":
            a, b, om = g
            return vff_Kuf_along_dim(a, b, om, ell, x)
        if self.basis == "b1":
            return b1_Kuf_along_dim(g, x)
        return pairwise(self.kind, g, x, ell, s)

    def jitters(self) -> Tuple[float, float]:
        th = self.theta()
        return tuple(factor_jitter(self._unit(self._Kuu_d(d), th[2 + d])) for d in (0, 1))

    def _unit(self, K: torch.Tensor, s) -> torch.Tensor:
        """The outputscale-free factor the jitter schedule is applied to: K / s, or K * s for the inter-domain bases whose
        Kuu scales with 1/s (vff, b1)."""
        return K * s if self.basis in ("vff", "b1") else K / s

    def _Kuu(self) -> torch.Tensor:
        """:792-806 -- torch.kron(Kuu_1, Kuu_2) (jittered factors, see header)."""
        Ks = []
        th = self.theta()
        for d in (0, 1):
            K = self._Kuu_d(d)
            jit = factor_jitter(self._unit(K, th[2 + d]))
            scale = 1.0 / th[2 + d] if self.basis in ("vff", "b1") else th[2 + d]
            Ks.append(K + (scale * jit) * torch.eye(K.shape[0], dtype=DT))
        return torch.kron(Ks[0], Ks[1])

    def _Kuf(self, x: torch.Tensor) -> torch.Tensor:
        """:808-823 -- stack([k1*k2 for k2 in Kuf_1 for k1 in Kuf_2]): row u=i1*m2+i2."""
        Kuf_1 = self._Kuf_d(0, x[:, 0])
        Kuf_2 = self._Kuf_d(1, x[:, 1])
        return (Kuf_1[:, None, :] * Kuf_2[None, :, :]).reshape(-1, x.shape[0])

    # -- gridded read-out q_u -> p(v|u) -> q_v (gridded_kronecker_structure.py:396-438, :613-654) --------------------
    def _Kvu_d(self, d: int, mesh: torch.Tensor) -> torch.Tensor:
        """Cov(v, u) along dimension d, v = B0 cell features on `mesh`: :281-338 (points: the B0 closed form at the
        inducing coordinates), :499-555 (VFF: cell integrals of the Fourier features; no outputscale)."""
        th = self.theta()
        ell, s = th[d:d + 1], th[2 + d]
        g = self.grid_1 if d == 0 else self.grid_2
        if self.basis == "points" and self.kind == "matern12":
            return b0_Kuf_along_dim(mesh, ell, s, g)
        if self.basis == "vff":
            a, _, om = g
            om = om.to(DT)
            me = mesh.to(DT)
            delta = me[1] - me[0]
            k0 = torch.ones(me.shape[0] - 1, 1, dtype=DT) * delta
            kc = (torch.sin(om[1:] * (me[1:] - a)[:, None]) - torch.sin(om[1:] * (me[:-1] - a)[:, None])) / om[1:]
            ks = -(torch.cos(om[1:] * (me[1:] - a)[:, None]) - torch.cos(om[1:] * (me[:-1] - a)[:, None])) / om[1:]
            return torch.cat([k0, kc, ks], dim=1)
        if self.basis == "b1":
            # GriddedMatern12ASVGP._Kvu_along_dim (gridded_kronecker_structure.py:831-838), literally: the B1 knots are the B0
            # mesh padded by `padding` knots on either side; row i holds delta at the two knots of cell i (no hyper-parameter)
            K = g.shape[0]
            ns = mesh.shape[0] - 1
            padding = (K - (ns + 1)) // 2
            delta = (g[1] - g[0]).to(DT)
            first = torch.zeros(K, dtype=DT)
            first[padding] = first[padding + 1] = delta
            return torch.vstack([torch.roll(first, i) for i in range(ns)])
        raise NotImplementedError("gridded read-out: points (Matern-1/2), vff and b1 inducing features")

    def q_v_gridded(self, mesh_1: torch.Tensor, mesh_2: torch.Tensor, literal: bool = True) -> MVN:
        """:417-438 / :634-654: mean = Kvu Kuu^-1 mu_u, cov = Kvv - Kvu Kuu^-1 Kuv + Kvu X Kuv with X = S_u^-1 (literal: what
        the reference computes) or X = Kuu^-1 S_u Kuu^-1 (the conditional covariance of v under q(u))."""
        th = self.theta()
        Kuu = self._Kuu()
        Kvu = torch.kron(self._Kvu_d(0, mesh_1), self._Kvu_d(1, mesh_2))
        Kvv = torch.kron(b0_Kuu_along_dim(mesh_1.shape[0] - 1, mesh_1[1] - mesh_1[0], th[0:1], th[2]),
                         b0_Kuu_along_dim(mesh_2.shape[0] - 1, mesh_2[1] - mesh_2[0], th[1:2], th[3]))
        qu = self.q_v()                                   # q(u) in these classes' naming: mean Kuu Sigma^-1 Kuf y / s2, cov Kuu Sigma^-1 Kuu
        Su = qu.covariance_matrix
        mean = Kvu @ inv_matmul(Kuu, qu.mean)
        KiKuv = inv_matmul(Kuu, Kvu.T)
        X = inv_matmul(Su, Kvu.T) if literal else inv_matmul(Kuu, Su @ KiKuv)
        cov = Kvv - Kvu @ KiKuv + Kvu @ X
        return MVN(mean, cov)

    def _sigma(self) -> torch.Tensor:
        """:134-150."""
        noise = self.theta()[4]
        Kuf = self._Kuf(self.train_inputs[0])
        return self._Kuu() + (Kuf @ Kuf.T) / noise

    def _elbo(self) -> torch.Tensor:
        """:249-278."""
        X, y = self.train_inputs[0], self.train_targets
        th = self.theta()
        noise = th[4]
        Kuu = self._Kuu()
        Kuf = self._Kuf(X)
        # diag of ProductKernel(ScaleKernel(.)*ScaleKernel(.)): s1*s2 at zero distance
        Kff_trace = X.shape[0] * th[2] * th[3]
        approx_prior = Kuf.T @ inv_matmul(Kuu, Kuf)
        evidence_cov = approx_prior + noise * torch.eye(X.shape[0], dtype=DT)
        evidence_term = mvn_log_prob(evidence_cov, y)
        trace_term = (Kff_trace - torch.trace(approx_prior)) / (2 * noise)
        return evidence_term - trace_term

    def elbo_and_grad(self):
        """ELBO and d ELBO / d raw (what Adam sees in the notebooks' loops)."""
        if self.raw.grad is not None:
            self.raw.grad = None
        e = self._elbo()
        (g,) = torch.autograd.grad(e, self.raw)
        return e.detach(), g.detach()

    def q_v(self) -> MVN:
        """:825-849."""
        X, y = self.train_inputs[0], self.train_targets
        noise = self.theta()[4]
        Kuu = self._Kuu()
        Kuf = self._Kuf(X)
        sigma = self._sigma()
        mu = (Kuu @ inv_matmul(sigma, Kuf) @ y) / noise
        S = Kuu @ inv_matmul(sigma, Kuu)
        return MVN(mu, S)

    def posterior(self, x_star) -> MVN:
        """:199-230."""
        x_star = torch.as_tensor(x_star, dtype=DT)
        X, y = self.train_inputs[0], self.train_targets
        th = self.theta()
        noise = th[4]
        Kuu = self._Kuu()
        Kuf = self._Kuf(X)
        Kuf_star = self._Kuf(x_star)
        sigma = self._sigma()
        cond_mu = (Kuf_star.T @ inv_matmul(sigma, Kuf) @ y) / noise
        k1 = pairwise(self.kind, x_star[:, 0], x_star[:, 0], th[0], th[2])
        k2 = pairwise(self.kind, x_star[:, 1], x_star[:, 1], th[1], th[3])
        term1 = k1 * k2
        term2 = Kuf_star.T @ inv_matmul(sigma, Kuf_star)
        term3 = Kuf_star.T @ inv_matmul(Kuu, Kuf_star)
        return MVN(cond_mu, term1 + term2 - term3)

    def posterior_predictive(self, x_star) -> MVN:
        """:232-247 -- GaussianLikelihood(posterior) adds noise to the diagonal."""
        p = self.posterior(x_star)
        n = p.mean.shape[0]
        return MVN(p.mean, p.covariance_matrix + self.theta()[4] * torch.eye(n, dtype=DT))


# ----------------------------------------------------------------------------
# 1-D model (univariate_structure.py: SparseGP + Matern12B0SplineGriddedGP / SVGP)
# ----------------------------------------------------------------------------
class Dense1D:
    """raw = [ell, s, noise]."""

    def __init__(self, X, y, basis: str, kind: str, grid, raw=None):
        self.X = torch.as_tensor(X, dtype=DT).reshape(-1)
        self.y = torch.as_tensor(y, dtype=DT).reshape(-1)
        self.basis, self.kind = basis, kind
        self.grid = torch.as_tensor(grid, dtype=None if basis == "b0" else DT)
        self.raw = (torch.zeros(3, dtype=DT) if raw is None else torch.as_tensor(raw, dtype=DT)).clone()
        self.raw.requires_grad_(True)

    def theta(self):
        th = softplus(self.raw)
        return torch.cat([th[:2], th[2:3] + NOISE_LOWER])

    def _Kuu(self):
        ell, s, _ = self.theta()
        if self.basis == "b0":       # univariate_structure.py:809-825
            K = b0_Kuu_along_dim(self.grid.shape[0] - 1, self.grid[1] - self.grid[0], ell, s)
        else:                        # :304-306
            K = pairwise(self.kind, self.grid, self.grid, ell, s)
        return K + (s * factor_jitter(K / s)) * torch.eye(K.shape[0], dtype=DT)

    def _Kuf(self, x):
        ell, s, _ = self.theta()
        if self.basis == "b0":       # :764-787
            return b0_Kuf_along_dim(self.grid, ell, s, x)
        return pairwise(self.kind, self.grid, x, ell, s)   # :320

    def _sigma(self):                # :104-120
        Kuf = self._Kuf(self.X)
        return self._Kuu() + (Kuf @ Kuf.T) / self.theta()[2]

    def _elbo(self):                 # :234-263
        th = self.theta()
        noise = th[2]
        Kuu, Kuf = self._Kuu(), self._Kuf(self.X)
        n = self.X.shape[0]
        approx_prior = Kuf.T @ inv_matmul(Kuu, Kuf)
        evidence_term = mvn_log_prob(approx_prior + noise * torch.eye(n, dtype=DT), self.y)
        trace_term = (n * th[1] - torch.trace(approx_prior)) / (2 * noise)
        return evidence_term - trace_term

    def elbo_and_grad(self):
        e = self._elbo()
        (g,) = torch.autograd.grad(e, self.raw)
        return e.detach(), g.detach()

    def q_v(self):                   # :693-717
        noise = self.theta()[2]
        Kuu, Kuf, sigma = self._Kuu(), self._Kuf(self.X), self._sigma()
        mu = (Kuu @ inv_matmul(sigma, Kuf) @ self.y) / noise
        return MVN(mu, Kuu @ inv_matmul(sigma, Kuu))

    def posterior(self, x_star):     # :184-215
        x_star = torch.as_tensor(x_star, dtype=DT).reshape(-1)
        th = self.theta()
        noise = th[2]
        Kuu, Kuf, Ks, sigma = self._Kuu(), self._Kuf(self.X), self._Kuf(x_star), self._sigma()
        mu = (Ks.T @ inv_matmul(sigma, Kuf) @ self.y) / noise
        term1 = pairwise(self.kind, x_star, x_star, th[0], th[1])
        cov = term1 + Ks.T @ inv_matmul(sigma, Ks) - Ks.T @ inv_matmul(Kuu, Ks)
        return MVN(mu, cov)


# ----------------------------------------------------------------------------
# synthetic gridded data in gen_2d's layout (src/utils/datagenerators.py:37-73)
# ----------------------------------------------------------------------------
def latent_2d(x1, x2):
    """The notebooks' test function (5_gridded_kronecker_structure_models.ipynb cell 3)."""
    import numpy as np
    return (np.sin(5 * x1) + np.cos(7 * x2) + 0.5 * np.sin(15 * x1) + 0.5 * np.cos(12 * x2)
            + 0.2 * np.sin(20 * x1) + 0.2 * np.cos(25 * x2))


def gen_grid(n1: int, n2: int, lims1=(0.0, 1.0), lims2=(0.0, 1.0), noise=0.05, seed=0):
    """X (N,2) with x1 fastest (p = j*n1 + i <-> (x1[i], x2[j])), y (N,), x1, x2."""
    import numpy as np
    x1 = np.linspace(lims1[0], lims1[1], n1)
    x2 = np.linspace(lims2[0], lims2[1], n2)
    X1, X2 = np.meshgrid(x1, x2)                    # 'xy' indexing: shape (n2, n1)
    X = np.vstack([X1.ravel(), X2.ravel()]).T
    rng = np.random.default_rng(seed)
    y = latent_2d(X[:, 0], X[:, 1]) + noise * rng.standard_normal(X.shape[0])
    return X, y, x1, x2
