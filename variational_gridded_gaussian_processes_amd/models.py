"""Drop-in host-side mirror of the reference's model API for the Kronecker ELBO hot path.

Same class names, constructor arguments, method names and attribute reads as the reference's
notebooks use:

    reference                                                    here
    src/models/sparse/gridded_kronecker_structure.py:1255-1433   Matern12GriddedGP
    src/models/sparse/kronecker_structure.py:671-849             Matern12B0SplineGriddedGP
    src/models/sparse/kronecker_structure.py:287-338             Matern12SVGP (+ Matern32/52/RBF variants, new)
    src/models/sparse/kronecker_structure.py:15-278              KroneckerStructure (base: _elbo, q_v, posterior, ...)
    src/models/sparse/univariate_structure.py:721-825, :325-354  univariate.Matern12B0SplineGriddedGP, univariate.*SVGP

`model._elbo()` returns a differentiable 0-d tensor, so the notebooks' loop
`optimizer.zero_grad(); loss = -model._elbo(); loss.backward(); optimizer.step()` with
`torch.optim.Adam(model.parameters(), lr)` runs unchanged -- but value and gradient come from ONE
call into libvggp_hip.so (analytic gradient), not from autograd over dense N x N algebra.
The gpytorch pieces the reference leans on (GaussianLikelihood, ScaleKernel, MaternKernel,
MultivariateNormal) are mirrored here only as far as the path reads them: parameter transforms
(softplus, noise >= 1e-4, raw init 0) and accessor names.  There is no CPU fallback.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import VggpError
from .basis import B0SplineBasis, B1SplineBasis, FourierBasisMatern12
from .engine import Engine

NOISE_LOWER = 1e-4


def _inv_softplus(y: torch.Tensor) -> torch.Tensor:
    return y + torch.log(-torch.expm1(-y))


# ---------------------------------------------------------------------------------------------
# minimal mirrors of the gpytorch modules whose attributes the notebooks read
# ---------------------------------------------------------------------------------------------
class _BaseKernel(torch.nn.Module):
    """MaternKernel(nu) / RBFKernel: only the lengthscale parameter lives here."""

    def __init__(self, kind: str):
        super().__init__()
        self.kind = kind
        self.nu = {"matern12": 0.5, "matern32": 1.5, "matern52": 2.5, "rbf": math.inf}[kind]
        self.raw_lengthscale = torch.nn.Parameter(torch.zeros(1, 1))

    @property
    def lengthscale(self) -> torch.Tensor:
        # a fresh tensor, like gpytorch's property: `kernel.lengthscale[0] = v` therefore does NOT
        # update the raw parameter (the reference's non_informative_initialise quirk)
        return torch.nn.functional.softplus(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        v = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype).reshape(1, 1)
        with torch.no_grad():
            self.raw_lengthscale.copy_(_inv_softplus(v))


class ScaleKernel(torch.nn.Module):
    def __init__(self, base_kernel: _BaseKernel):
        super().__init__()
        self.base_kernel = base_kernel
        self.raw_outputscale = torch.nn.Parameter(torch.zeros(()))

    @property
    def outputscale(self) -> torch.Tensor:
        return torch.nn.functional.softplus(self.raw_outputscale)

    @outputscale.setter
    def outputscale(self, value):
        v = torch.as_tensor(value, dtype=self.raw_outputscale.dtype).reshape(())
        with torch.no_grad():
            self.raw_outputscale.copy_(_inv_softplus(v))


class GaussianLikelihood(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.raw_noise = torch.nn.Parameter(torch.zeros(1))

    @property
    def noise(self) -> torch.Tensor:
        return torch.nn.functional.softplus(self.raw_noise) + NOISE_LOWER

    @noise.setter
    def noise(self, value):
        v = torch.as_tensor(value, dtype=self.raw_noise.dtype).reshape(1)
        with torch.no_grad():
            self.raw_noise.copy_(_inv_softplus(v - NOISE_LOWER))


class MultivariateNormal:
    """What q_v()/posterior() return: `.mean`, `.variance`, `.stddev`, `.confidence_region()` and a lazily
    materialised `.covariance_matrix` (dense, small problems only)."""

    def __init__(self, mean: torch.Tensor, variance: torch.Tensor, cov_fn=None):
        self.mean = mean
        self._variance = variance
        self._cov_fn = cov_fn
        self._cov = None

    @property
    def variance(self) -> torch.Tensor:
        return self._variance

    @property
    def stddev(self) -> torch.Tensor:
        return self._variance.clamp_min(0).sqrt()

    @property
    def covariance_matrix(self) -> torch.Tensor:
        if self._cov is None:
            if self._cov_fn is None:
                raise NotImplementedError("dense covariance is not available for this distribution; use .variance")
            self._cov = self._cov_fn()
        return self._cov

    def confidence_region(self) -> Tuple[torch.Tensor, torch.Tensor]:
        s2 = 2.0 * self.stddev
        return self.mean - s2, self.mean + s2


class _ElboFunction(torch.autograd.Function):
    """value + analytic gradient from one vggp_elbo_step; backward only scales the cached gradient."""

    @staticmethod
    def forward(ctx, theta: torch.Tensor, model: "KroneckerStructure", Z: Optional[torch.Tensor] = None):
        try:
            elbo, grad, info = model._engine_step([float(t) for t in theta.detach().cpu()])
        except VggpError as e:
            if e.code == _lib.VGGP_ENOTPD:       # what the reference's notebooks catch (61_envisat... cell 39)
                raise torch.linalg.LinAlgError(str(e)) from e
            raise
        model.last_info = info
        gz = None
        if Z is not None and Z.requires_grad:
            # trainable inducing points (the reference registers Z as a Parameter, kronecker_structure.py:303-304, and autograd
            # reaches it through kernel(Z)): analytic gradient from the engine's resident state (vggp_zgrad)
            g1, g2 = model._engine.zgrad_scattered(model._Y) if getattr(model, "_scattered", False) else model._engine.zgrad(model._Y)
            cols = [g1] if Z.shape[1] == 1 else [g1, g2]
            gz = torch.stack(cols, dim=1).to(dtype=Z.dtype, device=Z.device)
        ctx.save_for_backward(torch.as_tensor(grad, dtype=theta.dtype, device=theta.device), gz)
        return torch.as_tensor(elbo, dtype=theta.dtype, device=theta.device)

    @staticmethod
    def backward(ctx, grad_out):
        g, gz = ctx.saved_tensors
        return grad_out * g, None, (grad_out * gz if gz is not None else None)


def _detect_grid(X: torch.Tensor):
    """X (N,2) -> (x1, x2, W, flat).  Full grid in gen_2d layout (utils/datagenerators.py:70-72: x1 fastest): W and flat
    are None.  Otherwise X must be a subset of the cartesian grid of its own unique coordinates (a masked grid,
    BASELINE config 5): W[j, i] = 1 on observed points and flat[k] = j*n1 + i places y[k] on the grid."""
    Xn = X.detach().cpu().numpy().astype(np.float64)
    if Xn.ndim != 2 or Xn.shape[1] != 2:
        raise ValueError("X must be (N, 2)")
    N = Xn.shape[0]
    x2_first = Xn[0, 1]
    n1 = int(np.argmax(Xn[:, 1] != x2_first)) if np.any(Xn[:, 1] != x2_first) else N
    if n1 > 0 and N % n1 == 0:
        n2 = N // n1
        G = Xn.reshape(n2, n1, 2)
        x1, x2 = G[0, :, 0].copy(), G[:, 0, 1].copy()
        if (np.array_equal(G[:, :, 0], np.broadcast_to(x1, (n2, n1)))
                and np.array_equal(G[:, :, 1], np.broadcast_to(x2[:, None], (n2, n1)))):
            return x1, x2, None, None
    x1, i = np.unique(Xn[:, 0], return_inverse=True)
    x2, j = np.unique(Xn[:, 1], return_inverse=True)
    if len(x1) * len(x2) > 16 * N:
        # general scattered points (along-track data): no grid at all -- the per-point coordinates go to the engine as they are
        # (vggp_elbo_step_scattered: Khatri-Rao assembly in M-space)
        return Xn[:, 0].copy(), Xn[:, 1].copy(), "scattered", None
    flat = j.astype(np.int64) * len(x1) + i.astype(np.int64)
    W = np.zeros(len(x1) * len(x2))
    W[flat] = 1.0
    if int(W.sum()) != N:
        raise ValueError("X holds duplicated points")
    return x1, x2, W.reshape(len(x2), len(x1)), flat


class KroneckerStructure(torch.nn.Module):
    """kronecker_structure.py:15-278 -- the 2-D sparse-GP base class, structured engine inside."""

    kind = "matern12"

    def __init__(self, X: torch.Tensor, y: torch.Tensor, engine: Optional[Engine] = None, warm_start: bool = True):
        super().__init__()
        self.train_inputs = (X,)
        self.train_targets = y
        self.likelihood = GaussianLikelihood()
        self.kernel_1 = ScaleKernel(_BaseKernel(self.kind))
        self.kernel_2 = ScaleKernel(_BaseKernel(self.kind))
        self._engine = engine if engine is not None else Engine()
        self._warm = warm_start
        self._planned = False
        self._plan_token, self._plan_key = -1, None
        self.last_info = None
        self._x1, self._x2, W, flat = _detect_grid(X)
        n2, n1 = len(self._x2), len(self._x1)
        yd = torch.as_tensor(y, dtype=torch.float64).reshape(-1).to(self._engine.device)
        self._scattered = isinstance(W, str)
        self._masked = W is not None     # (read-outs of the scattered mode are the masked ones: dense M-space state)
        if self._scattered:              # scattered points: y stays a vector, one coordinate pair per point
            self._Y = yd.contiguous()
            self._nobs = float(yd.numel())
        elif self._masked:               # masked grid: scatter the observations onto the grid (zeros elsewhere)
            self._W = torch.as_tensor(W, device=self._engine.device)
            self._Y = torch.zeros(n2 * n1, dtype=torch.float64, device=self._engine.device)
            self._Y[torch.as_tensor(flat, device=self._engine.device)] = yd
            self._Y = self._Y.reshape(n2, n1)
            self._nobs = float(W.sum())
        else:
            self._Y = yd.reshape(n2, n1).contiguous()
        self._yy = float((yd * yd).sum().item()) if self._scattered else self._engine.sumsq(self._Y)

    def _as_scattered(self):
        """Switch a masked-grid model to the scattered representation of the same observations (one coordinate pair per point):
        the step with a Z-gradient on incomplete data is vggp_zgrad_scattered."""
        if self._scattered or not self._masked:
            return
        Xn = self.train_inputs[0].detach().cpu().numpy().astype(np.float64)
        yd = torch.as_tensor(self.train_targets, dtype=torch.float64).reshape(-1).to(self._engine.device)
        self._x1, self._x2 = Xn[:, 0].copy(), Xn[:, 1].copy()
        self._scattered, self._Y, self._nobs = True, yd.contiguous(), float(yd.numel())
        self._yy = float((yd * yd).sum().item())
        self._W = None
        self._planned = False

    # subclasses provide (basis, grid_1, grid_2)
    def _basis(self) -> Tuple[str, np.ndarray, np.ndarray]:
        raise NotImplementedError

    def _f32_mesh(self) -> bool:
        """True when the B0 mesh is a float32 tensor (the reference's default): reproduce its float32 k*delta."""
        mesh = getattr(self, "mesh_1", getattr(self, "mesh", None))
        return mesh is not None and mesh.dtype == torch.float32

    def _plan(self):
        """(Re)plan the engine when this model is not its last planner (an Engine may be shared between models: the plan --
        factors, basis, meshes -- lives in the engine) or when its inducing description changed since the last plan."""
        basis, g1, g2 = self._basis()
        key = (basis, np.asarray(g1).tobytes(), np.asarray(g2).tobytes())
        if self._planned and self._plan_token == self._engine.plan_token and key != self._plan_key and basis == "points" \
                and self._plan_key is not None and len(self._plan_key[1]) == len(key[1]) and len(self._plan_key[2]) == len(key[2]):
            # only the inducing points moved (an optimiser training Z): new coordinates in place, plan / graphs / warm start stay
            if key[1] != self._plan_key[1]:
                self._engine.set_inducing(0, g1)
            if key[2] != self._plan_key[2]:
                self._engine.set_inducing(1, g2)
            self._plan_key = key
        elif not self._planned or self._plan_token != self._engine.plan_token or key != self._plan_key:
            if self._scattered or self._masked:
                self._check_dense_workspace(basis, g1, g2)
            self._engine.plan(self.kind, basis, g1, self._x1, self.kind, basis, g2, self._x2, warm_start=self._warm,
                              b0_f32_kdelta=self._f32_mesh(), scattered=self._scattered)
            self._planned = True
            self._plan_token, self._plan_key = self._engine.plan_token, key

    def _check_dense_workspace(self, basis, g1, g2):
        """The masked / scattered steps work on dense M x M matrices (M = m1 m2 <= 16384; ~10 M^2 doubles) and, for scattered points,
        on four m_d^2 x N pair-product buffers that are not chunked over N: say so here, with the numbers, instead of failing in
        hipMalloc at the first step."""
        m1 = len(g1) - 1 if basis == "b0" else (2 * (len(g1) - 3) + 1 if basis == "vff" else len(g1))
        m2 = len(g2) - 1 if basis == "b0" else (2 * (len(g2) - 3) + 1 if basis == "vff" else len(g2))
        M, N = m1 * m2, int(self._nobs)
        if M > 16384:
            raise ValueError(f"{type(self).__name__}: X is {'scattered' if self._scattered else 'a grid with holes'}, which takes the dense "
                             f"M-space solver, and M = m1 * m2 = {m1} * {m2} = {M} exceeds its limit of 16384 "
                             f"(Engine.elbo_step_masked_iter handles larger M on masked grids)")
        n_pair = N if self._scattered else max(len(self._x1), len(self._x2))
        need = 8.0 * (10.0 * M * M + 2.0 * (m1 * m1 + m2 * m2) * n_pair)
        free = torch.cuda.mem_get_info(self._engine.device)[0] if torch.cuda.is_available() else None
        if free is not None and need > free:
            raise ValueError(f"{type(self).__name__}: the dense M-space workspace needs about {need / 2**30:.1f} GiB (M = {M}, N = {N}, "
                             f"m_d = {m1}, {m2}: ~10 M^2 + 2 (m1^2 + m2^2) N doubles) but {free / 2**30:.1f} GiB of device memory are free")

    def _theta(self) -> torch.Tensor:
        return torch.stack([self.kernel_1.base_kernel.lengthscale.reshape(()),
                            self.kernel_2.base_kernel.lengthscale.reshape(()),
                            self.kernel_1.outputscale.reshape(()), self.kernel_2.outputscale.reshape(()),
                            self.likelihood.noise.reshape(())]).to(torch.float64)

    def _engine_step(self, theta):
        self._plan()
        if self._scattered:
            return self._engine.elbo_step_scattered(self._Y, self._yy, theta)
        if self._masked:
            return self._engine.elbo_step_masked(self._Y, self._W, self._nobs, self._yy, theta)
        return self._engine.elbo_step(self._Y, self._yy, theta)

    # -- reference API ---------------------------------------------------------------------------
    def _elbo(self) -> torch.Tensor:
        """kronecker_structure.py:249-278 (collapsed Titsias bound at the optimal q(u))."""
        return _ElboFunction.apply(self._theta(), self)

    def _refresh(self):
        """q_v()/posterior() read the engine state of the CURRENT hyper-parameters."""
        with torch.no_grad():
            _ElboFunction.apply(self._theta(), self)

    def q_v(self) -> MultivariateNormal:
        """gridded_kronecker_structure.py:1409-1433 == kronecker_structure.py:825-849.
        mean is flat (M,) with u = i1*m2 + i2 (callers do `.mean.reshape(m, m).T`)."""
        self._refresh()
        if self._masked:
            mean, var = self._engine.qv_masked()
            return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu(),
                                      cov_fn=lambda: self._engine.qv_cov_masked().cpu())
        mean, var = self._engine.qv()
        return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu(),
                                  cov_fn=lambda: self._engine.qv_cov().cpu())

    def posterior(self, x_star: torch.Tensor) -> MultivariateNormal:
        """kronecker_structure.py:199-230: mean and variance at x_star (N*, 2); `.covariance_matrix` (the reference's dense
        N* x N* matrix, :223-229) is materialised on first access (vggp_posterior_cov: N* <= 8192, M N* <= 2^27)."""
        self._refresh()
        xs = torch.as_tensor(x_star, dtype=torch.float64)
        post = self._engine.posterior_masked if self._masked else self._engine.posterior
        mean, var = post(xs)

        def cov():
            self._refresh()          # the engine may have been re-planned by another model since
            return self._engine.posterior_cov(xs, masked=self._masked).cpu()
        return MultivariateNormal(mean.cpu(), var.cpu(), cov_fn=cov)

    def posterior_predictive(self, x_star: torch.Tensor) -> MultivariateNormal:
        """kronecker_structure.py:232-247: the likelihood adds the noise variance."""
        p = self.posterior(x_star)
        noise = self.likelihood.noise.detach().to(p.variance.dtype)
        return MultivariateNormal(p.mean, p.variance + noise,
                                  cov_fn=lambda: p.covariance_matrix + noise * torch.eye(p.mean.shape[0], dtype=p.variance.dtype))

    # -- dense views kept for small sizes / debugging (the hot path never forms them) ------------------------------
    def _factor(self, d: int, x: torch.Tensor):
        """Unit-outputscale per-dimension factors from the engine: A0 (m x n at coordinates x), K0 (m x m)."""
        basis, g1, g2 = self._basis()
        ell = (self.kernel_1 if d == 0 else self.kernel_2).base_kernel.lengthscale.reshape(()).item()
        grid = torch.as_tensor(g1 if d == 0 else g2, dtype=torch.float64, device=self._engine.device)
        xs = torch.as_tensor(x, dtype=torch.float64, device=self._engine.device).contiguous()
        flags = 1 if (basis == "b0" and self._f32_mesh()) else 0
        A, _, K, _ = self._engine.factor_build(self.kind, basis, xs, grid, ell, flags)
        return A, K

    def _Kuu_along_dim(self, d: int) -> torch.Tensor:
        """kronecker_structure.py:702-739 (B0) / :318-319 (points): m_d x m_d, outputscale applied (no jitter)."""
        s = (self.kernel_1 if d == 0 else self.kernel_2).outputscale.detach().to(torch.float64)
        _, K = self._factor(d, torch.zeros(1, dtype=torch.float64))
        if self._basis()[0] in ("vff", "b1"):          # inter-domain features: Kuu_d = K0 / s_d
            return (K / s.to(K.device)).cpu()
        return (s.to(K.device) * K).cpu()

    def _Kuf_along_dim(self, d: int, x: torch.Tensor) -> torch.Tensor:
        """kronecker_structure.py:741-790 (B0) / :336-337 (points): m_d x len(x)."""
        s = (self.kernel_1 if d == 0 else self.kernel_2).outputscale.detach().to(torch.float64)
        A, _ = self._factor(d, x)
        if self._basis()[0] in ("vff", "b1"):          # ... and Kuf_d carries no outputscale
            return A.cpu()
        return (s.to(A.device) * A).cpu()

    def _Kuu(self) -> torch.Tensor:
        """kronecker_structure.py:792-806: torch.kron(Kuu_1, Kuu_2), M x M dense -- small M only."""
        K1, K2 = self._Kuu_along_dim(0), self._Kuu_along_dim(1)
        if K1.shape[0] * K2.shape[0] > 8192:
            raise ValueError("_Kuu(): M = m1*m2 > 8192; the dense matrix is for small sizes / debugging only")
        return torch.kron(K1, K2)

    def _Kuf(self, x: torch.Tensor) -> torch.Tensor:
        """kronecker_structure.py:808-823: row-wise Khatri-Rao of the per-dimension factors, M x N, u = i1*m2 + i2."""
        x = torch.as_tensor(x, dtype=torch.float64)
        A1, A2 = self._Kuf_along_dim(0, x[:, 0]), self._Kuf_along_dim(1, x[:, 1])
        if A1.shape[0] * A2.shape[0] * x.shape[0] > (1 << 27):
            raise ValueError("_Kuf(): M*N too large for the dense matrix (small sizes / debugging only)")
        return (A1[:, None, :] * A2[None, :, :]).reshape(-1, x.shape[0])

    def _sigma(self) -> torch.Tensor:
        """kronecker_structure.py:134-150: Kuu + Kuf Kuf^T / sigma^2 (dense; debugging only)."""
        Kuf = self._Kuf(self.train_inputs[0])
        return self._Kuu() + Kuf @ Kuf.T / self.likelihood.noise.detach().to(torch.float64).cpu()

    def prior(self, x: torch.Tensor) -> MultivariateNormal:
        """kronecker_structure.py:90-103: zero mean, product kernel k1(x1,x1') k2(x2,x2') -- dense N x N, small N only."""
        x = torch.as_tensor(x, dtype=torch.float64)
        if x.shape[0] > 8192:
            raise ValueError("prior(): dense N x N covariance, N <= 8192")
        cov = torch.ones(x.shape[0], x.shape[0], dtype=torch.float64)
        for d, k in ((0, self.kernel_1), (1, self.kernel_2)):
            r = (x[:, d, None] - x[None, :, d]).abs() / k.base_kernel.lengthscale.reshape(()).item()
            kind = k.base_kernel.kind
            if kind == "matern12":
                kd = torch.exp(-r)
            elif kind == "matern32":
                kd = (1 + math.sqrt(3) * r) * torch.exp(-math.sqrt(3) * r)
            elif kind == "matern52":
                kd = (1 + math.sqrt(5) * r + 5 * r * r / 3) * torch.exp(-math.sqrt(5) * r)
            else:
                kd = torch.exp(-0.5 * r * r)
            cov = cov * k.outputscale.detach().to(torch.float64) * kd
        return MultivariateNormal(torch.zeros(x.shape[0], dtype=torch.float64), torch.diagonal(cov).clone(), cov_fn=lambda: cov)

    def non_informative_initialise(self, lmbda: float, kappa: float) -> None:
        """kronecker_structure.py:34-61.  As in the reference, the `lengthscale[0] = ...` assignments go
        through the property getter and leave the raw lengthscale untouched (documented quirk)."""
        X, y = self.train_inputs[0], self.train_targets
        self.kernel_1.outputscale = y.var()
        self.kernel_1.base_kernel.lengthscale[0] = (X[:, 0].std() / lmbda)
        self.kernel_2.outputscale = y.var()
        self.kernel_2.base_kernel.lengthscale[0] = (X[:, 1].std() / lmbda)
        self.likelihood.noise = ((self.kernel_1.outputscale + self.kernel_2.outputscale) / 2) / (kappa ** 2)

    def informative_initialise(self, prior_amplitude: float, lmbda: float) -> None:
        """kronecker_structure.py:63-88."""
        X, y = self.train_inputs[0], self.train_targets
        self.kernel_1.outputscale = (torch.tensor(prior_amplitude) / 2) ** 2
        self.kernel_1.base_kernel.lengthscale[0] = (X[:, 0].std() / lmbda)
        self.kernel_2.outputscale = (torch.tensor(prior_amplitude) / 2) ** 2
        self.kernel_2.base_kernel.lengthscale[0] = (X[:, 1].std() / lmbda)
        self.likelihood.noise = y.var() - ((self.kernel_1.outputscale + self.kernel_2.outputscale) / 2)

    # convenience: the notebooks' fit loop / grid prediction
    def fit(self, n_iter: int = 100, lr: float = 0.01):
        opt = torch.optim.Adam(self.parameters(), lr=lr)
        history = torch.empty(n_iter)
        for i in range(n_iter):
            opt.zero_grad()
            loss = -self._elbo()
            history[i] = loss.item()
            loss.backward()
            opt.step()
        return history

    def predict(self) -> MultivariateNormal:
        return self.q_v()


class _B0Gridded(KroneckerStructure):
    def __init__(self, X, y, nknots: int, dim1lims: Tuple[float, float], dim2lims: Tuple[float, float], **kw):
        super().__init__(X, y, **kw)
        self.nknots = nknots
        self.dim1lims, self.dim2lims = dim1lims, dim2lims
        self.mesh_1 = torch.linspace(dim1lims[0], dim1lims[1], nknots)   # default dtype, as the reference
        self.mesh_2 = torch.linspace(dim2lims[0], dim2lims[1], nknots)
        self.delta_1 = self.mesh_1[1] - self.mesh_1[0]
        self.delta_2 = self.mesh_2[1] - self.mesh_2[0]
        self.b0_mesh_1, self.b0_mesh_2 = self.mesh_1, self.mesh_2
        # gridded_kronecker_structure.py:1283-1284 / kronecker_structure.py:698-699 (bspline.py:81-103)
        self.basis_1 = B0SplineBasis(self.mesh_1, self._engine)
        self.basis_2 = B0SplineBasis(self.mesh_2, self._engine)

    def _basis(self):
        return "b0", self.mesh_1.double().numpy(), self.mesh_2.double().numpy()


class Matern12GriddedGP(_B0Gridded):
    """gridded_kronecker_structure.py:1255-1433 (the flagship model)."""


class Matern12B0SplineGriddedGP(_B0Gridded):
    """kronecker_structure.py:671-849 (identical maths)."""


class Matern12VFFGP(KroneckerStructure):
    """kronecker_structure.py:346-515: variational Fourier features, Matern-1/2.  Kuu_d = diag(alpha) + beta beta^T (:400-462,
    scales with 1/s_d), Kuf_d = Fourier basis (fourier.py:14-88, no s_d): only the per-dimension factors differ."""

    def __init__(self, X, y, nfrequencies: int, dim1lims: Tuple[float, float], dim2lims: Tuple[float, float], **kw):
        super().__init__(X, y, **kw)
        self.nfrequencies = nfrequencies
        self.dim1lims, self.dim2lims = dim1lims, dim2lims
        # fourier.py:13, float32 like the reference (python float * int64 arange / python float)
        self.omegas_1 = (2 * torch.pi) * torch.arange(nfrequencies + 1) / (dim1lims[1] - dim1lims[0])
        self.omegas_2 = (2 * torch.pi) * torch.arange(nfrequencies + 1) / (dim2lims[1] - dim2lims[0])

    @property
    def basis_1(self) -> FourierBasisMatern12:
        """kronecker_structure.py:464-470: the Fourier basis at the CURRENT lengthscale (the reference rebuilds it per call)."""
        return FourierBasisMatern12(self.nfrequencies, self.dim1lims[0], self.dim1lims[1],
                                    self.kernel_1.base_kernel.lengthscale.reshape(()).item(), self._engine)

    @property
    def basis_2(self) -> FourierBasisMatern12:
        return FourierBasisMatern12(self.nfrequencies, self.dim2lims[0], self.dim2lims[1],
                                    self.kernel_2.base_kernel.lengthscale.reshape(()).item(), self._engine)

    def _basis(self):
        g1 = np.concatenate([[self.dim1lims[0], self.dim1lims[1]], self.omegas_1.double().numpy()])
        g2 = np.concatenate([[self.dim2lims[0], self.dim2lims[1]], self.omegas_2.double().numpy()])
        return "vff", g1, g2


def _b0_kvv_diag_unit(delta: float, ell: float) -> float:
    """diag of the unit-outputscale B0 Gram matrix (gridded_kronecker_structure.py:341-394): ell^2 * 2 (e^{-d/l} + d/l - 1)."""
    r = delta / ell
    return ell * ell * 2.0 * (math.exp(-r) + r - 1.0)


class GriddedMatern12VFFGP(Matern12VFFGP):
    """gridded_kronecker_structure.py:470-654: the VFF model with a gridded read-out -- q_v() is the distribution of the
    B0 cell features v on an nsplines x nsplines grid, obtained from the inducing posterior q(u) through p(v | u).
    Mean and variance come from `vggp_readout` (Kronecker in the per-dimension cross-covariances :499-555); the variance is
    the reference's own expression (:431, literal=True) unless literal=False is asked for."""

    def __init__(self, X, y, nfrequencies: int, vffdim1lims, vffdim2lims, nsplines: int, griddim1lims, griddim2lims, **kw):
        super().__init__(X, y, nfrequencies, vffdim1lims, vffdim2lims, **kw)
        self.nsplines = nsplines
        self.nknots = nsplines + 1
        self.griddim1lims, self.griddim2lims = griddim1lims, griddim2lims
        self.mesh_1 = torch.linspace(griddim1lims[0], griddim1lims[1], self.nknots)
        self.mesh_2 = torch.linspace(griddim2lims[0], griddim2lims[1], self.nknots)
        self.delta_1 = self.mesh_1[1] - self.mesh_1[0]
        self.delta_2 = self.mesh_2[1] - self.mesh_2[0]

    @staticmethod
    def _Kvu_along_dim(mesh: torch.Tensor, a: float, omegas: torch.Tensor) -> torch.Tensor:
        """:499-541: cell integrals of 1, cos(w (t - a)), sin(w (t - a)); (nsplines, 2M + 1), no hyper-parameter."""
        me, om = mesh.double(), omegas.double()
        k0 = torch.ones(me.shape[0] - 1, 1, dtype=torch.float64) * (me[1] - me[0])
        kc = (torch.sin(om[1:] * (me[1:] - a)[:, None]) - torch.sin(om[1:] * (me[:-1] - a)[:, None])) / om[1:]
        ks = -(torch.cos(om[1:] * (me[1:] - a)[:, None]) - torch.cos(om[1:] * (me[:-1] - a)[:, None])) / om[1:]
        return torch.cat([k0, kc, ks], dim=1)

    def q_u(self) -> MultivariateNormal:
        """:613-624 -- the posterior over the Fourier features (what the parent class calls q_v)."""
        return super().q_v()

    def q_v(self, psd: bool = True, literal: bool = True) -> MultivariateNormal:
        """:634-654 (mean and the diagonal of the covariance; flat index a * nsplines + b)."""
        self._refresh()
        C1 = self._Kvu_along_dim(self.mesh_1, self.dim1lims[0], self.omegas_1)
        C2 = self._Kvu_along_dim(self.mesh_2, self.dim2lims[0], self.omegas_2)
        l1 = self.kernel_1.base_kernel.lengthscale.reshape(()).item()
        l2 = self.kernel_2.base_kernel.lengthscale.reshape(()).item()
        kd1 = torch.full((self.nsplines,), _b0_kvv_diag_unit(float(self.delta_1.double()), l1), dtype=torch.float64)
        kd2 = torch.full((self.nsplines,), _b0_kvv_diag_unit(float(self.delta_2.double()), l2), dtype=torch.float64)
        mean, var = self._engine.readout(C1, C2, kd1, kd2, literal=literal, masked=self._masked)
        return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu())


def _b0_cross_points(mesh: torch.Tensor, z: torch.Tensor, ell: float) -> torch.Tensor:
    """Unit-outputscale Cov(v_k, f(z)) for B0 cells on `mesh` (gridded_kronecker_structure.py:281-321): (nsplines, len(z)).
    Host-side set-up of the read-out (a few hundred numbers); cell k = (mesh[k], mesh[k+1]] like searchsorted(right=False)."""
    me, zz = mesh.double(), z.double()
    a, b = me[:-1, None], me[1:, None]
    E1, E2 = ell * torch.exp(-(zz[None, :] - a).abs() / ell), ell * torch.exp(-(zz[None, :] - b).abs() / ell)
    inside = (zz[None, :] > a) & (zz[None, :] <= b)
    sign = torch.where(zz[None, :] <= a, 1.0, -1.0)
    return torch.where(inside, 2 * ell - (E1 + E2), sign * (E1 - E2))


class _GriddedReadout:
    """q_u -> p(v | u) -> q_v on an nsplines x nsplines B0 grid through vggp_readout (mixin of the Gridded* classes)."""

    def _grid_init(self, nsplines: int, griddim1lims, griddim2lims):
        self.n_b0_splines = self.nsplines = nsplines
        self.dim1_grid_lims, self.dim2_grid_lims = griddim1lims, griddim2lims
        self.b0_mesh_1 = torch.linspace(griddim1lims[0], griddim1lims[1], nsplines + 1)
        self.b0_mesh_2 = torch.linspace(griddim2lims[0], griddim2lims[1], nsplines + 1)
        self.b0_delta_1 = self.b0_mesh_1[1] - self.b0_mesh_1[0]
        self.b0_delta_2 = self.b0_mesh_2[1] - self.b0_mesh_2[0]
        self.b0_basis_1 = B0SplineBasis(self.b0_mesh_1, self._engine)
        self.b0_basis_2 = B0SplineBasis(self.b0_mesh_2, self._engine)

    def _cross(self, d: int, ell: float) -> torch.Tensor:        # C_d (nsplines x m_d) at unit outputscale
        raise NotImplementedError

    def q_u(self) -> MultivariateNormal:
        """The posterior over the inducing features (what the Kronecker base class calls q_v)."""
        return KroneckerStructure.q_v(self)

    def q_v(self, psd: bool = True, literal: bool = True) -> MultivariateNormal:
        """mean and the diagonal of the covariance of the B0 cell features (flat index a * nsplines + b); the variance is the
        reference's own expression (literal=True) unless literal=False asks for the conditional variance under q(u)."""
        self._refresh()
        l1 = self.kernel_1.base_kernel.lengthscale.reshape(()).item()
        l2 = self.kernel_2.base_kernel.lengthscale.reshape(()).item()
        kd1 = torch.full((self.nsplines,), _b0_kvv_diag_unit(float(self.b0_delta_1.double()), l1), dtype=torch.float64)
        kd2 = torch.full((self.nsplines,), _b0_kvv_diag_unit(float(self.b0_delta_2.double()), l2), dtype=torch.float64)
        mean, var = self._engine.readout(self._cross(0, l1), self._cross(1, l2), kd1, kd2, literal=literal, masked=self._masked)
        return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu())


class Matern12B1SplineASVGP(KroneckerStructure):
    """kronecker_structure.py:524-660: B1-spline (hat function) inducing features, Matern-1/2.
    Kuu_d = (A ell + B / ell + BC) / (2 s_d) (:560-614, tridiagonal), Kuf_d = hats(x) (:616-628)."""

    def __init__(self, X, y, nknots: int, dim1lims: Tuple[float, float], dim2lims: Tuple[float, float], **kw):
        super().__init__(X, y, **kw)
        self.nknots = nknots
        self.dim1lims, self.dim2lims = dim1lims, dim2lims
        self.mesh_1 = torch.linspace(dim1lims[0], dim1lims[1], nknots)
        self.mesh_2 = torch.linspace(dim2lims[0], dim2lims[1], nknots)
        self.delta_1 = self.mesh_1[1] - self.mesh_1[0]
        self.delta_2 = self.mesh_2[1] - self.mesh_2[0]
        self.delta = self.delta_1
        self.basis_1 = B1SplineBasis(self.mesh_1, self._engine)      # kronecker_structure.py:548-549
        self.basis_2 = B1SplineBasis(self.mesh_2, self._engine)

    def _basis(self):
        return "b1", self.mesh_1.double().numpy(), self.mesh_2.double().numpy()

    def _f32_mesh(self) -> bool:
        return False                 # the float32 (k*delta) quirk belongs to the B0 closed forms only


class Matern12SVGP(KroneckerStructure):
    """kronecker_structure.py:287-338: Z (m, 2) holds the per-dimension inducing coordinates; the inducing set
    is cartesian_prod(Z[:,0], Z[:,1]) (:336), so Kuu = kron(K1(Z[:,0]), K2(Z[:,1])) (:318-321).
    Z is a trainable Parameter as in the reference (:303-304): `_elbo()` carries its analytic gradient (vggp_zgrad) unless
    `train_z=False`.  Moved inducing points reach the engine through vggp_set_inducing (no re-plan: graphs and warm start stay).  On
    incomplete data (scattered points, or a grid with holes, which is then treated as its observed points) the gradient is
    vggp_zgrad_scattered's."""

    def __init__(self, X, y, Z: torch.Tensor, train_z: bool = True, **kw):
        super().__init__(X, y, **kw)
        self.Z = torch.nn.Parameter(torch.as_tensor(Z, dtype=torch.float64).clone(), requires_grad=bool(train_z))
        if train_z:
            self._as_scattered()

    def _elbo(self) -> torch.Tensor:
        return _ElboFunction.apply(self._theta(), self, self.Z)

    def _basis(self):
        Z = self.Z.detach().cpu().numpy()
        return "points", Z[:, 0].copy(), Z[:, 1].copy()


class Matern32SVGP(Matern12SVGP):
    kind = "matern32"


class Matern52SVGP(Matern12SVGP):
    kind = "matern52"


class RBFSVGP(Matern12SVGP):
    kind = "rbf"


# ---------------------------------------------------------------------------------------------
# 1-D models: the same engine with a trivial second factor (VGGP_BASIS_ONE)
# ---------------------------------------------------------------------------------------------
class _SparseGP1D(torch.nn.Module):
    """univariate_structure.py:15-263 (SparseGP): one factor; kernel = ScaleKernel(Matern)."""

    kind = "matern12"

    def __init__(self, X: torch.Tensor, y: torch.Tensor, engine: Optional[Engine] = None, warm_start: bool = True):
        super().__init__()
        self.train_inputs = (X,)
        self.train_targets = y
        self.likelihood = GaussianLikelihood()
        self.kernel = ScaleKernel(_BaseKernel(self.kind))
        self._engine = engine if engine is not None else Engine()
        self._warm = warm_start
        self._planned = False
        self._plan_token, self._plan_key = -1, None
        self._masked = False
        self.last_info = None
        self._x = torch.as_tensor(X, dtype=torch.float64).reshape(-1).numpy().copy()
        self._Y = torch.as_tensor(y, dtype=torch.float64).reshape(1, -1).contiguous().to(self._engine.device)
        self._yy = self._engine.sumsq(self._Y)

    def _basis(self):
        raise NotImplementedError

    def _plan(self):
        basis, g = self._basis()
        key = (basis, np.asarray(g).tobytes())
        if self._planned and self._plan_token == self._engine.plan_token and key != self._plan_key and basis == "points" \
                and self._plan_key is not None and len(self._plan_key[1]) == len(key[1]):
            self._engine.set_inducing(0, g)          # only Z moved
            self._plan_key = key
        elif not self._planned or self._plan_token != self._engine.plan_token or key != self._plan_key:
            mesh = getattr(self, "mesh", None)
            self._engine.plan(self.kind, basis, g, self._x, "matern12", "one", None, np.zeros(1), warm_start=self._warm,
                              b0_f32_kdelta=mesh is not None and mesh.dtype == torch.float32)
            self._planned = True
            self._plan_token, self._plan_key = self._engine.plan_token, key

    def _theta(self):
        one = torch.ones((), dtype=torch.float64)
        return torch.stack([self.kernel.base_kernel.lengthscale.reshape(()).double(), one,
                            self.kernel.outputscale.reshape(()).double(), one,
                            self.likelihood.noise.reshape(()).double()])

    def _engine_step(self, theta):
        self._plan()
        if self._masked:
            return self._engine.elbo_step_masked(self._Y, self._W, self._nobs, self._yy, theta)
        return self._engine.elbo_step(self._Y, self._yy, theta)

    def _elbo(self):
        """univariate_structure.py:234-263."""
        return _ElboFunction.apply(self._theta(), self)

    def _refresh(self):
        with torch.no_grad():
            _ElboFunction.apply(self._theta(), self)

    def q_v(self) -> MultivariateNormal:
        """univariate_structure.py:693-717."""
        self._refresh()
        if self._masked:
            mean, var = self._engine.qv_masked()
            return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu())
        mean, var = self._engine.qv()
        return MultivariateNormal(mean.reshape(-1).cpu(), var.reshape(-1).cpu(),
                                  cov_fn=lambda: self._engine.qv_cov().cpu())

    def posterior(self, x_star) -> MultivariateNormal:
        """univariate_structure.py:184-215 (mean and diagonal)."""
        self._refresh()
        mean, var = self._engine.posterior(torch.as_tensor(x_star, dtype=torch.float64).reshape(-1))
        return MultivariateNormal(mean.cpu(), var.cpu())

    def posterior_predictive(self, x_star) -> MultivariateNormal:
        p = self.posterior(x_star)
        return MultivariateNormal(p.mean, p.variance + self.likelihood.noise.detach().to(p.variance.dtype))

    def non_informative_initialise(self, lmbda: float, kappa: float) -> None:
        """univariate_structure.py:45-66 (same lengthscale-getter quirk)."""
        X, y = self.train_inputs[0], self.train_targets
        self.kernel.outputscale = y.var()
        self.likelihood.noise = self.kernel.outputscale / (kappa ** 2)
        self.kernel.base_kernel.lengthscale[0] = (X.std() / lmbda)

    def informative_initialise(self, prior_amplitude: float, lmbda: float) -> None:
        """univariate_structure.py:68-87."""
        X, y = self.train_inputs[0], self.train_targets
        self.kernel.outputscale = (torch.tensor(prior_amplitude) / 2) ** 2
        self.likelihood.noise = y.var() - self.kernel.outputscale
        self.kernel.base_kernel.lengthscale[0] = (X.std() / lmbda)


class univariate:
    """Namespace for the 1-D classes (they share names with the 2-D ones in the reference's other module)."""

    class Matern12B0SplineGriddedGP(_SparseGP1D):
        """univariate_structure.py:721-825."""

        def __init__(self, X, y, nknots: int, dim1lims: Tuple[float, float], **kw):
            super().__init__(X, y, **kw)
            self.nknots = nknots
            self.alim, self.blim = dim1lims
            self.mesh = torch.linspace(self.alim, self.blim, nknots)
            self.delta = self.mesh[1] - self.mesh[0]
            self.n_splines = nknots - 1
            self.basis = B0SplineBasis(self.mesh, self._engine)       # univariate_structure.py:738

        def _basis(self):
            return "b0", self.mesh.double().numpy()

    class Matern12SVGP(_SparseGP1D):
        """univariate_structure.py:273-332."""

        def __init__(self, X, y, Z, train_z: bool = True, **kw):
            super().__init__(X, y, **kw)
            self.Z = torch.nn.Parameter(torch.as_tensor(Z, dtype=torch.float64).reshape(-1, 1).clone(),
                                        requires_grad=bool(train_z))

        def _elbo(self):
            return _ElboFunction.apply(self._theta(), self, self.Z)

        def _basis(self):
            return "points", self.Z.detach().cpu().numpy().reshape(-1).copy()

    class Matern32SVGP(Matern12SVGP):
        kind = "matern32"

    class Matern52SVGP(Matern12SVGP):
        kind = "matern52"


class GriddedMatern12SVGP(_GriddedReadout, KroneckerStructure):
    """gridded_kronecker_structure.py:222-460: inducing POINTS Z (M, 2) with a gridded read-out on B0 cells.  The reference
    evaluates the product kernel on the rows of Z (Kuu = k(Z, Z), :252-264): that is a Kronecker product exactly when Z is the
    cartesian product of its per-dimension coordinates, which is the case this engine covers -- Z must list
    cartesian_prod(z1, z2) (either coordinate fastest); arbitrary scattered Z has no per-dimension factors."""

    def __init__(self, X, y, Z: torch.Tensor, n_b0_splines: int, dim1_grid_lims, dim2_grid_lims, **kw):
        KroneckerStructure.__init__(self, X, y, **kw)
        Zt = torch.as_tensor(Z, dtype=torch.float64)
        self.Z = torch.nn.Parameter(Zt.clone(), requires_grad=False)
        self._z1, self._z2, self._u_of_row = _detect_cartesian(Zt)
        self._grid_init(n_b0_splines, dim1_grid_lims, dim2_grid_lims)

    def _basis(self):
        return "points", self._z1.copy(), self._z2.copy()

    def _cross(self, d: int, ell: float) -> torch.Tensor:
        return _b0_cross_points(self.b0_mesh_1 if d == 0 else self.b0_mesh_2, torch.as_tensor(self._z1 if d == 0 else self._z2), ell)

    def q_u(self) -> MultivariateNormal:
        """:396-405, in the row order of the caller's Z."""
        qu = KroneckerStructure.q_v(self)
        idx = torch.as_tensor(self._u_of_row)
        return MultivariateNormal(qu.mean[idx], qu.variance[idx])


def _cluster_coordinates(v: np.ndarray, rtol: float):
    """Distinct values of v up to rtol * span(v) (coordinates that went through a float32 round trip or a scaler differ in the
    last bits): (centres, index of each entry's centre); a centre is the mean of its cluster."""
    order = np.argsort(v, kind="stable")
    vs = v[order]
    tol = rtol * max(float(vs[-1] - vs[0]), np.finfo(np.float64).tiny) if len(vs) else 0.0
    new = np.concatenate([[True], np.diff(vs) > tol]) if len(vs) else np.zeros(0, bool)
    gid = np.cumsum(new) - 1
    centres = np.bincount(gid, weights=vs) / np.bincount(gid)
    inv = np.empty(len(v), dtype=np.int64)
    inv[order] = gid
    return centres, inv


def _detect_cartesian(Z: torch.Tensor, rtol: float = 1e-6):
    """Z (M, 2) listing every pair of cartesian_prod(z1, z2) exactly once, in any row order -> (z1, z2, u_of_row) with
    u_of_row[r] = i1 * m2 + i2, the engine's inducing index of row r; raises when Z is not such a grid.  Coordinates are
    matched up to rtol of their span, so a grid that passed through float32 or a scaler is still recognised."""
    Zn = Z.detach().cpu().numpy().astype(np.float64)
    if Zn.ndim != 2 or Zn.shape[1] != 2:
        raise ValueError("Z must be (M, 2)")
    z1, i1 = _cluster_coordinates(Zn[:, 0], rtol)
    z2, i2 = _cluster_coordinates(Zn[:, 1], rtol)
    u = i1 * len(z2) + i2
    if len(z1) * len(z2) != Zn.shape[0] or len(np.unique(u)) != Zn.shape[0]:
        raise ValueError("GriddedMatern12SVGP: Z must be a full cartesian grid cartesian_prod(z1, z2): only then is "
                         "k(Z, Z) = kron(K1, K2) (arbitrary scattered inducing points are out of scope)")
    return z1, z2, u


class GriddedMatern12ASVGP(_GriddedReadout, Matern12B1SplineASVGP):
    """gridded_kronecker_structure.py:685-969: B1-spline inducing features on the B0 mesh padded by `padding_factor` knots on
    either side (:699-724), gridded read-out with the reference's own Cov(v, u) -- delta at the two knots of each cell
    (:831-845; the reference's value, not the exact RKHS cross-covariance)."""

    def __init__(self, X, y, n_b0_splines: int, padding_factor: int, dim1_grid_lims, dim2_grid_lims, **kw):
        KroneckerStructure.__init__(self, X, y, **kw)
        self._grid_init(n_b0_splines, dim1_grid_lims, dim2_grid_lims)
        self.padding_factor = padding_factor

        def padded(mesh, delta):
            left = torch.tensor([(mesh[0] - i * delta).item() for i in range(padding_factor, 0, -1)])
            right = torch.tensor([(mesh[-1] + i * delta).item() for i in range(1, padding_factor + 1)])
            return torch.cat((left, mesh, right))
        self.b0_mesh_padded_1 = padded(self.b0_mesh_1, self.b0_delta_1)
        self.b0_mesh_padded_2 = padded(self.b0_mesh_2, self.b0_delta_2)
        self.mesh_1, self.mesh_2 = self.b0_mesh_padded_1, self.b0_mesh_padded_2          # the B1 knots of the parent class
        self.nknots = self.mesh_1.shape[0]
        self.delta_1, self.delta_2 = self.mesh_1[1] - self.mesh_1[0], self.mesh_2[1] - self.mesh_2[0]
        self.delta = self.delta_1
        self.b1_basis_1 = self.basis_1 = B1SplineBasis(self.mesh_1, self._engine)
        self.b1_basis_2 = self.basis_2 = B1SplineBasis(self.mesh_2, self._engine)

    def _cross(self, d: int, ell: float) -> torch.Tensor:
        mesh = self.mesh_1 if d == 0 else self.mesh_2
        K, ns, pad = mesh.shape[0], self.nsplines, self.padding_factor
        C = torch.zeros(ns, K, dtype=torch.float64)
        delta = float((mesh[1] - mesh[0]).double())
        idx = torch.arange(ns)
        C[idx, pad + idx] = delta
        C[idx, pad + idx + 1] = delta
        return C
