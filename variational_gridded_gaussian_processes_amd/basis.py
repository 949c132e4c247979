"""Host-side mirrors of the reference's inducing-feature basis objects, as far as the model classes expose them.

    reference                                   here
    src/basis/bspline.py:81-103   (SplineBasis / B0SplineBasis)          B0SplineBasis
    src/basis/bspline.py:106-112  (B1SplineBasis)                        B1SplineBasis
    src/basis/fourier.py:5-88     (FourierBasis / FourierBasisMatern12)  FourierBasisMatern12

The models read `.mesh`, `.m`, `.delta`, `.n_basis_functions`, `.order`, `.omegas`, `.a`, `.b`, `.M`, `.lengthscale` and
(B1 / Fourier only) call the basis on a coordinate vector.  `__call__` returns the (n_basis x n) feature matrix; with an
Engine attached the evaluation runs through vggp_factor_build (the HIP factor kernel) -- the same numbers the step uses --
otherwise through a few torch ops (the B0 indicators, which the hot path never evaluates).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


class SplineBasis:
    """bspline.py:81-94: `m = len(mesh) - (order + 1)`, `delta = mesh[1] - mesh[0]` (the mesh keeps its dtype: the reference's
    float32 meshes are what make the float32 k*delta quirk of the B0 closed forms)."""

    order = 0

    def __init__(self, mesh: torch.Tensor, engine=None):
        self.mesh = mesh
        self.m = mesh.size(0) - (self.order + 1)
        self.delta = mesh[1] - mesh[0]
        self._engine = engine

    def _features(self, kind: str, basis: str, grid: np.ndarray, x: torch.Tensor, ell: float) -> torch.Tensor:
        if self._engine is None:
            raise RuntimeError("this basis object has no Engine attached: evaluation runs on the GPU (no CPU path)")
        eng = self._engine
        xs = torch.as_tensor(x, dtype=torch.float64, device=eng.device).reshape(-1).contiguous()
        g = torch.as_tensor(np.asarray(grid, dtype=np.float64), device=eng.device)
        A, _, _, _ = eng.factor_build(kind, basis, xs, g, ell)
        return A.cpu()


class B0SplineBasis(SplineBasis):
    """bspline.py:97-103: order-0 B-splines (cell indicators on [v_k, v_k+1]); m = nknots - 1 cells."""

    order = 0

    def __init__(self, mesh: torch.Tensor, engine=None):
        super().__init__(mesh, engine)
        self.n_basis_functions = int(self.m)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = torch.as_tensor(x)
        lo, hi = self.mesh[:-1, None], self.mesh[1:, None]
        return torch.logical_and(x[None, :] >= lo, x[None, :] <= hi) * 1


class B1SplineBasis(SplineBasis):
    """bspline.py:106-112: hat functions on the knots (two half hats at the ends); n_basis_functions = nknots."""

    order = 1

    def __init__(self, mesh: torch.Tensor, engine=None):
        super().__init__(mesh, engine)
        self.n_basis_functions = int(mesh.size(0))

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return self._features("matern12", "b1", self.mesh.double().numpy(), x, 1.0)


class FourierBasisMatern12:
    """fourier.py:5-88: M + 1 cosine and M sine features on [a, b), exp(-r / ell) tails outside; omegas are float32 like the
    reference's (python float * int64 arange / python float)."""

    def __init__(self, n_frequencies: int, a: float, b: float, lengthscale: float, engine=None):
        self.M, self.a, self.b, self.lengthscale = n_frequencies, a, b, lengthscale
        self.lmbda = 1 / lengthscale
        self.omegas = (2 * torch.pi) * torch.arange(self.M + 1) / (b - a)
        self.n_basis_functions = 2 * self.M + 1
        self._engine = engine

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        grid = np.concatenate([[self.a, self.b], self.omegas.double().numpy()])
        return SplineBasis._features(self, "matern12", "vff", grid, x, float(self.lengthscale))
