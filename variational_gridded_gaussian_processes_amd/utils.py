"""Host-side helpers around the fit / predict path: the two predictive metrics the reference's notebook
61 imports but does not ship, and the gridding of a reference field into cell integrals.  Plain torch / numpy on the host -- none
of this is on the hot path.  (The reference's point metrics and scalers -- evaluationmetrics.py, dataprocessors.py -- are out of
scope, SURVEY.md section 2 rows 8-9: use the reference's own.)

Array-based: the reference's loaders wrap xarray datasets read from netCDF files (src/utils/dataloaders.py); neither xarray nor
the files exist here, so the functions take the arrays those loaders would hand over.  nlpd / msll: standard definitions
(Rasmussen & Williams section 2.5).  grid_cells mirrors GulfStream.grid_ref_data_simpson / grid_ref_data_trapz
(dataloaders.py:485-539).  The synthetic track generator (generate_track's array-level twin) lives in datagen.py."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch


def _pair(true: torch.Tensor, pred: torch.Tensor) -> torch.Tensor:
    """The reference's argument contract (evaluationmetrics.py:11-14): two 2-D tensors of one shape -> residuals."""
    for name, t in (("true", true), ("pred", pred)):
        assert t.dim() == 2, f"{name} tensor must be 2D, got {t.dim()}D"
    assert true.shape == pred.shape, f"true and pred must have the same shape, got {tuple(true.shape)} and {tuple(pred.shape)}"
    return true - pred


def nlpd(true: torch.Tensor, mean: torch.Tensor, var: torch.Tensor) -> torch.Tensor:
    """Negative log predictive density of Gaussian marginals, averaged over the points: mean_i [ (y_i - m_i)^2 / (2 v_i) +
    log(2 pi v_i) / 2 ].  var is the predictive variance INCLUDING the observation noise when `true` is noisy."""
    res = _pair(true, mean)
    assert var.shape == mean.shape and bool((var > 0).all()), "var must be positive and shaped like mean"
    return (0.5 * res.square() / var + 0.5 * torch.log(2.0 * math.pi * var)).mean()


def msll(true: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, train_targets: torch.Tensor) -> torch.Tensor:
    """Mean standardised log loss: nlpd of the model minus the nlpd of the trivial Gaussian with the training targets' mean and
    variance (negative = better than trivial)."""
    mu0, v0 = train_targets.mean(), train_targets.var(unbiased=False)
    return nlpd(true, mean, var) - nlpd(true, torch.full_like(mean, float(mu0)), torch.full_like(var, float(v0)))


# ---- reference field -> cell integrals (what q_v() of the Gridded* models is compared with) -------------------------------------
def grid_cells(field: np.ndarray, lon: np.ndarray, lat: np.ndarray, n_grids: int, rule: str = "simpson") -> np.ndarray:
    """Integrals of a regularly sampled field over an n_grids x n_grids partition of its index range
    (dataloaders.py:513-539 `grid_ref_data_simpson`, :485-511 `grid_ref_data_trapz`; there the field is the time-mean SSH of a
    600 x 600 dataset).  Cell (i, j) covers field[i*P:(i+1)*P, j*P:(j+1)*P] with P = field.shape[0] // n_grids; as in the
    reference the inner rule runs along axis 1 with spacing lon[1]-lon[0] of the cell's lon slice (indexed by i) and the outer
    one with the lat spacing (indexed by j).  -> (n_grids, n_grids)."""
    from scipy.integrate import simpson, trapezoid
    quad = {"simpson": simpson, "trapz": trapezoid}[rule]
    field = np.asarray(field, float)
    P = field.shape[0] // n_grids
    out = np.empty((n_grids, n_grids))
    for i in range(n_grids):
        dlon = float(lon[i * P + 1] - lon[i * P])
        for j in range(n_grids):
            dlat = float(lat[j * P + 1] - lat[j * P])
            cell = field[i * P:(i + 1) * P, j * P:(j + 1) * P]
            out[i, j] = quad(quad(cell, dx=dlon, axis=1), dx=dlat)
    return out
