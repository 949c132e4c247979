"""Synthetic data in the layouts the reference's harness produces (host side, numpy only; nothing here is on the hot path).

* `gen_2d` / `gen_grid` -- gridded observations in the point order of `src/utils/datagenerators.py:37-73` (`np.meshgrid(x1, x2)`
  in 'xy' indexing, ravelled: point p = j*n1 + i is (x1[i], x2[j]), x1 fastest), which is the order `Y[n2][n1]` of the engine.
* `generate_track` -- array-level twin of `SimulationDataHour.generate_track` (`src/utils/dataloaders.py:290-377`): the index
  sets of crossing "ascending / descending" satellite tracks over a regular (lat, lon) field, without xarray or netCDF
  (neither exists here, and the NATL60 files are git-ignored in the reference).  `track_mask` turns them into the (grid, mask)
  pair the masked step takes (BASELINE configs[4]'s "track-shaped" variant), `track_points` into the
  scattered (X, y) the reference's notebooks 6 / 61 hand to `Matern12GriddedGP`.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def latent_2d(x1, x2):
    """The notebooks' 2-D test function (5_gridded_kronecker_structure_models.ipynb cell 3)."""
    return (np.sin(5 * x1) + np.cos(7 * x2) + 0.5 * np.sin(15 * x1) + 0.5 * np.cos(12 * x2)
            + 0.2 * np.sin(20 * x1) + 0.2 * np.cos(25 * x2))


def gen_1d(fun: Callable, leftlim: float, rightlim: float, nobs: int, randomspacing: bool = False, rng=None):
    """datagenerators.py:8-34: (domain, fun(domain)); evenly spaced unless randomspacing."""
    if randomspacing:
        rng = np.random.default_rng() if rng is None else rng
        domain = rng.random(nobs) * (rightlim - leftlim) + leftlim
    else:
        domain = np.linspace(leftlim, rightlim, nobs)
    return domain, fun(domain)


def gen_2d(func: Callable, x1lims: Tuple[float, float], x2lims: Tuple[float, float], nobs: int, randomspacing: bool = False,
           rng=None):
    """datagenerators.py:37-73: X (nobs^2, 2) with x1 fastest, y = func(X[:,0], X[:,1])."""
    if randomspacing:
        rng = np.random.default_rng() if rng is None else rng
        d1 = rng.random(nobs) * (x1lims[1] - x1lims[0]) + x1lims[0]
        d2 = rng.random(nobs) * (x2lims[1] - x2lims[0]) + x2lims[0]
    else:
        d1, d2 = np.linspace(x1lims[0], x1lims[1], nobs), np.linspace(x2lims[0], x2lims[1], nobs)
    X1, X2 = np.meshgrid(d1, d2)
    X = np.vstack([X1.ravel(), X2.ravel()]).T
    return X, func(X[:, 0], X[:, 1])


def gen_grid(n1: int, n2: int, lims1=(0.0, 1.0), lims2=(0.0, 1.0), noise: float = 0.05, seed: int = 0, latent=latent_2d):
    """Rectangular n1 x n2 version of gen_2d with seeded observation noise: X (N, 2) with x1 fastest, y (N,), x1, x2."""
    x1 = np.linspace(lims1[0], lims1[1], n1)
    x2 = np.linspace(lims2[0], lims2[1], n2)
    X1, X2 = np.meshgrid(x1, x2)                    # 'xy' indexing: shape (n2, n1)
    X = np.vstack([X1.ravel(), X2.ravel()]).T
    y = latent(X[:, 0], X[:, 1]) + noise * np.random.default_rng(seed).standard_normal(X.shape[0])
    return X, y, x1, x2


# ---- satellite-track sampling of a gridded field ---------------------------------------------------------------------------------
def generate_track(lon_dim: int, lat_dim: int, trajectory_gradient: int, track_sparsity: float, observation_sparsity: int = 0,
                   degree_range: float = 10.0) -> Tuple[np.ndarray, np.ndarray]:
    """Index sets (lon_idx, lat_idx) of the synthetic tracks of dataloaders.py:290-377 on a lat_dim x lon_dim field
    (`field[lat_idx, lon_idx]` are the observed values; the reference hard-codes 600 x 600 points over 10 degrees).

    Tracks climb `trajectory_gradient` latitude rows per longitude column.  A first family starts on the bottom edge every
    `track_sparsity` degrees of longitude; each track is laid twice -- once upwards from row 0 ("forward") and once mirrored
    downwards from the top row ("backward": the reference indexes latitude with -1, -2, ..., i.e. from the far edge).  A
    second family starts on the left edge every `track_sparsity * trajectory_gradient` degrees of latitude.  Indices are
    returned non-negative; duplicates (crossings) are kept, in the reference's order; `observation_sparsity` k > 0 keeps every
    k-th point of the concatenated sequence, as the reference's final slice does."""
    if not (0 < track_sparsity <= degree_range):
        raise ValueError(f"Track sparsity must be between 0 and {degree_range:g}. Provided track sparsity: {track_sparsity}")
    if trajectory_gradient < 1:
        raise ValueError("trajectory_gradient must be a positive integer")
    g = int(trajectory_gradient)
    max_lon = int(lon_dim / g)
    lon_parts, lat_parts = [], []

    def lay(cols, rows):                      # one leg of a track: pair column / row indices, drop what leaves the field
        k = min(len(cols), len(rows))
        cols, rows = cols[:k], rows[:k]
        ok = (rows >= 0) & (rows < lat_dim) & (cols >= 0) & (cols < lon_dim)
        lon_parts.append(cols[ok])
        lat_parts.append(rows[ok])

    # family 1: tracks entering through the bottom edge, one every track_sparsity degrees of longitude
    n_lon_tracks = int(degree_range / track_sparsity)
    lon_shift = track_sparsity * (lon_dim / degree_range)
    for i in range(n_lon_tracks):
        start = int(i * lon_shift)
        end = min(max_lon + start, lon_dim)
        cols = np.repeat(np.arange(start, end), g)
        k = len(cols)
        lay(cols, np.arange(k))                                  # forward: rows 0, 1, ...
        lay(cols, lat_dim - 1 - np.arange(k))                    # backward: the reference's rows -1, -2, ... (from the far edge)
    # family 2: tracks entering through the left edge, one every track_sparsity * gradient degrees of latitude
    lat_sparsity = track_sparsity * g
    n_lat_tracks = int(degree_range / lat_sparsity)
    lat_shift = lat_sparsity * (lat_dim / degree_range)
    base_cols = np.repeat(np.arange(0, max_lon), g)
    for j in range(n_lat_tracks):
        start = int(j * lat_shift)
        lay(base_cols, np.arange(start, lat_dim))                              # forward: rows start, start + 1, ...
        lay(base_cols, np.mod(np.arange(-start, -lat_dim, -1), lat_dim))       # backward: the reference's rows -start, -start-1, ...
    lon_idx = np.concatenate(lon_parts).astype(np.int64) if lon_parts else np.empty(0, np.int64)
    lat_idx = np.concatenate(lat_parts).astype(np.int64) if lat_parts else np.empty(0, np.int64)
    if observation_sparsity:
        lon_idx, lat_idx = lon_idx[::observation_sparsity], lat_idx[::observation_sparsity]
    return lon_idx, lat_idx


def track_points(field: np.ndarray, lon: np.ndarray, lat: np.ndarray, trajectory_gradient: int, track_sparsity: float,
                 observation_sparsity: int = 0, degree_range: Optional[float] = None):
    """(track_lon, track_lat, track_values) as `generate_track` returns them in the reference: field is [lat, lon]."""
    field = np.asarray(field)
    dr = float(lon[-1] - lon[0]) if degree_range is None else degree_range
    li, la = generate_track(field.shape[1], field.shape[0], trajectory_gradient, track_sparsity, observation_sparsity, dr)
    return np.asarray(lon)[li], np.asarray(lat)[la], field[la, li]


def track_mask(n1: int, n2: int, trajectory_gradient: int = 2, track_sparsity: float = 0.1, observation_sparsity: int = 0,
               degree_range: float = 10.0) -> np.ndarray:
    """0/1 mask W [n2][n1] (dimension 1 = longitude = the fast axis of Y) of the grid points a track passes through: the
    (grid, mask) form of the same observations, which is what `vggp_elbo_step_masked` / the model classes take."""
    li, la = generate_track(n1, n2, trajectory_gradient, track_sparsity, observation_sparsity, degree_range)
    W = np.zeros((n2, n1))
    W[la, li] = 1.0
    return W
