"""CPU tests of the host-side data front-end: gen_2d's point order and the array-level twin of
SimulationDataHour.generate_track (dataloaders.py:290-377).  The reference's loader cannot be imported here (xarray is absent),
so the track generator is pinned on counts and index sequences derived BY HAND from the reference's loop bounds at its own
600 x 600 / 10-degree geometry, and on its structural properties."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dense as D                                                 # noqa: E402
from variational_gridded_gaussian_processes_amd import datagen as G          # noqa: E402


def test_gen_grid_is_the_oracles_and_gen_2ds_layout():
    X, y, x1, x2 = G.gen_grid(7, 5, lims2=(-1.0, 2.0), seed=3)
    Xo, yo, x1o, x2o = D.gen_grid(7, 5, lims2=(-1.0, 2.0), seed=3)
    assert np.array_equal(X, Xo) and np.array_equal(y, yo) and np.array_equal(x1, x1o) and np.array_equal(x2, x2o)
    X2, y2 = G.gen_2d(G.latent_2d, (0, 1), (0, 1), 6)
    assert X2.shape == (36, 2) and np.array_equal(X2[:6, 0], np.linspace(0, 1, 6)) and np.all(X2[:6, 1] == 0.0)   # x1 fastest
    assert np.allclose(y2, G.latent_2d(X2[:, 0], X2[:, 1]))


def test_generate_track_counts_at_the_reference_geometry():
    """600 x 600, gradient 2, one track per degree: family 1 lays 10 tracks of 2 * 2 * min(300, 600 - 60 i) points
    (6 * 1200 + 960 + 720 + 480 + 240 = 9600), family 2 five tracks of 2 * (600 - 120 j) points (3600)."""
    li, la = G.generate_track(600, 600, 2, 1.0)
    assert len(li) == len(la) == 13200
    assert li.min() >= 0 and li.max() <= 599 and la.min() >= 0 and la.max() <= 599
    # first track, forward leg: columns 0,0,1,1,... rows 0,1,2,3,...; backward leg: same columns, rows 599, 598, ...
    assert np.array_equal(li[:6], [0, 0, 1, 1, 2, 2]) and np.array_equal(la[:6], np.arange(6))
    assert np.array_equal(li[600:606], [0, 0, 1, 1, 2, 2]) and np.array_equal(la[600:606], 599 - np.arange(6))
    # second track starts one degree (60 columns) further east
    assert li[1200] == 60 and la[1200] == 0
    # family 2, second track (j = 1): rows 120.., columns 0,0,1,1,...; its mirror starts at row -120 = 480 and walks down
    o = 9600 + 1200
    assert la[o] == 120 and li[o] == 0 and la[o + 479] == 599 and li[o + 479] == 239
    assert la[o + 480] == 480 and la[o + 481] == 479 and li[o + 480] == 0
    # every k-th observation, as the reference's final slice
    l5, a5 = G.generate_track(600, 600, 2, 1.0, observation_sparsity=5)
    assert np.array_equal(l5, li[::5]) and np.array_equal(a5, la[::5])


def test_track_mask_and_points_agree():
    n = 120
    lon, lat = np.linspace(0, 10, n), np.linspace(30, 40, n)
    field = np.add.outer(np.sin(lat), np.cos(lon))                      # [lat, lon]
    tl, ta, tv = G.track_points(field, lon, lat, 2, 1.0)
    W = G.track_mask(n, n, 2, 1.0)
    li, la = G.generate_track(n, n, 2, 1.0)
    assert W.shape == (n, n) and set(np.unique(W)) == {0.0, 1.0}
    assert W.sum() == len(set(zip(la.tolist(), li.tolist())))          # crossings are merged in the mask, kept in the points
    assert np.allclose(tv, np.sin(ta) + np.cos(tl))
    assert 0.02 < W.mean() < 0.5
    try:
        G.generate_track(n, n, 2, 0.0)
        assert False
    except ValueError as e:
        assert "Track sparsity" in str(e)
