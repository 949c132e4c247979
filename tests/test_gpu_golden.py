"""GPU parity against the committed golden fixtures (tests/golden/*.npz), through the C-ABI."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "oracle_*x*.npz")) if "mask" not in p)
MASK_CASES = sorted(glob.glob(os.path.join(GOLD, "oracle_mask*.npz")))
RTOL = 1e-7          # north-star tolerance is 1e-5; float64 end to end does far better


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_engine_vs_golden(engine, path):
    g = np.load(path)
    basis, kind = str(g["basis"]), str(g["kind"])
    n1, n2 = len(g["x1"]), len(g["x2"])
    engine.plan(kind, basis, g["grid1"], g["x1"], kind, basis, g["grid2"], g["x2"],
                b0_f32_kdelta=bool(g["mesh_is_f32"]))
    Y = torch.tensor(g["y"].reshape(n2, n1), device="cuda")
    elbo, grad, info = engine.elbo_step(Y, engine.sumsq(Y), g["theta"])
    assert info["jitter"] == tuple(g["jitter"])
    assert abs(elbo - g["elbo"]) <= RTOL * abs(g["elbo"])
    graw = grad / (1.0 + np.exp(-g["raw"]))
    assert rel(graw, g["grad_raw"]) < RTOL
    mean, var = engine.qv()
    assert rel(mean.cpu().numpy().reshape(-1), g["qv_mean"]) < RTOL
    assert rel(var.cpu().numpy().reshape(-1), g["qv_var"]) < RTOL
    pm, pv = engine.posterior(torch.tensor(g["xs"], device="cuda"))
    assert rel(pm.cpu().numpy(), g["post_mean"]) < RTOL
    assert rel(pv.cpu().numpy(), g["post_var"]) < 1e-6


@pytest.mark.parametrize("path", MASK_CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_engine_masked_vs_golden(engine, path):
    """BASELINE config 5: golden = dense restatement run on the observed subset only."""
    g = np.load(path)
    basis, kind = str(g["basis"]), str(g["kind"])
    n1, n2 = len(g["x1"]), len(g["x2"])
    engine.plan(kind, basis, g["grid1"], g["x1"], kind, basis, g["grid2"], g["x2"])
    W = torch.tensor(g["W"].astype(np.float64), device="cuda")
    Ym = torch.tensor(g["y"].reshape(n2, n1), device="cuda") * W
    elbo, grad, info = engine.elbo_step_masked(Ym, W, float(g["W"].sum()), engine.sumsq(Ym), g["theta"])
    assert info["status"] == 0
    assert abs(elbo - g["elbo"]) <= RTOL * abs(g["elbo"])
    assert rel(grad / (1.0 + np.exp(-g["raw"])), g["grad_raw"]) < RTOL
    mean, var = engine.qv_masked()
    assert rel(mean.cpu().numpy().reshape(-1), g["qv_mean"]) < RTOL
    assert rel(var.cpu().numpy().reshape(-1), g["qv_var"]) < RTOL


def test_engine_vs_golden_1d(engine):
    g = np.load(os.path.join(GOLD, "oracle_1d_b0_256.npz"))
    th = g["theta"]
    engine.plan("matern12", "b0", g["mesh"], g["x"], "matern12", "one", None, np.zeros(1))
    Y = torch.tensor(g["y"].reshape(1, -1), device="cuda")
    elbo, grad, _ = engine.elbo_step(Y, engine.sumsq(Y), [th[0], 1.0, th[1], 1.0, th[2]])
    assert abs(elbo - g["elbo"]) <= RTOL * abs(g["elbo"])
    assert rel(grad[[0, 2, 4]] / (1.0 + np.exp(-g["raw"])), g["grad_raw"]) < RTOL
    mean, var = engine.qv()
    assert rel(mean.cpu().numpy().reshape(-1), g["qv_mean"]) < RTOL
    pm, pv = engine.posterior(torch.tensor(g["xs"], device="cuda"))
    assert rel(pm.cpu().numpy(), g["post_mean"]) < RTOL and rel(pv.cpu().numpy(), g["post_var"]) < 1e-6
