"""GPU parity of the ELBO step / q(v) / posterior against the CPU oracle (rtol 1e-5 is the
north-star tolerance; float64 agreement is far tighter and asserted at 1e-7 here)."""
import numpy as np
import pytest
import torch

from oracle import dense as D
from oracle import kron as Kr

pytestmark = pytest.mark.gpu
DEV = "cuda"
RTOL = 1e-7


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def run_case(engine, n1, n2, basis, kind, g1, g2, theta, warm=False):
    X, y, x1, x2 = D.gen_grid(n1, n2)
    f1 = Kr.Factor(basis, kind, np.asarray(g1, float), x1)
    f2 = Kr.Factor(basis, kind, np.asarray(g2, float), x2)
    st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
    engine.plan(kind, basis, g1, x1, kind, basis, g2, x2, warm_start=warm)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    elbo, grad, info = engine.elbo_step(Y, yy, theta)
    return st, elbo, grad, info, (f1, f2)


CASES = [
    (24, 20, "b0", "matern12", np.linspace(0, 1, 8), np.linspace(0, 1, 10), [0.2, 0.3, 1.0, 0.8, 0.01]),
    (16, 12, "points", "matern12", np.linspace(0, 1, 9), np.linspace(0, 1, 7), [0.2, 0.3, 1.0, 0.8, 0.01]),
    (32, 32, "points", "matern32", np.linspace(0, 1, 24), np.linspace(0, 1, 24), [0.2, 0.2, 1.0, 1.0, 0.0025]),
    (40, 33, "points", "matern52", np.linspace(0, 1, 17), np.linspace(0, 1, 12), [0.25, 0.2, 1.3, 0.7, 0.01]),
    (64, 64, "points", "rbf", np.linspace(0, 1, 32), np.linspace(0, 1, 32), [0.2, 0.2, 1.0, 1.0, 0.0025]),
    (256, 256, "points", "rbf", np.linspace(0, 1, 64), np.linspace(0, 1, 64), [0.2, 0.2, 1.0, 1.0, 0.0025]),
    (200, 300, "b0", "matern12", np.linspace(0, 1, 33), np.linspace(0, 1, 21), [0.6931, 0.6931, 0.6931, 0.6931, 0.6932]),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[2]}-{c[3]}-{c[0]}x{c[1]}")
def test_elbo_step_vs_oracle(engine, case):
    st, elbo, grad, info, fs = run_case(engine, *case)
    assert info["status"] == 0
    assert info["jitter"] == (st.d1.jit, st.d2.jit)
    assert abs(elbo - st.elbo) <= RTOL * abs(st.elbo)
    assert rel(grad, st.grad) < RTOL
    mean, var = engine.qv()
    rm, rv = Kr.q_v(st)
    assert rel(mean.cpu().numpy(), rm) < RTOL
    assert rel(var.cpu().numpy(), rv) < RTOL
    xs = np.random.default_rng(5).uniform(0, 1, (1000, 2))
    pm, pv = engine.posterior(torch.tensor(xs, device=DEV))
    om, ov = Kr.posterior(st, fs[0], fs[1], xs)
    assert rel(pm.cpu().numpy(), om) < RTOL
    assert rel(pv.cpu().numpy(), ov) < 1e-6


def test_qv_cov_small(engine):
    st, *_ = run_case(engine, 24, 20, "b0", "matern12", np.linspace(0, 1, 8), np.linspace(0, 1, 10),
                      [0.2, 0.3, 1.0, 0.8, 0.01])
    cov = engine.qv_cov().cpu().numpy()
    assert rel(cov, Kr.q_v_cov(st)) < RTOL


def test_warm_start_tracks_cold(engine):
    """A short hyper-parameter trajectory with warm-started eigensolves equals cold solves."""
    n, m = 128, 32
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    engine.plan("matern32", "points", g, x1, "matern32", "points", g, x2, warm_start=True)
    yy = engine.sumsq(Y)
    theta = np.array([0.2, 0.25, 1.0, 0.9, 0.01])
    f1, f2 = Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2)
    for it in range(4):
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        st = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
        assert abs(elbo - st.elbo) <= RTOL * abs(st.elbo)
        assert rel(grad, st.grad) < RTOL
        theta = theta * (1 + 0.01 * np.array([1, -1, 0.5, -0.5, 1]))


def test_split_partials_equal_step(engine):
    """partials + finish with an externally owned payload == elbo_step (the multi-GPU seam)."""
    case = CASES[2]
    st, elbo, grad, info, _ = run_case(engine, *case)
    X, y, x1, x2 = D.gen_grid(case[0], case[1])
    Y = torch.tensor(y.reshape(case[1], case[0]), device=DEV)
    pay = engine.elbo_partials(Y, case[6])
    e2, g2, _ = engine.elbo_finish(pay, engine.sumsq(Y), case[6])
    assert e2 == elbo and np.array_equal(g2, grad)


def test_block_jacobi_variant_matches(engine):
    """VGGP_FLAG_BLOCK_JACOBI (block eigensolver with MFMA updates) gives the same step as the default."""
    n, m = 96, 48
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    theta = [0.2, 0.25, 1.0, 0.9, 0.01]
    out = []
    for blk in (False, True):
        engine.plan("matern52", "points", g, x1, "matern52", "points", g, x2, warm_start=True, block_jacobi=blk)
        yy = engine.sumsq(Y)
        res = [engine.elbo_step(Y, yy, np.array(theta) * (1 + 0.01 * it)) for it in range(3)]
        out.append(res[-1])
    assert abs(out[0][0] - out[1][0]) <= 1e-10 * abs(out[0][0])
    assert rel(out[1][1], out[0][1]) < 1e-8


MASK_CASES = [
    (12, 10, "b0", "matern12", np.linspace(0, 1, 6), np.linspace(0, 1, 5), [0.2, 0.3, 1.1, 0.8, 0.02], 0.3),
    (32, 32, "b0", "matern12", np.linspace(0, 1, 9), np.linspace(0, 1, 9), [0.25, 0.2, 1.0, 1.2, 0.01], 0.3),
    (40, 36, "points", "matern32", np.linspace(0, 1, 12), np.linspace(0, 1, 13), [0.3, 0.25, 0.9, 1.2, 0.01], 0.3),
    (24, 20, "points", "rbf", np.linspace(0, 1, 5), np.linspace(0, 1, 6), [0.3, 0.25, 0.9, 1.2, 0.01], 0.0),
    # M = 45 * 47 = 2115: 17 panels of the blocked Cholesky, the last one ragged, trailing updates in two column strips
    (96, 90, "b0", "matern12", np.linspace(0, 1, 46), np.linspace(0, 1, 48), [0.3, 0.25, 0.9, 1.2, 0.01], 0.3),
]


@pytest.mark.parametrize("case", MASK_CASES, ids=lambda c: f"{c[2]}-{c[3]}-{c[0]}x{c[1]}-miss{c[7]}")
def test_masked_step_vs_oracle(engine, case):
    """BASELINE config 5 shape (missing observations under a mask): ELBO, gradient and q(v) against the structured
    masked oracle (itself == dense restatement on the observed subset to 1e-14, tests/test_oracle.py)."""
    n1, n2, basis, kind, g1, g2, theta, frac = case
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Wn = (np.random.default_rng(1).uniform(size=(n2, n1)) > frac).astype(np.float64)
    f1, f2 = Kr.Factor(basis, kind, np.asarray(g1, float), x1), Kr.Factor(basis, kind, np.asarray(g2, float), x2)
    st = Kr.elbo_step_masked(y.reshape(n2, n1), Wn, f1, f2, theta)
    engine.plan(kind, basis, g1, x1, kind, basis, g2, x2)
    W = torch.tensor(Wn, device=DEV)
    Ym = torch.tensor(y.reshape(n2, n1), device=DEV) * W
    elbo, grad, info = engine.elbo_step_masked(Ym, W, float(Wn.sum()), engine.sumsq(Ym), theta)
    assert info["status"] == 0
    assert abs(elbo - st.elbo) <= RTOL * abs(st.elbo)
    assert rel(grad, st.grad) < RTOL
    mean, var = engine.qv_masked()
    rm, rv = Kr.q_v_masked(st)
    assert rel(mean.cpu().numpy(), rm) < RTOL and rel(var.cpu().numpy(), rv) < RTOL
    xs = np.random.default_rng(9).uniform(0, 1, (3 * len(g1) * len(g2) + 5, 2))     # > M points: two chunks
    pm, pv = engine.posterior_masked(torch.tensor(xs, device=DEV))
    om, ov = Kr.posterior_masked(st, f1, f2, xs)
    assert rel(pm.cpu().numpy(), om) < RTOL and rel(pv.cpu().numpy(), ov) < 1e-6
    if frac == 0.0:      # full mask: the masked solver must agree with the Kronecker (eigen) path
        e2, g2_, _ = engine.elbo_step(Ym, engine.sumsq(Ym), theta)
        assert abs(e2 - elbo) <= 1e-9 * abs(elbo) and rel(grad, g2_) < 1e-7


def test_masked_blocked_cholesky_multi_panel(engine):
    """M = 18 x 17 = 306 > 128: exercises the blocked (3-panel) dense Cholesky / inverse."""
    n1, n2 = 48, 40
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Wn = (np.random.default_rng(3).uniform(size=(n2, n1)) > 0.3).astype(np.float64)
    g1, g2 = np.linspace(0, 1, 18), np.linspace(0, 1, 17)
    theta = [0.2, 0.25, 1.0, 1.1, 0.01]
    f1, f2 = Kr.Factor("points", "matern32", g1, x1), Kr.Factor("points", "matern32", g2, x2)
    st = Kr.elbo_step_masked(y.reshape(n2, n1), Wn, f1, f2, theta)
    engine.plan("matern32", "points", g1, x1, "matern32", "points", g2, x2)
    W = torch.tensor(Wn, device=DEV)
    Ym = torch.tensor(y.reshape(n2, n1), device=DEV) * W
    elbo, grad, info = engine.elbo_step_masked(Ym, W, float(Wn.sum()), engine.sumsq(Ym), theta)
    assert abs(elbo - st.elbo) <= RTOL * abs(st.elbo) and rel(grad, st.grad) < RTOL


def test_long_warm_trajectory_stays_on_the_oracle(engine):
    """60 consecutive steps along a smooth hyper-parameter path: the extrapolated warm start (basis predicted from the
    last two steps + one Newton-Schulz step) must not drift -- without the re-orthogonalisation the error grows ~2.4x
    per step and reaches NaN within ~40 steps."""
    n1, n2, m = 96, 80, 12
    X, y, x1, x2 = D.gen_grid(n1, n2)
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2)
    engine.plan("matern32", "points", g, x1, "matern32", "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    th0 = np.array([0.2, 0.25, 1.0, 1.1, 0.01])
    for k in range(60):
        th = th0 * (1 + 0.01 * k + 0.003 * np.sin(k))
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        if k % 10 == 9 or k < 4:
            ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
            assert abs(elbo - ref.elbo) <= 1e-9 * abs(ref.elbo), k
            assert rel(grad, ref.grad) < 1e-8, k
    mean, var = engine.qv()
    rm, rv = Kr.q_v(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-8 and rel(var.cpu().numpy(), rv) < 1e-8


def test_config3_full_size_matern32_vs_structured_oracle(engine):
    """BASELINE configs[2]: 1024 x 1024 grid, Matern-3/2 point factors, m_d = 128 -- directly against oracle/kron.py
    (the dense restatement cannot run at this size: three N x N float64 matrices would be 8.8 TB each)."""
    n, m = 1024, 128
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "matern32", g, x1), Kr.Factor("points", "matern32", g, x2)
    theta = [0.2, 0.2, 1.0, 1.0, 0.0025]
    ref = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
    engine.plan("matern32", "points", g, x1, "matern32", "points", g, x2)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    elbo, grad, info = engine.elbo_step(Y, engine.sumsq(Y), theta)
    assert (info["jitter"][0], info["jitter"][1]) == (ref.d1.jit, ref.d2.jit)
    assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo)
    assert rel(grad, ref.grad) < 1e-6
    mean, var = engine.qv()
    rm, rv = Kr.q_v(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6
    # size-independent property: the bound can only tighten when the noise variance is the one that generated the data
    e2, _, _ = engine.elbo_step(Y, engine.sumsq(Y), [0.2, 0.2, 1.0, 1.0, 0.25])
    assert e2 < elbo


def test_config5_shape_masked_b0_vs_structured_oracle(engine):
    """BASELINE configs[4] shape: Matern-1/2 B0 model (the reference's real one) on a masked grid, Bernoulli(0.7) keep with
    default_rng(1); 512 x 512 grid, m_d = 32 (M = 1024: blocked dense Cholesky with 8 panels) against the masked oracle."""
    n, nk = 512, 33
    X, y, x1, x2 = D.gen_grid(n, n)
    Wn = (np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64)
    g = np.linspace(0, 1, nk)
    f1, f2 = Kr.Factor("b0", "matern12", g, x1), Kr.Factor("b0", "matern12", g, x2)
    theta = [0.2, 0.2, 1.0, 1.0, 0.0025]
    ref = Kr.elbo_step_masked(y.reshape(n, n), Wn, f1, f2, theta)
    engine.plan("matern12", "b0", g, x1, "matern12", "b0", g, x2)
    W = torch.tensor(Wn, device=DEV)
    Ym = torch.tensor(y.reshape(n, n), device=DEV) * W
    elbo, grad, info = engine.elbo_step_masked(Ym, W, float(Wn.sum()), engine.sumsq(Ym), theta)
    assert info["status"] == 0
    assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo)
    assert rel(grad, ref.grad) < 1e-6
    mean, var = engine.qv_masked()
    rm, rv = Kr.q_v_masked(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6


def test_config5_full_size_masked_2048_vs_structured_oracle(engine):
    """BASELINE configs[4] at its size: 2048 x 2048 grid, Matern-1/2 B0 model, Bernoulli(0.7) keep with default_rng(1)
    (~30 % missing), m_d = 32 (M = 1024, exact masked mode) against Kr.elbo_step_masked; then two more steps of a
    hyper-parameter trajectory (the fit loop of the config) stay on the oracle."""
    n, nk = 2048, 33
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    Wn = (np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64)
    assert 0.29 < 1.0 - Wn.mean() < 0.31
    g = np.linspace(0, 1, nk)
    f1, f2 = Kr.Factor("b0", "matern12", g, x1), Kr.Factor("b0", "matern12", g, x2)
    engine.plan("matern12", "b0", g, x1, "matern12", "b0", g, x2)
    W = torch.tensor(Wn, device=DEV)
    Ym = torch.tensor(y.reshape(n, n), device=DEV) * W
    yy = engine.sumsq(Ym)
    for k in range(3):
        theta = np.array([0.2, 0.2, 1.0, 1.0, 0.0025]) * (1.0 + 0.02 * k)
        ref = Kr.elbo_step_masked(y.reshape(n, n), Wn, f1, f2, theta)
        elbo, grad, info = engine.elbo_step_masked(Ym, W, float(Wn.sum()), yy, theta)
        assert info["status"] == 0
        assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), k
        assert rel(grad, ref.grad) < 1e-6, k
    mean, var = engine.qv_masked()
    rm, rv = Kr.q_v_masked(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6
    # size-independent property: with everything observed the masked solver equals the Kronecker (eigen) path
    ones = torch.ones_like(W)
    Yf = torch.tensor(y.reshape(n, n), device=DEV)
    e_m, g_m, _ = engine.elbo_step_masked(Yf, ones, float(n * n), engine.sumsq(Yf), theta)
    e_k, g_k, _ = engine.elbo_step(Yf, engine.sumsq(Yf), theta)
    assert abs(e_m - e_k) <= 1e-8 * abs(e_k) and rel(g_m, g_k) < 1e-6


def test_masked_step_at_md128_and_md96(engine):
    """The dense M-space solver above M = 8192 (blocked Cholesky / inverse with the triangular skips, 64-bit offsets): m_d = 128
    (M = 16384, config 5's grid at the headline's inducing count) with everything observed equals the Kronecker path, which is
    pinned on the oracle; and at m_d = 96 (M = 9216) a grid with 30 % holes equals the scattered step on the observed points --
    two independent assemblies (mask-weighted grid products / Khatri-Rao over points) into the same solver.  No CPU oracle at
    these sizes (a dense 16384^3 factorisation in numpy takes minutes)."""
    th = [0.2, 0.3, 1.0, 0.8, 0.01]
    n, m = 256, 128
    X, y, x1, x2 = D.gen_grid(n, n)
    mesh = np.linspace(0, 1, m + 1)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    engine.plan("matern12", "b0", mesh, x1, "matern12", "b0", mesh, x2)
    e_k, g_k, _ = engine.elbo_step(Y, engine.sumsq(Y), th)
    mean_k, var_k = engine.qv()
    e_m, g_m, _ = engine.elbo_step_masked(Y, torch.ones_like(Y), float(n * n), engine.sumsq(Y), th)
    assert abs(e_m - e_k) <= 1e-7 * abs(e_k) and rel(g_m, g_k) < 1e-6
    mean_m, var_m = engine.qv_masked()
    assert rel(mean_m.cpu().numpy(), mean_k.cpu().numpy()) < 1e-6 and rel(var_m.cpu().numpy(), var_k.cpu().numpy()) < 1e-6
    n, m = 192, 96
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    Wn = np.random.default_rng(2).uniform(size=(n, n)) < 0.7
    W = torch.tensor(Wn.astype(np.float64), device=DEV)
    Ym = torch.tensor(y.reshape(n, n), device=DEV) * W
    engine.plan("matern32", "points", g, x1, "matern32", "points", g, x2)
    e_m, g_m, _ = engine.elbo_step_masked(Ym, W, float(Wn.sum()), engine.sumsq(Ym), th)
    obs = Wn.reshape(-1)
    engine.plan("matern32", "points", g, X[obs, 0].copy(), "matern32", "points", g, X[obs, 1].copy(), scattered=True)
    yo = y[obs]
    e_s, g_s, _ = engine.elbo_step_scattered(torch.tensor(yo, device=DEV), float(yo @ yo), th)
    assert abs(e_m - e_s) <= 1e-8 * abs(e_s) and rel(g_m, g_s) < 1e-6


@pytest.mark.parametrize("kind,n", [("matern32", 512), ("rbf", 384), ("matern12", 300)])
def test_md256_elbo_step_vs_structured_oracle(engine, kind, n):
    """m_d = 256 (the top of the m_d sweep of SURVEY.md section 8d and of vggp_plan's range): the factors no longer fit one
    workgroup's LDS, so the blocked Cholesky and the global-memory Jacobi body (vg_jacobi_body<false>, 184 < m <= 256)
    carry the step; cold step + warm-started trajectory against oracle/kron.py."""
    m = 256
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
    engine.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    for k in range(4):
        theta = np.array([0.2, 0.25, 1.0, 0.9, 0.01]) * (1.0 + 0.01 * k)
        ref = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        assert info["status"] == 0 and info["jitter"] == (ref.d1.jit, ref.d2.jit), (k, info)
        assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, elbo, ref.elbo)
        assert rel(grad, ref.grad) < 1e-6, k
    mean, var = engine.qv()
    rm, rv = Kr.q_v(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6


def test_failed_step_resets_the_warm_start(engine):
    """A step that fails (NaN data -> not positive definite is not reachable from data, so poison the grid instead) must
    not leave its bases behind: the following valid steps start cold and are correct."""
    n1, n2, m = 40, 36, 8
    X, y, x1, x2 = D.gen_grid(n1, n2)
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "rbf", g, x1), Kr.Factor("points", "rbf", g, x2)
    theta = [0.3, 0.25, 0.9, 1.2, 0.01]
    ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
    engine.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    for _ in range(3):
        engine.elbo_step(Y, yy, theta)
    Ybad = Y.clone()
    Ybad[3, 4] = float("nan")
    from variational_gridded_gaussian_processes_amd import VggpError
    try:
        engine.elbo_step(Ybad, yy, theta)          # NaN propagates into G?  (G depends on the factors only) -> may succeed
    except VggpError:
        pass
    for _ in range(3):
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        assert abs(elbo - ref.elbo) <= 1e-9 * abs(ref.elbo) and rel(grad, ref.grad) < 1e-7


def _new_basis_case(basis):
    n1, n2 = 40, 34
    X, y, x1, x2 = D.gen_grid(n1, n2)
    if basis == "vff":
        a, b, M = -0.1, 1.1, 6                        # domain strictly larger than the data (as VFF requires)
        om = D.vff_omegas(M, a, b).double().numpy()   # the reference's float32 omegas
        g = np.concatenate([[a, b], om])
        dgrid = (a, b, M)
    else:
        g = np.linspace(0, 1, 9)
        dgrid = torch.tensor(g)
    return n1, n2, X, y, x1, x2, g, dgrid


@pytest.mark.parametrize("basis", ["vff", "b1"])
def test_interdomain_bases_vs_oracles(engine, basis):
    """Matern12VFFGP (kronecker_structure.py:346-515) and Matern12B1SplineASVGP (:524-660) only swap the
    per-dimension factors (Kuu_d scales with 1/s_d, Kuf_d carries no s_d): ELBO, gradient, q(v) and posterior against the
    structured oracle and the literal dense restatement."""
    n1, n2, X, y, x1, x2, g, dgrid = _new_basis_case(basis)
    theta = [0.3, 0.25, 0.9, 1.2, 0.02]
    f1, f2 = Kr.Factor(basis, "matern12", g, x1), Kr.Factor(basis, "matern12", g, x2)
    st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
    dm = D.DenseKron(X, y, basis, "matern12", dgrid, dgrid, raw=D.raw_from_constrained(theta))
    ed, gd = dm.elbo_and_grad()
    engine.plan("matern12", basis, g, x1, "matern12", basis, g, x2)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    elbo, grad, info = engine.elbo_step(Y, engine.sumsq(Y), theta)
    assert info["jitter"] == (st.d1.jit, st.d2.jit)
    assert abs(elbo - st.elbo) <= RTOL * abs(st.elbo) and abs(elbo - ed.item()) <= 1e-6 * abs(ed.item())
    assert rel(grad, st.grad) < RTOL
    assert rel(Kr.grad_raw(grad, dm.raw.detach().numpy()), gd.numpy()) < 1e-6
    mean, var = engine.qv()
    rm, rv = Kr.q_v(st)
    assert rel(mean.cpu().numpy(), rm) < RTOL and rel(var.cpu().numpy(), rv) < RTOL
    qd = dm.q_v()
    assert rel(mean.cpu().numpy().reshape(-1), qd.mean.detach().numpy()) < 1e-6
    cov = engine.qv_cov().cpu().numpy()
    assert rel(cov, Kr.q_v_cov(st)) < 1e-6
    xs = np.random.default_rng(2).uniform(-0.05, 1.05, (64, 2))          # a few points outside the data range
    pm, pv = engine.posterior(torch.tensor(xs, device=DEV))
    om_, ov_ = Kr.posterior(st, f1, f2, xs)
    assert rel(pm.cpu().numpy(), om_) < RTOL and rel(pv.cpu().numpy(), ov_) < 1e-6
    # masked grid through the same factors
    Wn = (np.random.default_rng(5).uniform(size=(n2, n1)) > 0.3).astype(np.float64)
    stm = Kr.elbo_step_masked(y.reshape(n2, n1), Wn, f1, f2, theta)
    W = torch.tensor(Wn, device=DEV)
    Ym = Y * W
    em, gm, im = engine.elbo_step_masked(Ym, W, float(Wn.sum()), engine.sumsq(Ym), theta)
    assert abs(em - stm.elbo) <= RTOL * abs(stm.elbo) and rel(gm, stm.grad) < RTOL
    mm, mv = engine.qv_masked()
    rmm, rmv = Kr.q_v_masked(stm, f1, f2)
    assert rel(mm.cpu().numpy(), rmm) < RTOL and rel(mv.cpu().numpy(), rmv) < RTOL


@pytest.mark.parametrize("basis", ["vff", "points"])
@pytest.mark.parametrize("literal", [True, False], ids=["literal", "conditional"])
def test_gridded_readout_vs_oracle(engine, basis, literal):
    """q_u -> p(v|u) -> q_v for B0 cell features (gridded_kronecker_structure.py:396-438, :613-654),
    Kronecker in the per-dimension cross-covariances, against the structured oracle (== the dense literal formulas to 1e-14,
    checked on the CPU in tests/test_oracle.py)."""
    n1, n2 = 40, 34
    X, y, x1, x2 = D.gen_grid(n1, n2)
    if basis == "vff":
        a, b, M = -0.1, 1.1, 6
        g = np.concatenate([[a, b], D.vff_omegas(M, a, b).double().numpy()])
    else:
        g = np.linspace(0, 1, 11)
    theta = [0.3, 0.25, 0.9, 1.2, 0.02]
    f1, f2 = Kr.Factor(basis, "matern12", g, x1), Kr.Factor(basis, "matern12", g, x2)
    st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
    mesh1, mesh2 = np.linspace(0, 1, 8), np.linspace(0.1, 0.9, 6)
    C1, kd1 = Kr.cross_b0(f1, mesh1, theta[0])
    C2, kd2 = Kr.cross_b0(f2, mesh2, theta[1])
    rm, rv = Kr.readout(st, f1, f2, C1, C2, kd1, kd2, literal=literal)
    engine.plan("matern12", basis, g, x1, "matern12", basis, g, x2)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    engine.elbo_step(Y, engine.sumsq(Y), theta)
    mean, var = engine.readout(torch.tensor(C1), torch.tensor(C2), torch.tensor(kd1), torch.tensor(kd2), literal=literal)
    assert mean.shape == (7, 5)
    assert rel(mean.cpu().numpy(), rm) < RTOL and rel(var.cpu().numpy(), rv) < 1e-6


def test_polish_ends_warm_matern_steps(engine):
    """Eigensolver: on a well-separated spectrum (Matern) a warm-started step ends in the first-order polish instead of a
    second dense sweep (eigh.hip; the a-priori bound keeps the result inside the same tolerance), and never on RBF's
    clustered spectrum.  Values and gradients of the polished steps are checked against the oracle."""
    n1 = n2 = 96
    X, y, x1, x2 = D.gen_grid(n1, n2)
    g = np.linspace(0, 1, 64)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    for kind, expect in (("matern32", True), ("rbf", False)):
        engine.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
        yy = engine.sumsq(Y)
        f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
        hits = 0
        for t in range(8):
            theta = [0.3 * 1.01 ** t, 0.25 * 1.01 ** t, 0.9, 1.2, 0.02]
            elbo, grad, info = engine.elbo_step(Y, yy, theta)
            st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
            assert rel(elbo, st.elbo) < RTOL and rel(grad, st.grad) < 1e-6, (kind, t)
            hits += int(all(info["polished"]))
        assert (hits >= 3) == expect, (kind, hits)
        if kind == "rbf":        # rank-deficient Gram matrices: the subspace start leaves the eigensolver a handful of rounds
            assert sum(info["rounds"]) < 60, info


def test_polish_odd_m_vff_trajectory(engine):
    """The polish on an odd-sized problem (VFF: m = 2M + 1 = 21, padded to 22 inside the eigensolver), warm trajectory."""
    n1, n2, M = 80, 72, 10
    X, y, x1, x2 = D.gen_grid(n1, n2)
    a, b = -0.1, 1.1
    g = np.concatenate([[a, b], D.vff_omegas(M, a, b).double().numpy()])
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    engine.plan("matern12", "vff", g, x1, "matern12", "vff", g, x2, warm_start=True)
    yy = engine.sumsq(Y)
    f1, f2 = Kr.Factor("vff", "matern12", g, x1), Kr.Factor("vff", "matern12", g, x2)
    hits = 0
    for t in range(8):
        theta = [0.3 * 1.01 ** t, 0.25 * 0.99 ** t, 0.9, 1.2, 0.02]
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, theta)
        assert rel(elbo, st.elbo) < RTOL and rel(grad, st.grad) < 1e-6, t
        hits += int(any(info["polished"]))
    mean, var = engine.qv()
    qm, qv_ = Kr.q_v(st)
    assert rel(mean.cpu().numpy(), qm) < 1e-6 and rel(var.cpu().numpy(), qv_) < 1e-6
    assert hits >= 1


def test_readouts_after_warm_rbf_steps_have_cold_accuracy(engine):
    """RBF, 1024 x 1024, m_d = 128, warm-started fit-loop steps (subspace start): the ELBO / gradient need the Gram matrices
    diagonalised to an absolute threshold only, but q(v)'s VARIANCE sees the tiny eigenvalues of the numerically-null block
    through D = 1 + lam1 lam2 / sigma^2 (5e-4 relative when read from the warm basis).  The first read-out after a warm step
    therefore re-runs the finish half cold (vg_accurate_state): q(v) and posterior(x*) match the oracle like a cold step, and
    the next steps keep tracking the oracle from the replaced basis."""
    n, m = 1024, 128
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "rbf", g, x1), Kr.Factor("points", "rbf", g, x2)
    engine.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    th0 = np.array([0.2, 0.2, 1.0, 1.0, 0.0025])
    xs = np.random.default_rng(8).uniform(0, 1, (500, 2))
    for k in range(8):
        th = th0 * (1 + 0.01 * k)
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        if k in (4, 7):
            assert sum(info["rounds"]) < 200, info            # a warm step (a cold one needs > 1000 rotation rounds)
            ref = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
            assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo) and rel(grad, ref.grad) < 1e-6
            mean, var = engine.qv()
            rm, rv = Kr.q_v(ref)
            assert rel(mean.cpu().numpy(), rm) < 1e-6
            assert rel(var.cpu().numpy(), rv) < 1e-6, (k, rel(var.cpu().numpy(), rv))
            pm, pv = engine.posterior(torch.tensor(xs, device=DEV))
            om, ov = Kr.posterior(ref, f1, f2, xs)
            assert rel(pm.cpu().numpy(), om) < 1e-6 and rel(pv.cpu().numpy(), ov) < 1e-5


@pytest.mark.parametrize("n,m,warm_plan", [(512, 128, False), (384, 96, True), (256, 192, False)])
def test_cold_rbf_steps_take_the_range_finder(engine, n, m, warm_plan):
    """A COLD step of an RBF plan (first step, warm start off, a jump) does not run the full Jacobi solve: pivoted-Cholesky range
    finder + thin chain (vg_cold_thin_prepass).  Value, gradient, q(v), posterior and the inducing-point gradient (which rebuilds the
    full state) against the oracle; a spectrum that does not fit the 32 rows (short lengthscale) must fall back to the full solve
    and still be right."""
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", "rbf", g, x1), Kr.Factor("points", "rbf", g, x2)
    engine.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=warm_plan)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    xs = np.random.default_rng(8).uniform(0, 1, (200, 2))
    for k, ell in enumerate((0.2, 0.26, 0.15)):                # 30 % / 40 % apart: each one a cold step in a warm plan as well
        th = np.array([ell, 1.1 * ell, 1.0, 0.9, 0.01])
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        ref = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
        assert sum(info["rounds"]) < 400, info                 # (the full solve of two 128 x 128 matrices: ~1900 rounds)
        assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, info)
        assert rel(grad, ref.grad) < RTOL, (k, info)
        mean, var = engine.qv()
        rm, rv = Kr.q_v(ref)
        assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6
        pm, pv = engine.posterior(torch.tensor(xs, device=DEV))
        om, ov = Kr.posterior(ref, f1, f2, xs)
        assert rel(pm.cpu().numpy(), om) < 1e-6 and rel(pv.cpu().numpy(), ov) < 1e-5
    # a spectrum beyond 32 directions: refused by the miss check, repeated with the full solve
    th = np.array([0.03, 0.035, 1.0, 0.9, 0.01])
    elbo, grad, info = engine.elbo_step(Y, yy, th)
    ref = Kr.elbo_step(y.reshape(n, n), f1, f2, th)
    # (1e-6 / 3e-4: ~60 range directions whose K0-eigenvalues pass through the jitter level -- the regime is ill-conditioned for every
    #  path: the ORACLE's analytic d/d ell and its own central differences agree to 3e-5 here, the full cold solve with the range
    #  finder switched off differs from the oracle by 2e-7 / 9e-5 at 384 x 384, m = 96; this assertion is about the fall-back working)
    assert abs(elbo - ref.elbo) <= 1e-6 * abs(ref.elbo) and rel(grad, ref.grad) < 3e-4, info
    assert sum(info["rounds"]) > 400, info


def _rbf_trajectory(engine, profile, steps=26, m=64):
    """A smooth 1 %-per-step path with one 30 % jump: subspace start, Newton Ritz solve, riders -- and their fall-backs."""
    n = 192
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    engine.plan("rbf", "points", g, x1, "rbf", "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    engine.profile(profile)
    out = []
    th0 = np.array([0.2, 0.22, 1.0, 1.1, 0.01])
    for k in range(steps):
        th = th0 * (1 + 0.01 * k) * (1.3 if k >= 15 else 1.0)
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        out.append((th, elbo, grad.copy(), info))
    engine.profile(False)
    return out, (y.reshape(n, n), Kr.Factor("points", "rbf", g, x1), Kr.Factor("points", "rbf", g, x2))


def test_rbf_warm_chain_with_jump_vs_oracle(engine):
    """The RBF fit-loop chain (subspace start -> row QR -> Newton iteration on the Ritz matrix -> sparse-first main solve, the
    projection launches riding in the chain's single-workgroup launches) against the structured oracle along a path that
    contains a jump the warm start cannot follow smoothly."""
    out, (Yn, f1, f2) = _rbf_trajectory(engine, profile=False)
    for k in (0, 1, 2, 5, 14, 15, 16, 17, 25):
        th, elbo, grad, info = out[k]
        ref = Kr.elbo_step(Yn, f1, f2, th)
        # (1e-8: after the jump the lengthscales are 0.3 -- cond(K + 1e-8 I) ~ 1e10 -- and GPU and numpy Cholesky differ by
        #  their rounding; measured 6e-10 ... 2.3e-9 depending on the elimination order of the 16 x 16 blocks)
        assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, info)
        assert rel(grad, ref.grad) < RTOL, (k, info)
    assert all(o[3]["status"] == 0 for o in out)


@pytest.mark.parametrize("m", [24, 32, 40, 48])
def test_rbf_warm_chain_at_small_inducing_counts(engine, m):
    """m_d <= 48, where the Gram matrices are not rank-deficient by the 3 rank <= m rule but still carry the near-null cluster:
    the relaxed subspace start (Ritz problem of rank + 2 rows, at least 8 rows left to sweep) along the same path with a jump,
    against the oracle; and it must actually be taken (no Jacobi rounds on the smooth stretch)."""
    out, (Yn, f1, f2) = _rbf_trajectory(engine, profile=False, m=m)
    for k in (0, 1, 2, 5, 9, 14, 15, 16, 17, 25):
        th, elbo, grad, info = out[k]
        ref = Kr.elbo_step(Yn, f1, f2, th)
        # (5e-8: the thin chain takes its pass over Y before the whitening, S = L^-1 (A Y) instead of (L^-1 A) Y -- against a
        #  long-double reference that S is good to 2e-11 instead of 2e-13 (tools/studies/early_projection_accuracy.py), and after the
        #  jump, cond(K + 1e-8 I) ~ 1e10, the bound's cancellation y^T y / v - P beta / v^2 turns it into 1.6e-8 at m_d = 24)
        assert abs(elbo - ref.elbo) <= 5e-8 * abs(ref.elbo), (k, info)
        assert rel(grad, ref.grad) < 3e-7, (k, info)          # (measured 1.03e-7 at k = 17, m_d = 24; 1.18e-7 at k = 14, m_d = 48)
    assert all(o[3]["status"] == 0 for o in out)
    if m >= 32:
        assert any(sum(o[3]["rounds"]) == 0 for o in out[6:15]), [o[3]["rounds"] for o in out[6:15]]


@pytest.mark.parametrize("n1,n2,m1,m2", [(2048, 96, 64, 48), (96, 35, 88, 80), (1536, 200, 128, 64)])
def test_thin_chain_on_slab_shapes(engine, n1, n2, m1, m2):
    """Wide / short slabs (the shape of a rank's shard): here the plan's split of S = [B2;V2] Y has FEWER slabs than the early
    projection S' = [A2;dA2] Y leaves behind (its own split is fixed), so the slabs of S must have room of their own -- a step on
    such a shape once wrote S past its buffer into S'.  Warm RBF trajectory against the oracle."""
    X, y, x1, x2 = D.gen_grid(n1, n2)
    del X
    g1, g2 = np.linspace(0, 1, m1), np.linspace(0, 1, m2)
    engine.plan("rbf", "points", g1, x1, "rbf", "points", g2, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    f1, f2 = Kr.Factor("points", "rbf", g1, x1), Kr.Factor("points", "rbf", g2, x2)
    quiet = 0
    for k in range(12):
        th = np.array([0.2, 0.3, 1.0, 0.8, 0.01]) * (1.0 + 0.01 * k)
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        quiet += sum(info["rounds"]) == 0
        if k in (0, 3, 6, 9, 11):
            ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
            # (5e-8 / 3e-7: the accuracy of the early projection on ill-conditioned factors, see
            #  test_rbf_warm_chain_at_small_inducing_counts; measured 1.3e-8 at 2048 x 96, m = 64 / 48)
            assert abs(elbo - ref.elbo) <= 5e-8 * max(abs(ref.elbo), 0.5 * n1 * n2), (k, info)
            assert rel(grad, ref.grad) < 3e-7, (k, info)
    assert quiet >= 6          # the warm chain was taken


def test_riders_and_graph_equal_plain_launches(engine):
    """Profiling mode runs every launch group by itself on one stream (no graph replay, no riders): the same trajectory must
    give the same numbers as the default mode, to rounding of the reduction orders that differ (none should)."""
    a, _ = _rbf_trajectory(engine, profile=False, steps=20)
    b, _ = _rbf_trajectory(engine, profile=True, steps=20)
    for k, (ra, rb) in enumerate(zip(a, b)):
        assert abs(ra[1] - rb[1]) <= 1e-11 * abs(rb[1]), k
        assert rel(ra[2], rb[2]) < 1e-9, k


@pytest.mark.parametrize("kind,n1,n2,m1,m2", [("matern32", 40, 33, 7, 6), ("rbf", 96, 80, 24, 20), ("matern12", 64, 64, 16, 16),
                                             ("matern52", 300, 257, 33, 48)])
def test_zgrad_vs_oracle(engine, kind, n1, n2, m1, m2):
    """vggp_zgrad (inducing-point gradient of SVGP's trainable Z, kronecker_structure.py:303-304) against oracle/kron.py z_grad,
    itself checked against central differences of the ELBO in tests/test_oracle.py; cold and warm steps."""
    rng = np.random.default_rng(3)
    X, y, x1, x2 = D.gen_grid(n1, n2)
    z1, z2 = np.sort(rng.uniform(0.02, 0.98, m1)), np.sort(rng.uniform(0.02, 0.98, m2))
    f1, f2 = Kr.Factor("points", kind, z1, x1), Kr.Factor("points", kind, z2, x2)
    engine.plan(kind, "points", z1, x1, kind, "points", z2, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    th0 = np.array([0.21, 0.27, 1.2, 0.9, 0.02])
    for k in range(4):
        th = th0 * (1 + 0.01 * k)
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        g1, g2 = engine.zgrad(Y)
        if k in (0, 3):
            ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
            r1, r2 = Kr.z_grad(ref, f1, f2, y.reshape(n2, n1))
            # (RBF with irregular inducing points: cond(K + 1e-8 I) ~ 1e10 and the sensitivities are divided by z_i - z_j;
            #  measured 4e-6 between the GPU's and scipy's substitutions)
            tol = 1e-5 if kind == "rbf" else 1e-6
            assert rel(g1.cpu().numpy(), r1) < tol, (k, kind)
            assert rel(g2.cpu().numpy(), r2) < tol, (k, kind)


@pytest.mark.parametrize("kind", ["rbf", "matern32"])
def test_set_inducing_keeps_plan_and_tracks_the_oracle(engine, kind):
    """vggp_set_inducing: the inducing points drift along with the hyper-parameters (what an optimiser training Z does); every
    step -- warm-started, graphs replayed, no re-plan -- against the oracle built for the moved points, the Z-gradient included."""
    n1, n2, m = 128, 96, 32
    rng = np.random.default_rng(7)
    X, y, x1, x2 = D.gen_grid(n1, n2)
    z1, z2 = np.linspace(0.01, 0.99, m), np.linspace(0.02, 0.98, m)
    engine.plan(kind, "points", z1, x1, kind, "points", z2, x2, warm_start=True)
    token = engine.plan_token
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    yy = engine.sumsq(Y)
    th0 = np.array([0.2, 0.22, 1.0, 1.1, 0.01])
    d1, d2 = 1e-3 * rng.standard_normal(m), 1e-3 * rng.standard_normal(m)
    for k in range(12):
        th = th0 * (1 + 0.008 * k)
        if k:
            z1, z2 = z1 + d1, z2 + d2
            engine.set_inducing(0, z1)
            engine.set_inducing(1, z2)
        elbo, grad, info = engine.elbo_step(Y, yy, th)
        if k in (0, 1, 6, 11):
            f1, f2 = Kr.Factor("points", kind, z1, x1), Kr.Factor("points", kind, z2, x2)
            ref = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
            assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, info)
            assert rel(grad, ref.grad) < RTOL, (k, info)
            g1, g2 = engine.zgrad(Y)
            r1, r2 = Kr.z_grad(ref, f1, f2, y.reshape(n2, n1))
            # (32 RBF points at lengthscale 0.2 are numerically rank deficient -- jitter 1e-8, cond 1e10 -- and d ELBO / d z carries
            #  L^-T . L^-1: two float64 implementations agree to ~1e-4 there, measured 2e-4; Matern-3/2: 1e-6)
            tol = 1e-3 if kind == "rbf" else 1e-5
            assert rel(g1.cpu().numpy(), r1) < tol and rel(g2.cpu().numpy(), r2) < tol, k
    assert engine.plan_token == token
    with pytest.raises(Exception):
        engine.set_inducing(0, z1[:-1])


@pytest.mark.parametrize("basis,kind,g1,g2,N", [
    ("b0", "matern12", np.linspace(0, 1, 9), np.linspace(0, 1, 7), 300),
    ("points", "matern32", np.linspace(0, 1, 12), np.linspace(0.05, 0.95, 10), 1500),
    ("points", "rbf", np.linspace(0, 1, 16), np.linspace(0, 1, 16), 5000),
    ("b0", "matern12", np.linspace(0, 1, 33), np.linspace(0, 1, 33), 20000)])
def test_scattered_step_vs_oracle(engine, basis, kind, g1, g2, N):
    """vggp_elbo_step_scattered: N points that form no grid (the reference's _elbo() on along-track data, kronecker_structure.py
    :249-278 with _Kuf(x) :808-823 for arbitrary x) against oracle/kron.py elbo_step_scattered, which equals the literal dense
    restatement's autograd to 1e-14 (tests/test_oracle.py); q(v) and the posterior read-outs through the masked entries."""
    rng = np.random.default_rng(11)
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.1 * rng.standard_normal(N)
    th = np.array([0.3, 0.25, 1.2, 0.8, 0.05])
    f1, f2 = Kr.Factor(basis, kind, g1, np.zeros(1)), Kr.Factor(basis, kind, g2, np.zeros(1))
    ref = Kr.elbo_step_scattered(X, y, f1, f2, th)
    engine.plan(kind, basis, g1, X[:, 0].copy(), kind, basis, g2, X[:, 1].copy(), scattered=True)
    yd = torch.tensor(y, device=DEV)
    elbo, grad, info = engine.elbo_step_scattered(yd, float(y @ y), th)
    assert abs(elbo - ref.elbo) <= RTOL * abs(ref.elbo)
    assert rel(grad, ref.grad) < RTOL
    mean, var = engine.qv_masked()
    rm, rv = Kr.q_v_masked(ref, f1, f2)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-6
    xs = rng.uniform(0, 1, (40, 2))
    pm, pv = engine.posterior_masked(torch.tensor(xs))
    qm, qv = Kr.posterior_masked(ref, f1, f2, xs)
    assert rel(pm.cpu().numpy(), qm) < 1e-6 and rel(pv.cpu().numpy(), qv) < 1e-6
    with pytest.raises(Exception):
        engine.elbo_step(torch.zeros(N, N, dtype=torch.float64, device=DEV)[:2], 0.0, th)      # grid step on a scattered plan


@pytest.mark.parametrize("literal", [True, False], ids=["literal", "conditional"])
def test_readout_masked_equals_full_grid_readout(engine, literal):
    """vggp_readout_masked (gridded read-out from the dense M-space state of a masked / scattered step) on a fully observed grid
    -- as a masked step with W = 1 and as a scattered step over the grid's points -- must reproduce vggp_readout of the
    Kronecker path, which is pinned on the oracle (test_gridded_readout_vs_oracle)."""
    n1, n2, m1, m2, mv = 40, 36, 9, 8, 12
    X, y, x1, x2 = D.gen_grid(n1, n2)
    g1, g2 = np.linspace(0, 1, m1), np.linspace(0, 1, m2)
    th = np.array([0.25, 0.3, 1.1, 0.9, 0.02])
    mesh = np.linspace(0, 1, mv + 1)
    f1, f2 = Kr.Factor("points", "matern12", g1, x1), Kr.Factor("points", "matern12", g2, x2)
    C1, kdv1 = Kr.cross_b0(f1, mesh, th[0])
    C2, kdv2 = Kr.cross_b0(f2, mesh, th[1])
    kd1, kd2 = torch.tensor(kdv1), torch.tensor(kdv2)
    C1t, C2t = torch.tensor(C1), torch.tensor(C2)
    Y = torch.tensor(y.reshape(n2, n1), device=DEV)
    engine.plan("matern12", "points", g1, x1, "matern12", "points", g2, x2)
    engine.elbo_step(Y, engine.sumsq(Y), th)
    mean0, var0 = engine.readout(C1t, C2t, kd1, kd2, literal=literal)
    W = torch.ones_like(Y)
    engine.elbo_step_masked(Y, W, float(n1 * n2), engine.sumsq(Y), th)
    mean1, var1 = engine.readout(C1t, C2t, kd1, kd2, literal=literal, masked=True)
    assert rel(mean1.cpu().numpy(), mean0.cpu().numpy()) < 1e-8 and rel(var1.cpu().numpy(), var0.cpu().numpy()) < 1e-8
    engine.plan("matern12", "points", g1, X[:, 0].copy(), "matern12", "points", g2, X[:, 1].copy(), scattered=True)
    yd = torch.tensor(y, device=DEV)
    engine.elbo_step_scattered(yd, float(y @ y), th)
    mean2, var2 = engine.readout(C1t, C2t, kd1, kd2, literal=literal, masked=True)
    assert rel(mean2.cpu().numpy(), mean0.cpu().numpy()) < 1e-8 and rel(var2.cpu().numpy(), var0.cpu().numpy()) < 1e-8


@pytest.mark.parametrize("kind,tol", [("matern32", 1e-6), ("matern12", 1e-6), ("rbf", 1e-4)])
def test_zgrad_scattered_vs_oracle(engine, kind, tol):
    """vggp_zgrad_scattered (gradient of the scattered ELBO w.r.t. the inducing coordinates: what autograd gives the reference's
    SVGP classes on along-track data) against oracle/kron.py z_grad_scattered, which matches central differences of the
    scattered oracle (CPU suite); then vggp_set_inducing moves Z and both track the oracle again."""
    rng = np.random.default_rng(8)
    N, m1, m2 = 5000, 12, 10
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.05 * rng.normal(size=N)
    z1 = np.linspace(0.03, 0.97, m1) + rng.uniform(-0.02, 0.02, m1)
    z2 = np.linspace(0.03, 0.97, m2) + rng.uniform(-0.02, 0.02, m2)
    th = np.array([0.21, 0.27, 1.2, 0.9, 0.02])
    yd = torch.tensor(y, device=DEV)
    engine.plan(kind, "points", z1, X[:, 0].copy(), kind, "points", z2, X[:, 1].copy(), scattered=True)
    for rep in range(2):
        f1, f2 = Kr.Factor("points", kind, z1, X[:, 0].copy()), Kr.Factor("points", kind, z2, X[:, 1].copy())
        ref = Kr.elbo_step_scattered(X, y, f1, f2, th)
        r1, r2 = Kr.z_grad_scattered(ref, X, y, f1, f2)
        elbo, grad, _ = engine.elbo_step_scattered(yd, float(y @ y), th)
        g1, g2 = engine.zgrad_scattered(yd)
        assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo)
        assert rel(g1.cpu().numpy(), r1) < tol and rel(g2.cpu().numpy(), r2) < tol, (rep, rel(g1.cpu().numpy(), r1), rel(g2.cpu().numpy(), r2))
        z1 = z1 + 1e-3 * np.sign(r1)
        z2 = z2 + 1e-3 * np.sign(r2)
        engine.set_inducing(0, z1)
        engine.set_inducing(1, z2)
    with pytest.raises(Exception):
        engine.zgrad_scattered(yd)                     # set_inducing invalidated the state: a new step first


def test_error_paths_of_the_masked_and_scattered_entries(engine):
    """Error behaviour at the C-ABI: read-outs and gradients refuse to run without the step whose state they read, steps refuse
    a context planned for the other data layout, and a failed call leaves a message behind (vggp_last_error)."""
    from variational_gridded_gaussian_processes_amd import VggpError
    from variational_gridded_gaussian_processes_amd import _lib
    n, m = 24, 6
    X, y, x1, x2 = D.gen_grid(n, n)
    g = np.linspace(0, 1, m)
    th = [0.2, 0.3, 1.0, 0.8, 0.01]
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yd = torch.tensor(y, device=DEV)
    C = torch.tensor(Kr.cross_b0(Kr.Factor("points", "matern12", g, x1), np.linspace(0, 1, 5), th[0])[0])
    kd = torch.ones(4, dtype=torch.float64)

    def code(fn):
        with pytest.raises(VggpError) as ei:
            fn()
        assert str(ei.value)                    # the message of vggp_last_error travels with the exception
        return ei.value.code

    engine.plan("matern12", "points", g, x1, "matern12", "points", g, x2)
    assert code(lambda: engine.readout(C, C, kd, kd, masked=True)) == _lib.VGGP_ESTATE          # no masked step yet
    assert code(lambda: engine.qv_masked()) == _lib.VGGP_ESTATE
    assert code(lambda: engine.zgrad_scattered(yd)) == _lib.VGGP_ESTATE                        # not a scattered context
    assert code(lambda: engine.elbo_step_scattered(yd[:n].contiguous(), 1.0, th)) != 0           # grid plan, scattered step
    engine.elbo_step(Y, engine.sumsq(Y), th)
    assert code(lambda: engine.readout(C, C, kd, kd, masked=True)) == _lib.VGGP_ESTATE          # a full-grid step is no masked state
    engine.plan("matern12", "points", g, X[:, 0].copy(), "matern12", "points", g, X[:, 1].copy(), scattered=True)
    assert code(lambda: engine.zgrad_scattered(yd)) == _lib.VGGP_ESTATE                        # planned, but no step yet
    Yb = torch.zeros(n * n, n * n, dtype=torch.float64, device=DEV)        # (the shape a grid step would want from this plan)
    assert code(lambda: engine.elbo_step_masked(Yb, torch.ones_like(Yb), float(n * n), 1.0, th)) != 0
    assert code(lambda: engine.elbo_step(Yb, 1.0, th)) != 0
    assert code(lambda: engine.elbo_step_scattered(yd, float(y @ y), [0.2, 0.3, -1.0, 0.8, 0.01])) != 0      # theta must be positive
    engine.elbo_step_scattered(yd, float(y @ y), th)
    g1, g2 = engine.zgrad_scattered(yd)
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all()


# ---- iterative masked step: PCG + Lanczos quadrature + control-variate traces, no M x M matrix ----------
def test_iterative_masked_step_vs_dense_small(engine):
    """vggp_elbo_step_masked_iter against the dense masked oracle and against its own numpy specification (different probes:
    same tolerances), B0 / Matern-1/2 and points / Matern-3/2, Bernoulli and track-shaped masks.  Stated tolerances: ELBO 1e-5,
    gradient 1e-4 of its largest component.  Deterministic: the same sequence of calls gives the same bits."""
    from variational_gridded_gaussian_processes_amd import datagen as G
    n = 96
    X, y, x1, x2 = D.gen_grid(n, n)
    theta = [0.2, 0.3, 1.0, 0.8, 0.01]
    masks = [(np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64), G.track_mask(n, n, 2, 0.5)]
    for basis, kind, g in (("b0", "matern12", np.linspace(0, 1, 13)), ("points", "matern32", np.linspace(0, 1, 10))):
        f1, f2 = Kr.Factor(basis, kind, g, x1), Kr.Factor(basis, kind, g, x2)
        engine.plan(kind, basis, g, x1, kind, basis, g, x2)
        for Wn in masks:
            ref = Kr.elbo_step_masked(y.reshape(n, n), Wn, f1, f2, theta)
            W = torch.tensor(Wn, device=DEV)
            Ym = torch.tensor(y.reshape(n, n), device=DEV) * W
            yy = engine.sumsq(Ym)
            elbo, grad, info = engine.elbo_step_masked_iter(Ym, W, float(Wn.sum()), yy, theta, n_probes=16)
            assert info["status"] == 0 and 0 < info["rounds"][0] < 60
            # (the bound is a sum of terms of size N / 2 that may cancel -- it is -35 in one of these cases -- so the relative
            #  tolerance is taken on max(|ELBO|, N / 2); and at M = 100 .. 144 the probe estimate of log|Sigma~| is at its worst
            #  -- few directions to average over, and the track mask is far from the preconditioner's "fraction p observed
            #  everywhere" -- so this small case allows 2e-4; the stated 1e-5 / 1e-4 are asserted at M = 4096 and 16384 below)
            assert abs(elbo - ref.elbo) <= 2e-4 * max(abs(ref.elbo), 0.5 * Wn.sum()), (basis, elbo, ref.elbo)
            assert np.abs(grad - ref.grad).max() <= 2e-4 * np.abs(ref.grad).max(), (basis, grad, ref.grad)
            # repeated calls: the second one reuses the preconditioner's basis with Rayleigh quotients in place of the eigenvalues
            # (another preconditioner: the estimate moves within its own noise), the third repeats the second bit by bit
            e2, g2, _ = engine.elbo_step_masked_iter(Ym, W, float(Wn.sum()), yy, theta, n_probes=16)
            e3, g3, _ = engine.elbo_step_masked_iter(Ym, W, float(Wn.sum()), yy, theta, n_probes=16)
            assert e3 == e2 and np.array_equal(g3, g2)
            assert abs(e2 - elbo) <= 1e-6 * max(abs(elbo), 0.5 * Wn.sum()) and np.abs(g2 - grad).max() <= 1e-6 * np.abs(grad).max()


def test_iterative_masked_step_vs_dense_at_M4096_and_beyond(engine):
    """the iterative step against the dense solver at M = 4096 (2048 x 2048 grid, 30 % missing, m_d = 64) and
    M = 16384 (m_d = 128) -- ELBO 1e-5, gradient 1e-4 -- and a run where the dense solver refuses (M = 36864 > 16384: m_d = 192),
    checked through a size-independent property: with everything observed the iterative step equals the Kronecker (eigen) path,
    for which its preconditioner is exact."""
    n = 2048
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    Wn = (np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(np.float64)
    W = torch.tensor(Wn, device=DEV)
    Yf = torch.tensor(y.reshape(n, n), device=DEV)
    Ym = Yf * W
    nobs = float(Wn.sum())
    theta = [0.2, 0.2, 1.0, 1.0, 0.0025]
    for m in (64, 128):
        mesh = np.linspace(0, 1, m + 1)
        engine.plan("matern12", "b0", mesh, x1, "matern12", "b0", mesh, x2)
        yy = engine.sumsq(Ym)
        e_d, g_d, _ = engine.elbo_step_masked(Ym, W, nobs, yy, theta)
        e_i, g_i, info = engine.elbo_step_masked_iter(Ym, W, nobs, yy, theta, n_probes=16)
        assert abs(e_i - e_d) <= 1e-5 * abs(e_d), (m, e_i, e_d)
        assert np.abs(g_i - g_d).max() <= 1e-4 * np.abs(g_d).max(), (m, g_i, g_d)
    m = 192
    mesh = np.linspace(0, 1, m + 1)
    engine.plan("matern12", "b0", mesh, x1, "matern12", "b0", mesh, x2)
    with pytest.raises(Exception):
        engine.elbo_step_masked(Ym, W, nobs, engine.sumsq(Ym), theta)                 # M = 36864: the dense solver refuses
    e_i, g_i, info = engine.elbo_step_masked_iter(Ym, W, nobs, engine.sumsq(Ym), theta, n_probes=8)
    assert np.isfinite(e_i) and np.all(np.isfinite(g_i)) and info["rounds"][0] < 80
    ones = torch.ones_like(W)
    yyf = engine.sumsq(Yf)
    e_a, g_a, info_a = engine.elbo_step_masked_iter(Yf, ones, float(n * n), yyf, theta, n_probes=8)
    e_k, g_k, _ = engine.elbo_step(Yf, yyf, theta)
    assert info_a["rounds"][0] <= 3                                                    # P = Sigma~: PCG converges at once
    assert abs(e_a - e_k) <= 1e-8 * abs(e_k) and rel(g_a, g_k) < 1e-6


@pytest.mark.parametrize("kind,basis,m", [("rbf", "points", 64), ("matern32", "points", 48), ("matern12", "b0", 64)])
def test_iterative_masked_step_keeps_its_preconditioner_basis(engine, kind, basis, m):
    """Along a fit-loop trajectory the iterative step solves the preconditioner's eigenproblem once and then reuses that basis with
    the Rayleigh quotients of the current Gram matrices (any orthonormal basis gives an SPD preconditioner with a known determinant,
    so nothing but the PCG's iteration count depends on it; RBF factors, whose spectrum spans too many decades for a stale basis,
    keep solving every step): every step against the dense masked step, 1e-5 / 1e-4 as for a single
    step; and the count must not run away while the hyper-parameters drift by 20 %."""
    n = 768
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    Wn = (np.random.default_rng(3).uniform(size=(n, n)) < 0.7).astype(np.float64)
    W = torch.tensor(Wn, device=DEV)
    Ym = torch.tensor(y.reshape(n, n), device=DEV) * W
    nobs = float(Wn.sum())
    g = np.linspace(0, 1, m + 1 if basis == "b0" else m)
    engine.plan(kind, basis, g, x1, kind, basis, g, x2)
    yy = engine.sumsq(Ym)
    its = []
    for k in range(10):
        theta = np.array([0.2, 0.22, 1.0, 0.9, 0.01]) * (1.0 + 0.02 * k)
        e_i, g_i, info = engine.elbo_step_masked_iter(Ym, W, nobs, yy, theta, n_probes=16)
        its.append(info["rounds"][0])
        if k in (0, 1, 4, 9):
            e_d, g_d, _ = engine.elbo_step_masked(Ym, W, nobs, yy, theta)
            assert abs(e_i - e_d) <= 1e-5 * abs(e_d), (k, e_i, e_d)
            assert np.abs(g_i - g_d).max() <= 1e-4 * np.abs(g_d).max(), (k, g_i, g_d)
    assert max(its) <= its[0] + 8, its


def test_b1_hats_on_a_padded_mesh_take_the_newton_chain(engine):
    """B1 hats on a mesh padded beyond the data (the reference's Gridded ASVGP, gridded_kronecker_structure.py:699-724): the hats
    without data span an EXACT null space of the Gram matrix that turns with the lengthscale.  Its block of S G S^T has O(1)
    first-order quotients inside -- which used to reject every warm start (two dense sweeps per step) -- but only the range-null
    rotations matter: the Newton chain leaves pairs of previously-null rows alone (VgRefineJob::lam_prev) and converges
    quadratically.  Trajectory against the oracle; the chain must actually be taken."""
    n, m, pad = 512, 96, 6
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    d = 1.0 / (m - 1 - 2 * pad)
    g = np.linspace(-pad * d, 1 + pad * d, m)
    f1, f2 = Kr.Factor("b1", "matern12", g, x1), Kr.Factor("b1", "matern12", g, x2)
    engine.plan("matern12", "b1", g, x1, "matern12", "b1", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    base = np.array([0.2, 0.25, 1.0, 0.9, 0.01])
    newton_steps = 0
    for k in range(16):
        theta = base * (1.0 + 0.004 * k)
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        assert info["status"] == 0
        if k in (0, 6, 11, 15):
            ref = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
            assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, elbo, ref.elbo)
            assert rel(grad, ref.grad) < 1e-6, (k, rel(grad, ref.grad))
        if k >= 4 and sum(info["rounds"]) == 0:
            newton_steps += 1
    assert newton_steps >= 6, newton_steps
    mean, var = engine.qv()
    rm, rv = Kr.q_v(Kr.elbo_step(y.reshape(n, n), f1, f2, theta))
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-5


# ---- Newton chain: warm full-rank Gram matrices without a single-workgroup sweep ------------------------------------------------
@pytest.mark.parametrize("kind,m,n", [("matern32", 256, 512), ("matern12", 192, 384)])
def test_newton_chain_tracks_the_oracle(engine, kind, m, n):
    """Matern factors at m_d > 128 (beyond the LDS eigensolver: 10 ms of global-memory Jacobi per step at 256): once two bases are
    known the step runs the Newton chain -- GEMM iterations from the extrapolated basis, no rotation rounds -- and stays on the
    oracle across a smooth trajectory and a 20 % jump of the hyper-parameters (where the chain misses and the step is repeated on
    the regular chain, unnoticed by the caller).  (Matern-5/2 at m_d = 128 takes the same chain in the 1024^2 Adam loop of
    tools/time_families.py; on this test's smaller grid its warm steps end in the polish instead.)"""
    X, y, x1, x2 = D.gen_grid(n, n)
    del X
    g = np.linspace(0, 1, m)
    f1, f2 = Kr.Factor("points", kind, g, x1), Kr.Factor("points", kind, g, x2)
    engine.plan(kind, "points", g, x1, kind, "points", g, x2, warm_start=True)
    Y = torch.tensor(y.reshape(n, n), device=DEV)
    yy = engine.sumsq(Y)
    base = np.array([0.2, 0.25, 1.0, 0.9, 0.01])
    newton_steps = 0
    for k in range(14):
        theta = base * (1.0 + 0.005 * k) * (1.2 if k >= 9 else 1.0)
        elbo, grad, info = engine.elbo_step(Y, yy, theta)
        assert info["status"] == 0
        if k in (0, 5, 8, 9, 13):
            ref = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
            assert abs(elbo - ref.elbo) <= 1e-8 * abs(ref.elbo), (k, elbo, ref.elbo)
            assert rel(grad, ref.grad) < 1e-6, (k, rel(grad, ref.grad))
        if k >= 3 and sum(info["rounds"]) == 0 and not any(info["polished"]):
            newton_steps += 1            # no rotation round and no polish: the Newton chain
    assert newton_steps >= 5, newton_steps
    mean, var = engine.qv()
    ref = Kr.elbo_step(y.reshape(n, n), f1, f2, theta)
    rm, rv = Kr.q_v(ref)
    assert rel(mean.cpu().numpy(), rm) < 1e-6 and rel(var.cpu().numpy(), rv) < 1e-5
