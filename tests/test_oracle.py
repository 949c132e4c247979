"""CPU tests (no GPU): the oracle against the committed golden vectors, the reference pins, and itself."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import dense as D
from oracle import kron as Kr

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "oracle_*x*.npz")) if "mask" not in p)
MASK_CASES = sorted(glob.glob(os.path.join(GOLD, "oracle_mask*.npz")))


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def factors(g):
    basis, kind = str(g["basis"]), str(g["kind"])
    q = bool(g["mesh_is_f32"])
    return (Kr.Factor(basis, kind, g["grid1"], g["x1"], q), Kr.Factor(basis, kind, g["grid2"], g["x2"], q))


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_structured_oracle_reproduces_dense_golden(path):
    """oracle/kron.py (the algorithm the GPU runs) == golden vectors from oracle/dense.py (reference-literal)."""
    g = np.load(path)
    f1, f2 = factors(g)
    n1, n2 = len(g["x1"]), len(g["x2"])
    st = Kr.elbo_step(g["y"].reshape(n2, n1), f1, f2, g["theta"])
    assert (st.d1.jit, st.d2.jit) == tuple(g["jitter"])
    assert abs(st.elbo - g["elbo"]) <= 1e-9 * abs(g["elbo"])          # (b1: 1.1e-10, everything else <= 1e-12)
    assert rel(Kr.grad_raw(st.grad, g["raw"]), g["grad_raw"]) < 1e-8
    mean, var = Kr.q_v(st)
    assert rel(mean.reshape(-1), g["qv_mean"]) < 1e-8
    assert rel(var.reshape(-1), g["qv_var"]) < 1e-8
    pm, pv = Kr.posterior(st, f1, f2, g["xs"])
    assert rel(pm, g["post_mean"]) < 1e-8
    assert rel(pv, g["post_var"]) < 1e-7


@pytest.mark.parametrize("path", CASES[:3], ids=lambda p: os.path.basename(p)[7:-4])
def test_dense_oracle_regenerates_golden(path):
    g = np.load(path)
    mesh_dt = torch.float32 if bool(g["mesh_is_f32"]) else torch.float64
    dm = D.DenseKron(g["X"], g["y"], str(g["basis"]), str(g["kind"]), torch.tensor(g["grid1"]).to(mesh_dt),
                     torch.tensor(g["grid2"]).to(mesh_dt), raw=g["raw"])
    e, gr = dm.elbo_and_grad()
    assert abs(e.item() - g["elbo"]) <= 1e-12 * abs(g["elbo"])
    assert rel(gr.numpy(), g["grad_raw"]) < 1e-10


@pytest.mark.parametrize("path", MASK_CASES, ids=lambda p: os.path.basename(p)[7:-4])
def test_masked_structured_oracle_reproduces_dense_golden(path):
    """Masked mode: M-space assembly (oracle/kron.py elbo_step_masked) == dense restatement on the observed subset."""
    g = np.load(path)
    f1, f2 = factors(g)
    n1, n2 = len(g["x1"]), len(g["x2"])
    st = Kr.elbo_step_masked(g["y"].reshape(n2, n1), g["W"].astype(np.float64), f1, f2, g["theta"])
    assert abs(st.elbo - g["elbo"]) <= 1e-10 * abs(g["elbo"])
    assert rel(Kr.grad_raw(st.grad, g["raw"]), g["grad_raw"]) < 1e-8
    mean, var = Kr.q_v_masked(st)
    assert rel(mean.reshape(-1), g["qv_mean"]) < 1e-8
    assert rel(var.reshape(-1), g["qv_var"]) < 1e-8


def test_masked_oracle_with_full_mask_equals_grid_oracle():
    X, y, x1, x2 = D.gen_grid(14, 11)
    f1, f2 = Kr.Factor("points", "matern32", np.linspace(0, 1, 6), x1), Kr.Factor("points", "matern32", np.linspace(0, 1, 5), x2)
    th = [0.3, 0.2, 1.1, 0.9, 0.02]
    a = Kr.elbo_step(y.reshape(11, 14), f1, f2, th)
    b = Kr.elbo_step_masked(y.reshape(11, 14), np.ones((11, 14)), f1, f2, th)
    assert abs(a.elbo - b.elbo) <= 1e-11 * abs(a.elbo) and rel(b.grad, a.grad) < 1e-9


def test_gradient_matches_finite_differences():
    g = np.load(os.path.join(GOLD, "oracle_pts_m32_32x32.npz"))
    f1, f2 = factors(g)
    Y = g["y"].reshape(32, 32)
    th = g["theta"].copy()
    st = Kr.elbo_step(Y, f1, f2, th)
    for i in range(5):
        h = 1e-6 * th[i]
        tp, tm = th.copy(), th.copy()
        tp[i] += h
        tm[i] -= h
        fd = (Kr.elbo_step(Y, f1, f2, tp).elbo - Kr.elbo_step(Y, f1, f2, tm).elbo) / (2 * h)
        assert abs(fd - st.grad[i]) <= 1e-6 * max(1.0, abs(st.grad[i]))


def test_row_shard_emulation_equals_single_rank():
    """The multi-GPU decomposition (dim-2 row shards + one summed payload) is exact."""
    g = np.load(os.path.join(GOLD, "oracle_b0_m12_32x32.npz"))
    f1, f2 = factors(g)
    Y = g["y"].reshape(32, 32)
    s1 = Kr.elbo_step(Y, f1, f2, g["theta"], n_ranks=1)
    for r in (2, 3, 4):
        sr = Kr.elbo_step(Y, f1, f2, g["theta"], n_ranks=r)
        assert abs(sr.elbo - s1.elbo) <= 1e-12 * abs(s1.elbo)
        assert rel(sr.grad, s1.grad) < 1e-10


def test_elbo_is_a_lower_bound_and_tightens():
    """Check (b) of SURVEY.md section 4: ELBO <= exact log marginal likelihood, -> equality as M grows."""
    n = 12
    X, y, x1, x2 = D.gen_grid(n, n)
    th = np.array([0.3, 0.3, 1.0, 1.0, 0.05])
    K = (D.pairwise("matern12", torch.tensor(X[:, 0]), torch.tensor(X[:, 0]), th[0], th[2])
         * D.pairwise("matern12", torch.tensor(X[:, 1]), torch.tensor(X[:, 1]), th[1], th[3]))
    lml = D.mvn_log_prob(K + th[4] * torch.eye(n * n, dtype=torch.float64), torch.tensor(y)).item()
    prev = -np.inf
    for m in (3, 6, 12):
        z = np.linspace(0, 1, m)
        e = Kr.elbo_step(y.reshape(n, n), Kr.Factor("points", "matern12", z, x1),
                         Kr.Factor("points", "matern12", z, x2), th).elbo
        assert prev <= e <= lml + 1e-9
        prev = e
    assert abs(prev - lml) < 1e-6 * abs(lml)      # inducing points == data grid -> bound is tight


def test_one_d_oracle_and_trivial_second_factor():
    """Config 1: the 1-D dense oracle == the 2-D structured algorithm with a trivial second factor."""
    g = np.load(os.path.join(GOLD, "oracle_1d_b0_256.npz"))
    th = g["theta"]
    f1 = Kr.Factor("b0", "matern12", g["mesh"], g["x"])
    f2 = Kr.Factor("one", "matern12", np.ones(1), np.zeros(1))
    st = Kr.elbo_step(g["y"].reshape(1, -1), f1, f2, [th[0], 1.0, th[1], 1.0, th[2]])
    assert abs(st.elbo - g["elbo"]) <= 1e-10 * abs(g["elbo"])
    graw = Kr.grad_raw(st.grad[[0, 2, 4]], g["raw"])
    assert rel(graw, g["grad_raw"]) < 1e-8
    mean, var = Kr.q_v(st)
    assert rel(mean.reshape(-1), g["qv_mean"]) < 1e-8 and rel(var.reshape(-1), g["qv_var"]) < 1e-8


def test_reference_pins():
    """Outputs of the reference modules that import here (generated by tests/golden/make_golden.py)."""
    p = np.load(os.path.join(GOLD, "ref_pins.npz"))
    # point ordering of gen_2d (src/utils/datagenerators.py:70-72): x1 fastest
    X, y, x1, x2 = D.gen_grid(5, 5, lims1=(0.0, 1.0), lims2=(-1.0, 2.0), noise=0.0)
    assert np.array_equal(X, p["gen2d_X"]) and np.allclose(y, p["gen2d_y"], rtol=0, atol=1e-15)
    # B0SplineBasis bookkeeping (src/basis/bspline.py:81-103): m = nknots - 1, float32 delta
    mesh = torch.linspace(0, 1, 11)
    assert np.array_equal(mesh.numpy(), p["b0_mesh"]) and int(p["b0_m"]) == 10 and int(p["b0_nbasis"]) == 10
    assert float(mesh[1] - mesh[0]) == float(p["b0_delta"])
    # quad known-answer (src/utils/integrators.py:10-30): exact cell integrals of sin + cos
    m = p["quad_mesh"]
    exact = (-np.cos(m[1:]) + np.sin(m[1:])) - (-np.cos(m[:-1]) + np.sin(m[:-1]))
    assert np.abs(exact - p["quad_areas"]).max() < 1e-12


def _vff_tol(om, x, a):
    """The reference evaluates the Fourier features in float32 (float32 omegas against 0-dim float64 points,
    fourier.py:33-41), so its cos/sin arguments carry a float32 rounding of relative size 6e-8."""
    return 2e-7 * max(1.0, float(np.abs(om).max() * np.abs(x - a).max()))


def test_reference_basis_pins():
    """The oracle's VFF / B1 Kuf builders against outputs of the reference's own importable basis classes
    (FourierBasisMatern12(M, a, b, ell)(x), fourier.py:58-88; B1SplineBasis(mesh)(x), bspline.py:106-112) committed in
    tests/golden/ref_pins_basis.npz by make_golden.py."""
    p = np.load(os.path.join(GOLD, "ref_pins_basis.npz"))
    for tag in ("vff_a", "vff_b"):
        M, a, b, ell = int(p[tag + "_M"]), float(p[tag + "_a"]), float(p[tag + "_b"]), float(p[tag + "_ell"])
        x, om32 = p[tag + "_x"], p[tag + "_omegas"]
        assert om32.dtype == np.float32 and np.array_equal(om32, D.vff_omegas(M, a, b).numpy())
        ref = p[tag + "_Phi"]
        assert ref.shape == (2 * M + 1, len(x))
        tol = _vff_tol(om32, x, a)
        got = D.vff_Kuf_along_dim(a, b, torch.tensor(om32), torch.tensor(ell, dtype=torch.float64), torch.tensor(x)).numpy()
        assert np.abs(got - ref).max() < tol
        A, _ = Kr.vff_A(a, b, om32.astype(np.float64), x, ell)
        assert np.abs(A - ref).max() < tol
        outside = (x < a) | (x >= b)          # the exp(-r/ell) tails carry no float32 argument blow-up
        assert outside.sum() >= 4 and np.abs(A[:, outside] - ref[:, outside]).max() < 2e-7
    for tag, tol in (("b1_f64", 1e-15), ("b1_f32", 1e-6)):
        mesh, x, ref = p[tag + "_mesh"], p[tag + "_x"], p[tag + "_Phi"]
        got = D.b1_Kuf_along_dim(torch.tensor(mesh), torch.tensor(x)).numpy()
        assert np.array_equal(got, ref)       # same formula on the same (possibly float32-born) knots: exact
        A, _ = Kr.b1_A(mesh.astype(np.float64), x)
        assert np.abs(A - ref).max() < tol    # kron.b1_A assumes a uniform spacing; a float32 linspace is uniform to 1e-7


def test_b0_closed_forms_are_cell_integrals():
    """_Kuu_along_dim / _Kuf_along_dim (kronecker_structure.py:723-790) == numerical cell integrals of exp(-|.|/l)."""
    import scipy.integrate as si
    mesh, ell = np.linspace(0, 1, 6), 0.37
    K, _ = Kr.b0_K(5, mesh[1] - mesh[0], ell)
    A, _ = Kr.b0_A(mesh, np.array([0.0, 0.13, 0.4, 0.77, 1.0]), ell)
    for i, j in ((0, 0), (0, 1), (1, 4), (3, 2)):
        val = si.dblquad(lambda t, u: np.exp(-abs(t - u) / ell), mesh[i], mesh[i + 1], mesh[j], mesh[j + 1])[0]
        assert abs(val - K[i, j]) < 1e-8
    for k, (xp, col) in ((1, (0.13, 1)), (3, (0.77, 3)), (0, (1.0, 4))):
        val = si.quad(lambda t: np.exp(-abs(t - xp) / ell), mesh[k], mesh[k + 1], points=[xp] if mesh[k] < xp < mesh[k + 1] else None)[0]
        assert abs(val - A[k, col]) < 1e-10
    # derivative formulas
    h = 1e-6
    Kp, _ = Kr.b0_K(5, mesh[1] - mesh[0], ell + h)
    Km, _ = Kr.b0_K(5, mesh[1] - mesh[0], ell - h)
    _, dK = Kr.b0_K(5, mesh[1] - mesh[0], ell)
    assert np.abs((Kp - Km) / (2 * h) - dK).max() < 1e-8


@pytest.mark.parametrize("basis", ["vff", "points"])
@pytest.mark.parametrize("literal", [True, False])
def test_gridded_readout_structured_equals_dense(basis, literal):
    """q_u -> p(v|u) -> q_v (gridded_kronecker_structure.py:396-438, :613-654): Kronecker read-out == dense literal formulas."""
    n1, n2 = 18, 14
    X, y, x1, x2 = D.gen_grid(n1, n2)
    th = [0.3, 0.25, 0.9, 1.2, 0.02]
    if basis == "vff":
        a, b, M = -0.1, 1.1, 4
        g, dg = np.concatenate([[a, b], D.vff_omegas(M, a, b).double().numpy()]), (a, b, M)
    else:
        g = np.linspace(0, 1, 7)
        dg = torch.tensor(g)
    mesh = np.linspace(0, 1, 6)
    dm = D.DenseKron(X, y, basis, "matern12", dg, dg, raw=D.raw_from_constrained(th))
    f1, f2 = Kr.Factor(basis, "matern12", g, x1), Kr.Factor(basis, "matern12", g, x2)
    st = Kr.elbo_step(y.reshape(n2, n1), f1, f2, th)
    C1, kd1 = Kr.cross_b0(f1, mesh, th[0])
    C2, kd2 = Kr.cross_b0(f2, mesh, th[1])
    q = dm.q_v_gridded(torch.tensor(mesh), torch.tensor(mesh), literal=literal)
    m_, v_ = Kr.readout(st, f1, f2, C1, C2, kd1, kd2, literal=literal)
    assert rel(m_.reshape(-1), q.mean.detach().numpy()) < 1e-10
    assert rel(v_.reshape(-1), q.variance.detach().numpy()) < 1e-9



@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern52", "matern12"])
def test_z_grad_matches_central_differences(kind):
    """oracle/kron.py z_grad (the spec of vggp_zgrad: gradient w.r.t. SVGP's trainable inducing coordinates,
    kronecker_structure.py:303-304) against central differences of the ELBO."""
    rng = np.random.default_rng(0)
    n1, n2, m1, m2 = 40, 33, 7, 6
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Y = y.reshape(n2, n1)
    z1, z2 = np.sort(rng.uniform(0.03, 0.97, m1)), np.sort(rng.uniform(0.03, 0.97, m2))
    th = np.array([0.21, 0.27, 1.2, 0.9, 0.02])
    f1, f2 = Kr.Factor("points", kind, z1, x1), Kr.Factor("points", kind, z2, x2)
    g1, g2 = Kr.z_grad(Kr.elbo_step(Y, f1, f2, th), f1, f2, Y)
    h = 1e-6
    for dim, (z, g) in enumerate(((z1, g1), (z2, g2))):
        fd = np.zeros(len(z))
        for i in range(len(z)):
            zp, zm = z.copy(), z.copy()
            zp[i] += h
            zm[i] -= h
            fp = (Kr.Factor("points", kind, zp, x1), f2) if dim == 0 else (f1, Kr.Factor("points", kind, zp, x2))
            fm = (Kr.Factor("points", kind, zm, x1), f2) if dim == 0 else (f1, Kr.Factor("points", kind, zm, x2))
            fd[i] = (Kr.elbo_step(Y, *fp, th).elbo - Kr.elbo_step(Y, *fm, th).elbo) / (2 * h)
        assert np.abs(g - fd).max() <= 2e-6 * np.abs(fd).max()



@pytest.mark.parametrize("kind", ["rbf", "matern32", "matern12"])
def test_z_grad_scattered_matches_central_differences(kind):
    """oracle/kron.py z_grad_scattered (the spec of vggp_zgrad_scattered) against central differences of the scattered ELBO."""
    rng = np.random.default_rng(4)
    N, m1, m2 = 300, 6, 5
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(4 * X[:, 1]) + 0.05 * rng.normal(size=N)
    z1 = np.linspace(0.05, 0.95, m1) + rng.uniform(-0.03, 0.03, m1)      # (irregular, but no near-coincident pair: central
    z2 = np.linspace(0.05, 0.95, m2) + rng.uniform(-0.03, 0.03, m2)      #  differences through an RBF Kuu of cond 1e10 are noise)
    th = np.array([0.21, 0.27, 1.2, 0.9, 0.02])
    mk = lambda za, zb: (Kr.Factor("points", kind, za, X[:, 0].copy()), Kr.Factor("points", kind, zb, X[:, 1].copy()))
    f1, f2 = mk(z1, z2)
    g1, g2 = Kr.z_grad_scattered(Kr.elbo_step_scattered(X, y, f1, f2, th), X, y, f1, f2)
    h = 1e-6
    for dim, (z, g) in enumerate(((z1, g1), (z2, g2))):
        fd = np.zeros(len(z))
        for i in range(len(z)):
            zp, zm = z.copy(), z.copy()
            zp[i] += h
            zm[i] -= h
            fp = mk(zp, z2) if dim == 0 else mk(z1, zp)
            fm = mk(zm, z2) if dim == 0 else mk(z1, zm)
            fd[i] = (Kr.elbo_step_scattered(X, y, *fp, th).elbo - Kr.elbo_step_scattered(X, y, *fm, th).elbo) / (2 * h)
        assert np.abs(g - fd).max() <= 5e-6 * np.abs(fd).max()


@pytest.mark.parametrize("basis,kind,g1,g2", [("b0", "matern12", np.linspace(0, 1, 7), np.linspace(0, 1, 6)),
                                              ("points", "matern32", np.linspace(0, 1, 6), np.linspace(0.05, 0.95, 5)),
                                              ("points", "rbf", np.linspace(0, 1, 5), np.linspace(0, 1, 5))])
def test_scattered_oracle_equals_dense_restatement(basis, kind, g1, g2):
    """oracle/kron.py elbo_step_scattered (Khatri-Rao assembly in M-space, analytic gradient) against the literal dense
    restatement with autograd on random scattered points: the reference's _elbo() takes any X (kronecker_structure.py:808-823)."""
    rng = np.random.default_rng(0)
    N = 57
    X = rng.uniform(0, 1, (N, 2))
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.1 * rng.standard_normal(N)
    th = np.array([0.3, 0.25, 1.2, 0.8, 0.05])
    dm = D.DenseKron(X, y, basis, kind, torch.tensor(g1), torch.tensor(g2), raw=D.raw_from_constrained(th))
    ed, gd_raw = dm.elbo_and_grad()
    f1, f2 = Kr.Factor(basis, kind, g1, np.zeros(1)), Kr.Factor(basis, kind, g2, np.zeros(1))
    st = Kr.elbo_step_scattered(X, y, f1, f2, th)
    graw = Kr.grad_raw(st.grad, D.raw_from_constrained(th).numpy())
    assert abs(st.elbo - ed.item()) <= 1e-11 * abs(ed.item())
    assert np.abs(graw - gd_raw.numpy()).max() <= 1e-10 * np.abs(gd_raw.numpy()).max()


# ---- a15: the six third-party primitives the oracle RESTATES, checked against independent torch / scipy implementations --------
# (gpytorch / linear_operator are not importable here, so parity at that boundary stays "unpinned"; but gpytorch's classes wrap
# torch.distributions.MultivariateNormal, torch.nn.functional.softplus and dense Cholesky solves, all of which ARE importable.)
def test_restated_primitives_against_torch_and_scipy():
    import scipy.linalg
    import scipy.special
    rng = np.random.default_rng(7)
    n = 40
    A = rng.standard_normal((n, n))
    cov = torch.tensor(A @ A.T / n + 0.3 * np.eye(n))
    y = torch.tensor(rng.standard_normal(n))
    # MultivariateNormal(0, C).log_prob(y)  (kronecker_structure.py:273)
    ref = torch.distributions.MultivariateNormal(torch.zeros(n, dtype=torch.float64), covariance_matrix=cov).log_prob(y)
    assert abs(float(D.mvn_log_prob(cov, y)) - float(ref)) <= 1e-12 * abs(float(ref))
    # lazify(A).inv_matmul(B) = A^-1 B  (:269)
    B = torch.tensor(rng.standard_normal((n, 7)))
    assert torch.allclose(D.inv_matmul(cov, B), torch.linalg.solve(cov, B), rtol=1e-11, atol=1e-13)
    assert torch.allclose(D.inv_matmul(cov, y), torch.linalg.solve(cov, y), rtol=1e-11, atol=1e-13)
    # softplus / inverse, and gpytorch's constraints: Positive() = softplus, GreaterThan(1e-4) = softplus + 1e-4  (:263)
    x = torch.tensor(rng.uniform(-30, 30, 50))
    assert torch.allclose(D.softplus(x), torch.nn.functional.softplus(x), rtol=1e-14, atol=0)
    assert torch.allclose(D.inv_softplus(D.softplus(x[x > -20])), x[x > -20], rtol=1e-9, atol=1e-9)
    raw = torch.tensor(rng.uniform(-3, 3, 5))
    th = D.constrained_from_raw(raw)
    assert torch.allclose(th[:4], torch.nn.functional.softplus(raw[:4])) and abs(float(th[4] - torch.nn.functional.softplus(raw[4])) - 1e-4) < 1e-15
    assert torch.allclose(D.raw_from_constrained(th.tolist()), raw, rtol=1e-9, atol=1e-9)
    # ToeplitzLinearOperator(first_row).to_dense()[i, j] = r[|i - j|]  (:737) -- the B0 Kuu builder against scipy's toeplitz
    m, delta, ell, s = 9, 0.125, 0.37, 1.3
    K = D.b0_Kuu_along_dim(m, torch.tensor(delta), torch.tensor(ell), torch.tensor(s)).numpy()
    assert np.allclose(K, scipy.linalg.toeplitz(K[0]), rtol=0, atol=0) and np.allclose(K, K.T)
    # Matern / RBF kernels against their Bessel-function definition  k_nu(r) = 2^(1-nu)/Gamma(nu) (sqrt(2 nu) r)^nu K_nu(sqrt(2 nu) r)
    r = np.linspace(1e-3, 6.0, 60)
    for kind, nu in (("matern12", 0.5), ("matern32", 1.5), ("matern52", 2.5)):
        z = np.sqrt(2 * nu) * r
        bessel = 2.0 ** (1 - nu) / scipy.special.gamma(nu) * z ** nu * scipy.special.kv(nu, z)
        assert np.allclose(D.kappa(kind, torch.tensor(r)).numpy(), bessel, rtol=1e-10, atol=1e-14), kind
    assert np.allclose(D.kappa("rbf", torch.tensor(r)).numpy(), np.exp(-0.5 * r * r), rtol=1e-14)
    # psd_safe_cholesky: no jitter on a PD matrix, the 1e-8 .. 1e-6 ladder on a singular one
    assert D.psd_safe_cholesky(cov)[1] == 0.0
    sing = torch.ones(6, 6, dtype=torch.float64)
    assert D.psd_safe_cholesky(sing)[1] in (1e-8, 1e-7, 1e-6)


def test_iterative_masked_step_matches_the_dense_masked_step():
    """oracle/kron.py elbo_step_masked_iter (PCG + stochastic Lanczos quadrature + control-variate trace estimators, fixed
    probes) against the dense masked oracle: ELBO 1e-5, gradient 1e-4 of its largest component -- the tolerances stated for
    vggp_elbo_step_masked_iter -- on a Bernoulli mask and on a track-shaped one, and reproducible bit for bit."""
    from variational_gridded_gaussian_processes_amd import datagen as G
    n, m = 96, 12
    X, y, x1, x2 = D.gen_grid(n, n)
    mesh = np.linspace(0, 1, m + 1)
    f1, f2 = Kr.Factor("b0", "matern12", mesh, x1), Kr.Factor("b0", "matern12", mesh, x2)
    theta = [0.2, 0.3, 1.0, 0.8, 0.01]
    for Wn in ((np.random.default_rng(1).uniform(size=(n, n)) < 0.7).astype(float), G.track_mask(n, n, 2, 0.5)):
        ref = Kr.elbo_step_masked(y.reshape(n, n), Wn, f1, f2, theta)
        it = Kr.elbo_step_masked_iter(y.reshape(n, n), Wn, f1, f2, theta, nprobe=16, seed=0)
        assert abs(it.elbo - ref.elbo) <= 1e-5 * abs(ref.elbo), (it.elbo, ref.elbo)
        assert np.abs(it.grad - ref.grad).max() <= 1e-4 * np.abs(ref.grad).max(), (it.grad, ref.grad)
        assert np.abs(it.A0 - ref.A0).max() <= 1e-8 * np.abs(ref.A0).max()
        assert it.iters < 60
    again = Kr.elbo_step_masked_iter(y.reshape(n, n), Wn, f1, f2, theta, nprobe=16, seed=0)
    assert again.elbo == it.elbo and np.array_equal(again.grad, it.grad)
