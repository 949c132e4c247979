"""GPU tests of the drop-in model API (the notebooks' call sequence) against the dense oracle."""
import numpy as np
import pytest
import torch

from oracle import dense as D
from oracle import kron as Kr

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_smoke_entry():
    import __graft_entry__ as g
    g.smoke()


def test_gridded_gp_notebook_loop_matches_dense_oracle(engine):
    """5_gridded_kronecker_structure_models.ipynb cells 24-29 on a 25x25 grid, 11 knots, 5 Adam steps:
    loss trajectory and q(v) equal the dense float64 restatement driven by the same optimiser."""
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    n, nknots = 25, 11
    X, y, x1, x2 = D.gen_grid(n, n)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    model = Matern12GriddedGP(Xt, yt, nknots, (0, 1), (0, 1), engine=engine).to(torch.float64)
    dm = D.DenseKron(X, y, "b0", "matern12", torch.linspace(0, 1, nknots), torch.linspace(0, 1, nknots))
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    opt_d = torch.optim.Adam([dm.raw], lr=0.01)
    for it in range(5):
        opt.zero_grad()
        loss = -model._elbo()
        loss.backward()
        opt.step()
        opt_d.zero_grad()
        loss_d = -dm._elbo()
        loss_d.backward()
        opt_d.step()
        assert abs(loss.item() - loss_d.item()) <= 1e-8 * abs(loss_d.item()), it
    # accessors the notebooks read every iteration (61_envisat... cell 54)
    th = dm.theta().detach().numpy()
    assert rel(model.kernel_1.base_kernel.lengthscale.item(), th[0]) < 1e-5
    assert rel(model.kernel_2.outputscale.item(), th[3]) < 1e-5
    assert rel(model.likelihood.noise.item(), th[4]) < 1e-5
    qv, qd = model.q_v(), dm.q_v()
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-7
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-7
    assert rel(qv.covariance_matrix.numpy(), qd.covariance_matrix.detach().numpy()) < 1e-5
    grid_mean = qv.mean.reshape(nknots - 1, nknots - 1).T        # how cell 29 consumes it
    assert grid_mean.shape == (10, 10)
    xs = torch.tensor(np.random.default_rng(2).uniform(0, 1, (50, 2)))
    po, pd = model.posterior(xs), dm.posterior(xs)
    assert rel(po.mean.numpy(), pd.mean.detach().numpy()) < 1e-5
    assert rel(po.variance.numpy(), pd.variance.detach().numpy()) < 1e-5
    lo, hi = po.confidence_region()
    assert bool((hi > lo).all())
    pp = model.posterior_predictive(xs)
    assert rel(pp.variance.numpy(), (pd.variance + dm.theta()[4]).detach().numpy()) < 1e-5


@pytest.mark.parametrize("cls,kind", [("Matern12SVGP", "matern12"), ("Matern32SVGP", "matern32"),
                                      ("Matern52SVGP", "matern52"), ("RBFSVGP", "rbf")])
def test_svgp_family_vs_dense(engine, cls, kind):
    import variational_gridded_gaussian_processes_amd.models as M
    n1, n2, m = 20, 16, 7
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Z = torch.tensor(np.stack([np.linspace(0, 1, m), np.linspace(0.05, 0.95, m)], axis=1))
    model = getattr(M, cls)(torch.tensor(X), torch.tensor(y), Z, engine=engine).to(torch.float64)
    dm = D.DenseKron(X, y, "points", kind, Z[:, 0], Z[:, 1])
    e = model._elbo()
    e.backward()
    ed, gd = dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    got = np.array([model.kernel_1.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_2.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_1.raw_outputscale.grad.item(), model.kernel_2.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert rel(got, gd.numpy()) < 1e-5


def test_univariate_b0_config1(engine):
    """BASELINE config 1: 1-D, 256 gridded inputs (notebook-1 shape), vs the 1-D dense oracle; plus the
    reference's only known-answer check: q(v) means ~ per-cell integrals of the latent function
    (4_gridded_univariate_structure_models.ipynb cells 26-29, src/utils/integrators.py:10-30)."""
    import scipy.integrate as integrate
    from variational_gridded_gaussian_processes_amd.models import univariate
    n, nknots = 256, 33
    x = np.linspace(0, 2 * np.pi, n)
    f = lambda t: np.sin(t) + np.cos(t)
    y = f(x) + 0.05 * np.random.default_rng(0).standard_normal(n)
    model = univariate.Matern12B0SplineGriddedGP(torch.tensor(x), torch.tensor(y), nknots, (0, 2 * np.pi),
                                                 engine=engine).to(torch.float64)
    mesh = torch.linspace(0, 2 * np.pi, nknots)
    dm = D.Dense1D(x, y, "b0", "matern12", mesh)
    e = model._elbo()
    e.backward()
    ed, gd = dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    got = np.array([model.kernel.base_kernel.raw_lengthscale.grad.item(), model.kernel.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert rel(got, gd.numpy()) < 1e-5
    qv, qd = model.q_v(), dm.q_v()
    # 1e-5, not 1e-7: in the 1-D reference `lengthscale.squeeze()` is 0-dim, so the Toeplitz first row is
    # evaluated in float32 (univariate_structure.py:813-820 under torch promotion); the engine keeps float64
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    # train a little, then compare cell means with quad integrals of the latent function
    model.likelihood.noise = 0.05 ** 2
    opt = torch.optim.Adam(model.parameters(), lr=0.05)
    for _ in range(150):
        opt.zero_grad()
        (-model._elbo()).backward()
        opt.step()
    m = mesh.double().numpy()
    areas = np.array([integrate.quad(f, m[i], m[i + 1])[0] for i in range(nknots - 1)])
    assert np.abs(model.q_v().mean.numpy() - areas).max() < 0.03
    po, pd = model.posterior(torch.tensor(x[::7])), None
    assert po.mean.shape == (len(x[::7]),)


def test_initialisers_and_errors(engine):
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    X, y, *_ = D.gen_grid(12, 10)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    model = Matern12GriddedGP(Xt, yt, 6, (0, 1), (0, 1), engine=engine).to(torch.float64)
    ell_before = model.kernel_1.base_kernel.lengthscale.item()
    model.non_informative_initialise(lmbda=5.0, kappa=10.0)
    assert abs(model.kernel_1.outputscale.item() - yt.var().item()) < 1e-9
    assert model.kernel_1.base_kernel.lengthscale.item() == ell_before       # the reference's getter quirk
    assert abs(model.likelihood.noise.item() - yt.var().item() / 100.0) < 1e-9
    assert np.isfinite(model._elbo().item())
    # general scattered inputs (no underlying grid): routed to the scattered step (test_scattered_inputs_model_matches_dense_oracle)
    Xr = torch.tensor(np.random.default_rng(0).uniform(size=(200, 2)))
    sc = Matern12GriddedGP(Xr, yt[:200] if len(yt) >= 200 else torch.zeros(200), 6, (0, 1), (0, 1), engine=engine)
    assert sc._scattered and np.isfinite(sc._elbo().item())
    with pytest.raises(ValueError):      # duplicated points
        Matern12GriddedGP(torch.cat([Xt[:-3], Xt[:1]]), torch.cat([yt[:-3], yt[:1]]), 6, (0, 1), (0, 1), engine=engine)


def test_masked_grid_model_matches_dense_oracle(engine):
    """BASELINE config 5 through the model API: the reference gets the observed subset as X, y (any order)."""
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    n1, n2, nk = 32, 32, 9
    X, y, x1, x2 = D.gen_grid(n1, n2)
    rng = np.random.default_rng(4)
    keep = rng.permutation(np.flatnonzero(rng.uniform(size=n1 * n2) > 0.3))      # 30% missing, shuffled order
    Xo, yo = X[keep], y[keep]
    model = Matern12GriddedGP(torch.tensor(Xo), torch.tensor(yo), nk, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model._masked
    dm = D.DenseKron(Xo, yo, "b0", "matern12", torch.linspace(0, 1, nk), torch.linspace(0, 1, nk))
    e = model._elbo()
    e.backward()
    ed, gd = dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    got = np.array([model.kernel_1.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_2.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_1.raw_outputscale.grad.item(), model.kernel_2.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert rel(got, gd.numpy()) < 1e-5
    qv, qd = model.q_v(), dm.q_v()
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    xs = rng.uniform(0, 1, (300, 2))
    po, pd = model.posterior(torch.tensor(xs)), dm.posterior(xs)
    assert rel(po.mean.numpy(), pd.mean.detach().numpy()) < 1e-5
    assert rel(po.variance.numpy(), pd.variance.detach().numpy()) < 1e-5
    hist = model.fit(n_iter=5, lr=0.05)            # the fit loop runs and improves the bound
    assert hist[-1] < hist[0]


def test_config5_track_shaped_mask_through_the_model_class(engine):
    """SURVEY.md section 8d's "track-shaped" variant of BASELINE configs[4]: a 2048 x 2048 lat / lon grid observed only along
    crossing satellite tracks (datagen.generate_track, the array-level twin of dataloaders.py:290-377: gradient 2, one track every
    0.1 degrees of a 10-degree box -> ~9.5 % of the grid), handed to Matern12GriddedGP as the observed subset X, y exactly as
    the reference's notebooks 6 / 61 do.  ELBO, raw-parameter gradients and q(v) against the structured masked oracle on the
    same (grid, mask); a small case of the same construction against the literal dense restatement."""
    from variational_gridded_gaussian_processes_amd import datagen as G
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    n, nk = 2048, 33
    X, y, x1, x2 = D.gen_grid(n, n)
    Wn = G.track_mask(n, n, trajectory_gradient=2, track_sparsity=0.1)
    assert 0.08 < Wn.mean() < 0.11
    obs = np.flatnonzero(Wn.reshape(-1) > 0)
    model = Matern12GriddedGP(torch.tensor(X[obs]), torch.tensor(y[obs]), nk, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model._masked and not model._scattered
    e = model._elbo()
    e.backward()
    mesh = torch.linspace(0, 1, nk).double().numpy()                    # the reference's float32-born mesh values
    # every row and column of the grid carries a track point here, so the model's grid of unique coordinates is the full one
    assert len(model._x1) == n and len(model._x2) == n
    f1, f2 = Kr.Factor("b0", "matern12", mesh, x1, True), Kr.Factor("b0", "matern12", mesh, x2, True)
    raw = np.zeros(5)
    ref = Kr.elbo_step_masked(y.reshape(n, n), Wn, f1, f2, Kr.theta_from_raw(raw))
    g_raw = Kr.grad_raw(ref.grad, raw)
    got = np.array([model.kernel_1.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_2.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_1.raw_outputscale.grad.item(), model.kernel_2.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert abs(e.item() - ref.elbo) <= 1e-8 * abs(ref.elbo)
    assert rel(got, g_raw) < 1e-6
    qv = model.q_v()
    rm, rv = Kr.q_v_masked(ref)
    assert rel(qv.mean.numpy(), rm.reshape(-1)) < 1e-6 and rel(qv.variance.numpy(), rv.reshape(-1)) < 1e-6
    # the same construction at a size the literal dense restatement can run
    n, nk = 48, 7
    X, y, x1, x2 = D.gen_grid(n, n)
    Wn = G.track_mask(n, n, trajectory_gradient=2, track_sparsity=2.0)
    obs = np.flatnonzero(Wn.reshape(-1) > 0)
    model = Matern12GriddedGP(torch.tensor(X[obs]), torch.tensor(y[obs]), nk, (0, 1), (0, 1), engine=engine).to(torch.float64)
    dm = D.DenseKron(X[obs], y[obs], "b0", "matern12", torch.linspace(0, 1, nk), torch.linspace(0, 1, nk))
    ed, gd = dm.elbo_and_grad()
    e = model._elbo()
    e.backward()
    assert abs(e.item() - ed.item()) <= 1e-6 * abs(ed.item())
    qv, qd = model.q_v(), dm.q_v()
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5 and rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5


def test_dense_debug_views_and_prior(engine):
    """_Kuu / _Kuf / _sigma / prior (kept for small sizes) equal the dense restatement."""
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    X, y, *_ = D.gen_grid(10, 8)
    model = Matern12GriddedGP(torch.tensor(X), torch.tensor(y), 6, (0, 1), (0, 1), engine=engine).to(torch.float64)
    dm = D.DenseKron(X, y, "b0", "matern12", torch.linspace(0, 1, 6), torch.linspace(0, 1, 6))
    th = dm.theta().detach()
    Kuu_ref = torch.kron(dm._Kuu_d(0), dm._Kuu_d(1)).detach()
    assert rel(model._Kuu().numpy(), Kuu_ref.numpy()) < 1e-5                 # float32 k*delta of the reference mesh
    Kuf = model._Kuf(torch.tensor(X))
    assert Kuf.shape == (25, 80)
    Kuf_ref = (dm._Kuf_d(0, torch.tensor(X[:, 0]))[:, None, :] * dm._Kuf_d(1, torch.tensor(X[:, 1]))[None, :, :]).reshape(25, 80).detach()
    assert rel(Kuf.numpy(), Kuf_ref.numpy()) < 1e-5
    S = model._sigma()
    assert rel(S.numpy(), (Kuu_ref + Kuf_ref @ Kuf_ref.T / th[4]).numpy()) < 1e-5
    pr = model.prior(torch.tensor(X[:7]))
    k = th[2] * th[3] * np.exp(-abs(X[0, 0] - X[3, 0]) / th[0].item() - abs(X[0, 1] - X[3, 1]) / th[1].item())
    assert abs(pr.covariance_matrix[0, 3].item() - float(k)) < 1e-12 and pr.mean.abs().max() == 0


def test_not_positive_definite_raises_linalgerror(engine):
    """ENOTPD after the jitter schedule surfaces as torch.linalg.LinAlgError (what notebook 61 cell 39 catches)."""
    from variational_gridded_gaussian_processes_amd.models import RBFSVGP
    X, y, *_ = D.gen_grid(12, 10)
    z1 = np.linspace(0, 1, 8)
    z1[3] = np.nan                                       # poisons K1: no jitter level can make it positive definite
    Z = torch.tensor(np.stack([z1, np.linspace(0, 1, 8)], axis=1))
    model = RBFSVGP(torch.tensor(X), torch.tensor(y), Z, engine=engine).to(torch.float64)
    with pytest.raises(torch.linalg.LinAlgError):
        model._elbo()


@pytest.mark.parametrize("cls,basis", [("Matern12VFFGP", "vff"), ("Matern12B1SplineASVGP", "b1")])
def test_interdomain_models_vs_dense(engine, cls, basis):
    """The reference's other two 2-D Kronecker models through the model API, against the literal dense
    restatement (float32 omegas / mesh as the reference builds them)."""
    import variational_gridded_gaussian_processes_amd.models as M
    n1, n2 = 24, 20
    X, y, x1, x2 = D.gen_grid(n1, n2)
    if basis == "vff":
        lims, nf = (-0.1, 1.1), 5
        model = M.Matern12VFFGP(torch.tensor(X), torch.tensor(y), nf, lims, lims, engine=engine).to(torch.float64)
        dm = D.DenseKron(X, y, "vff", "matern12", (lims[0], lims[1], nf), (lims[0], lims[1], nf))
    else:
        model = M.Matern12B1SplineASVGP(torch.tensor(X), torch.tensor(y), 8, (0, 1), (0, 1), engine=engine).to(torch.float64)
        dm = D.DenseKron(X, y, "b1", "matern12", torch.linspace(0, 1, 8), torch.linspace(0, 1, 8))
    e = model._elbo()
    e.backward()
    ed, gd = dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    got = np.array([model.kernel_1.base_kernel.raw_lengthscale.grad.item(), model.kernel_2.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_1.raw_outputscale.grad.item(), model.kernel_2.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert rel(got, gd.numpy()) < 1e-5
    qv, qd = model.q_v(), dm.q_v()
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5 and rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    xs = np.random.default_rng(1).uniform(0, 1, (50, 2))
    po, pd = model.posterior(torch.tensor(xs)), dm.posterior(xs)
    assert rel(po.mean.numpy(), pd.mean.detach().numpy()) < 1e-5 and rel(po.variance.numpy(), pd.variance.detach().numpy()) < 1e-5
    assert rel(model._Kuu().numpy(), torch.kron(dm._Kuu_d(0), dm._Kuu_d(1)).detach().numpy()) < 1e-5
    hist = model.fit(n_iter=4, lr=0.05)
    assert hist[-1] < hist[0]


def test_gridded_vff_model_vs_dense(engine):
    """GriddedMatern12VFFGP (gridded_kronecker_structure.py:470-654): the gridded read-out through the model API against
    the literal dense formulas."""
    from variational_gridded_gaussian_processes_amd.models import GriddedMatern12VFFGP
    n1, n2, nf, ns = 24, 20, 5, 7
    X, y, x1, x2 = D.gen_grid(n1, n2)
    lims = (-0.1, 1.1)
    model = GriddedMatern12VFFGP(torch.tensor(X), torch.tensor(y), nf, lims, lims, ns, (0, 1), (0, 1), engine=engine).to(torch.float64)
    dm = D.DenseKron(X, y, "vff", "matern12", (lims[0], lims[1], nf), (lims[0], lims[1], nf))
    mesh = torch.linspace(0, 1, ns + 1)
    qd = dm.q_v_gridded(mesh, mesh, literal=True)
    qv = model.q_v()
    assert qv.mean.shape == (ns * ns,)
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    qu, qud = model.q_u(), dm.q_v()
    assert rel(qu.mean.numpy(), qud.mean.detach().numpy()) < 1e-5


@pytest.mark.parametrize("data", ["masked", "scattered"])
def test_gridded_vff_model_readout_on_incomplete_data_vs_dense(engine, data):
    """GriddedMatern12VFFGP.q_v() when the observations are a grid with holes or scattered points (vggp_readout_masked: the
    read-out from the dense M-space state) against the literal dense formulas on the same points."""
    import variational_gridded_gaussian_processes_amd.models as M
    n1, n2, nf, ns = 24, 20, 5, 7
    X, y, x1, x2 = D.gen_grid(n1, n2)
    rng = np.random.default_rng(5)
    keep = rng.random(len(y)) > 0.3
    X, y = X[keep], y[keep]
    if data == "scattered":
        X = np.clip(X + rng.normal(scale=4e-3, size=X.shape), 0.0, 1.0)
    lims = (-0.1, 1.1)
    model = M.GriddedMatern12VFFGP(torch.tensor(X), torch.tensor(y), nf, lims, lims, ns, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model._masked and model._scattered == (data == "scattered")
    dm = D.DenseKron(X, y, "vff", "matern12", (lims[0], lims[1], nf), (lims[0], lims[1], nf))
    mesh = torch.linspace(0, 1, ns + 1)
    for literal in (True, False):
        qd = dm.q_v_gridded(mesh, mesh, literal=literal)
        qv = model.q_v(literal=literal)
        assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
        assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    # the SVGP-in-points gridded class (gridded_kronecker_structure.py:222-460) on the same points
    z = torch.linspace(0, 1, 6, dtype=torch.float64)
    sv = M.GriddedMatern12SVGP(torch.tensor(X), torch.tensor(y), torch.cartesian_prod(z, z), ns, (0, 1), (0, 1), engine=engine).to(torch.float64)
    ds = D.DenseKron(X, y, "points", "matern12", z, z)
    qd, qv = ds.q_v_gridded(mesh, mesh), sv.q_v()
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5 and rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5


@pytest.mark.parametrize("cls", ["b0", "vff", "points"])
def test_posterior_dense_covariance_vs_dense_restatement(engine, cls):
    """posterior(x*).covariance_matrix -- the reference's dense N* x N* matrix (kronecker_structure.py:223-229) -- and
    posterior_predictive's (+ noise on the diagonal, :232-247) against the literal dense restatement, for the B0 flagship,
    a VFF model (Kuu scales with 1/s) and an SVGP model."""
    import variational_gridded_gaussian_processes_amd.models as M
    n1, n2 = 20, 16
    X, y, x1, x2 = D.gen_grid(n1, n2)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    if cls == "b0":
        model = M.Matern12GriddedGP(Xt, yt, 7, (0, 1), (0, 1), engine=engine).to(torch.float64)
        dm = D.DenseKron(X, y, "b0", "matern12", torch.linspace(0, 1, 7), torch.linspace(0, 1, 7))
    elif cls == "vff":
        lims, nf = (-0.1, 1.1), 4
        model = M.Matern12VFFGP(Xt, yt, nf, lims, lims, engine=engine).to(torch.float64)
        dm = D.DenseKron(X, y, "vff", "matern12", (lims[0], lims[1], nf), (lims[0], lims[1], nf))
    else:
        z = np.linspace(0, 1, 6)
        model = M.Matern32SVGP(Xt, yt, torch.tensor(np.stack([z, z], axis=1)), engine=engine).to(torch.float64)
        dm = D.DenseKron(X, y, "points", "matern32", torch.tensor(z), torch.tensor(z))
    xs = np.random.default_rng(4).uniform(0, 1, (37, 2))
    po, pd = model.posterior(torch.tensor(xs)), dm.posterior(xs)
    cov, ref = po.covariance_matrix.numpy(), pd.covariance_matrix.detach().numpy()
    assert cov.shape == (37, 37)
    assert rel(cov, ref) < 1e-5
    assert np.abs(np.diag(cov) - po.variance.numpy()).max() <= 1e-9 * np.abs(ref).max()
    pp, ppd = model.posterior_predictive(torch.tensor(xs)), dm.posterior_predictive(xs)
    assert rel(pp.covariance_matrix.numpy(), ppd.covariance_matrix.detach().numpy()) < 1e-5


def test_masked_dense_covariances_vs_dense_restatement(engine):
    """Masked grid (the observed subset as X, y): dense covariances of posterior(x*) and of q(v) from Sigma~^{-1}."""
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP
    n1, n2, nk = 18, 14, 6
    X, y, x1, x2 = D.gen_grid(n1, n2)
    keep = np.random.default_rng(2).uniform(size=n1 * n2) < 0.7
    Xo, yo = X[keep], y[keep]
    model = Matern12GriddedGP(torch.tensor(Xo), torch.tensor(yo), nk, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model._masked
    dm = D.DenseKron(Xo, yo, "b0", "matern12", torch.linspace(0, 1, nk), torch.linspace(0, 1, nk))
    xs = np.random.default_rng(5).uniform(0, 1, (21, 2))
    po, pd = model.posterior(torch.tensor(xs)), dm.posterior(xs)
    assert rel(po.covariance_matrix.numpy(), pd.covariance_matrix.detach().numpy()) < 1e-5
    qv, qd = model.q_v(), dm.q_v()
    assert rel(qv.covariance_matrix.numpy(), qd.covariance_matrix.detach().numpy()) < 1e-5


def test_basis_objects_mirror_the_reference(engine):
    """basis_1 / basis_2 (gridded_kronecker_structure.py:1283-1284, kronecker_structure.py:548-549, :464-470; bspline.py:81-112,
    fourier.py:5-88): bookkeeping attributes against the reference-produced pins, evaluation through the HIP factor kernel."""
    import os
    import variational_gridded_gaussian_processes_amd.models as M
    pins = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pins.npz"))
    bp = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pins_basis.npz"))
    X, y, *_ = D.gen_grid(8, 6)
    Xt, yt = torch.tensor(X), torch.tensor(y)
    g = M.Matern12GriddedGP(Xt, yt, 11, (0, 1), (0, 1), engine=engine)
    b = g.basis_1
    assert b.m == int(pins["b0_m"]) == b.n_basis_functions == int(pins["b0_nbasis"]) and b.order == 0
    assert float(b.delta) == float(pins["b0_delta"]) and np.array_equal(b.mesh.numpy(), pins["b0_mesh"])
    ind = g.basis_2(torch.tensor([0.05, 0.55]))
    assert ind.shape == (10, 2) and int(ind[0, 0]) == 1 and int(ind[5, 1]) == 1 and int(ind.sum()) == 2
    # B1: hats on the reference's float64 pin mesh
    mesh = torch.tensor(bp["b1_f64_mesh"])
    a = M.Matern12B1SplineASVGP(Xt, yt, len(mesh), (float(mesh[0]), float(mesh[-1])), (0, 1), engine=engine)
    assert a.basis_1.n_basis_functions == len(mesh) and a.basis_1.order == 1
    assert np.abs(a.basis_1(torch.tensor(bp["b1_f64_x"])).numpy() - bp["b1_f64_Phi"]).max() < 1e-6      # float32 linspace mesh
    # Fourier: the pinned (M, a, b, ell) case through the model's basis property
    Mf, fa, fb, ell = int(bp["vff_a_M"]), float(bp["vff_a_a"]), float(bp["vff_a_b"]), float(bp["vff_a_ell"])
    v = M.Matern12VFFGP(Xt, yt, Mf, (fa, fb), (fa, fb), engine=engine).to(torch.float64)
    v.kernel_1.base_kernel.lengthscale = ell
    fbasis = v.basis_1
    assert fbasis.M == Mf and fbasis.a == fa and fbasis.b == fb and np.array_equal(fbasis.omegas.numpy(), bp["vff_a_omegas"])
    assert abs(fbasis.lengthscale - ell) < 1e-12
    Phi = fbasis(torch.tensor(bp["vff_a_x"])).numpy()
    assert np.abs(Phi - bp["vff_a_Phi"]).max() < 2e-7 * max(1.0, float(np.abs(bp["vff_a_omegas"]).max() * np.abs(bp["vff_a_x"] - fa).max()))


@pytest.mark.parametrize("z_fastest", ["second", "first"])
def test_gridded_svgp_model_vs_dense(engine, z_fastest):
    """GriddedMatern12SVGP (gridded_kronecker_structure.py:222-460) with Z = cartesian_prod(z1, z2) in either row order:
    q_v (the gridded read-out: mean and the reference's literal variance) and q_u against the literal dense formulas."""
    from variational_gridded_gaussian_processes_amd.models import GriddedMatern12SVGP
    n1, n2, ns = 24, 20, 6
    X, y, x1, x2 = D.gen_grid(n1, n2)
    z1, z2 = np.linspace(0.05, 0.95, 7), np.linspace(0.0, 1.0, 5)
    if z_fastest == "second":
        Z = torch.cartesian_prod(torch.tensor(z1), torch.tensor(z2))
    else:
        Z = torch.tensor(np.stack(np.meshgrid(z1, z2), axis=-1).reshape(-1, 2))          # gen_2d layout: first coordinate fastest
    model = GriddedMatern12SVGP(torch.tensor(X), torch.tensor(y), Z, ns, (0, 1), (0.1, 0.9), engine=engine).to(torch.float64)
    dm = D.DenseKron(X, y, "points", "matern12", torch.tensor(z1), torch.tensor(z2))
    mesh1, mesh2 = torch.linspace(0, 1, ns + 1), torch.linspace(0.1, 0.9, ns + 1)
    qd = dm.q_v_gridded(mesh1, mesh2, literal=True)
    qv = model.q_v()
    assert qv.mean.shape == (ns * ns,)
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    qc, qcd = model.q_v(literal=False), dm.q_v_gridded(mesh1, mesh2, literal=False)
    assert rel(qc.variance.numpy(), qcd.variance.detach().numpy()) < 1e-5
    # q_u in the row order of the caller's Z: the dense restatement orders u = i1 * m2 + i2
    qu, qud = model.q_u(), dm.q_v()
    ref_mean = qud.mean.detach().numpy().reshape(len(z1), len(z2))
    got = qu.mean.numpy().reshape(len(z1), len(z2)) if z_fastest == "second" else qu.mean.numpy().reshape(len(z2), len(z1)).T
    assert rel(got, ref_mean) < 1e-5
    e, (ed, _) = model._elbo(), dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    with pytest.raises(ValueError):
        GriddedMatern12SVGP(torch.tensor(X), torch.tensor(y), Z[:-1], ns, (0, 1), (0, 1), engine=engine)


def test_gridded_asvgp_model_vs_dense(engine):
    """GriddedMatern12ASVGP (gridded_kronecker_structure.py:685-969): B1 features on the padded mesh, the reference's own Kvu."""
    from variational_gridded_gaussian_processes_amd.models import GriddedMatern12ASVGP
    n1, n2, ns, pad = 24, 20, 7, 2
    X, y, x1, x2 = D.gen_grid(n1, n2)
    model = GriddedMatern12ASVGP(torch.tensor(X), torch.tensor(y), ns, pad, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model.b0_mesh_padded_1.shape[0] == ns + 1 + 2 * pad and model.b1_basis_1.n_basis_functions == ns + 1 + 2 * pad
    dm = D.DenseKron(X, y, "b1", "matern12", model.b0_mesh_padded_1, model.b0_mesh_padded_2)
    qd = dm.q_v_gridded(model.b0_mesh_1, model.b0_mesh_2, literal=True)
    qv = model.q_v()
    assert qv.mean.shape == (ns * ns,)
    assert rel(qv.mean.numpy(), qd.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), qd.variance.detach().numpy()) < 1e-5
    e, (ed, _) = model._elbo(), dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    qu, qud = model.q_u(), dm.q_v()
    assert rel(qu.mean.numpy(), qud.mean.detach().numpy()) < 1e-5


@pytest.mark.parametrize("cls,kind", [("Matern12SVGP", "matern12"), ("Matern32SVGP", "matern32"), ("Matern52SVGP", "matern52")])
def test_svgp_trainable_inducing_points_vs_dense_autograd(engine, cls, kind):
    """The reference registers Z as a trainable Parameter (kronecker_structure.py:303-304) and autograd differentiates the ELBO
    through kernel(Z) and kernel(cartesian_prod(Z), x): Z.grad from the engine's analytic vggp_zgrad against autograd through the
    literal dense restatement, and an Adam loop that moves Z (re-planning the engine every step)."""
    import variational_gridded_gaussian_processes_amd.models as M
    n1, n2, m = 18, 15, 6
    X, y, x1, x2 = D.gen_grid(n1, n2)
    rng = np.random.default_rng(5)
    Z = torch.tensor(np.stack([np.sort(rng.uniform(0.05, 0.95, m)), np.sort(rng.uniform(0.05, 0.95, m))], axis=1))
    model = getattr(M, cls)(torch.tensor(X), torch.tensor(y), Z, engine=engine).to(torch.float64)
    assert model.Z.requires_grad
    dm = D.DenseKron(X, y, "points", kind, Z[:, 0].clone(), Z[:, 1].clone())
    dm.grid_1.requires_grad_(True)
    dm.grid_2.requires_grad_(True)
    e = model._elbo()
    e.backward()
    ed = dm._elbo()
    ed.backward()
    want = torch.stack([dm.grid_1.grad, dm.grid_2.grad], dim=1).numpy()
    assert abs(e.item() - ed.item()) <= 1e-5 * abs(ed.item())
    assert rel(model.Z.grad.numpy(), want) < 1e-5
    # a few optimiser steps: every parameter, Z included, moves and the bound improves
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    z0, first = model.Z.detach().clone(), None
    for it in range(6):
        opt.zero_grad()
        loss = -model._elbo()
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < first and (model.Z.detach() - z0).abs().max() > 1e-3
    # train_z=False keeps the inducing points out of the graph
    fixed = getattr(M, cls)(torch.tensor(X), torch.tensor(y), Z, engine=engine, train_z=False).to(torch.float64)
    fixed._elbo().backward()
    assert fixed.Z.grad is None


def test_univariate_svgp_trainable_inducing_points(engine):
    """1-D twin (univariate_structure.py:273-332, Z a trainable Parameter): Z.grad against autograd through the dense 1-D
    restatement."""
    from variational_gridded_gaussian_processes_amd.models import univariate
    rng = np.random.default_rng(2)
    n, m = 60, 9
    x = np.linspace(0, 1, n)
    y = np.sin(7 * x) + 0.1 * rng.standard_normal(n)
    z = np.sort(rng.uniform(0.05, 0.95, m))
    model = univariate.Matern32SVGP(torch.tensor(x), torch.tensor(y), torch.tensor(z), engine=engine).to(torch.float64)
    dm = D.Dense1D(x, y, "points", "matern32", torch.tensor(z))
    dm.grid.requires_grad_(True)
    e = model._elbo()
    e.backward()
    ed = dm._elbo()
    ed.backward()
    assert abs(e.item() - ed.item()) <= 1e-6 * abs(ed.item())
    assert rel(model.Z.grad.numpy().reshape(-1), dm.grid.grad.numpy()) < 1e-5


def test_scattered_inputs_model_matches_dense_oracle(engine):
    """X that is no (masked) grid at all -- along-track style points, what the reference's notebooks 6 / 61 / 7 feed to the same
    classes: the model routes to vggp_elbo_step_scattered; ELBO, raw gradients, q_v and posterior against the literal dense
    restatement on the same points."""
    from variational_gridded_gaussian_processes_amd.models import Matern12GriddedGP, Matern32SVGP
    rng = np.random.default_rng(4)
    N, nk = 400, 8
    t = np.linspace(0, 1, N)
    X = np.stack([(0.5 + 0.45 * np.sin(9 * t) + 0.01 * rng.standard_normal(N)).clip(0, 1),
                  (t + 0.02 * rng.standard_normal(N)).clip(0, 1)], axis=1)          # a wiggly track
    y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.05 * rng.standard_normal(N)
    model = Matern12GriddedGP(torch.tensor(X), torch.tensor(y), nk, (0, 1), (0, 1), engine=engine).to(torch.float64)
    assert model._scattered
    dm = D.DenseKron(X, y, "b0", "matern12", torch.linspace(0, 1, nk), torch.linspace(0, 1, nk))
    e = model._elbo()
    e.backward()
    ed, gd = dm.elbo_and_grad()
    assert abs(e.item() - ed.item()) <= 1e-6 * abs(ed.item())
    got = np.array([model.kernel_1.base_kernel.raw_lengthscale.grad.item(), model.kernel_2.base_kernel.raw_lengthscale.grad.item(),
                    model.kernel_1.raw_outputscale.grad.item(), model.kernel_2.raw_outputscale.grad.item(),
                    model.likelihood.raw_noise.grad.item()])
    assert rel(got, gd.numpy()) < 1e-5
    qv, dq = model.q_v(), dm.q_v()
    assert rel(qv.mean.numpy(), dq.mean.detach().numpy()) < 1e-5
    assert rel(qv.variance.numpy(), torch.diagonal(dq.covariance_matrix).detach().numpy()) < 1e-5
    xs = rng.uniform(0, 1, (25, 2))
    p, dp = model.posterior(torch.tensor(xs)), dm.posterior(torch.tensor(xs))
    assert rel(p.mean.numpy(), dp.mean.detach().numpy()) < 1e-5
    assert rel(p.variance.numpy(), torch.diagonal(dp.covariance_matrix).detach().numpy()) < 1e-5
    # an Adam loop on scattered data (points basis, fixed Z)
    Z = torch.tensor(np.stack([np.linspace(0, 1, 7), np.linspace(0, 1, 7)], axis=1))
    sv = Matern32SVGP(torch.tensor(X), torch.tensor(y), Z, engine=engine, train_z=False).to(torch.float64)
    opt = torch.optim.Adam(sv.parameters(), lr=0.05)
    first = None
    for it in range(8):
        opt.zero_grad()
        loss = -sv._elbo()
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < first


@pytest.mark.parametrize("data", ["scattered", "holes"])
def test_svgp_trainable_inducing_points_on_incomplete_data(engine, data):
    """Matern32SVGP with its default trainable Z on scattered points and on a grid with holes (treated as its observed points):
    Z.grad (vggp_zgrad_scattered) and the hyper-parameter gradients against autograd through the literal dense restatement on
    the same points, then an Adam loop that moves everything."""
    import variational_gridded_gaussian_processes_amd.models as M
    rng = np.random.default_rng(6)
    if data == "scattered":
        N = 350
        X = rng.uniform(0, 1, (N, 2))
        y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.05 * rng.standard_normal(N)
    else:
        X, y, x1, x2 = D.gen_grid(20, 17)
        keep = rng.random(len(y)) > 0.3
        X, y = X[keep], y[keep]
    m = 6
    Z = torch.tensor(np.stack([np.linspace(0.05, 0.95, m) + rng.uniform(-0.03, 0.03, m),
                               np.linspace(0.05, 0.95, m) + rng.uniform(-0.03, 0.03, m)], axis=1))
    model = M.Matern32SVGP(torch.tensor(X), torch.tensor(y), Z, engine=engine).to(torch.float64)
    assert model.Z.requires_grad and model._scattered
    dm = D.DenseKron(X, y, "points", "matern32", Z[:, 0].clone(), Z[:, 1].clone())
    dm.grid_1.requires_grad_(True)
    dm.grid_2.requires_grad_(True)
    e = model._elbo()
    e.backward()
    ed = dm._elbo()
    ed.backward()
    want = torch.stack([dm.grid_1.grad, dm.grid_2.grad], dim=1).numpy()
    assert abs(e.item() - ed.item()) <= 1e-6 * abs(ed.item())
    assert rel(model.Z.grad.numpy(), want) < 1e-5
    opt = torch.optim.Adam(model.parameters(), lr=0.01)
    z0, first = model.Z.detach().clone(), None
    for it in range(6):
        opt.zero_grad()
        loss = -model._elbo()
        loss.backward()
        opt.step()
        first = loss.item() if first is None else first
    assert loss.item() < first and (model.Z.detach() - z0).abs().max() > 1e-3
    p = model.posterior(torch.tensor(rng.uniform(0, 1, (10, 2))))
    assert torch.isfinite(p.mean).all() and (p.variance > 0).all()
