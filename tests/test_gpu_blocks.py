"""GPU unit tests of the exported building blocks (C-ABI) against float64 torch/numpy references."""
import numpy as np
import pytest
import torch

from oracle import kron as Kr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 1024, 128), (50, 37, 29), (256, 100, 1000), (1, 5, 3),
                                   (1024, 1024, 96), (640, 1344, 160)])
def test_gemm_layouts(engine, shape):
    """All four operand orientations; asymmetric data so a transposed fragment map cannot hide.  The last two shapes are plain
    launches of >= 128 whole tiles with k a multiple of 32: they run on the deep-stage tile (gemm.hip vg_gemm_deep_body)."""
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(1)
    A = torch.randn(M, K, generator=g, dtype=torch.float64).to(DEV)
    B = torch.randn(K, N, generator=g, dtype=torch.float64).to(DEV)
    At = A.t().contiguous().t()      # M-contiguous view of the same values
    Bt = B.t().contiguous().t()      # K-contiguous view
    ref = (A.cpu() @ B.cpu()).numpy()
    for a in (A, At):
        for b in (B, Bt):
            out = engine.gemm(a, b).cpu().numpy()
            assert rel(out, ref) < 1e-13, (a.stride(), b.stride())


def test_gemm_identity_asymmetric(engine):
    A = torch.eye(64, dtype=torch.float64, device=DEV)
    B = (torch.arange(64 * 64, dtype=torch.float64, device=DEV).reshape(64, 64) * 1.0)
    assert torch.equal(engine.gemm(A, B), B)
    assert torch.equal(engine.gemm(B, A), B)


@pytest.mark.parametrize("kind", ["matern12", "matern32", "matern52", "rbf"])
def test_factor_build_points(engine, kind):
    x = np.linspace(0, 1, 301)
    z = np.linspace(0, 1, 40)
    ell = 0.23
    A, dA, K, dK = engine.factor_build(kind, "points", torch.tensor(x, device=DEV), torch.tensor(z, device=DEV), ell)
    rA, rdA = Kr.points_factor(kind, z, x, ell)
    rK, rdK = Kr.points_factor(kind, z, z, ell)
    for got, ref in ((A, rA), (dA, rdA), (K, rK), (dK, rdK)):
        assert rel(got.cpu().numpy(), ref) < 1e-13


def test_factor_build_b0(engine):
    x = np.concatenate([np.linspace(0, 1, 257), [0.0, 1.0, 0.25, 0.5]])   # includes knots and both ends
    mesh = np.linspace(0, 1, 33)
    ell = 0.31
    A, dA, K, dK = engine.factor_build("matern12", "b0", torch.tensor(x, device=DEV), torch.tensor(mesh, device=DEV), ell)
    rA, rdA = Kr.b0_A(mesh, x, ell)
    rK, rdK = Kr.b0_K(32, mesh[1] - mesh[0], ell)
    for got, ref in ((A, rA), (dA, rdA), (K, rK), (dK, rdK)):
        assert rel(got.cpu().numpy(), ref) < 1e-12


def test_factor_build_vs_reference_basis_pins(engine):
    """HIP vggp_factor_build(VGGP_BASIS_VFF / VGGP_BASIS_B1) A-factors against outputs of the reference's own importable
    basis classes (tests/golden/ref_pins_basis.npz: FourierBasisMatern12(M, a, b, ell)(x), fourier.py:58-88 -- float32
    arithmetic there, hence the tolerance -- and B1SplineBasis(mesh)(x), bspline.py:106-112)."""
    import os
    p = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_pins_basis.npz"))
    for tag in ("vff_a", "vff_b"):
        M, a, b, ell = int(p[tag + "_M"]), float(p[tag + "_a"]), float(p[tag + "_b"]), float(p[tag + "_ell"])
        x, om32, ref = p[tag + "_x"], p[tag + "_omegas"], p[tag + "_Phi"]
        grid = np.concatenate([[a, b], om32.astype(np.float64)])
        A, dA, K, dK = engine.factor_build("matern12", "vff", torch.tensor(x, device=DEV), torch.tensor(grid, device=DEV), ell)
        tol = 2e-7 * max(1.0, float(np.abs(om32).max() * np.abs(x - a).max()))
        assert A.shape == ref.shape and np.abs(A.cpu().numpy() - ref).max() < tol
        outside = (x < a) | (x >= b)
        assert np.abs(A.cpu().numpy()[:, outside] - ref[:, outside]).max() < 2e-7
    for tag, tol in (("b1_f64", 1e-14), ("b1_f32", 1e-6)):
        mesh, x, ref = p[tag + "_mesh"].astype(np.float64), p[tag + "_x"], p[tag + "_Phi"]
        A, dA, K, dK = engine.factor_build("matern12", "b1", torch.tensor(x, device=DEV), torch.tensor(mesh, device=DEV), 0.3)
        assert A.shape == ref.shape and np.abs(A.cpu().numpy() - ref).max() < tol
        assert float(dA.abs().max()) == 0.0


@pytest.mark.parametrize("m,kind,ell", [(7, "matern12", 0.3), (64, "matern32", 0.2), (128, "matern52", 0.2),
                                        (128, "rbf", 0.2), (150, "matern32", 0.1), (300, "rbf", 0.05),
                                        (1024, "matern32", 0.05)])
def test_cholesky_inverse(engine, m, kind, ell):
    z = np.linspace(0, 1, m)
    K, _ = Kr.points_factor(kind, z, z, ell)
    L, Li, jit = engine.cholesky_inverse(torch.tensor(K, device=DEV))
    Lr, jr = Kr.chol_jitter(K)
    assert jit == jr
    L, Li = L.cpu().numpy(), Li.cpu().numpy()
    Kj = K + jit * np.eye(m)
    # m > 128: blocked path, panels solved with the explicit inverse of the (ill-conditioned) diagonal blocks
    assert rel(L @ L.T, Kj) < (3e-13 if m <= 128 else 2e-12)
    assert np.abs(np.triu(L, 1)).max() == 0 and np.abs(np.triu(Li, 1)).max() == 0
    # L Li = I to conditioning
    assert np.abs(L @ Li - np.eye(m)).max() < 1e-9 * max(1.0, np.linalg.cond(Lr) * 1e-4)


@pytest.mark.parametrize("block", [False, True], ids=["scalar", "block"])
@pytest.mark.parametrize("m,kind", [(1, "matern12"), (9, "matern12"), (64, "rbf"), (128, "matern32"), (150, "matern12"),
                                    (185, "matern12"), (200, "matern32"), (256, "matern12"), (256, "rbf")])
def test_eigh(engine, m, kind, block):
    if block and m > 150:
        pytest.skip("the block-Jacobi variant exists for m <= 128 (larger m falls back to the scalar solver: covered at 150)")
    xx = np.linspace(0, 1, 4 * m + 3)
    f = Kr.Factor("points", kind, np.linspace(0, 1, m), xx)
    d = Kr.dim_prepare(f, 0.2, 1.0)
    G = d.B @ d.B.T
    lam, Qt, sweeps = engine.eigh(torch.tensor(G, device=DEV), block=block)
    lam, Qt = lam.cpu().numpy(), Qt.cpu().numpy()
    w = np.linalg.eigvalsh(G)
    assert np.abs(np.sort(lam) - w).max() < 1e-12 * w.max()
    assert np.abs(Qt @ Qt.T - np.eye(m)).max() < 1e-11
    assert np.linalg.norm(Qt @ G @ Qt.T - np.diag(lam)) < 1e-11 * np.linalg.norm(G)
    assert 1 <= sweeps < 60
    if not block:            # the scalar variant hands the eigenpairs back sorted (the next warm start relies on it)
        assert np.all(np.diff(lam) <= 0)


@pytest.mark.parametrize("trans", [False, True], ids=["L", "Lt"])
@pytest.mark.parametrize("m,kind,ell", [(7, "matern12", 0.3), (64, "rbf", 0.2), (128, "rbf", 0.2), (130, "matern32", 0.2),
                                        (256, "matern52", 0.1), (300, "matern12", 0.2), (1024, "matern32", 0.05)])
def test_trsm_vs_scipy(engine, m, kind, ell, trans):
    """vggp_trsm (substitution on the matrix cores: strip kernel for the 128 x 128 diagonal blocks, MFMA GEMM updates
    between them) against scipy.linalg.solve_triangular, both orientations, ragged sizes, ill-conditioned RBF factors."""
    import scipy.linalg as sla
    z = np.linspace(0, 1, m)
    K, _ = Kr.points_factor(kind, z, z, ell)
    L, _ = Kr.chol_jitter(K)
    R = np.random.default_rng(m).standard_normal((m, 77))
    ref = sla.solve_triangular(L, R, lower=True, trans="T" if trans else "N")
    X = engine.trsm(torch.tensor(L, device=DEV), torch.tensor(R, device=DEV), trans=trans).cpu().numpy()
    # forward error relative to the solution scale (cond(L) up to 1e5 here) and the backward-stable residual
    assert rel(X, ref) < 1e-9
    resid = (L.T if trans else L) @ X - R
    assert np.abs(resid).max() < 1e-10 * max(1.0, np.abs(L).max() * np.abs(X).max())


def test_trsm_in_place_and_wide(engine):
    """ncols not a multiple of the 64-column workgroup tile, X aliasing R."""
    import scipy.linalg as sla
    m, ncols = 96, 1000
    z = np.linspace(0, 1, m)
    K, _ = Kr.points_factor("matern32", z, z, 0.15)
    L = np.linalg.cholesky(K)
    R = np.random.default_rng(1).standard_normal((m, ncols))
    Rt = torch.tensor(R, device=DEV)
    from variational_gridded_gaussian_processes_amd._lib import check
    check(engine.lib.vggp_trsm(engine._h, torch.tensor(L, device=DEV).data_ptr(), m, Rt.data_ptr(), ncols, Rt.data_ptr(), 0, 0))
    torch.cuda.synchronize()
    assert rel(Rt.cpu().numpy(), sla.solve_triangular(L, R, lower=True)) < 1e-11


@pytest.mark.parametrize("n1,n2", [(96, 130), (1024, 1024), (300, 520), (1000, 648)])
def test_kron_solve(engine, n1, n2):
    """BASELINE metric (ii) from the CHOLESKY FACTORS: X = K1^{-1} Y K2^{-T} (substitution for factors within one 128-block, explicit
    block-doubling inverses and K_d^-1 = Linv^T Linv products beyond; partial blocks included), against the CPU oracle
    (scipy.linalg.solve_triangular, oracle/kron.py kron_solve) and the size-independent residual K1 X K2^T == Y."""
    K1, _ = Kr.points_factor("matern32", np.linspace(0, 1, n1), np.linspace(0, 1, n1), 0.1 if n1 < 500 else 0.05)
    K2, _ = Kr.points_factor("matern12", np.linspace(0, 1, n2), np.linspace(0, 1, n2), 0.3 if n2 < 500 else 0.2)
    Y = np.random.default_rng(0).standard_normal((n1, n2))
    L1, L2 = np.linalg.cholesky(K1), np.linalg.cholesky(K2)
    X = engine.kron_solve(torch.tensor(L1, device=DEV), torch.tensor(L2, device=DEV), torch.tensor(Y, device=DEV)).cpu().numpy()
    ref = Kr.kron_solve(L1, L2, Y)
    assert rel(X, ref) < 1e-9
    # size-independent property: K1 X K2^T == Y, to the residual a backward-stable solve leaves (eps cond |Y|)
    r_gpu, r_ref = np.abs(K1 @ X @ K2.T - Y).max(), np.abs(K1 @ ref @ K2.T - Y).max()
    assert r_gpu < 1e-9 * np.abs(X).max() and r_gpu < 20 * max(r_ref, 1e-12)
    # factors straight from the engine's own Cholesky
    Lg1, _, _ = engine.cholesky_inverse(torch.tensor(K1, device=DEV))
    Lg2, _, _ = engine.cholesky_inverse(torch.tensor(K2, device=DEV))
    X2 = engine.kron_solve(Lg1, Lg2, torch.tensor(Y, device=DEV)).cpu().numpy()
    assert rel(X2, ref) < 1e-8


def test_sumsq(engine):
    y = torch.randn(100003, dtype=torch.float64, device=DEV)
    assert abs(engine.sumsq(y) - float((y * y).sum())) < 1e-9 * float((y * y).sum())


@pytest.mark.parametrize("block", [False, True], ids=["scalar", "block"])
@pytest.mark.parametrize("m", [128, 130, 150])
def test_eigh_bitwise_repeatable(engine, m, block):
    """Fixed reduction/rotation order: repeated solves are bit-identical (also exercises odd pair counts,
    two angle-phase waves, sparse rounds and the concurrent log hand-off to the replay workgroups)."""
    f = Kr.Factor("points", "matern12", np.linspace(0, 1, m), np.linspace(0, 1, 4 * m + 3))
    d = Kr.dim_prepare(f, 0.2, 1.0)
    G = torch.tensor(d.B @ d.B.T, device=DEV)
    lam0, Qt0, _ = engine.eigh(G, block=block)
    for _ in range(4):
        lam, Qt, _ = engine.eigh(G, block=block)
        assert torch.equal(lam, lam0) and torch.equal(Qt, Qt0)
    Gn, Q = G.cpu().numpy(), Qt0.cpu().numpy()
    assert np.linalg.norm(Q @ Gn @ Q.T - np.diag(lam0.cpu().numpy())) < 2e-13 * np.linalg.norm(Gn)
