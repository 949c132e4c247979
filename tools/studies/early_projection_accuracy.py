"""S = (L^-1 A) Y  versus  S = L^-1 (A Y): both in float64 against a long-double reference, on the RBF factors of
test_rbf_warm_chain_at_small_inducing_counts (m = 24, ell = 0.3, 192 x 192 grid) and of the headline (m = 128, ell = 0.2, 1024)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as sla
from oracle import dense as D, kron as Kr

def trsm_ld(L, R):           # forward substitution in long double
    L = L.astype(np.longdouble); X = R.astype(np.longdouble).copy()
    for i in range(L.shape[0]):
        X[i] = (X[i] - L[i, :i] @ X[:i]) / L[i, i]
    return X

for n, m, ell in ((192, 24, 0.3042), (512, 128, 0.2)):
    X, y, x1, x2 = D.gen_grid(n, n)
    Y = y.reshape(n, n)
    g = np.linspace(0, 1, m)
    f = Kr.Factor("points", "rbf", g, x2)
    K0, dK0, A0, dA0 = f.build(ell)
    L, jit = Kr.chol_jitter(K0)
    ref = trsm_ld(L, A0.astype(np.longdouble) @ Y.astype(np.longdouble))          # exact-ish S for THIS L
    B = sla.solve_triangular(L, A0, lower=True)
    S_orig = B @ Y
    S_early = sla.solve_triangular(L, A0 @ Y, lower=True)
    sc = np.abs(ref).max()
    print(f"n={n} m={m} jitter={jit:g}  |S|max={float(sc):.3e}   orig err {float(np.abs(S_orig - ref).max() / sc):.2e}   early err {float(np.abs(S_early - ref).max() / sc):.2e}")
