"""How many cyclic sweeps does the r x r Ritz problem of the subspace start need (RBF, m = 128, 1 % change of the
lengthscale per step, r = 24), for which threshold, and does an ordered diagonal or a first-order pre-rotation help?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import kron as Kr
m, r = 128, 24
f = Kr.Factor("points", "rbf", np.linspace(0, 1, m), np.linspace(0, 1, 1024))
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0); return d.B @ d.B.T
def jacobi(H, thr, maxsweep=12):
    H = H.copy(); n = H.shape[0]; n1 = n - 1; out = []
    for sweep in range(maxsweep):
        nrot = 0
        for rd in range(n1):
            for k in range(n // 2):
                if k == 0: p, q = rd, n1
                else: p = (rd + k) % n1; q = (rd - k) % n1
                g = H[p, q]
                if abs(g) > thr:
                    nrot += 1
                    dd = H[q, q] - H[p, p]; o = 2 * g
                    t = abs(o) / (abs(dd) + np.hypot(dd, o))
                    if (dd >= 0) != (o >= 0): t = -t
                    c = 1 / np.sqrt(1 + t * t); s = t * c
                    Hp = H[:, p].copy(); Hq = H[:, q].copy()
                    H[:, p] = c * Hp - s * Hq; H[:, q] = s * Hp + c * Hq
                    Hp = H[p, :].copy(); Hq = H[q, :].copy()
                    H[p, :] = c * Hp - s * Hq; H[q, :] = s * Hp + c * Hq
        out.append(nrot)
        if nrot == 0: break
    return out
G0 = gram(0.2); lam, Q = np.linalg.eigh(G0); V = Q[:, ::-1].T[:r].copy()
for t in range(1, 4):
    G = gram(0.2 * 1.01 ** t)
    Z = V @ G
    V1 = np.linalg.qr((Z / np.linalg.norm(Z, axis=1)[:, None]).T)[0].T
    H = V1 @ G @ V1.T; H = (H + H.T) / 2
    nH = np.linalg.norm(H)
    off = np.abs(H - np.diag(np.diag(H)))
    print(f"step {t}: ||H|| {nH:.3e}, max offdiag {off.max():.2e}; diag sorted descending: {bool(np.all(np.diff(np.diag(H)) <= 0))}")
    for tol in (1e-13, 1e-12, 1e-11, 1e-10):
        thr = tol * nH / r
        print(f"   tol {tol:.0e} (thr {thr:.1e}): rotations per sweep {jacobi(H, thr)}")
    # first-order pre-rotation on well-separated pairs only
    d = np.diag(H)
    E = np.zeros_like(H)
    for i in range(r):
        for j in range(r):
            if i != j and abs(H[i, j]) < 1e-3 * abs(d[i] - d[j]): E[i, j] = H[i, j] / (d[j] - d[i])
    R = np.eye(r) + E + E @ E / 2
    R = np.linalg.qr(R)[0] * np.sign(np.diag(np.linalg.qr(R)[1]))
    H2 = R.T @ H @ R; H2 = (H2 + H2.T) / 2
    print(f"   after a first-order pre-rotation of the separated pairs: max offdiag {np.abs(H2 - np.diag(np.diag(H2))).max():.2e}; tol 1e-13: {jacobi(H2, 1e-13 * nH / r)}")
    w, W = np.linalg.eigh(H); V = (W[:, ::-1].T @ V1)

# ---- iterated first-order refinement (a Newton iteration on the diagonalising rotation) instead of Jacobi sweeps ---------------
print("\niterated refinement  R = I + E + E^2/2,  E_ij = h_ij / (h_jj - h_ii)  (pairs above thr; |E_ij| > 0.3 left to Jacobi)")
G0 = gram(0.2); lam, Q = np.linalg.eigh(G0); V = Q[:, ::-1].T[:r].copy()
for t in range(1, 4):
    G = gram(0.2 * 1.01 ** t)
    Z = V @ G
    V1 = np.linalg.qr((Z / np.linalg.norm(Z, axis=1)[:, None]).T)[0].T
    H0 = V1 @ G @ V1.T; H0 = (H0 + H0.T) / 2
    nH = np.linalg.norm(H0); thr = 1e-13 * nH / r
    H = H0.copy(); W = np.eye(r)
    for it in range(6):
        d = np.diag(H)
        E = np.zeros_like(H); skipped = 0
        for i in range(r):
            for j in range(i):
                if abs(H[i, j]) > thr:
                    e = H[i, j] / (d[j] - d[i]) if d[j] != d[i] else np.inf
                    if abs(e) < 0.3: E[i, j] = e; E[j, i] = -e
                    else: skipped += 1
        R = np.eye(r) + E + E @ E / 2
        H = R.T @ H @ R; H = (H + H.T) / 2
        W = W @ R
        off = np.abs(H - np.diag(np.diag(H)))
        print(f"   step {t} iteration {it}: max|E| {np.abs(E).max():.1e}, pairs skipped {skipped}, max offdiag after {off.max():.2e} (thr {thr:.1e}), elements above thr {int((np.tril(off, -1) > thr).sum())}, orth err {np.abs(W.T @ W - np.eye(r)).max():.1e}")
        if np.abs(E).max() == 0: break
    # one Newton-Schulz step, then what is left for Jacobi on the true matrix
    W = W @ (1.5 * np.eye(r) - 0.5 * (W.T @ W))
    Hf = W.T @ H0 @ W; Hf = (Hf + Hf.T) / 2
    print(f"   step {t}: after Newton-Schulz orth err {np.abs(W.T @ W - np.eye(r)).max():.1e}; rotations left for Jacobi {jacobi(Hf, thr)}")
    w, Wx = np.linalg.eigh(H0); V = (Wx[:, ::-1].T @ V1)
