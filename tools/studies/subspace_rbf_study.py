"""RBF (numerical rank ~20 of 128): one step of subspace iteration + Rayleigh-Ritz on the r leading eigenvectors of the
previous step, complement = previous null rows projected off the new leading rows and re-orthonormalised.
How block-diagonal is Q G Q^T afterwards (cross block, null block against m thr), and what is left for Jacobi?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'jacobi_ordering.py')).read().split("for kind in")[0])
f = Kr.Factor("points", "rbf", np.linspace(0, 1, m), np.linspace(0, 1, 1024))
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0); return d.B @ d.B.T
G0 = gram(0.2); lam, Q = np.linalg.eigh(G0); Qp = Q[:, ::-1].T.copy(); lamp = lam[::-1]
for t in range(1, 5):
    G = gram(0.2 * 1.01 ** t); nG = np.linalg.norm(G); thr = 1e-13 * nG / m
    for cut in (1e-14, 1e-15):
        r = int((lamp > cut * np.linalg.norm(lamp)).sum())
        r = min(m, ((r + 7) // 8) * 8)
        V = Qp[:r]
        Z = V @ G
        Z = Z / np.linalg.norm(Z, axis=1)[:, None]
        # Gram-Schmidt in the given (descending) order == Cholesky-QR of the normalised rows
        V1 = np.linalg.qr(Z.T)[0].T                               # rows orthonormal (Householder: tolerates the nearly parallel edge rows)
        H = V1 @ G @ V1.T; H = (H + H.T) / 2
        w, W = np.linalg.eigh(H); W = W[:, ::-1]; V2 = W.T @ V1
        N = Qp[r:] - (Qp[r:] @ V2.T) @ V2
        N = N - (N @ V2.T) @ V2                                    # second projection (twice is enough)
        N = np.linalg.qr(N.T)[0].T
        Qn = np.vstack([V2, N])
        R = Qn @ G @ Qn.T; R = (R + R.T) / 2
        off = R - np.diag(np.diag(R))
        TT = np.abs(off[:r, :r]).max() / thr; TN = np.abs(off[:r, r:]).max() / thr
        NNF = np.linalg.norm(off[r:, r:]) / (m * thr); NNd = np.abs(np.diag(R)[r:]).max() / nG
        orth = np.abs(Qn @ Qn.T - np.eye(m)).max()
        rots = jacobi(R, thr, 6)
        print(f"step {t} cut {cut:.0e} r {r}: max|off|/thr top-top {TT:.1e} top-null {TN:.1e}; null block ||.||_F/(m thr) {NNF:.2f} max diag/||G|| {NNd:.1e}; orth {orth:.1e}; Jacobi left {rots}")
    lam, Q = np.linalg.eigh(G); Qp = Q[:, ::-1].T.copy(); lamp = lam[::-1]
