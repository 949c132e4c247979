"""CPU study (numpy): the orthonormalisation of Z = V G in the warm thin chain -- row-by-row Gram-Schmidt (what vg_rowqr_kernel does)
against a symmetric (Loewdin / Newton-Schulz) orthonormalisation of the row-normalised Z, which has no sequential row loop.
Measures what the chain needs: the miss quantity (tr G - sum theta) / lam_max and the error of the Ritz values.
usage: lowdin_rowqr_study.py [relative lengthscale step] [r] [NS iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense as D, kron as Kr
dl = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
r = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nit = int(sys.argv[3]) if len(sys.argv) > 3 else 5
n, m = 1024, 128
x = D.gen_grid(n, 4)[2]
f = Kr.Factor("points", "rbf", np.linspace(0, 1, m), x)
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0)
    return d.B @ d.B.T
G0, G1 = gram(0.2), gram(0.2 * (1 + dl))
l0, Q0 = np.linalg.eigh(G0); V = Q0[:, ::-1][:, :r].T            # previous range basis, rows sorted by decreasing eigenvalue
lt = np.linalg.eigvalsh(G1)[::-1]
Z = V @ G1
def gs(Z):
    V1 = np.zeros_like(Z)
    for i in range(Z.shape[0]):
        v = Z[i].copy()
        for _ in range(2):
            v -= V1[:i].T @ (V1[:i] @ v)
        V1[i] = v / np.linalg.norm(v)
    return V1
def lowdin(Z, nit):
    X = Z / np.linalg.norm(Z, axis=1)[:, None]
    for it in range(nit):
        W = X @ X.T
        print(f"      NS it {it}: ||X X^T - I||_max = {np.abs(W - np.eye(len(W))).max():.2e}")
        X = 1.5 * X - 0.5 * W @ X
    return X
def judge(V1, tag):
    H = V1 @ G1 @ V1.T
    th = np.linalg.eigvalsh((H + H.T) / 2)[::-1]
    print(f"{tag}: orth err {np.abs(V1 @ V1.T - np.eye(r)).max():.1e}   miss (tr G - sum theta)/lam_max = {(np.trace(G1) - th.sum()) / lt[0]:.2e}"
          f"   max |theta_k - lam_k| / lam_max = {np.abs(th - lt[:r]).max() / lt[0]:.2e}   (true tail beyond r: {lt[r:].sum() / lt[0]:.1e})")
print("row norms of Z / lam_max:", np.array2string(np.linalg.norm(Z, axis=1) / lt[0], precision=1))
judge(gs(Z), "Gram-Schmidt (CGS2)   ")
judge(lowdin(Z, nit), "normalise + NS        ")
# hybrid: GS for nothing, but blocks: normalise, then NS
