"""CPU study (numpy): how far is the previous step's eigenbasis from diagonalising the next step's Gram matrix for the feature
families the warm chains do not carry (B1 hats on a padded mesh, 127 Fourier features), and which pairs defeat the first-order
rotation E_ij = g_ij / (g_ii - g_jj)?   usage: family_warm_study.py [b1|vff|m32] [relative step of the lengthscale]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense as D, kron as Kr
fam = sys.argv[1] if len(sys.argv) > 1 else "b1"
dl = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
n, m = 1024, 128
x = D.gen_grid(n, 4)[2]
if fam == "b1":
    pad = 8; d = 1.0 / (m - 1 - 2 * pad)
    f = Kr.Factor("b1", "matern12", np.linspace(-pad * d, 1 + pad * d, m), x)
elif fam == "vff":
    M = 63; f = Kr.Factor("vff", "matern12", np.concatenate([[-0.1, 1.1], np.arange(M + 1) * 2 * np.pi / 1.2]), x)
else:
    f = Kr.Factor("points", "matern32", np.linspace(0, 1, m), x)
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0)
    return d.B @ d.B.T
ell = 0.2
G0, G1, G2 = gram(ell), gram(ell * (1 + dl)), gram(ell * (1 + 2 * dl))
lam0, Q0 = np.linalg.eigh(G0); lam1, Q1 = np.linalg.eigh(G1)
print("spectrum: max %.3e, min %.3e; eigenvalues below 1e-13 max: %d; smallest relative gap %.2e" % (
    lam1.max(), lam1.min(), (lam1 < 1e-13 * lam1.max()).sum(), np.min(np.diff(lam1)[lam1[1:] > 1e-12 * lam1.max()] / lam1.max())))
def report(Gw, tag):
    dg = np.diag(Gw); off = Gw - np.diag(dg)
    thr = 1e-13 * np.linalg.norm(Gw) / m
    with np.errstate(divide="ignore", invalid="ignore"):
        E = np.where(np.abs(off) > thr, off / (dg[:, None] - dg[None, :]), 0.0)
    E[np.isnan(E)] = 0
    big = np.argwhere(np.abs(E) > 1e-3)
    print(f"{tag}: max |offdiag| / ||G|| = {np.abs(off).max() / np.linalg.norm(Gw):.2e}; pairs with |E| > 1e-3: {len(big) // 2}, > 0.3: {(np.abs(E) > 0.3).sum() // 2}, max |E| = {np.abs(E).max():.2e}")
    for i, j in big[:6]:
        if i < j: print(f"     pair ({i},{j}): g_ii {dg[i]:.3e} g_jj {dg[j]:.3e} g_ij {off[i, j]:.3e} E {E[i, j]:.2e}")
report(Q0.T @ G1 @ Q0, "previous basis")
# extrapolated basis: Q1 from Q0 and the step before (first order in the rotation generator)
lamm, Qm = np.linalg.eigh(gram(ell * (1 - dl)))
# align signs / order
def align(Qa, Qb):
    S = Qa.T @ Qb
    idx = np.argmax(np.abs(S), axis=1)
    Qb2 = Qb[:, idx] * np.sign(S[np.arange(m), idx])
    return Qb2
Q0a = align(Qm, Q0)
U = Qm.T @ Q0a                      # rotation between the two previous bases
Qx = Q0a @ U                        # extrapolation
Qx, _ = np.linalg.qr(Qx)
report(Qx.T @ G1 @ Qx, "extrapolated basis")
