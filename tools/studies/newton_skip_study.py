"""CPU emulation (numpy) of the Newton chain on the Gram matrix of one dimension: extrapolated start from the two previous bases,
iterations E_ij = g_ij / (g_jj - g_ii) with large quotients SKIPPED, S <- (I + E + E^2/2) S, Newton-Schulz, and the largest
off-diagonal element against the chain's acceptance threshold after every iteration.
usage: newton_skip_study.py [b1|vff|m52] [relative lengthscale step] [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import dense as D, kron as Kr
fam = sys.argv[1] if len(sys.argv) > 1 else "b1"
dl = float(sys.argv[2]) if len(sys.argv) > 2 else 0.002
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n, m = 1024, 128
x = D.gen_grid(n, 4)[2]
if fam == "b1":
    pad = 8; d = 1.0 / (m - 1 - 2 * pad)
    f = Kr.Factor("b1", "matern12", np.linspace(-pad * d, 1 + pad * d, m), x)
elif fam == "vff":
    M = 63; f = Kr.Factor("vff", "matern12", np.concatenate([[-0.1, 1.1], np.arange(M + 1) * 2 * np.pi / 1.2]), x); m = 127
else:
    f = Kr.Factor("points", "matern52", np.linspace(0, 1, m), x)
def gram(ell):
    d = Kr.dim_prepare(f, ell, 1.0)
    return d.B @ d.B.T
ell = 0.2
Gs = [gram(ell * (1 + dl) ** k) for k in range(3)]
Qa = np.linalg.eigh(Gs[0])[1].T          # rows = eigenvectors
Qb = np.linalg.eigh(Gs[1])[1].T
def newton(S, G, tag, emax=0.3, noise=1e-13, skip=True, nullset=None, cluster=None, cluster_its=1):
    for it in range(iters + 1):
        Gw = S @ G @ S.T
        dg = np.diag(Gw).copy(); off = Gw - np.diag(dg)
        acc = 1e-11 * np.linalg.norm(Gw) / m
        nfl = noise * np.abs(dg).max()
        live = ~((np.abs(dg)[:, None] <= nfl) & (np.abs(dg)[None, :] <= nfl))
        if nullset is not None: live &= ~(nullset[:, None] & nullset[None, :])
        live_rot = live.copy()
        if cluster is not None and it < cluster_its: live_rot &= ~(cluster[:, None] & cluster[None, :])
        worst = np.abs(off * live).max()
        i, j = np.unravel_index(np.argmax(np.abs(off * live)), off.shape)
        print(f"  {tag} it {it}: max offdiag / accept = {worst / acc:9.2e}  at ({i},{j}) g_ii {dg[i]:.2e} g_jj {dg[j]:.2e}   orth err {np.abs(S @ S.T - np.eye(m)).max():.1e}")
        if it == iters: break
        thr = 1e-12 * np.linalg.norm(Gw) / m
        with np.errstate(divide="ignore", invalid="ignore"):
            E = np.where((np.abs(off) > thr) & live_rot, off / (dg[:, None] - dg[None, :]), 0.0)   # E_ij = g_ij / (g_ii - g_jj), skew
        E[~np.isfinite(E)] = 0.0
        nbig = (np.abs(E) > emax).sum() // 2
        if skip: E[np.abs(E) > emax] = 0.0
        S = (np.eye(m) + E + E @ E / 2) @ S
        S = 1.5 * S - 0.5 * (S @ S.T) @ S
        print(f"       pairs skipped {nbig}")
    return S
U = Qb @ Qa.T
print("previous basis as the start:"); newton(Qb.copy(), Gs[2], "prev")
print("extrapolated start U Qb:"); S0 = U @ Qb; newton(S0, Gs[2], "extr")

# static null set: rows whose eigenvalue was numerically zero in the previous step (sorted ascending by eigh: the first ones)
lamb = np.linalg.eigvalsh(Gs[1])
nullset = lamb <= 1e-12 * lamb.max()
print("static null set of", nullset.sum(), "rows:")
newton(Qb.copy(), Gs[2], "prev+nullset", nullset=nullset)
newton(U @ Qb, Gs[2], "extr+nullset", nullset=nullset)

cluster = lamb <= 1e-7 * lamb.max()
print("cluster of", cluster.sum(), "rows (<= 1e-7 of the largest) deferred for the first iteration(s):")
for ci in (1, 2):
    newton(Qb.copy(), Gs[2], f"prev+cluster{ci}", nullset=nullset, cluster=cluster, cluster_its=ci)
    newton(U @ Qb, Gs[2], f"extr+cluster{ci}", nullset=nullset, cluster=cluster, cluster_its=ci)
