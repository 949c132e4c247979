"""Why B = L^-1 A is formed by blocked substitution and not as (explicit inverse) x A  (DESIGN.md section 2).

numpy float64 on the CPU: an RBF factor of the 4096-point axis with 128 inducing points (cond(K + 1e-8 I) ~ 1e10), the three
ways of forming B and the Gram matrix G = B B^T next to a float128 reference, plus the effect on the posterior-variance
kernel  diag(L Q diag(1/(1 + lam lam'/s2)) Q^T L^T)  of one dimension.
"""
import numpy as np
import scipy.linalg as sl


def factor(n=4096, m=128, ell=0.3, eps=1e-8):
    x = (np.arange(n) + 0.5) / n
    z = np.linspace(0.0, 1.0, m)
    k = lambda a, b: np.exp(-0.5 * ((a[:, None] - b[None, :]) / ell) ** 2)
    return k(z, z) + eps * np.eye(m), k(z, x)


def blocked_substitution(L, A, nb=16):
    m = L.shape[0]
    X = np.empty_like(A)
    for i in range(0, m, nb):
        T = A[i:i + nb] - L[i:i + nb, :i] @ X[:i]
        X[i:i + nb] = np.linalg.inv(L[i:i + nb, i:i + nb]) @ T        # 16x16 diagonal-block inverse, as the kernel does
    return X


def main():
    K, A = factor()
    L = np.linalg.cholesky(K)
    Lq = L.astype(np.longdouble)
    # long-double substitution as the reference
    Bq = np.empty(A.shape, dtype=np.longdouble)
    for i in range(L.shape[0]):
        Bq[i] = (A[i].astype(np.longdouble) - Lq[i, :i] @ Bq[:i]) / Lq[i, i]
    Gq = (Bq @ Bq.T).astype(np.float64)
    B_inv = np.linalg.inv(L) @ A
    B_sub = sl.solve_triangular(L, A, lower=True)
    B_blk = blocked_substitution(L, A)
    print(f"cond(L) = {np.linalg.cond(L):.2e}")
    for name, B in (("inverse x A", B_inv), ("scalar substitution", B_sub), ("16-block substitution", B_blk)):
        G = B @ B.T
        lam, Q = np.linalg.eigh(G)
        lamq, Qq = np.linalg.eigh(Gq)
        d, dq = 1.0 / (1.0 + lam * 1e3 / 0.01), 1.0 / (1.0 + lamq * 1e3 / 0.01)
        var = np.einsum("ij,j,ij->i", L @ Q, d, L @ Q)
        varq = np.einsum("ij,j,ij->i", L @ Qq, dq, L @ Qq)
        print(f"{name:24s} |B - Bref|/|B| = {np.abs(B - Bq).max() / np.abs(Bq).max():.2e}   "
              f"|G - Gref|/|G| = {np.abs(G - Gq).max() / np.abs(Gq).max():.2e}   "
              f"variance rel err = {np.abs(var - varq).max() / np.abs(varq).max():.2e}")


if __name__ == "__main__":
    main()
