#!/usr/bin/env python3
"""Regenerates the committed golden fixtures (run in the build container; /root/reference optional).

1. oracle_*.npz -- inputs + expected outputs of the dense float64 restatement (oracle/dense.py) for the
   cases of SURVEY.md section 8c.  PARITY UNPINNED: the reference's model classes need gpytorch (absent), so these
   vectors come from the build's own restatement, not from running the reference.
2. ref_pins.npz -- outputs of the reference modules that DO import here (torch/numpy/scipy only):
   src/utils/datagenerators.gen_2d (point ordering), src/basis/bspline.B0SplineBasis (mesh bookkeeping),
   src/utils/integrators.integrate_1d (the quad known-answer check).  Data only -- no reference source.
3. ref_pins_basis.npz -- outputs of the reference's importable inducing-feature bases at fixed inputs:
   src/basis/fourier.FourierBasisMatern12(M, a, b, ell)(x)  (fourier.py:58-88; float32 arithmetic in the reference) and
   src/basis/bspline.B1SplineBasis(mesh)(x)  (bspline.py:106-112).  They pin the oracle's vff/b1 Kuf builders and the
   HIP vggp_factor_build(VGGP_BASIS_VFF / VGGP_BASIS_B1) A-factors on reference-produced numbers.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dense as D  # noqa: E402

CASES = {
    # name: (n1, n2, basis, kind, grid1, grid2, theta, mesh_dtype)
    "b0_m12_f32mesh_16x12": (16, 12, "b0", "matern12", ("lin", 0, 1, 8), ("lin", 0, 1, 6), None, "f32"),
    "b0_m12_32x32": (32, 32, "b0", "matern12", ("lin", 0, 1, 11), ("lin", 0, 1, 9), [0.2, 0.3, 1.0, 0.8, 0.01], "f64"),
    "pts_m12_16x12": (16, 12, "points", "matern12", ("lin", 0, 1, 9), ("lin", 0, 1, 7), [0.2, 0.3, 1.0, 0.8, 0.01], "f64"),
    "pts_m32_32x32": (32, 32, "points", "matern32", ("lin", 0, 1, 12), ("lin", 0, 1, 10), [0.25, 0.2, 1.2, 0.9, 0.0025], "f64"),
    "pts_m52_20x24": (20, 24, "points", "matern52", ("lin", 0, 1, 8), ("lin", 0, 1, 9), [0.3, 0.25, 0.7, 1.1, 0.01], "f64"),
    "pts_rbf_32x32": (32, 32, "points", "rbf", ("lin", 0, 1, 8), ("lin", 0, 1, 8), [0.2, 0.2, 1.0, 1.0, 0.0025], "f64"),
}


def grid(spec, dtype):
    _, a, b, n = spec
    if dtype == "f32":
        return torch.linspace(a, b, n)             # the reference's float32 mesh
    return torch.tensor(np.linspace(a, b, n))


def make_case(name, spec):
    n1, n2, basis, kind, g1s, g2s, theta, mdt = spec
    X, y, x1, x2 = D.gen_grid(n1, n2, seed=abs(hash(name)) % 1000 if False else len(name))
    g1, g2 = grid(g1s, mdt), grid(g2s, mdt)
    raw = torch.zeros(5, dtype=torch.float64) if theta is None else D.raw_from_constrained(theta)
    dm = D.DenseKron(X, y, basis, kind, g1, g2, raw=raw)
    elbo, graw = dm.elbo_and_grad()
    qv = dm.q_v()
    xs = np.random.default_rng(11).uniform(0, 1, (25, 2))
    po = dm.posterior(xs)
    np.savez(os.path.join(HERE, f"oracle_{name}.npz"),
             X=X, y=y, x1=x1, x2=x2, grid1=g1.double().numpy(), grid2=g2.double().numpy(),
             mesh_is_f32=np.array(mdt == "f32"), basis=np.array(basis), kind=np.array(kind),
             raw=raw.numpy(), theta=dm.theta().detach().numpy(), jitter=np.array(dm.jitters()),
             elbo=elbo.numpy(), grad_raw=graw.numpy(),
             qv_mean=qv.mean.detach().numpy(), qv_var=qv.variance.detach().numpy(),
             xs=xs, post_mean=po.mean.detach().numpy(), post_var=po.variance.detach().numpy())
    print(name, float(elbo))


NEW_BASIS_CASES = {
    # the reference's VFF and B1-spline models (only the per-dimension factors differ)
    "vff_m12_24x20": (24, 20, "vff", (-0.1, 1.1, 5), [0.25, 0.2, 1.0, 1.2, 0.01]),
    "b1_m12_24x20": (24, 20, "b1", 8, [0.25, 0.2, 1.0, 1.2, 0.01]),
}


def make_new_basis_case(name, spec):
    n1, n2, basis, gs, theta = spec
    X, y, x1, x2 = D.gen_grid(n1, n2, seed=len(name))
    raw = D.raw_from_constrained(theta)
    if basis == "vff":
        a, b, M = gs
        dg = (a, b, M)
        kgrid = np.concatenate([[a, b], D.vff_omegas(M, a, b).double().numpy()])
    else:
        mesh = torch.tensor(np.linspace(0, 1, gs))      # float64 mesh: the reference's float32 delta (1e-8 relative) is not reproduced
        dg = mesh
        kgrid = mesh.double().numpy()
    dm = D.DenseKron(X, y, basis, "matern12", dg, dg, raw=raw)
    elbo, graw = dm.elbo_and_grad()
    qv = dm.q_v()
    xs = np.random.default_rng(11).uniform(0, 1, (25, 2))
    po = dm.posterior(xs)
    np.savez(os.path.join(HERE, f"oracle_{name}.npz"),
             X=X, y=y, x1=x1, x2=x2, grid1=kgrid, grid2=kgrid, mesh_is_f32=np.array(False), basis=np.array(basis),
             kind=np.array("matern12"), raw=raw.numpy(), theta=dm.theta().detach().numpy(), jitter=np.array(dm.jitters()),
             elbo=elbo.numpy(), grad_raw=graw.numpy(), qv_mean=qv.mean.detach().numpy(), qv_var=qv.variance.detach().numpy(),
             xs=xs, post_mean=po.mean.detach().numpy(), post_var=po.variance.detach().numpy())
    print(name, float(elbo))


MASK_CASES = {
    # BASELINE.json configs[4]: missing observations under a mask (the reference receives only the observed subset)
    "mask30_b0_m12_32x32": (32, 32, "b0", "matern12", ("lin", 0, 1, 9), ("lin", 0, 1, 9), [0.25, 0.2, 1.0, 1.2, 0.01], "f64", 0.3),
    "mask30_pts_m32_24x20": (24, 20, "points", "matern32", ("lin", 0, 1, 10), ("lin", 0, 1, 8), [0.3, 0.25, 0.9, 1.2, 0.01], "f64", 0.3),
}


def make_mask_case(name, spec):
    n1, n2, basis, kind, g1s, g2s, theta, mdt, frac = spec
    X, y, x1, x2 = D.gen_grid(n1, n2, seed=len(name))
    W = np.random.default_rng(5).uniform(size=(n2, n1)) > frac          # observed grid points
    g1, g2 = grid(g1s, mdt), grid(g2s, mdt)
    raw = D.raw_from_constrained(theta)
    dm = D.DenseKron(X, y, basis, kind, g1, g2, raw=raw, mask=W.reshape(-1))
    elbo, graw = dm.elbo_and_grad()
    qv = dm.q_v()
    np.savez(os.path.join(HERE, f"oracle_{name}.npz"),
             X=X, y=y, x1=x1, x2=x2, W=W, grid1=g1.double().numpy(), grid2=g2.double().numpy(),
             mesh_is_f32=np.array(False), basis=np.array(basis), kind=np.array(kind),
             raw=raw.numpy(), theta=dm.theta().detach().numpy(), jitter=np.array(dm.jitters()),
             elbo=elbo.numpy(), grad_raw=graw.numpy(),
             qv_mean=qv.mean.detach().numpy(), qv_var=qv.variance.detach().numpy())
    print(name, float(elbo))


def make_1d():
    n, nknots = 256, 33
    x = np.linspace(0, 2 * np.pi, n)
    y = np.sin(x) + np.cos(x) + 0.05 * np.random.default_rng(0).standard_normal(n)
    mesh = torch.tensor(np.linspace(0, 2 * np.pi, nknots))
    dm = D.Dense1D(x, y, "b0", "matern12", mesh, raw=torch.tensor([0.3, -0.2, -2.0], dtype=torch.float64))
    e, g = dm.elbo_and_grad()
    qv = dm.q_v()
    xs = np.linspace(0.1, 6.0, 17)
    po = dm.posterior(xs)
    np.savez(os.path.join(HERE, "oracle_1d_b0_256.npz"), x=x, y=y, mesh=mesh.numpy(), raw=dm.raw.detach().numpy(),
             theta=dm.theta().detach().numpy(), elbo=e.numpy(), grad_raw=g.numpy(), qv_mean=qv.mean.detach().numpy(),
             qv_var=qv.variance.detach().numpy(), xs=xs, post_mean=po.mean.detach().numpy(),
             post_var=po.variance.detach().numpy())
    print("1d", float(e))


def make_ref_pins():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("no /root/reference: ref_pins.npz left as committed")
        return
    sys.path.insert(0, ref)
    from src.basis.bspline import B0SplineBasis
    from src.utils.datagenerators import gen_2d
    from src.utils.integrators import integrate_1d
    X, yv = gen_2d(D.latent_2d, (0.0, 1.0), (-1.0, 2.0), 5)
    mesh = torch.linspace(0, 1, 11)
    b = B0SplineBasis(mesh)
    f = lambda t: np.sin(t) + np.cos(t)
    m1 = np.linspace(0, 2 * np.pi, 33)
    areas, errs = integrate_1d(f, m1)
    np.savez(os.path.join(HERE, "ref_pins.npz"), gen2d_X=X, gen2d_y=yv, b0_mesh=mesh.numpy(), b0_m=np.array(b.m),
             b0_delta=np.array(float(b.delta)), b0_nbasis=np.array(b.n_basis_functions), quad_mesh=m1,
             quad_areas=areas)
    print("ref pins written")


def make_ref_pins_basis():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        print("no /root/reference: ref_pins_basis.npz left as committed")
        return
    sys.path.insert(0, ref)
    from src.basis.bspline import B1SplineBasis
    from src.basis.fourier import FourierBasisMatern12
    rng = np.random.default_rng(3)
    out = {}
    # VFF: points inside [a, b), on both boundaries, and outside on either side (the exp(-r/ell) tails)
    for tag, (M, a, b, ell) in {"vff_a": (5, -0.1, 1.1, 0.25), "vff_b": (12, 0.0, 2.0, 0.6931)}.items():
        x = np.concatenate([rng.uniform(a, b, 40), [a, b, a - 0.3, a - 0.01, b + 0.02, b + 0.5, 0.5 * (a + b)]])
        fb = FourierBasisMatern12(M, a, b, ell)
        Phi = fb(torch.tensor(x))                 # (2M+1) x n; float32 omegas promote against the float64 points
        out[tag + "_M"], out[tag + "_a"], out[tag + "_b"], out[tag + "_ell"] = np.array(M), np.array(a), np.array(b), np.array(ell)
        out[tag + "_x"] = x
        out[tag + "_omegas"] = fb.omegas.numpy()
        out[tag + "_Phi"] = Phi.double().numpy()
        out[tag + "_dtype"] = np.array(str(Phi.dtype))
    # B1: interior points, every knot, both ends, and points outside the mesh
    for tag, (lo, hi, nk, dt) in {"b1_f64": (0.0, 1.0, 8, torch.float64), "b1_f32": (-1.0, 2.0, 11, torch.float32)}.items():
        mesh = torch.linspace(lo, hi, nk, dtype=dt)
        x = np.concatenate([rng.uniform(lo, hi, 40), mesh.double().numpy(), [lo - 0.1, hi + 0.1]])
        Phi = B1SplineBasis(mesh)(torch.tensor(x))
        out[tag + "_mesh"] = mesh.numpy()
        out[tag + "_x"] = x
        out[tag + "_Phi"] = Phi.double().numpy()
    np.savez(os.path.join(HERE, "ref_pins_basis.npz"), **out)
    print("ref basis pins written:", {k: v.shape for k, v in out.items() if k.endswith("_Phi")})


if __name__ == "__main__":
    if "--basis-pins-only" in sys.argv:
        make_ref_pins_basis()
        sys.exit(0)
    for k, v in CASES.items():
        make_case(k, v)
    for k, v in MASK_CASES.items():
        make_mask_case(k, v)
    for k, v in NEW_BASIS_CASES.items():
        make_new_basis_case(k, v)
    make_1d()
    make_ref_pins()
    make_ref_pins_basis()
