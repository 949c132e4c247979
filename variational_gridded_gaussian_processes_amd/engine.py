"""Thin host wrapper over the C-ABI: one `Engine` = one vggp_ctx on one GPU.

PyTorch is used only as the array container (device memory, streams); every number comes out
of libvggp_hip.so.  All tensors handed to the engine must be CUDA(ROCm) float64 and contiguous.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import BASIS, KIND, Desc, Info, check


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()):
        raise TypeError("engine tensors must be contiguous float64 on the GPU")
    return t.data_ptr()


def _stream(device=None) -> int:
    """torch's current stream ON THE ENGINE'S DEVICE (not on torch's current device)."""
    return torch.cuda.current_stream(device).cuda_stream


def _dvec(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def _basis_m(basis: str, g: np.ndarray) -> int:
    """Number of inducing features of a dimension from its grid array."""
    if basis == "one":
        return 1
    if basis == "b0":
        return len(g) - 1
    if basis == "vff":
        return 2 * (len(g) - 3) + 1
    return len(g)


class Engine:
    """Owns a vggp_ctx.  See include/vggp.h for the contract of every call."""

    def __init__(self, device: Optional[int] = None, n_ranks: int = 1, rank: int = 0, unique_id: Optional[bytes] = None,
                 allreduce=None):
        """n_ranks > 1: this process is rank `rank` of a row-sharded job (one Engine per GPU).  `unique_id` (bytes from
        Engine.unique_id() on rank 0, distributed by the host program) gives the context its RCCL communicator; or
        `allreduce(numpy_view)` -- a Python function summing a float64 array over the ranks in place -- installs the host
        callback transport (the multi-rank rehearsal on one GPU, where RCCL refuses duplicate devices)."""
        if not torch.cuda.is_available():
            raise RuntimeError("variational_gridded_gaussian_processes_amd needs a gfx950 GPU: "
                               "torch.cuda.is_available() is False and there is no CPU path")
        self.lib = _lib.load()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.n_ranks, self.rank = int(n_ranks), int(rank)
        if unique_id is not None and len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.UNIQUE_ID_BYTES} bytes")
        h = C.c_void_p()
        uid = C.create_string_buffer(bytes(unique_id), _lib.UNIQUE_ID_BYTES) if unique_id is not None else None
        check(self.lib.vggp_create(C.byref(h), self.device_index, self.n_ranks, self.rank, C.cast(uid, C.c_void_p) if uid else None))
        self._h = h
        self._cb = None
        if allreduce is not None:
            def _trampoline(_user, buf, count, fn=allreduce):
                try:
                    fn(np.ctypeslib.as_array(buf, shape=(int(count),)))
                    return 0
                except Exception:          # never let a Python exception cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            self._cb = _lib.ALLREDUCE_FN(_trampoline)          # keep the thunk alive as long as the context
            check(self.lib.vggp_set_allreduce(self._h, self._cb, None))
        self.m1 = self.m2 = self.n1 = self.n2 = 0
        self.planned = False
        self._step_bufs = None
        self.plan_token = 0          # bumped by every plan(): a model sharing this engine re-plans when it is not the last planner

    @staticmethod
    def unique_id() -> bytes:
        """RCCL unique id for Engine(..., n_ranks, rank, unique_id): call on rank 0, ship the bytes to the other ranks."""
        buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        check(_lib.load().vggp_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    @property
    def transport(self) -> str:
        t = C.c_int()
        check(self.lib.vggp_comm_info(self._h, None, None, C.byref(t)))
        return {0: "none", 1: "rccl", 2: "callback"}[t.value]

    def comm_info(self) -> dict:
        """What the CONTEXT reports about its collective (vggp_comm_info): communicator size, this rank, transport."""
        n, r, t = C.c_int(), C.c_int(), C.c_int()
        check(self.lib.vggp_comm_info(self._h, C.byref(n), C.byref(r), C.byref(t)))
        return {"n_ranks": n.value, "rank": r.value, "transport": {0: "none", 1: "rccl", 2: "callback"}[t.value]}

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        """The context's sum all-reduce on a contiguous float64 GPU tensor, in place."""
        check(self.lib.vggp_allreduce(self._h, _ptr(t), t.numel(), _stream(self.device)))
        return t

    def close(self):
        if getattr(self, "_h", None):
            self.lib.vggp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- planning -------------------------------------------------------------------------------
    def plan(self, kind1: str, basis1: str, grid1, x1, kind2: str, basis2: str, grid2, x2,
             n_total: Optional[int] = None, warm_start: bool = False, b0_f32_kdelta: bool = False,
             block_jacobi: bool = False, scattered: bool = False) -> None:
        """grid_d: mesh (m+1 knots, basis 'b0'), inducing coordinates (m, 'points'), knot mesh (m, 'b1') or
        [a, b, omega_0 .. omega_M] (m = 2M + 1, 'vff'); x_d: the n_d unique (local) observation coordinates along d."""
        x1, x2 = _dvec(x1), _dvec(x2)
        g1 = _dvec(grid1) if basis1 != "one" else np.ones(1)
        g2 = _dvec(grid2) if basis2 != "one" else np.ones(1)
        m1, m2 = _basis_m(basis1, g1), _basis_m(basis2, g2)
        d = Desc()
        d.kind1, d.basis1, d.kind2, d.basis2 = KIND[kind1], BASIS[basis1], KIND[kind2], BASIS[basis2]
        d.n1, d.n2, d.m1, d.m2 = len(x1), len(x2), m1, m2
        d.n_total = int(n_total) if n_total is not None else len(x1) * len(x2)
        d.x1, d.x2 = x1.ctypes.data, x2.ctypes.data
        d.grid1, d.grid2 = g1.ctypes.data, g2.ctypes.data
        d.warm_start = 1 if warm_start else 0
        d.flags = (_lib.FLAG_B0_F32_KDELTA if b0_f32_kdelta else 0) | (_lib.FLAG_BLOCK_JACOBI if block_jacobi else 0) | \
                  (_lib.FLAG_SCATTERED if scattered else 0)
        if scattered:          # x1[k], x2[k]: the coordinates of point k (no grid)
            if len(x1) != len(x2):
                raise ValueError("scattered plan: x1 and x2 must hold one coordinate pair per point")
            d.n_total = int(n_total) if n_total is not None else len(x1)      # (point-sharded job: points over all ranks)
        self.scattered = bool(scattered)
        with torch.cuda.device(self.device):
            check(self.lib.vggp_plan(self._h, C.byref(d)))
        self.m1, self.m2, self.n1, self.n2 = m1, m2, len(x1), len(x2)
        self.payload_len = int(self.lib.vggp_payload_len(self._h))
        self.planned = True
        self.plan_token += 1

    @property
    def workspace_bytes(self) -> int:
        return int(self.lib.vggp_workspace_bytes(self._h))

    # -- hot path ---------------------------------------------------------------------------------
    def _check_Y(self, Y: torch.Tensor):
        if tuple(Y.shape) != (self.n2, self.n1):
            raise ValueError(f"Y must be [n2={self.n2}, n1={self.n1}] (x1 fastest), got {tuple(Y.shape)}")

    def elbo_step(self, Y: torch.Tensor, yy_total: float, theta: Sequence[float]):
        """-> (elbo, grad[5] wrt (ell1, ell2, s1, s2, sigma2), info dict)."""
        self._check_Y(Y)
        b = self._step_bufs           # (host argument blocks of the fit loop's call, allocated once: the call is made ~4000 times a second)
        if b is None:
            th, elbo, grad, info = (C.c_double * 5)(), C.c_double(), (C.c_double * 5)(), Info()
            b = self._step_bufs = (th, elbo, grad, info, C.byref(elbo), C.byref(info), np.frombuffer(grad, dtype=np.float64))
        th = b[0]
        th[0], th[1], th[2], th[3], th[4] = theta
        check(self.lib.vggp_elbo_step(self._h, _ptr(Y), float(yy_total), th, b[4], b[2], b[5], _stream(self.device)))
        return b[1].value, b[6].copy(), self._info(b[3])

    def elbo_step_masked(self, Ym: torch.Tensor, W: torch.Tensor, n_obs: float, yy_obs: float, theta: Sequence[float]):
        """Masked grid: Ym = W*Y and W (0/1 float64) are [n2, n1]; -> (elbo, grad[5], info)."""
        self._check_Y(Ym)
        self._check_Y(W)
        th = (C.c_double * 5)(*[float(t) for t in theta])
        elbo = C.c_double()
        grad = (C.c_double * 5)()
        info = Info()
        check(self.lib.vggp_elbo_step_masked(self._h, _ptr(Ym), _ptr(W), float(n_obs), float(yy_obs), th, C.byref(elbo),
                                             grad, C.byref(info), _stream(self.device)))
        return elbo.value, np.array(list(grad)), self._info(info)

    def elbo_step_masked_iter(self, Ym: torch.Tensor, W: torch.Tensor, n_obs: float, yy_obs: float, theta: Sequence[float],
                              n_probes: int = 16, tol: float = 1e-10, max_iter: int = 100):
        """The masked step without any M x M matrix (PCG + Lanczos quadrature + control-variate trace estimators, fixed probes;
        include/vggp.h): -> (elbo, grad[5], info) with info['rounds'][0] = PCG iterations."""
        self._check_Y(Ym)
        self._check_Y(W)
        th = (C.c_double * 5)(*[float(t) for t in theta])
        elbo = C.c_double()
        grad = (C.c_double * 5)()
        info = Info()
        check(self.lib.vggp_elbo_step_masked_iter(self._h, _ptr(Ym), _ptr(W), float(n_obs), float(yy_obs), th, int(n_probes), float(tol),
                                                  int(max_iter), C.byref(elbo), grad, C.byref(info), _stream(self.device)))
        return elbo.value, np.array(list(grad)), self._info(info)

    def elbo_step_scattered(self, y: torch.Tensor, yy: float, theta: Sequence[float]):
        """N scattered points (plan(..., scattered=True) with their coordinate pairs): y [N] float64 GPU tensor, yy = sum y^2;
        -> (elbo, grad[5], info).  qv_masked / posterior_masked / the *_cov_masked read-outs apply afterwards."""
        if not (y.is_cuda and y.dtype == torch.float64 and y.is_contiguous() and y.numel() == self.n1):
            raise TypeError("y must be a contiguous float64 GPU tensor with one value per planned point")
        th = (C.c_double * 5)(*[float(t) for t in theta])
        elbo = C.c_double()
        grad = (C.c_double * 5)()
        info = Info()
        check(self.lib.vggp_elbo_step_scattered(self._h, _ptr(y), float(yy), th, C.byref(elbo), grad, C.byref(info),
                                                _stream(self.device)))
        return elbo.value, np.array(list(grad)), self._info(info)

    def qv_masked(self) -> Tuple[torch.Tensor, torch.Tensor]:
        mean = torch.empty(self.m1, self.m2, dtype=torch.float64, device=self.device)
        var = torch.empty_like(mean)
        check(self.lib.vggp_qv_masked(self._h, _ptr(mean), _ptr(var), _stream(self.device)))
        return mean, var

    def elbo_partials(self, Y: torch.Tensor, theta: Sequence[float], payload: Optional[torch.Tensor] = None):
        self._check_Y(Y)
        if payload is None:
            payload = torch.empty(self.payload_len, dtype=torch.float64, device=self.device)
        th = (C.c_double * 5)(*[float(t) for t in theta])
        check(self.lib.vggp_elbo_partials(self._h, _ptr(Y), th, _ptr(payload), _stream(self.device)))
        return payload

    def elbo_finish(self, payload: torch.Tensor, yy_total: float, theta: Sequence[float]):
        th = (C.c_double * 5)(*[float(t) for t in theta])
        elbo = C.c_double()
        grad = (C.c_double * 5)()
        info = Info()
        check(self.lib.vggp_elbo_finish(self._h, _ptr(payload), float(yy_total), th, C.byref(elbo), grad,
                                        C.byref(info), _stream(self.device)))
        return elbo.value, np.array(list(grad)), self._info(info)

    @staticmethod
    def _info(i: Info) -> dict:
        return dict(jitter=(i.jitter1, i.jitter2), sweeps=(i.sweeps1, i.sweeps2), rounds=(i.rounds1, i.rounds2),
                    status=i.status, polished=(bool(i.polished & 1), bool(i.polished & 2)))

    def qv(self) -> Tuple[torch.Tensor, torch.Tensor]:
        mean = torch.empty(self.m1, self.m2, dtype=torch.float64, device=self.device)
        var = torch.empty_like(mean)
        check(self.lib.vggp_qv(self._h, _ptr(mean), _ptr(var), _stream(self.device)))
        return mean, var

    def qv_cov(self) -> torch.Tensor:
        M = self.m1 * self.m2
        cov = torch.empty(M, M, dtype=torch.float64, device=self.device)
        check(self.lib.vggp_qv_cov(self._h, _ptr(cov), _stream(self.device)))
        return cov

    def readout(self, C1: torch.Tensor, C2: torch.Tensor, kd1: torch.Tensor, kd2: torch.Tensor, literal: bool = True,
                masked: bool = False):
        """Gridded read-out q(v) of B0 cell features from the inducing posterior of the last step (include/vggp.h):
        C_d [mv_d, m_d] unit-outputscale cross-covariances, kd_d [mv_d] unit diagonals of Kvv_d -> mean, var [mv1, mv2]."""
        C1, C2 = C1.to(self.device, torch.float64).contiguous(), C2.to(self.device, torch.float64).contiguous()
        kd1, kd2 = kd1.to(self.device, torch.float64).contiguous(), kd2.to(self.device, torch.float64).contiguous()
        if C1.shape[1] != self.m1 or C2.shape[1] != self.m2:
            raise ValueError("C_d must be [mv_d, m_d]")
        mean = torch.empty(C1.shape[0], C2.shape[0], dtype=torch.float64, device=self.device)
        var = torch.empty_like(mean)
        fn = self.lib.vggp_readout_masked if masked else self.lib.vggp_readout      # masked: state of a masked / scattered step
        check(fn(self._h, _ptr(C1), C1.shape[0], _ptr(C2), C2.shape[0], _ptr(kd1), _ptr(kd2), _ptr(mean),
                 _ptr(var), 1 if literal else 0, _stream(self.device)))
        return mean, var

    def posterior_cov(self, x_star: torch.Tensor, masked: bool = False) -> torch.Tensor:
        """Dense covariance of posterior(x*) (kronecker_structure.py:223-229), x_star [ns, 2] -> [ns, ns]."""
        xs = x_star.to(self.device, torch.float64)
        if xs.dim() == 1:
            xs = torch.stack([xs, torch.zeros_like(xs)], dim=1)
        xs1, xs2 = xs[:, 0].contiguous(), xs[:, 1].contiguous()
        ns = xs1.shape[0]
        cov = torch.empty(ns, ns, dtype=torch.float64, device=self.device)
        fn = self.lib.vggp_posterior_cov_masked if masked else self.lib.vggp_posterior_cov
        check(fn(self._h, _ptr(xs1), _ptr(xs2), ns, _ptr(cov), _stream(self.device)))
        return cov

    def set_inducing(self, dim: int, z) -> None:
        """Move the inducing points of a 'points' basis (dimension 0 or 1) without re-planning: graphs and warm start stay."""
        zz = _dvec(z)
        with torch.cuda.device(self.device):
            check(self.lib.vggp_set_inducing(self._h, int(dim), zz.ctypes.data, len(zz)))

    def zgrad(self, Y: torch.Tensor):
        """d ELBO / d z of the last elbo_step(Y, ...) for the inducing coordinates of both dimensions ("points" bases; zeros
        otherwise) -> (g1 [m1], g2 [m2]) device tensors."""
        g1 = torch.empty(self.m1, dtype=torch.float64, device=self.device)
        g2 = torch.empty(self.m2, dtype=torch.float64, device=self.device)
        check(self.lib.vggp_zgrad(self._h, _ptr(Y), _ptr(g1), _ptr(g2), _stream(self.device)))
        return g1, g2

    def zgrad_scattered(self, y: torch.Tensor):
        """d ELBO / d z of the last elbo_step_scattered(y, ...) -> (g1 [m1], g2 [m2]) device tensors."""
        g1 = torch.empty(self.m1, dtype=torch.float64, device=self.device)
        g2 = torch.empty(self.m2, dtype=torch.float64, device=self.device)
        check(self.lib.vggp_zgrad_scattered(self._h, _ptr(y), _ptr(g1), _ptr(g2), _stream(self.device)))
        return g1, g2

    def qv_cov_masked(self) -> torch.Tensor:
        M = self.m1 * self.m2
        cov = torch.empty(M, M, dtype=torch.float64, device=self.device)
        check(self.lib.vggp_qv_cov_masked(self._h, _ptr(cov), _stream(self.device)))
        return cov

    def posterior_masked(self, x_star: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """posterior(x*) of the last masked step; x_star [ns, 2] -> mean[ns], var[ns]."""
        return self.posterior(x_star, _fn="vggp_posterior_masked")

    def posterior(self, x_star: torch.Tensor, _fn: str = "vggp_posterior") -> Tuple[torch.Tensor, torch.Tensor]:
        """x_star [ns, 2] (or [ns] with a trivial second dimension) -> mean[ns], var[ns]."""
        xs = x_star.to(self.device, torch.float64)
        if xs.dim() == 1:
            xs = torch.stack([xs, torch.zeros_like(xs)], dim=1)
        xs1, xs2 = xs[:, 0].contiguous(), xs[:, 1].contiguous()
        ns = xs1.shape[0]
        mean = torch.empty(ns, dtype=torch.float64, device=self.device)
        var = torch.empty_like(mean)
        check(getattr(self.lib, _fn)(self._h, _ptr(xs1), _ptr(xs2), ns, _ptr(mean), _ptr(var), _stream(self.device)))
        return mean, var

    # -- building blocks ------------------------------------------------------------------------------
    def factor_build(self, kind: str, basis: str, x: torch.Tensor, grid: torch.Tensor, ell: float, flags: int = 0):
        """-> A0[m,n], dA0[m,n], K0[m,m], dK0[m,m] at unit outputscale."""
        n = x.shape[0]
        m = _basis_m(basis, np.empty(grid.shape[0]))
        o = dict(dtype=torch.float64, device=self.device)
        A, dA, K, dK = torch.empty(m, n, **o), torch.empty(m, n, **o), torch.empty(m, m, **o), torch.empty(m, m, **o)
        check(self.lib.vggp_factor_build(self._h, KIND[kind], BASIS[basis], _ptr(x), n, _ptr(grid), m, float(ell), int(flags),
                                         _ptr(A), _ptr(dA), _ptr(K), _ptr(dK), _stream(self.device)))
        return A, dA, K, dK

    def cholesky_inverse(self, K: torch.Tensor):
        m = K.shape[0]
        L, Li = torch.empty_like(K), torch.empty_like(K)
        jit = C.c_double()
        check(self.lib.vggp_cholesky_inverse(self._h, _ptr(K), m, _ptr(L), _ptr(Li), C.byref(jit), _stream(self.device)))
        return L, Li, jit.value

    def eigh(self, G: torch.Tensor, block: bool = False):
        """-> lam[m], Qt[m,m] (row j = eigenvector j), sweeps."""
        m = G.shape[0]
        lam = torch.empty(m, dtype=torch.float64, device=self.device)
        Qt = torch.empty_like(G)
        sw = C.c_int32()
        check(self.lib.vggp_eigh(self._h, _ptr(G), m, _ptr(lam), _ptr(Qt), C.byref(sw),
                                 _lib.FLAG_BLOCK_JACOBI if block else 0, _stream(self.device)))
        return lam, Qt, sw.value

    def gemm(self, A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
        """C = A @ B for 2-D float64 GPU tensors with arbitrary strides (no copies are made)."""
        M, K = A.shape
        K2, N = B.shape
        assert K == K2
        Cm = torch.empty(M, N, dtype=torch.float64, device=self.device)
        for t in (A, B):
            if not (t.is_cuda and t.dtype == torch.float64):
                raise TypeError("gemm operands must be float64 GPU tensors")
        check(self.lib.vggp_gemm(self._h, A.data_ptr(), A.stride(0), A.stride(1), B.data_ptr(), B.stride(0),
                                 B.stride(1), Cm.data_ptr(), N, M, N, K, _stream(self.device)))
        return Cm

    def kron_solve(self, L1: torch.Tensor, L2: torch.Tensor, Y: torch.Tensor) -> torch.Tensor:
        """X = K1^{-1} Y K2^{-T} with K_d = L_d L_d^T from the Cholesky factors (four triangular solves by substitution)."""
        n1, n2 = Y.shape
        X = torch.empty_like(Y)
        check(self.lib.vggp_kron_solve(self._h, _ptr(L1), n1, _ptr(L2), n2, _ptr(Y), _ptr(X), _stream(self.device)))
        return X

    def trsm(self, L: torch.Tensor, R: torch.Tensor, trans: bool = False) -> torch.Tensor:
        """Solve L X = R (or L^T X = R) for lower-triangular L [m, m] and R [m, ncols] by substitution."""
        m, ncols = R.shape
        X = torch.empty_like(R)
        check(self.lib.vggp_trsm(self._h, _ptr(L), m, _ptr(R), ncols, _ptr(X), 1 if trans else 0, _stream(self.device)))
        return X

    def profile(self, enable: bool = True) -> None:
        check(self.lib.vggp_profile(self._h, 1 if enable else 0))

    def profile_read(self, reset: bool = True):
        """-> ({stage name: accumulated ms}, steps) measured with HIP events on the launch stream."""
        ms = (C.c_double * _lib.NSTAGE)()
        steps = C.c_int32()
        check(self.lib.vggp_profile_read(self._h, ms, C.byref(steps), 1 if reset else 0))
        names = [self.lib.vggp_stage_name(i).decode() for i in range(_lib.NSTAGE)]
        return dict(zip(names, list(ms))), steps.value

    def project_kernel_name(self) -> str:
        return self.lib.vggp_project_kernel_name().decode()

    def sumsq(self, y: torch.Tensor) -> float:
        out = C.c_double()
        check(self.lib.vggp_sumsq(self._h, _ptr(y), y.numel(), C.byref(out), _stream(self.device)))
        return out.value
