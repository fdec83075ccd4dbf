"""ctypes front-end of the CPU oracle (oracle/gp_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gp_oracle.c.  Imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
Parity: the covariance functions are pinned bit-exactly to the reference's own classes (oracle/_ref,
tests/golden/ref_kernels.npz); the Eigen-dependent rest is PARITY UNPINNED (no golden vectors in the reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

GAUSSIAN, LAPLACE, THINPLATE, MATERN32, MATERN52 = range(5)
KERNEL_IDS = {"gaussian": GAUSSIAN, "laplace": LAPLACE, "thinplate": THINPLATE,
              "matern32": MATERN32, "matern52": MATERN52}
DIST_DIRECT, DIST_EXPANSION = 0, 1


class OrcKernel(C.Structure):
    _fields_ = [("id", C.c_int), ("p0", C.c_double), ("p1", C.c_double)]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    targets = [os.path.join(_HERE, n) for n in ("libgp_oracle.so", "libgp_oracle_omp.so")]
    src = os.path.join(_HERE, "gp_oracle.c")
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


_libs = {}


def _lib(omp=False):
    key = bool(omp)
    if key in _libs:
        return _libs[key]
    path = os.path.join(_HERE, "libgp_oracle_omp.so" if omp else "libgp_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    kp = C.POINTER(OrcKernel)
    L.orc_k.restype = C.c_double
    L.orc_k.argtypes = [kp, C.c_double]
    L.orc_kdiff.restype = C.c_double
    L.orc_kdiff.argtypes = [kp, C.c_double]
    L.orc_kdiffdiff.restype = C.c_double
    L.orc_kdiffdiff.argtypes = [kp, C.c_double]
    L.orc_dist_matrix.restype = None
    L.orc_dist_matrix.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, dp, dp, dp, dp, dp]
    L.orc_ldlt.restype = C.c_int
    L.orc_ldlt.argtypes = [C.c_int, dp, ip]
    L.orc_ldlt_solve.restype = None
    L.orc_ldlt_solve.argtypes = [C.c_int, dp, ip, C.c_int, dp]
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [kp, C.c_int, dp, dp, dp, dp, dp, C.c_int, C.c_int]
    L.orc_update.restype = C.c_int
    L.orc_update.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    L.orc_evaluate.restype = None
    L.orc_evaluate.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp]
    L.orc_project.restype = None
    L.orc_project.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, C.c_double, C.c_double, C.c_int, C.c_double, dp,
                              dp, ip, ip]
    L.orc_march_surface.restype = C.c_long
    L.orc_march_surface.argtypes = [C.c_void_p, dp, C.c_double, C.c_double, C.c_double, C.c_long, C.c_long, dp, dp, dp,
                                    C.POINTER(C.c_long)]
    L.orc_evaluate_fullcov.restype = None
    L.orc_evaluate_fullcov.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    L.orc_tangent_basis.restype = None
    L.orc_tangent_basis.argtypes = [dp, dp, dp, dp]
    L.orc_free.restype = None
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_n.restype = C.c_int
    L.orc_n.argtypes = [C.c_void_p]
    L.orc_R.restype = C.c_double
    L.orc_R.argtypes = [C.c_void_p]
    L.orc_ldlt_info.restype = C.c_int
    L.orc_ldlt_info.argtypes = [C.c_void_p]
    L.orc_get_alpha.argtypes = [C.c_void_p, dp]
    L.orc_get_Kpp.argtypes = [C.c_void_p, dp]
    L.orc_get_ldlt.argtypes = [C.c_void_p, dp, ip]
    L.orc_get_normals.restype = C.c_int
    L.orc_get_normals.argtypes = [C.c_void_p, dp]
    L.orc_dgp_create.restype = C.c_void_p
    L.orc_dgp_create.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, dp, dp, dp, dp, dp]
    L.orc_dgp_free.argtypes = [C.c_void_p]
    L.orc_dgp_info.restype = C.c_int
    L.orc_dgp_info.argtypes = [C.c_void_p]
    L.orc_dgp_loglik.restype = C.c_double
    L.orc_dgp_loglik.argtypes = [C.c_void_p]
    L.orc_dgp_get_alpha.argtypes = [C.c_void_p, dp]
    L.orc_dgp_get_K.argtypes = [C.c_void_p, dp]
    L.orc_dgp_evaluate.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    if omp:
        # never oversubscribe: a GPU box exposes many more hardware threads than its CPU share
        try:
            avail = len(os.sched_getaffinity(0))
        except Exception:
            avail = os.cpu_count() or 1
        L.orc_set_num_threads(max(1, min(avail, int(os.environ.get("GP_ORACLE_THREADS", "8")))))
    _libs[key] = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def make_kernel(name, *params):
    """make_kernel('gaussian', sigma, length) / ('thinplate', R) / ('matern52', sigma, length)."""
    kid = KERNEL_IDS[name] if isinstance(name, str) else int(name)
    p = list(params) + [1.0] * (2 - len(params))
    return OrcKernel(kid, float(p[0]), float(p[1]))


def k(kern, d, omp=False):
    L = _lib(omp)
    return np.array([L.orc_k(C.byref(kern), float(x)) for x in np.atleast_1d(d)])


def kdiff(kern, d, omp=False):
    L = _lib(omp)
    return np.array([L.orc_kdiff(C.byref(kern), float(x)) for x in np.atleast_1d(d)])


def kdiffdiff(kern, d):
    L = _lib()
    return np.array([L.orc_kdiffdiff(C.byref(kern), float(x)) for x in np.atleast_1d(d)])


def dist_matrix(A, B, mode=DIST_DIRECT):
    """A: (m,3), B: (n,3) -> (m,n) distances (gp_regressor.hpp:548-557)."""
    L = _lib()
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64)
    m, n = len(A), len(B)
    D = np.empty((n, m), dtype=np.float64)  # column-major m x n
    ax, pax = _d(A[:, 0]); ay, pay = _d(A[:, 1]); az, paz = _d(A[:, 2])
    bx, pbx = _d(B[:, 0]); by, pby = _d(B[:, 1]); bz, pbz = _d(B[:, 2])
    L.orc_dist_matrix(mode, m, n, pax, pay, paz, pbx, pby, pbz, D.ctypes.data_as(C.POINTER(C.c_double)))
    return D.T.copy()


def ldlt(A):
    """Eigen-style LDLT of symmetric A. Returns (packed lower/diag matrix, transpositions, info)."""
    L = _lib()
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    t = np.zeros(n, dtype=np.int32)
    info = L.orc_ldlt(n, A.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_int)))
    return A, t, info


def ldlt_solve(F, t, b):
    L = _lib()
    F = np.asfortranarray(F, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    one = b.ndim == 1
    B = np.array(b.reshape(len(b), -1), dtype=np.float64, order="F")
    t = np.ascontiguousarray(t, dtype=np.int32)
    L.orc_ldlt_solve(F.shape[0], F.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_int)),
                     B.shape[1], B.ctypes.data_as(C.POINTER(C.c_double)))
    return B[:, 0].copy() if one else np.ascontiguousarray(B)


def tangent_basis(g):
    L = _lib()
    g, pg = _d(g)
    N = np.zeros(3); Tx = np.zeros(3); Ty = np.zeros(3)
    dp = C.POINTER(C.c_double)
    L.orc_tangent_basis(pg, N.ctypes.data_as(dp), Tx.ctypes.data_as(dp), Ty.ctypes.data_as(dp))
    return N, Tx, Ty


class Model:
    """gp_regression::Model + GPRegressor<Cov>::{create,evaluate,update} (oracle side)."""

    def __init__(self, kern, x, y, z, label, sigma2=None, with_normals=False, dist_mode=DIST_DIRECT, omp=False):
        self._L = _lib(omp)
        self.kern = kern
        x, px = _d(x); y, py = _d(y); z, pz = _d(z); lab, pl = _d(label)
        if sigma2 is not None and len(sigma2):
            s2, ps = _d(sigma2)
        else:
            ps = None
        self._h = self._L.orc_create(C.byref(kern), len(x), px, py, pz, pl, ps, int(with_normals), int(dist_mode))

    def __del__(self):
        try:
            if self._h:
                self._L.orc_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def n(self):
        return self._L.orc_n(self._h)

    @property
    def R(self):
        return self._L.orc_R(self._h)

    @property
    def ldlt_info(self):
        return self._L.orc_ldlt_info(self._h)

    @property
    def alpha(self):
        a = np.empty(self.n)
        self._L.orc_get_alpha(self._h, a.ctypes.data_as(C.POINTER(C.c_double)))
        return a

    @property
    def Kpp(self):
        n = self.n
        a = np.empty((n, n))
        self._L.orc_get_Kpp(self._h, a.ctypes.data_as(C.POINTER(C.c_double)))
        return a.T.copy()

    def ldlt(self):
        n = self.n
        a = np.empty((n, n))
        t = np.empty(n, dtype=np.int32)
        self._L.orc_get_ldlt(self._h, a.ctypes.data_as(C.POINTER(C.c_double)), t.ctypes.data_as(C.POINTER(C.c_int)))
        return a.T.copy(), t

    @property
    def normals(self):
        a = np.empty((self.n, 3))
        rc = self._L.orc_get_normals(self._h, a.ctypes.data_as(C.POINTER(C.c_double)))
        return a if rc == 0 else None

    def update(self, x, y, z, label, sigma2=None):
        x, px = _d(x); y, py = _d(y); z, pz = _d(z); lab, pl = _d(label)
        if sigma2 is not None and len(sigma2):
            s2, ps = _d(sigma2)
        else:
            ps = None
        return self._L.orc_update(self._h, len(x), px, py, pz, pl, ps)

    def evaluate(self, qx, qy, qz, want_v=False, want_grad=False, want_basis=False):
        """Returns dict(f=..., v=..., grad=..., tx=..., ty=...)."""
        qx, px = _d(qx); qy, py = _d(qy); qz, pz = _d(qz)
        nq = len(qx)
        dp = C.POINTER(C.c_double)
        f = np.empty(nq)
        out = {"f": f}
        v = np.empty(nq) if want_v else None
        g = np.empty((nq, 3)) if (want_grad or want_basis) else None
        tx = np.empty((nq, 3)) if want_basis else None
        ty = np.empty((nq, 3)) if want_basis else None
        ptr = lambda a: a.ctypes.data_as(dp) if a is not None else None
        self._L.orc_evaluate(self._h, nq, px, py, pz, ptr(f), ptr(v), ptr(g), ptr(tx), ptr(ty))
        if v is not None:
            out["v"] = v
        if g is not None:
            out["grad"] = g
        if tx is not None:
            out["tx"] = tx
            out["ty"] = ty
        return out

    def project(self, x, y, z, normal, f_tol=1e-2, improve_tol=1e-7, max_iter=500, step_mul=0.001):
        """AtlasBase::project (atlas.hpp:201-276) for every start point; returns dict(xyz, f, iter, status)."""
        x, px = _d(x); y, py = _d(y); z, pz = _d(z)
        n = len(x)
        nrm = np.ascontiguousarray(np.asarray(normal, dtype=np.float64).reshape(n, 3))
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        out = np.empty((n, 3)); f = np.empty(n)
        it = np.empty(n, dtype=np.int32); st = np.empty(n, dtype=np.int32)
        self._L.orc_project(self._h, n, px, py, pz, nrm.ctypes.data_as(dp), f_tol, improve_tol, int(max_iter),
                            step_mul, out.ctypes.data_as(dp), f.ctypes.data_as(dp), it.ctypes.data_as(ip),
                            st.ctypes.data_as(ip))
        return {"xyz": out, "f": f, "iter": it, "status": st}

    def march_surface(self, leaf, step, f_tol=0.01, start=None, max_cubes=1 << 20, capacity=1 << 20):
        """marchingSampling / marchingCubes (src/gp_node.cpp:1102-1291): dict(xyz, f, v, n_total, n_cubes)."""
        dp = C.POINTER(C.c_double)
        xyz = np.empty((capacity, 3)); f = np.empty(capacity); v = np.empty(capacity)
        nc = C.c_long(0)
        sp = None
        if start is not None:
            st = np.ascontiguousarray(start, dtype=np.float64)
            sp = st.ctypes.data_as(dp)
        n = self._L.orc_march_surface(self._h, sp, float(leaf), float(step), float(f_tol), int(max_cubes), int(capacity),
                                      xyz.ctypes.data_as(dp), f.ctypes.data_as(dp), v.ctypes.data_as(dp), C.byref(nc))
        k = max(0, min(int(n), capacity))
        return {"xyz": xyz[:k], "f": f[:k], "v": v[:k], "n_total": int(n), "n_cubes": int(nc.value)}

    def evaluate_fullcov(self, qx, qy, qz):
        qx, px = _d(qx); qy, py = _d(qy); qz, pz = _d(qz)
        nq = len(qx)
        dp = C.POINTER(C.c_double)
        v = np.empty(nq)
        V = np.empty((nq, nq))
        self._L.orc_evaluate_fullcov(self._h, nq, px, py, pz, v.ctypes.data_as(dp), V.ctypes.data_as(dp))
        return v, V.T.copy()


def num_threads(omp=True):
    return _lib(omp).orc_num_threads()


def set_num_threads(t):
    _lib(True).orc_set_num_threads(int(t))


SE = 5  # gp::CovSE of the reference's second library (include/gp/CovSE.h)


class DerivativeGP:
    """gp::GaussianProcess restated (oracle/gp_oracle.c, second half): a GP trained on values and gradients.
    kernel: ('se', sf, ell) or ('thinplate', R)."""

    def __init__(self, kernel, noise, x, y, z, target, normals=None, omp=False):
        self._L = _lib(omp)
        kid = SE if kernel[0] == "se" else THINPLATE
        p0 = float(kernel[1])
        p1 = float(kernel[2]) if len(kernel) > 2 else 1.0
        x, px = _d(x)
        y, py = _d(y)
        z, pz = _d(z)
        t, pt = _d(target)
        self.n = len(x)
        pn = None
        if normals is not None:
            nr, pn = _d(np.asarray(normals, dtype=np.float64).reshape(-1))
        self._h = C.c_void_p(self._L.orc_dgp_create(kid, p0, p1, float(noise), self.n, px, py, pz, pt, pn))
        self.info = int(self._L.orc_dgp_info(self._h))
        self._kernel, self._noise = (kernel[0], p0, p1), float(noise)
        self._pts = np.stack([x, y, z], axis=1)
        self._yv = np.concatenate([t, nr if normals is not None else np.zeros(3 * self.n)])

    def __del__(self):
        try:
            if self._h:
                self._L.orc_dgp_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def alpha(self):
        a = np.zeros(4 * self.n)
        self._L.orc_dgp_get_alpha(self._h, a.ctypes.data_as(C.POINTER(C.c_double)))
        return a

    @property
    def K(self):
        k = np.zeros((4 * self.n, 4 * self.n))
        self._L.orc_dgp_get_K(self._h, k.ctypes.data_as(C.POINTER(C.c_double)))
        return k

    @property
    def loglik(self):
        return float(self._L.orc_dgp_loglik(self._h))

    def dK_dlogl(self):
        """d K / d log(l) of the 4n x 4n covariance of a CovSE model, block by block: CovSE::grad's d k / d log(l) = k z with
        z = r^2 / l^2 (include/gp/CovSE.h:96-101) carried through the derivative blocks of compute() (GaussianProcess.h:545-567)
        -- g = -k / l^2 -> g (z - 2), h = k / l^4 -> h (z - 4).  Plain NumPy; checked against central differences of K in
        tests/test_oracle.py."""
        assert self._kernel[0] == "se"
        sf, l = self._kernel[1], self._kernel[2]
        n, P = self.n, self._pts
        U = P[:, None, :] - P[None, :, :]          # u = x_a - x_b
        r2 = (U ** 2).sum(-1)
        k = sf * sf * np.exp(-0.5 * r2 / l ** 2)
        g, h, z = -k / l ** 2, k / l ** 4, r2 / l ** 2
        D = np.zeros((4 * n, 4 * n))
        D[:n, :n] = k * z
        for d in range(3):
            D[n + d::3, :n] = g * (z - 2) * U[:, :, d]           # cov(d_d f(x_a), f(x_b)) = g u_d
            D[:n, n + d::3] = -g * (z - 2) * U[:, :, d]          # cov(f(x_a), d_e f(x_b)) = -g u_e
            for e in range(3):
                D[n + d::3, n + e::3] = -(g * (z - 2) if d == e else 0.0) - h * (z - 4) * U[:, :, d] * U[:, :, e]
        return D

    def loglik_gradient(self):
        """logLikelihoodGradient (include/gp/GaussianProcess.h:387-410) in CovSE's log hyper-parameters (log l, log sf)
        (getLogHyper's order, CovSE.h:108-118): W = alpha alpha^T - K^-1 (:398-400), grad_j = 1/2 sum_ab W_ab dK_ab / d theta_j
        (:402-408 -- the lower triangle with the diagonal halved is that sum), over the whole 4n x 4n matrix, i.e. the
        gradient of logLikelihood() :376-385 as this oracle defines it (the reference mixes n x n and 4n: it does not
        build).  d K / d log(sf) = 2 (K - sn^2 I) (CovSE::grad's 2 k).  tests/test_oracle.py holds it to central differences
        of loglik."""
        K = self.K
        a = self.alpha
        W = np.outer(a, a) - np.linalg.inv(K)
        dl = self.dK_dlogl()
        ds = 2.0 * (K - self._noise ** 2 * np.eye(4 * self.n))
        return np.array([0.5 * np.sum(W * dl), 0.5 * np.sum(W * ds)])

    def evaluate(self, qx, qy, qz, want_v=True):
        qx, px = _d(qx)
        qy, py = _d(qy)
        qz, pz = _d(qz)
        nq = len(qx)
        f4 = np.zeros((nq, 4))
        v = np.zeros(nq) if want_v else None
        self._L.orc_dgp_evaluate(self._h, nq, px, py, pz, f4.ctypes.data_as(C.POINTER(C.c_double)),
                                 v.ctypes.data_as(C.POINTER(C.c_double)) if want_v else None)
        out = {"f": f4[:, 0].copy(), "grad": f4[:, 1:].copy()}
        if want_v:
            out["v"] = v
        return out


RPROP_DEFAULT = dict(delta0=0.1, delta_min=1e-6, delta_max=50.0, eta_minus=0.5, eta_plus=1.2, eps_stop=1e-4, max_iter=100)


def rprop_find(sf, l, noise, x, y, z, target, normals=None, **desc):
    """Optimisation::find (include/gp/GaussianProcess.h:86-122) restated statement by statement on a CovSE DerivativeGP,
    in the log hyper-parameters p = (log l, log sf); Desc defaults :64-73.  Returns the best parameters, their likelihood,
    the number of applied steps and the (params, lik) trace.  A step onto parameters whose matrix is not positive definite
    ends the search (the reference's llt() would carry NaNs on)."""
    d = dict(RPROP_DEFAULT, **desc)
    sign = lambda v: 1.0 if v > 0 else (-1.0 if v < 0 else 0.0)
    make = lambda p: DerivativeGP(("se", float(np.exp(p[1])), float(np.exp(p[0]))), noise, x, y, z, target, normals)
    delta = np.ones(2) * d["delta0"]                      # :89
    grad_old = np.zeros(2)                                # :90
    params = np.array([np.log(l), np.log(sf)])            # :91 getLogHyper
    best_params, best = params.copy(), -np.inf            # :92-93
    gp = make(params)
    start_lik = gp.loglik
    trace, done = [], 0
    for _ in range(int(d["max_iter"])):                   # :96
        grad = -gp.loglik_gradient()                      # :97
        grad_old = grad_old * grad                        # :98
        for j in range(2):                                # :99-108
            if grad_old[j] > 0:
                delta[j] = min(delta[j] * d["eta_plus"], d["delta_max"])
            elif grad_old[j] < 0:
                delta[j] = max(delta[j] * d["eta_minus"], d["delta_min"])
                grad[j] = 0.0
            params[j] += -sign(grad[j]) * delta[j]
        grad_old = grad.copy()                            # :109
        if np.linalg.norm(grad_old) < d["eps_stop"]:      # :110
            break
        gp = make(params)                                 # :111 setLogHyper, :112 logLikelihood() -> compute()
        if gp.info != 0:
            break
        done += 1
        lik = gp.loglik
        trace.append((params.copy(), lik))
        if lik > best:                                    # :116-119
            best, best_params = lik, params.copy()
    return {"loghyper": best_params, "loglik": best if done else start_lik, "iterations": done, "trace": trace}
