"""ctypes binding of libgpx.so (include/gpx.h) for tests and bench.

This is plumbing over the C ABI, not a compute path: every numeric call goes to the HIP library and
raises GpxError when the library or a GPU is missing -- there is no Python / NumPy fallback.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPX_LIB") or os.path.join(_HERE, "lib", "libgpx.so")  # GPX_LIB: A/B runs of library variants

GAUSSIAN, LAPLACE, THINPLATE, MATERN32, MATERN52, SE = range(6)
KERNEL_IDS = {"gaussian": GAUSSIAN, "laplace": LAPLACE, "thinplate": THINPLATE,
              "matern32": MATERN32, "matern52": MATERN52, "se": SE}
F32, F64, MIXED, F32_SPLIT = 0, 1, 2, 3

OK = 0
E_NULL, E_EMPTY, E_LABELED_QUERY, E_SIZE_MISMATCH, E_SINGULAR, E_NAN_INPUT, E_HIP, E_OOM, E_NO_DEVICE, \
    E_BAD_ARG, E_STATE = (-1, -2, -3, -4, -5, -6, -7, -8, -9, -10, -11)

(FIELD_N, FIELD_R, FIELD_ALPHA, FIELD_P, FIELD_Y, FIELD_S2, FIELD_NORMALS, FIELD_STATS, FIELD_D, FIELD_PERM,
 FIELD_KPP) = range(11)


class GpxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("gpx error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


class Kernel(C.Structure):
    _fields_ = [("id", C.c_int32), ("reserved", C.c_int32), ("p", C.c_double * 4)]


class Options(C.Structure):
    _fields_ = [("precision", C.c_int32), ("device", C.c_int32), ("with_normals", C.c_int32),
                ("ir_steps", C.c_int32), ("prepare_variance", C.c_int32), ("query_batch", C.c_int32),
                ("reserved", C.c_int32 * 6)]


class ProjectOptions(C.Structure):
    _fields_ = [("f_tol", C.c_double), ("improve_tol", C.c_double), ("step_mul", C.c_double),
                ("max_iter", C.c_int32), ("reserved", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("t_kbuild_ms", C.c_double), ("t_factor_ms", C.c_double), ("t_solve_ms", C.c_double),
                ("t_inverse_ms", C.c_double), ("t_normals_ms", C.c_double), ("t_mean_ms", C.c_double),
                ("t_var_ms", C.c_double), ("t_var_gemm_ms", C.c_double), ("t_factor_gemm_ms", C.c_double),
                ("n", C.c_int64), ("n_padded", C.c_int64), ("n_negative_pivots", C.c_int64),
                ("ir_steps_done", C.c_int64), ("alpha_residual", C.c_double),
                ("var_gemm_launches", C.c_int64), ("factor_gemm_launches", C.c_int64),
                ("solve_fallbacks", C.c_int64), ("t_var_kqp_ms", C.c_double), ("factor_gemm_flops", C.c_double),
                ("surface_candidates", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


# every symbol include/gpx.h declares
EXPORTS = [
    "gpx_last_error", "gpx_version", "gpx_device_count", "gpx_model_create", "gpx_model_update",
    "gpx_model_evaluate", "gpx_model_evaluate_device", "gpx_model_sample_surface", "gpx_model_project",
    "gpx_model_prepare_variance", "gpx_model_get",
    "gpx_model_sync", "gpx_model_destroy", "gpx_model_create_shell", "gpx_model_state_blob", "gpx_model_commit",
    "gpx_model_replicate", "gpx_slab_range", "gpx_model_evaluate_sharded", "gpx_model_sample_surface_sharded", "gpx_trim", "gpx_debug_reload", "gpx_model_march_surface",
    "gpx_dev_kbuild", "gpx_dev_kqp", "gpx_dev_kqp_f32", "gpx_padded_n", "gpx_dgp_create", "gpx_dgp_evaluate", "gpx_dgp_get",
    "gpx_dgp_add", "gpx_dgp_destroy", "gpx_dgp_loglik_gradient", "gpx_rprop_default", "gpx_dgp_optimise", "gpx_pcd_read", "gpx_node_training_set",
]

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, found through $ORIGIN).
    If libgpx.so pulled in /opt/rocm's copy first, a later `import torch` would load a SECOND HIP runtime
    into the process and see no GPU.  Loading torch's copy first makes both resolve to one runtime,
    whatever the import order.  Without torch installed the system runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load libgpx.so (raises if the HIP extension has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GpxError(E_STATE, "libgpx.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`"
                       % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.gpx_last_error.restype = C.c_char_p
    L.gpx_version.restype = C.c_char_p
    L.gpx_device_count.restype = C.c_int
    L.gpx_padded_n.restype = C.c_size_t
    L.gpx_padded_n.argtypes = [C.c_size_t]
    L.gpx_model_create.restype = C.c_int
    L.gpx_model_create.argtypes = [C.POINTER(Kernel), C.c_size_t, dp, dp, dp, dp, dp, C.POINTER(Options),
                                   C.POINTER(vp)]
    L.gpx_model_update.restype = C.c_int
    L.gpx_model_update.argtypes = [vp, C.c_size_t, dp, dp, dp, dp, dp]
    L.gpx_model_evaluate.restype = C.c_int
    L.gpx_model_evaluate.argtypes = [vp, C.c_size_t, dp, dp, dp, dp, dp, dp, dp, dp]
    L.gpx_model_evaluate_device.restype = C.c_int
    L.gpx_model_evaluate_device.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.gpx_model_sample_surface.restype = C.c_int
    L.gpx_model_sample_surface.argtypes = [vp, C.c_size_t, dp, dp, dp, C.c_double, C.c_size_t,
                                           C.POINTER(C.c_int64), dp, dp, C.POINTER(C.c_size_t)]
    L.gpx_slab_range.restype = None
    L.gpx_slab_range.argtypes = [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.gpx_model_evaluate_sharded.restype = C.c_int
    L.gpx_model_evaluate_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_size_t, dp, dp, dp, dp, dp, dp, dp, dp]
    L.gpx_model_sample_surface_sharded.restype = C.c_int
    L.gpx_model_sample_surface_sharded.argtypes = [C.POINTER(vp), C.c_int, C.c_size_t, dp, dp, dp, C.c_double, C.c_size_t,
                                                   C.POINTER(C.c_int64), dp, dp, C.POINTER(C.c_size_t)]
    L.gpx_model_project.restype = C.c_int
    L.gpx_model_project.argtypes = [vp, C.c_size_t, dp, dp, dp, dp, C.POINTER(ProjectOptions), dp, dp,
                                    C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.gpx_model_prepare_variance.restype = C.c_int
    L.gpx_model_prepare_variance.argtypes = [vp]
    L.gpx_model_get.restype = C.c_int
    L.gpx_model_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.gpx_model_sync.restype = C.c_int
    L.gpx_model_sync.argtypes = [vp]
    L.gpx_model_destroy.restype = None
    L.gpx_model_destroy.argtypes = [vp]
    L.gpx_model_create_shell.restype = C.c_int
    L.gpx_model_create_shell.argtypes = [C.POINTER(Kernel), C.c_size_t, C.POINTER(Options), C.POINTER(vp)]
    L.gpx_model_state_blob.restype = C.c_int
    L.gpx_model_state_blob.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.gpx_model_commit.restype = C.c_int
    L.gpx_model_commit.argtypes = [vp, C.c_int]
    L.gpx_model_march_surface.restype = C.c_int
    L.gpx_model_march_surface.argtypes = [vp, dp, C.c_double, C.c_double, C.c_double, C.c_size_t, C.c_size_t, dp, dp, dp,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.gpx_trim.restype = None
    L.gpx_debug_reload.restype = None
    L.gpx_model_replicate.restype = C.c_int
    L.gpx_model_replicate.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    L.gpx_dev_kbuild.restype = C.c_int
    L.gpx_dev_kbuild.argtypes = [C.POINTER(Kernel), C.c_int, C.c_size_t, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.gpx_dev_kqp.restype = C.c_int
    L.gpx_dev_kqp.argtypes = [C.POINTER(Kernel), C.c_int, C.c_size_t, C.c_size_t, vp, vp, vp, C.c_size_t, vp, vp, vp, vp,
                              vp, vp]
    L.gpx_dev_kqp_f32.restype = C.c_int
    L.gpx_dev_kqp_f32.argtypes = [C.POINTER(Kernel), C.c_size_t, C.c_size_t, vp, vp, vp, vp, C.c_size_t, vp, vp, vp, vp, vp,
                                  vp]
    L.gpx_dgp_create.restype = C.c_int
    L.gpx_dgp_create.argtypes = [C.POINTER(Kernel), C.c_double, C.c_size_t, dp, dp, dp, dp, dp, C.POINTER(Options), C.POINTER(vp)]
    L.gpx_dgp_evaluate.restype = C.c_int
    L.gpx_dgp_evaluate.argtypes = [vp, C.c_size_t, dp, dp, dp, dp, dp]
    L.gpx_dgp_add.restype = C.c_int
    L.gpx_dgp_add.argtypes = [vp, C.c_size_t, dp, dp, dp, dp, dp]
    L.gpx_dgp_get.restype = C.c_int
    L.gpx_dgp_get.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.gpx_dgp_destroy.restype = None
    L.gpx_dgp_destroy.argtypes = [vp]
    L.gpx_dgp_loglik_gradient.restype = C.c_int
    L.gpx_dgp_loglik_gradient.argtypes = [vp, dp]
    L.gpx_rprop_default.restype = None
    L.gpx_rprop_default.argtypes = [vp]
    L.gpx_dgp_optimise.restype = C.c_int
    L.gpx_dgp_optimise.argtypes = [vp, vp, vp]
    L.gpx_pcd_read.restype = C.c_long
    L.gpx_pcd_read.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_size_t]
    L.gpx_node_training_set.restype = C.c_int
    L.gpx_node_training_set.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_double, C.c_double, dp, dp, dp, dp, dp]
    _lib = L
    return L


def _check(rc):
    if rc != OK:
        raise GpxError(rc, lib().gpx_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().gpx_device_count()


def trim():
    """Free the pool of parked large device buffers (gpx_trim)."""
    lib().gpx_trim()


def debug_reload():
    """gpx_debug_reload: parse the GPX_* switches from the environment again (the library reads them once per process)."""
    lib().gpx_debug_reload()


class switches:
    """with gpx.switches(GPX_DATAFLOW="0", ...): set the named switches, reload; restore the environment and reload on exit
    (the test suites hold every twin path to the default one this way).  A value of None unsets the variable."""

    def __init__(self, **kv):
        self.kv, self.old = kv, {}

    def __enter__(self):
        for k, v in self.kv.items():
            assert k.startswith("GPX_"), k
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        debug_reload()
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        debug_reload()
        return False


def make_kernel(name, *params):
    kid = KERNEL_IDS[name] if isinstance(name, str) else int(name)
    k = Kernel()
    k.id = kid
    vals = list(params) + [1.0] * (2 - len(params))
    for i, v in enumerate(vals[:4]):
        k.p[i] = float(v)
    return k


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _as_d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Model:
    """gp_regression::Model + GPRegressor<Cov>::{create, evaluate, update} over the C ABI."""

    def __init__(self, kernel, x, y, z, label, sigma2=None, precision=F32, with_normals=False, ir_steps=-1,
                 prepare_variance=False, query_batch=0, device=-1, _shell_n=None):
        self._L = lib()
        self._h = C.c_void_p(None)
        self.kernel = kernel
        self.precision = precision
        opt = Options()
        opt.precision = int(precision)
        opt.device = int(device)
        opt.with_normals = int(bool(with_normals))
        opt.ir_steps = int(ir_steps)
        opt.prepare_variance = int(bool(prepare_variance))
        opt.query_batch = int(query_batch)
        if _shell_n is not None:
            _check(self._L.gpx_model_create_shell(C.byref(kernel), int(_shell_n), C.byref(opt), C.byref(self._h)))
            return
        x, y, z, label = _as_d(x), _as_d(y), _as_d(z), _as_d(label)
        if not (len(x) == len(y) == len(z) == len(label)):
            raise GpxError(E_SIZE_MISMATCH, "coordinate / label length mismatch")
        s2p = None
        if sigma2 is not None and len(sigma2):
            sigma2 = _as_d(sigma2)
            if len(sigma2) != len(x):
                raise GpxError(E_SIZE_MISMATCH, "sigma2 length mismatch")
            s2p = _dptr(sigma2)
        _check(self._L.gpx_model_create(C.byref(kernel), len(x), _dptr(x), _dptr(y), _dptr(z), _dptr(label), s2p,
                                        C.byref(opt), C.byref(self._h)))

    @classmethod
    def shell(cls, kernel, n, precision=F32, query_batch=0, device=-1):
        return cls(kernel, None, None, None, None, precision=precision, query_batch=query_batch, device=device,
                   _shell_n=n)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.gpx_model_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- accessors ----
    def _get(self, field, arr):
        _check(self._L.gpx_model_get(self._h, field, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return arr

    @property
    def n(self):
        return int(self._get(FIELD_N, np.zeros(1, dtype=np.int64))[0])

    @property
    def R(self):
        return float(self._get(FIELD_R, np.zeros(1))[0])

    @property
    def alpha(self):
        return self._get(FIELD_ALPHA, np.zeros(self.n))

    @property
    def P(self):
        return self._get(FIELD_P, np.zeros((self.n, 3)))

    @property
    def Y(self):
        return self._get(FIELD_Y, np.zeros(self.n))

    @property
    def S2(self):
        return self._get(FIELD_S2, np.zeros(self.n))

    @property
    def D(self):
        return self._get(FIELD_D, np.zeros(self.n))

    @property
    def perm(self):
        return self._get(FIELD_PERM, np.zeros(self.n, dtype=np.int32))

    @property
    def normals(self):
        return self._get(FIELD_NORMALS, np.zeros((self.n, 3)))

    @property
    def Kpp(self):
        n = self.n
        return self._get(FIELD_KPP, np.zeros((n, n)))

    @property
    def stats(self):
        s = Stats()
        _check(self._L.gpx_model_get(self._h, FIELD_STATS, C.byref(s), C.sizeof(s)))
        return s.as_dict()

    # ---- operations ----
    def update(self, x, y, z, label, sigma2=None):
        x, y, z, label = _as_d(x), _as_d(y), _as_d(z), _as_d(label)
        s2p = None
        if sigma2 is not None and len(sigma2):
            sigma2 = _as_d(sigma2)
            s2p = _dptr(sigma2)
        _check(self._L.gpx_model_update(self._h, len(x), _dptr(x), _dptr(y), _dptr(z), _dptr(label), s2p))

    def prepare_variance(self):
        _check(self._L.gpx_model_prepare_variance(self._h))

    def sync(self):
        _check(self._L.gpx_model_sync(self._h))

    def evaluate(self, qx, qy, qz, want_v=False, want_grad=False, want_basis=False, label=None):
        """Host arrays in, dict of host arrays out (f [, v, grad, tx, ty])."""
        if label is not None and len(label):
            raise GpxError(E_LABELED_QUERY, "Query is already labeled!")
        qx, qy, qz = _as_d(qx), _as_d(qy), _as_d(qz)
        nq = len(qx)
        f = np.empty(nq)
        v = np.empty(nq) if want_v else None
        g = np.empty((nq, 3)) if want_grad else None
        tx = np.empty((nq, 3)) if want_basis else None
        ty = np.empty((nq, 3)) if want_basis else None
        p = lambda a: _dptr(a) if a is not None else None
        _check(self._L.gpx_model_evaluate(self._h, nq, _dptr(qx), _dptr(qy), _dptr(qz), p(f), p(v), p(g), p(tx),
                                          p(ty)))
        out = {"f": f}
        if v is not None:
            out["v"] = v
        if g is not None:
            out["grad"] = g
        if tx is not None:
            out["tx"], out["ty"] = tx, ty
        return out

    def sample_surface(self, qx, qy, qz, f_tol=0.01, capacity=None, want_v=True):
        """Batched fakeDeterministicSampling: indices, f and v of the queries with |f| <= f_tol."""
        qx, qy, qz = _as_d(qx), _as_d(qy), _as_d(qz)
        nq = len(qx)
        cap = nq if capacity is None else int(capacity)
        idx = np.empty(cap, dtype=np.int64)
        f = np.empty(cap)
        v = np.empty(cap) if want_v else None
        n_out = C.c_size_t(0)
        rc = self._L.gpx_model_sample_surface(self._h, nq, _dptr(qx), _dptr(qy), _dptr(qz), float(f_tol), cap,
                                              idx.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(f),
                                              _dptr(v) if v is not None else None, C.byref(n_out))
        n = int(n_out.value)
        if rc != OK and rc != E_SIZE_MISMATCH:
            _check(rc)
        k = min(n, cap)
        out = {"idx": idx[:k], "f": f[:k], "n_total": n, "truncated": rc == E_SIZE_MISMATCH}
        if v is not None:
            out["v"] = v[:k]
        return out

    def march_surface(self, leaf, step, f_tol=0.01, start=None, max_cubes=1 << 20, capacity=1 << 20, want_v=True):
        """marchingSampling / marchingCubes (src/gp_node.cpp:1102-1291) as a frontier of device batches:
        dict(xyz, f, v, n_total, n_cubes, truncated)."""
        xyz = np.empty((capacity, 3)); f = np.empty(capacity)
        v = np.empty(capacity) if want_v else None
        n_out, n_cubes = C.c_size_t(0), C.c_size_t(0)
        sp = None
        if start is not None:
            st = np.ascontiguousarray(start, dtype=np.float64)
            sp = _dptr(st)
        rc = self._L.gpx_model_march_surface(self._h, sp, float(leaf), float(step), float(f_tol), int(max_cubes),
                                             int(capacity), _dptr(xyz), _dptr(f), _dptr(v) if v is not None else None,
                                             C.byref(n_out), C.byref(n_cubes))
        if rc != OK and rc != E_SIZE_MISMATCH:
            _check(rc)
        k = min(int(n_out.value), capacity)
        out = {"xyz": xyz[:k], "f": f[:k], "n_total": int(n_out.value), "n_cubes": int(n_cubes.value),
               "truncated": rc == E_SIZE_MISMATCH}
        if v is not None:
            out["v"] = v[:k]
        return out

    def project(self, x, y, z, normal, f_tol=1e-2, improve_tol=1e-7, max_iter=500, step_mul=0.001):
        """Batched AtlasBase::project (atlas.hpp:201-276): dict(xyz, f, iter, status) for every start point."""
        x, y, z = _as_d(x), _as_d(y), _as_d(z)
        nq = len(x)
        nrm = np.ascontiguousarray(np.asarray(normal, dtype=np.float64).reshape(nq, 3))
        opt = ProjectOptions(float(f_tol), float(improve_tol), float(step_mul), int(max_iter), (C.c_int32 * 3)())
        out = np.empty((nq, 3)); f = np.empty(nq)
        it = np.empty(nq, dtype=np.int32); st = np.empty(nq, dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        _check(self._L.gpx_model_project(self._h, nq, _dptr(x), _dptr(y), _dptr(z), _dptr(nrm), C.byref(opt),
                                         _dptr(out), _dptr(f), it.ctypes.data_as(i32), st.ctypes.data_as(i32)))
        return {"xyz": out, "f": f, "iter": it, "status": st}

    def evaluate_device(self, nq, d_qx, d_qy, d_qz, d_f, d_v=None, d_grad=None, d_tx=None, d_ty=None, stream=None):
        """Raw device pointers (ints, e.g. torch.Tensor.data_ptr()) of fp64 arrays; asynchronous."""
        vp = lambda a: C.c_void_p(int(a)) if a else None
        _check(self._L.gpx_model_evaluate_device(self._h, int(nq), vp(d_qx), vp(d_qy), vp(d_qz), vp(d_f), vp(d_v),
                                                 vp(d_grad), vp(d_tx), vp(d_ty), vp(stream)))

    def state_blob(self, part):
        ptr = C.c_void_p(None)
        nbytes = C.c_size_t(0)
        _check(self._L.gpx_model_state_blob(self._h, int(part), C.byref(ptr), C.byref(nbytes)))
        return int(ptr.value), int(nbytes.value)

    def commit(self, with_variance=True):
        _check(self._L.gpx_model_commit(self._h, int(bool(with_variance))))

    def replicate(self, devices):
        """gpx_model_replicate: read-only replicas of this model on the given HIP devices (list of Model)."""
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        outs = (C.c_void_p * len(devices))()
        _check(self._L.gpx_model_replicate(self._h, len(devices), devs, outs))
        reps = []
        for h in outs:
            r = Model.__new__(Model)
            r._L, r._h, r.kernel, r.precision = self._L, C.c_void_p(h), self.kernel, self.precision
            reps.append(r)
        return reps


def slab_range(nq, rank, world):
    """gpx_slab_range: the contiguous slab [lo, hi) of nq queries that replica `rank` of `world` receives."""
    lo, hi = C.c_size_t(0), C.c_size_t(0)
    lib().gpx_slab_range(int(nq), int(rank), int(world), C.byref(lo), C.byref(hi))
    return int(lo.value), int(hi.value)


def _handles(models):
    arr = (C.c_void_p * len(models))(*[m._h.value if isinstance(m._h, C.c_void_p) else m._h for m in models])
    return arr


def evaluate_sharded(models, qx, qy, qz, want_v=False, want_grad=False, want_basis=False):
    """gpx_model_evaluate_sharded: ONE evaluate call cut into contiguous slabs over `models` (a model and its replicas)."""
    L = lib()
    qx, qy, qz = _as_d(qx), _as_d(qy), _as_d(qz)
    nq = len(qx)
    f = np.empty(nq)
    v = np.empty(nq) if want_v else None
    g = np.empty((nq, 3)) if want_grad else None
    tx = np.empty((nq, 3)) if want_basis else None
    ty = np.empty((nq, 3)) if want_basis else None
    p = lambda a: _dptr(a) if a is not None else None
    _check(L.gpx_model_evaluate_sharded(_handles(models), len(models), nq, _dptr(qx), _dptr(qy), _dptr(qz), p(f), p(v), p(g),
                                        p(tx), p(ty)))
    out = {"f": f}
    if v is not None:
        out["v"] = v
    if g is not None:
        out["grad"] = g
    if tx is not None:
        out["tx"], out["ty"] = tx, ty
    return out


def sample_surface_sharded(models, qx, qy, qz, f_tol=0.01, capacity=None, want_v=True):
    """gpx_model_sample_surface_sharded: ONE sampleSurface call over `models`; same dict as Model.sample_surface."""
    L = lib()
    qx, qy, qz = _as_d(qx), _as_d(qy), _as_d(qz)
    nq = len(qx)
    cap = nq if capacity is None else int(capacity)
    idx = np.empty(max(cap, 1), dtype=np.int64)
    f = np.empty(max(cap, 1))
    v = np.empty(max(cap, 1)) if want_v else None
    n_out = C.c_size_t(0)
    rc = L.gpx_model_sample_surface_sharded(_handles(models), len(models), nq, _dptr(qx), _dptr(qy), _dptr(qz), float(f_tol), cap,
                                            idx.ctypes.data_as(C.POINTER(C.c_int64)), _dptr(f),
                                            _dptr(v) if v is not None else None, C.byref(n_out))
    n = int(n_out.value)
    if rc != OK and rc != E_SIZE_MISMATCH:
        _check(rc)
    k = min(n, cap)
    out = {"idx": idx[:k], "f": f[:k], "n_total": n, "truncated": rc == E_SIZE_MISMATCH}
    if v is not None:
        out["v"] = v[:k]
    return out


class RProp(C.Structure):
    """gpx_rprop = Optimisation::Desc (include/gp/GaussianProcess.h:49-62)."""
    _fields_ = [("delta0", C.c_double), ("delta_min", C.c_double), ("delta_max", C.c_double), ("eta_minus", C.c_double),
                ("eta_plus", C.c_double), ("eps_stop", C.c_double), ("max_iter", C.c_uint64)]


class RPropResult(C.Structure):
    _fields_ = [("loghyper", C.c_double * 2), ("loglik", C.c_double), ("iterations", C.c_uint64)]


class DerivativeGP:
    """gpx_dgp_*: the first slice of the reference's second library, gp::GaussianProcess (values + gradients)."""

    def __init__(self, kernel, noise, x, y, z, target, normals=None, device=-1):
        self._L = lib()
        self._h = C.c_void_p(None)
        x, y, z, target = _as_d(x), _as_d(y), _as_d(z), _as_d(target)
        if not (len(x) == len(y) == len(z) == len(target)):
            raise GpxError(E_SIZE_MISMATCH, "coordinate / target length mismatch")
        nr = None
        if normals is not None:
            nr = _as_d(np.asarray(normals, dtype=np.float64).reshape(-1))
            if len(nr) != 3 * len(x):
                raise GpxError(E_SIZE_MISMATCH, "normals must be n x 3")
        opt = Options()
        opt.precision = F64
        opt.device = int(device)
        opt.ir_steps = -1
        self.n = len(x)
        _check(self._L.gpx_dgp_create(C.byref(kernel), float(noise), len(x), _dptr(x), _dptr(y), _dptr(z), _dptr(target),
                                      _dptr(nr) if nr is not None else None, C.byref(opt), C.byref(self._h)))

    def add(self, x, y, z, target, normals=None):
        """add_patterns: append samples and rebuild on the union (include/gp/GaussianProcess.h:340-374)."""
        x, y, z, target = _as_d(x), _as_d(y), _as_d(z), _as_d(target)
        if not (len(x) == len(y) == len(z) == len(target)):
            raise GpxError(E_SIZE_MISMATCH, "coordinate / target length mismatch")
        nr = None
        if normals is not None:
            nr = _as_d(np.asarray(normals, dtype=np.float64).reshape(-1))
            if len(nr) != 3 * len(x):
                raise GpxError(E_SIZE_MISMATCH, "normals must be n x 3")
        _check(self._L.gpx_dgp_add(self._h, len(x), _dptr(x), _dptr(y), _dptr(z), _dptr(target),
                                   _dptr(nr) if nr is not None else None))
        self.n += len(x)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.gpx_dgp_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def alpha(self):
        a = np.zeros(4 * self.n)
        _check(self._L.gpx_dgp_get(self._h, 1, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    @property
    def loglik(self):
        a = np.zeros(1)
        _check(self._L.gpx_dgp_get(self._h, 2, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return float(a[0])

    @property
    def stats(self):
        s = Stats()
        _check(self._L.gpx_dgp_get(self._h, 3, C.byref(s), C.sizeof(s)))
        return s.as_dict()

    @property
    def appended_from(self):
        a = np.zeros(1, dtype=np.int64)
        _check(self._L.gpx_dgp_get(self._h, 4, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return int(a[0])

    def loglik_gradient(self):
        """logLikelihoodGradient (include/gp/GaussianProcess.h:387-410): d loglik / d (log l, log sf); SE models only."""
        g = np.zeros(2)
        _check(self._L.gpx_dgp_loglik_gradient(self._h, _dptr(g)))
        return g

    def optimise(self, **desc):
        """Optimisation::find (RProp, include/gp/GaussianProcess.h:86-122) in place; keyword arguments override
        Optimisation::Desc's defaults (delta0, delta_min, delta_max, eta_minus, eta_plus, eps_stop, max_iter)."""
        d = RProp()
        self._L.gpx_rprop_default(C.byref(d))
        for k, v in desc.items():
            if not hasattr(d, k):
                raise TypeError("unknown RProp field %r" % k)
            setattr(d, k, v)
        r = RPropResult()
        _check(self._L.gpx_dgp_optimise(self._h, C.byref(d), C.byref(r)))
        return {"loghyper": np.array([r.loghyper[0], r.loghyper[1]]), "loglik": float(r.loglik), "iterations": int(r.iterations)}

    def evaluate(self, qx, qy, qz, want_v=True):
        qx, qy, qz = _as_d(qx), _as_d(qy), _as_d(qz)
        nq = len(qx)
        f4 = np.zeros((nq, 4))
        v = np.zeros(nq) if want_v else None
        _check(self._L.gpx_dgp_evaluate(self._h, nq, _dptr(qx), _dptr(qy), _dptr(qz), _dptr(f4.reshape(-1)),
                                        _dptr(v) if want_v else None))
        out = {"f": f4[:, 0].copy(), "grad": f4[:, 1:].copy()}
        if want_v:
            out["v"] = v
        return out


def pcd_read(path):
    L = lib()
    n = L.gpx_pcd_read(path.encode(), None, 0)
    if n < 0:
        raise GpxError(int(n), "cannot read PCD file %s" % path)
    xyz = np.zeros((n, 3), dtype=np.float32)
    n2 = L.gpx_pcd_read(path.encode(), xyz.ctypes.data_as(C.POINTER(C.c_float)), n)
    if n2 != n:
        raise GpxError(int(n2), "cannot decode PCD file %s" % path)
    return xyz


def node_training_set(xyz, sigma2=0.1, rad=2.0):
    L = lib()
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    n = len(xyz)
    out = [np.zeros(n + 15) for _ in range(5)]
    rc = L.gpx_node_training_set(xyz.ctypes.data_as(C.POINTER(C.c_float)), n, float(sigma2), float(rad),
                                 *[_dptr(a) for a in out])
    if rc < 0:
        raise GpxError(rc, "gpx_node_training_set failed")
    return tuple(a[:n + rc] for a in out)
