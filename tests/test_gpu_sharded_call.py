"""ONE evaluate / sampleSurface call cut over several replicas from the C boundary (VERDICT r5 missing 4; north star: "query-grid
shards"): gpx_model_evaluate_sharded / gpx_model_sample_surface_sharded and the header shim's GPX_DEVICES route.  The one-GPU
test box runs replicas on the SAME ordinal (devs = {0, 0}); a second ordinal has not run (include/gpx.h says so).  The bar is
bit-identity with the single call: every value is computed per query, independent of the batch it sits in."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "gaussian-object-modelling_amd")


def _queries(nq, seed):
    rng = np.random.default_rng(seed)
    return tuple(rng.uniform(-1.05, 1.05, nq) for _ in range(3))


SLICE = 1 << 18  # the single call's own pipeline unit (csrc/gpx_eval.hip: EVAL_SLICE): the sharded evaluate distributes whole slices


@pytest.mark.parametrize("kernel, n, prec, nrep, nq", [
    (("matern52", 1.0, 1.0), 300, "F32", 2, 2 * SLICE + 10007),   # small-model fp32 kernel; slices 2 + 1 (the partial one last)
    (("gaussian", 1.0, 1.0), 277, "F64", 3, 3 * SLICE),           # small fp64 kernel (mean fused into the variance launch); 1 + 1 + 1
    (("thinplate", 2.0), 277, "F64", 2, SLICE + 1),               # indefinite model; a last slice of ONE query
    (("gaussian", 1.0, 1.0), 700, "F32_SPLIT", 4, 2 * SLICE + 5),  # split-fp16 small kernel; more replicas than slices: 1 + 1 + 1 + 0
    (("matern32", 1.0, 1.0), 1500, "F32", 3, 4 * SLICE + 4099),   # large-model tiles (the mean's point split and the paired
    (("laplace", 1.0, 1.0), 1500, "F64", 2, 2 * SLICE + 70001),   #  variance launch depend on the batch: whole slices keep them)
    (("matern52", 1.0, 1.0), 2100, "F32_SPLIT", 2, 2 * SLICE + 333),
])
def test_sharded_evaluate_is_bit_identical_to_the_single_call(gpu, ds, kernel, n, prec, nrep, nq):
    m = gpu.Model(gpu.make_kernel(*kernel), *ds.fibonacci_training_set(n), precision=getattr(gpu, prec), prepare_variance=True)
    reps = [m] + m.replicate([0] * (nrep - 1))
    q = _queries(nq, n + nq)
    one = m.evaluate(*q, want_v=True, want_grad=True, want_basis=True)
    many = gpu.evaluate_sharded(reps, *q, want_v=True, want_grad=True, want_basis=True)
    for key in ("f", "v", "grad", "tx", "ty"):
        np.testing.assert_array_equal(one[key], many[key], err_msg=key)
    del one, many
    # the other overloads: mean only; mean + variance (small fp64 models take their mean from the variance kernel there)
    np.testing.assert_array_equal(m.evaluate(*q)["f"], gpu.evaluate_sharded(reps, *q)["f"])
    a, b = m.evaluate(*q, want_v=True), gpu.evaluate_sharded(reps, *q, want_v=True)
    np.testing.assert_array_equal(a["f"], b["f"])
    np.testing.assert_array_equal(a["v"], b["v"])
    for r in reps[1:]:
        r.close()
    m.close()


def test_sharded_evaluate_small_calls_and_slab_rule(gpu, ds):
    """Calls of one slice stay whole on replicas[0] (nq < n_replicas, a handful of queries, a full slice); the slab rule."""
    m = gpu.Model(gpu.make_kernel("gaussian", 1.0, 1.0), *ds.fibonacci_training_set(277), precision=gpu.F64, prepare_variance=True)
    reps = [m] + m.replicate([0, 0, 0])
    for nq in (1, 3, 5, 257, 4097, SLICE):
        q = _queries(nq, nq)
        one = m.evaluate(*q, want_v=True, want_grad=True)
        many = gpu.evaluate_sharded(reps, *q, want_v=True, want_grad=True)
        for key in ("f", "v", "grad"):
            np.testing.assert_array_equal(one[key], many[key], err_msg="%s nq=%d" % (key, nq))
    import importlib
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    for nq, w in ((10, 3), (1 << 20, 8), (7, 8), (16777216, 4), (0, 2)):
        assert [gpu.slab_range(nq, r, w) for r in range(w)] == [sh.slab_range(nq, r, w) for r in range(w)]
    # errors: the same handle twice, a null entry, an empty call, a null array
    L = gpu.lib()
    one = (C.c_double * 1)(0.5)
    arr = (C.c_void_p * 2)(m._h.value, m._h.value)
    assert L.gpx_model_evaluate_sharded(arr, 2, 1, one, one, one, one, None, None, None, None) == gpu.E_BAD_ARG
    arr = (C.c_void_p * 2)(m._h.value, None)
    assert L.gpx_model_evaluate_sharded(arr, 2, 1, one, one, one, one, None, None, None, None) == gpu.E_NULL
    assert L.gpx_last_error() == b"Empty Model pointer"
    arr = (C.c_void_p * 2)(reps[0]._h.value, reps[1]._h.value)
    assert L.gpx_model_evaluate_sharded(arr, 2, 0, one, one, one, one, None, None, None, None) == gpu.E_EMPTY
    assert L.gpx_model_evaluate_sharded(arr, 0, 1, one, one, one, one, None, None, None, None) == gpu.E_BAD_ARG
    q = _queries(1000, 5)
    assert L.gpx_model_evaluate_sharded(arr, 2, 1000, q[0].ctypes.data_as(C.POINTER(C.c_double)), None,
                                        q[2].ctypes.data_as(C.POINTER(C.c_double)), one, None, None, None, None) == gpu.E_NULL
    for r in reps[1:]:
        r.close()
    m.close()


@pytest.mark.parametrize("kernel, n, prec, g", [
    (("gaussian", 1.0, 1.0), 724, "F32", 48),   # 110592 lattice points: slabs of 36864 take the fp32 screen, like the whole
    (("matern52", 1.0, 1.0), 300, "F64", 24),   # 13824: no screen
    (("thinplate", 2.0), 277, "F64", 30),
    # large models: the mean of the whole grid, of the candidates and of a slab's candidates, and the variance of differently
    # composed survivor batches must not depend on the batch (evaluate_locked's fixed_order form)
    (("matern52", 1.0, 1.0), 3000, "F32", 45),  # 91125: whole grid screened, slabs of 30375 not
    (("gaussian", 1.0, 1.0), 2500, "F64", 40),
    (("thinplate", 4.0), 1800, "F32", 36),      # no screen for the thin plate: the fp64 mean of the whole grid / of the slabs
])
def test_sharded_sample_surface_returns_the_single_calls_set(gpu, ds, kernel, n, prec, g):
    m = gpu.Model(gpu.make_kernel(*kernel), *ds.fibonacci_training_set(n), precision=getattr(gpu, prec), prepare_variance=True)
    reps = [m] + m.replicate([0, 0])
    t = np.linspace(-1.01, 1.01, g)
    qx, qy, qz = (a.ravel().copy() for a in np.meshgrid(t, t, t, indexing="ij"))
    tol = 0.02
    one = m.sample_surface(qx, qy, qz, f_tol=tol)
    many = gpu.sample_surface_sharded(reps, qx, qy, qz, f_tol=tol)
    assert one["n_total"] == many["n_total"] > 20
    for key in ("idx", "f", "v"):
        np.testing.assert_array_equal(one[key], many[key], err_msg=key)
    # capacity below the number of survivors: the first `capacity` in query order, the full count, the truncation status
    cap = one["n_total"] // 2
    one_c = m.sample_surface(qx, qy, qz, f_tol=tol, capacity=cap)
    many_c = gpu.sample_surface_sharded(reps, qx, qy, qz, f_tol=tol, capacity=cap)
    assert many_c["truncated"] and one_c["truncated"] and many_c["n_total"] == one["n_total"]
    for key in ("idx", "f", "v"):
        np.testing.assert_array_equal(one_c[key], many_c[key], err_msg=key)
        np.testing.assert_array_equal(one[key][:cap], many_c[key], err_msg=key)
    # selection only
    sel = gpu.sample_surface_sharded(reps, qx, qy, qz, f_tol=tol, want_v=False)
    np.testing.assert_array_equal(sel["idx"], one["idx"])
    for r in reps[1:]:
        r.close()
    m.close()


def test_unchanged_caller_with_gpx_devices(gpu, tmp_path):
    """tests/cpp/sharded_shim.cpp: a caller of the reference's shape (GPRegressor<Gaussian>, create, the evaluate overloads,
    update) built against the header shim; with GPX_DEVICES=0,0,0 in its environment the same binary cuts its large calls over
    three replicas and prints the same checksums, bit for bit."""
    exe = str(tmp_path / "sharded_shim")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(PKG, "include"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "sharded_shim.cpp"), "-o", exe, "-L", os.path.join(PKG, "lib"), "-lgpx",
           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("GPX_DEVICES", "GPX_SHARD_MIN_NQ")}
    single = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert single.returncode == 0, single.stdout + single.stderr
    many = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=dict(env, GPX_DEVICES="0,0,0", GPX_SHARD_MIN_NQ="1000"))
    assert many.returncode == 0, many.stdout + many.stderr
    s_lines = [ln for ln in single.stdout.splitlines() if ln.startswith("sum ")]
    m_lines = [ln for ln in many.stdout.splitlines() if ln.startswith("sum ")]
    assert len(s_lines) >= 6 and s_lines == m_lines
    assert "shards 1" in single.stdout and "shards 3" in many.stdout
