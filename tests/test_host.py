"""CPU suite, part 2: host logic (input recipes, PCD readers, sharding) and the C-ABI surface of
libgpx.so (loads, exports every symbol of include/gpx.h, validates arguments, and FAILS LOUDLY without a
GPU -- there is no CPU compute path)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT

PCD_COUNTS = {"bowlA": 251, "bowlB": 466, "containerA": 451, "containerB": 709, "jug": 432, "kettle": 697,
              "kitchenUtensilB": 152, "mugD": 262, "pot": 339}


def test_pcd_fixture_census(ds, golden):
    for name, cnt in PCD_COUNTS.items():
        p = ds.read_pcd(os.path.join(GOLDEN_DIR, "pcd", name + ".pcd"))
        assert p.shape == (cnt, 3) and p.dtype == np.float32
        assert np.all(np.isfinite(p))
    assert dict(zip([str(n) for n in golden["pcd/names"]], [int(c) for c in golden["pcd/counts"]])) == PCD_COUNTS


def test_node_training_set_recipe(ds):
    """src/gp_node.cpp:85-117, :793-914: unit-ball normalisation, 15 exterior points on r = 2."""
    pts = ds.read_pcd(os.path.join(GOLDEN_DIR, "pcd", "mugD.pcd"))
    x, y, z, lab, s2 = ds.node_training_set(pts)
    assert len(x) == 262 + 15
    r = np.sqrt(x * x + y * y + z * z)
    assert abs(r[:262].max() - 1.0) < 1e-6
    np.testing.assert_allclose(r[262:], 2.0, rtol=1e-14)
    assert np.all(lab[:262] == 0) and np.all(lab[262:] == 1) and np.all(s2 == 0.1)
    ext = ds.exterior_points()
    assert ext.shape == (15, 3)
    assert sorted(set(np.round(ext[:, 2], 6))) == [-1.333333, -0.0, 1.333333] or len(set(np.round(ext[:, 2], 6))) == 3


def test_grid_matches_reference_loop(ds):
    """fakeDeterministicSampling at scale 1.01 / pass 0.07 visits 29 points per axis (src/gp_node.cpp:1025)."""
    cnt, xx = 0, -1.01
    while xx <= 1.01:
        cnt += 1
        xx += 0.07
    assert cnt == 29
    qx, qy, qz = ds.query_grid(5)
    assert len(qx) == 125 and qx[0] == -1.01 and qz[4] == 1.01 and qx[24] == qx[0] and qz[1] > qz[0]
    lo_hi = [ds.query_grid_slab(7, r, 3) for r in range(3)]
    assert sum(len(s[0]) for s in lo_hi) == 343
    full = ds.query_grid(7)
    np.testing.assert_array_equal(np.concatenate([s[0] for s in lo_hi]), full[0])
    np.testing.assert_array_equal(np.concatenate([s[2] for s in lo_hi]), full[2])


def test_mt19937_64_known_answer(ds):
    g = ds.MT19937_64(5489)  # std::mt19937_64 default seed: 10000th output is 9981545732273789042
    v = 0
    for _ in range(10000):
        v = g.next()
    assert v == 9981545732273789042
    x, y, z, lab, s2 = ds.fibonacci_training_set(100)
    x2 = ds.fibonacci_training_set(100)[0]
    np.testing.assert_array_equal(x, x2)
    assert len(x) == 100 and lab.sum() == 15
    assert np.max(np.abs(np.sqrt(x[:85] ** 2 + y[:85] ** 2 + z[:85] ** 2) - 1)) < 2e-3


def test_slab_ranges(ds):
    import importlib
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    for nq, w in ((10, 3), (1 << 20, 8), (7, 8), (16777216, 4)):
        spans = [sh.slab_range(nq, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == nq
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sh.slab_range(10, 3, 3)


# ------------------------------------------------------------------------------------------ C ABI
def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "gpx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpx_[a-z_0-9]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol(gpx):
    lib = gpx.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for sym in declared:
        assert hasattr(lib, sym), "libgpx.so does not export %s (declared in include/gpx.h)" % sym
    assert sorted(gpx.EXPORTS) == declared
    assert b"gfx950" in lib.gpx_version()
    assert lib.gpx_padded_n(277) == 512 and lib.gpx_padded_n(16384) == 16384 and lib.gpx_padded_n(1) == 256


def test_argument_validation_and_reference_messages(gpx):
    """Error behaviour of assertData (gp_regressor.hpp:563-572) at the C boundary; no GPU needed."""
    lib = gpx.lib()
    k = gpx.make_kernel("gaussian", 1, 1)
    h = C.c_void_p(None)
    one = (C.c_double * 1)(0.5)
    rc = lib.gpx_model_create(C.byref(k), 0, None, None, None, None, None, None, C.byref(h))
    assert rc == gpx.E_EMPTY and lib.gpx_last_error() == b"All input data is empty!"
    rc = lib.gpx_model_create(C.byref(k), 1, None, one, one, one, None, None, C.byref(h))
    assert rc == gpx.E_NULL and lib.gpx_last_error() == b"Empty data pointer"
    rc = lib.gpx_model_create(C.byref(k), 1, one, one, one, one, None, None, None)
    assert rc == gpx.E_NULL and lib.gpx_last_error() == b"Empty Model pointer"
    nan = (C.c_double * 1)(float("nan"))
    rc = lib.gpx_model_create(C.byref(k), 1, nan, one, one, one, None, None, C.byref(h))
    assert rc == gpx.E_NAN_INPUT
    bad = gpx.Options()
    bad.precision = 7
    rc = lib.gpx_model_create(C.byref(k), 1, one, one, one, one, None, C.byref(bad), C.byref(h))
    assert rc == gpx.E_BAD_ARG
    rc = lib.gpx_model_evaluate(None, 1, one, one, one, one, None, None, None, None)
    assert rc == gpx.E_NULL and lib.gpx_last_error() == b"Empty Model pointer"
    rc = lib.gpx_model_update(None, 1, one, one, one, one, None)
    assert rc == gpx.E_NULL and lib.gpx_last_error() == b"Empty model pointer"
    assert h.value is None


def test_slab_rule_and_sharded_call_validation_at_the_c_boundary(gpx):
    """gpx_slab_range is sharding.slab_range (the one-process-per-GPU form cuts the grid by the same rule); the sharded entry
    points validate their handle list before anything touches a device."""
    import importlib
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    for nq, w in ((10, 3), (1 << 20, 8), (7, 8), (16777216, 4), (0, 2), (5, 1)):
        assert [gpx.slab_range(nq, r, w) for r in range(w)] == [sh.slab_range(nq, r, w) for r in range(w)]
    assert gpx.slab_range(10, 3, 3) == (0, 0) and gpx.slab_range(10, -1, 3) == (0, 0)  # out of range: the empty slab
    lib = gpx.lib()
    one = (C.c_double * 1)(0.5)
    n_out = C.c_size_t(7)
    idx = (C.c_int64 * 1)()
    assert lib.gpx_model_evaluate_sharded(None, 2, 1, one, one, one, one, None, None, None, None) == gpx.E_NULL
    assert lib.gpx_last_error() == b"Empty Model pointer"
    arr = (C.c_void_p * 2)(None, None)
    assert lib.gpx_model_evaluate_sharded(arr, 2, 1, one, one, one, one, None, None, None, None) == gpx.E_NULL
    assert lib.gpx_model_evaluate_sharded(arr, 0, 1, one, one, one, one, None, None, None, None) == gpx.E_BAD_ARG
    assert lib.gpx_model_sample_surface_sharded(arr, 2, 1, one, one, one, 0.01, 1, idx, one, one, C.byref(n_out)) == gpx.E_NULL
    assert n_out.value == 0
    assert lib.gpx_model_sample_surface_sharded(arr, 2, 1, one, one, one, 0.01, 1, idx, one, one, None) == gpx.E_NULL


def test_derivative_gp_argument_validation(gpx):
    """gpx_dgp_* (first slice of the reference's gp::GaussianProcess): argument checks at the C boundary, no GPU needed.
    "No training data available" is the reference's message (include/gp/GaussianProcess.h:239)."""
    lib = gpx.lib()
    h = C.c_void_p(None)
    one = (C.c_double * 3)(0.5, 0.5, 0.5)
    k = gpx.make_kernel("se", 1.0, 1.0)
    assert lib.gpx_dgp_create(C.byref(k), 0.1, 0, one, one, one, one, None, None, C.byref(h)) == gpx.E_EMPTY
    assert lib.gpx_last_error() == b"No training data available"
    assert lib.gpx_dgp_create(C.byref(k), 0.1, 1, None, one, one, one, None, None, C.byref(h)) == gpx.E_NULL
    assert lib.gpx_dgp_create(C.byref(k), 0.1, 1, one, one, one, one, None, None, None) == gpx.E_NULL
    assert lib.gpx_dgp_create(None, 0.1, 1, one, one, one, one, None, None, C.byref(h)) == gpx.E_NULL
    assert lib.gpx_dgp_create(C.byref(k), -1.0, 1, one, one, one, one, None, None, C.byref(h)) == gpx.E_BAD_ARG
    km = gpx.make_kernel("matern52", 1.0, 1.0)
    assert lib.gpx_dgp_create(C.byref(km), 0.1, 1, one, one, one, one, None, None, C.byref(h)) == gpx.E_BAD_ARG
    nan = (C.c_double * 1)(float("nan"))
    assert lib.gpx_dgp_create(C.byref(k), 0.1, 1, nan, one, one, one, None, None, C.byref(h)) == gpx.E_NAN_INPUT
    assert lib.gpx_dgp_get(None, 0, one, 8) == gpx.E_NULL
    assert lib.gpx_dgp_add(None, 1, one, one, one, one, None) == gpx.E_NULL
    lib.gpx_dgp_destroy(None)
    assert h.value is None


def test_no_cpu_fallback(gpx):
    """Without a HIP device every compute entry point fails loudly (E_NO_DEVICE)."""
    if gpx.device_count() > 0:
        pytest.skip("a GPU is visible; the loud-failure path is exercised on CPU-only hosts")
    with pytest.raises(gpx.GpxError) as ei:
        gpx.Model(gpx.make_kernel("gaussian", 1, 1), [0.0, 1.0], [0.0, 0.0], [0.0, 0.0], [0.0, 1.0], [0.1, 0.1])
    assert ei.value.code == gpx.E_NO_DEVICE


def test_host_pcd_reader_matches_python(gpx, ds):
    """gpx_pcd_read / gpx_node_training_set (host C++) against the NumPy recipe, bit for bit."""
    for name in PCD_COUNTS:
        path = os.path.join(GOLDEN_DIR, "pcd", name + ".pcd")
        a = gpx.pcd_read(path)
        b = ds.read_pcd(path)
        np.testing.assert_array_equal(a, b)
        ta = gpx.node_training_set(a)
        tb = ds.node_training_set(b)
        for u, v in zip(ta, tb):
            np.testing.assert_array_equal(u, v)
    with pytest.raises(gpx.GpxError):
        gpx.pcd_read(os.path.join(GOLDEN_DIR, "pcd", "does_not_exist.pcd"))


def test_pcd_ascii_and_binary_modes(gpx, ds, tmp_path):
    pts = np.array([[0.5, -1.25, 2.0], [1.0, 2.0, 3.0], [-4.0, 0.125, 8.5]], dtype=np.float32)
    hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgba\nSIZE 4 4 4 4\nTYPE F F F U\nCOUNT 1 1 1 1\nWIDTH 3\n"
           "HEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 3\nDATA %s\n")
    pa = tmp_path / "a.pcd"
    pa.write_text(hdr % "ascii" + "".join("%r %r %r 4278190335\n" % tuple(float(v) for v in p) for p in pts))
    pb = tmp_path / "b.pcd"
    rec = np.zeros(3, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")])
    rec["x"], rec["y"], rec["z"], rec["rgba"] = pts[:, 0], pts[:, 1], pts[:, 2], 7
    pb.write_bytes((hdr % "binary").encode() + rec.tobytes())
    for p in (pa, pb):
        np.testing.assert_array_equal(gpx.pcd_read(str(p)), pts)
        np.testing.assert_array_equal(ds.read_pcd(str(p)), pts)


def test_pcd_reader_survives_mutated_files_under_sanitizers(tmp_path):
    """gpx_pcd_read parses untrusted files: 900 mutated inputs (byte flips, truncation, corrupted header numbers,
    absurd POINTS counts, garbage bodies) under AddressSanitizer + UBSan on the CPU build of the reader."""
    import glob
    import subprocess
    exe = str(tmp_path / "pcd_fuzz")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "pcd_fuzz.cpp"),
           os.path.join(ROOT, "gaussian-object-modelling_amd", "csrc", "gpx_pcd.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "pcd", "*.pcd")))
    env = dict(os.environ, PCD_FUZZ_TMP=str(tmp_path / "fuzz.pcd"), ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe, "100"] + files, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert "no crash" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_host_concurrency_under_sanitizers(tmp_path, san):
    """SURVEY section 5 / VERDICT r2 #6: the host-side concurrency machinery of libgpx.so -- flat combining of concurrent
    evaluate() calls, the pool of large device buffers, the per-device once flags, Eigen's pivot order
    (csrc/gpx_host.{hpp,cpp}, the very translation unit the library links) -- built by g++ with ThreadSanitizer and
    with AddressSanitizer + UBSan against a stub device backend: 841 threads x single-point requests (the node's
    29 x 29 threads per x-slice, src/gp_node.cpp:1027-1038), concurrent create / destroy with trims, a capped and a
    disabled pool (GPX_POOL_MB=0), injected allocation failures (every fifth, always), leak check."""
    import subprocess
    exe = str(tmp_path / "host_concurrency")
    csrc = os.path.join(ROOT, "gaussian-object-modelling_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-pthread", "-I", csrc,
           os.path.join(ROOT, "tests", "cpp", "host_concurrency.cpp"), os.path.join(csrc, "gpx_host.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and "sanitize" in (r.stderr or ""):
        pytest.skip("sanitizer runtime not available: " + r.stderr[-200:])
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([exe, "841"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "host concurrency ok" in r.stdout
    assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


def test_eigen_typed_overloads_compile_where_eigen_exists(tmp_path):
    """the Eigen::MatrixXd overloads of evaluate and computeTangentBasis(Eigen::Vector3d...) of the header
    shim sit behind __has_include(<Eigen/Core>).  Eigen 3 is not in this image, so here the test records
    that and skips; on a machine that has it the overloads must compile against the calls the reference makes."""
    import subprocess
    probe = tmp_path / "probe.cpp"
    probe.write_text("#if __has_include(<Eigen/Core>)\nint have_eigen = 1;\n#else\n#error no Eigen\n#endif\n")
    inc = []
    for cand in ("/usr/include/eigen3", "/usr/local/include/eigen3"):
        if os.path.isdir(cand):
            inc += ["-I", cand]
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only"] + inc + [str(probe)], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("<Eigen/Core> not available in this image: the Eigen-typed overloads stay uncompiled here")
    pkg = os.path.join(ROOT, "gaussian-object-modelling_amd")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall"] + inc + ["-I", os.path.join(pkg, "include"), "-I",
                        os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "eigen_overloads.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
