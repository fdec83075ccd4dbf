"""Builds tests/cpp/caller_shape.cpp -- a caller shaped like the reference's src/gp_node.cpp and
include/atlas/atlas_variance.hpp -- against the header shim and libgpx.so, runs it on the GPU and
checks what it computed against the oracle."""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT, nerr

pytestmark = pytest.mark.gpu
PKG = os.path.join(ROOT, "gaussian-object-modelling_amd")


def test_reference_shaped_caller(gpu, orc, tmp_path):
    exe = str(tmp_path / "caller_shape")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(PKG, "include"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "caller_shape.cpp"), "-o", exe, "-L", os.path.join(PKG, "lib"), "-lgpx",
           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out_txt = str(tmp_path / "out.txt")
    grid = 7
    r = subprocess.run([exe, os.path.join(GOLDEN_DIR, "pcd", "mugD.pcd"), out_txt, str(grid)], capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "caller_shape: OK" in r.stdout
    rows = [list(map(float, ln.split())) for ln in open(out_txt)]
    n, ng = int(rows[0][0]), int(rows[0][1])
    tr = np.array(rows[1:1 + n])
    g = np.array(rows[1 + n:1 + n + ng])
    fv, gg, fvu = rows[1 + n + ng], rows[2 + n + ng], rows[3 + n + ng]
    assert n == 277 and ng == grid ** 3
    om = orc.Model(orc.make_kernel("thinplate", 2.0), tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3], tr[:, 4])
    ref = om.evaluate(g[:, 0], g[:, 1], g[:, 2], want_v=True)
    assert nerr(g[:, 3], ref["f"]) < 1e-10 and nerr(g[:, 4], ref["v"]) < 1e-10  # shim default = fp64
    c = np.array([[0.3, -0.2, 0.6], tr[5, :3]])
    rc = om.evaluate(c[:, 0], c[:, 1], c[:, 2], want_v=True, want_grad=True)
    assert nerr([fv[0], fv[2]], rc["f"]) < 1e-10 and nerr([fv[1], fv[3]], rc["v"]) < 1e-10
    assert nerr(np.array(gg).reshape(2, 3), rc["grad"]) < 1e-10
    om.update([0.05, -0.4, 0.7], [0.9, 0.1, -0.3], [-0.2, 0.85, 0.55], [0.0, 0.0, 0.0], [0.1, 0.1, 0.1])
    ru = om.evaluate(c[:, 0], c[:, 1], c[:, 2], want_v=True)
    assert nerr([fvu[0], fvu[2]], ru["f"]) < 1e-10 and nerr([fvu[1], fvu[3]], ru["v"]) < 1e-10


def test_plain_c_caller_of_the_c_abi(gpu, tmp_path):
    """include/gpx.h is C99: examples/c_abi_example.c (create, evaluate, project, update from plain C) builds with
    gcc -std=c99 -pedantic and runs against libgpx.so."""
    exe = str(tmp_path / "c_abi_example")
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", os.path.join(ROOT, "examples", "c_abi_example.c"),
           "-I", os.path.join(ROOT, "include"), "-L", os.path.join(PKG, "lib"), "-lgpx",
           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "projected to" in r.stdout and "(status 1)" in r.stdout and "after update (+16 points)" in r.stdout


def test_eigen_typed_overloads_run(gpu, tmp_path):
    """evaluate(gp, query, f, v, MatrixXd &N [, Tx, Ty]) and computeTangentBasis(Vector3d...) of the header
    shim, compiled and RUN against their std::vector twins.  With the real Eigen where it is installed; this image has
    none, so here the adapters are built against tests/cpp/eigen_iface -- an interface stand-in with the few
    members they touch, which is not Eigen, is on no other include path and takes no part in any reference build."""
    inc = None
    for cand in ("/usr/include/eigen3", "/usr/local/include/eigen3"):
        if os.path.exists(os.path.join(cand, "Eigen", "Core")):
            inc = cand
    which = "Eigen" if inc else "interface stand-in"
    # (round 4: the GPU boxes were probed -- scripts/r4 first call, bench.py's `eigen_on_box` record: no Eigen there either,
    # so the oracle's LDL^T stays pinned by NumPy / LAPACK fixtures only and this row stays on the stand-in)
    print("eigen_on_box: %s" % ({"present": bool(inc), "path": inc},))
    inc = inc or os.path.join(ROOT, "tests", "cpp", "eigen_iface")
    exe = str(tmp_path / "eigen_adapters")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", inc, "-I", os.path.join(PKG, "include"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "eigen_adapters.cpp"), "-o", exe, "-L", os.path.join(PKG, "lib"), "-lgpx",
           "-Wl,-rpath," + os.path.join(PKG, "lib"), "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, which + ": " + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "eigen_adapters: OK" in r.stdout, which + ": " + r.stdout + r.stderr

