"""End-to-end on one of the reference's point clouds, the way its ROS node does it (src/gp_node.cpp:528-585 load,
:85-117 normalise, :821-849 exterior sphere, :916-922 create, :998-1100 sample the grid and keep |f| <= 0.01):

    python examples/reconstruct_pcd.py tests/golden/pcd/mugD.pcd [grid_per_axis=64] [f32|f64|mixed|f32split]

Needs an MI355X (libgpx has no CPU path).  Prints the stage timings and writes the surface samples
(x y z variance) next to the input as <name>.surface.xyzv."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "pcd", "mugD.pcd")
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    prec = {"f32": gpx.F32, "f64": gpx.F64, "mixed": gpx.MIXED, "f32split": gpx.F32_SPLIT}[sys.argv[3] if len(sys.argv) > 3 else "f64"]
    xyz = gpx.pcd_read(path)                                  # loadPCDFile
    x, y, z, label, sigma2 = gpx.node_training_set(xyz)       # deMeanAndNormalizeData + prepareExtData + prepareData
    print("%s: %d points -> %d training points (15 exterior)" % (os.path.basename(path), len(xyz), len(x)))
    t = time.perf_counter()
    gp = gpx.Model(gpx.make_kernel("thinplate", 2.0), x, y, z, label, sigma2, precision=prec)   # the node's kernel
    t_create = time.perf_counter() - t
    qx, qy, qz = ds.query_grid(grid)                           # [-1.01, 1.01]^3 lattice
    t = time.perf_counter()
    s = gp.sample_surface(qx, qy, qz, f_tol=0.01)              # mean everywhere, variance on the survivors
    t_sample = time.perf_counter() - t
    st = gp.stats
    print("create %.2f ms (kernel matrix %.3f, LDL^T %.3f, weights %.3f ms on the device; %d negative pivots)" % (
        t_create * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], st["n_negative_pivots"]))
    print("%d^3 = %d grid points sampled in %.2f ms: %d on the surface (|f| <= 0.01), variance %.4f .. %.4f" % (
        grid, len(qx), t_sample * 1e3, len(s["idx"]), s["v"].min() if len(s["v"]) else 0, s["v"].max() if len(s["v"]) else 0))
    out = os.path.splitext(path)[0] + ".surface.xyzv"
    try:
        np.savetxt(out, np.stack([qx[s["idx"]], qy[s["idx"]], qz[s["idx"]], s["v"]], 1), fmt="%.6f")
        print("wrote", out)
    except OSError as e:
        print("not written:", e)
    gp.close()


if __name__ == "__main__":
    main()
