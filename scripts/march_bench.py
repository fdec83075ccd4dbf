"""The node's surface-following sampler (src/gp_node.cpp:258: marchingSampling(false, 0.06, 0.02)) on the node's own
N = 277 model of resources/mugD.pcd and on a synthetic N = 4096 sphere: gpx_model_march_surface (one device batch per
frontier) against the reference's call pattern, one single-point evaluate(f, v) per lattice point."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
golden = np.load(os.path.join(ROOT, "tests", "golden", "gp_golden.npz"))
cases = [("mugD N=277 thinplate(2.0) f64", tuple(golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2")), ("thinplate", (2.0,)), gpx.F64),
         ("mugD N=277 thinplate(2.0) f32", tuple(golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2")), ("thinplate", (2.0,)), gpx.F32),
         ("sphere N=4096 matern52 f32", ds.fibonacci_training_set(4096), ("matern52", (1.0, 1.0)), gpx.F32)]
for name, (x, y, z, lab, s2), (kn, par), prec in cases:
    gm = gpx.Model(gpx.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec, prepare_variance=True)
    for leaf, step in ((0.06, 0.02), (0.15, 0.03)):
        gm.march_surface(leaf, step)  # warm-up (workspaces)
        t = time.perf_counter()
        r = gm.march_surface(leaf, step)
        dt = time.perf_counter() - t
        per_cube = (round(leaf / step) + 1) ** 3
        print("%-34s leaf %.2f pass %.2f: %6d cubes, %8d lattice points, %7d kept: %8.2f ms" % (
            name, leaf, step, r["n_cubes"], r["n_cubes"] * per_cube, r["n_total"], dt * 1e3), flush=True)
    # the reference's pattern on the first 2000 lattice points of the walk: one evaluate(f, v) call per point
    pts = r["xyz"][:2000]
    t = time.perf_counter()
    for p in pts:
        gm.evaluate(p[0:1], p[1:2], p[2:3], want_v=True)
    dt1 = (time.perf_counter() - t) / len(pts)
    print("%-34s single-point evaluate(f, v) calls: %.1f us per call -> %.1f s for the %d lattice points of the last walk" % (
        name, dt1 * 1e6, dt1 * r["n_cubes"] * per_cube, r["n_cubes"] * per_cube), flush=True)
    gm.close()
