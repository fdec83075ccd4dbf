"""create() stage times (device events) for the three precisions at a few sizes, through the C ABI."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
cases = [(16384, gpx.F32, "f32"), (16384, gpx.F64, "f64"), (4096, gpx.F64, "f64"), (277, gpx.F64, "f64"), (277, gpx.F32, "f32")]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] == int(sys.argv[1])]
for n, prec, pname in cases:
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpx.make_kernel("matern52", 1.0, 1.0)
    best = None
    for rep in range(3):
        t = time.perf_counter()
        gm = gpx.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=True)
        wall = time.perf_counter() - t
        st = gm.stats
        gm.close()
        if best is None or wall < best[0]:
            best = (wall, st)
    wall, st = best
    print("N %5d %s: create %.2f ms wall; device: kbuild %.2f  LDL^T %.2f (GEMM %.2f)  alpha %.2f  inverse %.2f   residual %.1e" % (
        n, pname, wall * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_factor_gemm_ms"], st["t_solve_ms"], st["t_inverse_ms"],
        st["alpha_residual"]), flush=True)
