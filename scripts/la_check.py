import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
x, y, z, lab, s2 = ds.fibonacci_training_set(16384)
kern = gpx.make_kernel("matern52", 1.0, 1.0)
for rep in range(5):
    t = time.perf_counter()
    gm = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F32)
    w = time.perf_counter() - t
    st = gm.stats
    t = time.perf_counter()
    gm.close()
    c = time.perf_counter() - t
    print("rep %d: create %.1f ms wall, LDL %.2f (gemm %.2f) alpha %.2f, close %.1f ms" % (rep, w*1e3, st["t_factor_ms"], st["t_factor_gemm_ms"], st["t_solve_ms"], c*1e3), flush=True)
