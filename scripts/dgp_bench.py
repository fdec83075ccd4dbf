"""The derivative-observation GP (gpx_dgp_*, first slice of the reference's gp::GaussianProcess): create and evaluate times.
Usage: python scripts/dgp_bench.py"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
for n in (277, 1024, 4096):
    r = np.random.default_rng(n)
    d = r.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    for kn, kern in (("se(1, 0.3)", gpx.make_kernel("se", 1.0, 0.3)), ("thinplate(4)", gpx.make_kernel("thinplate", 4.0))):
        for rep in range(2):
            t0 = time.perf_counter()
            g = gpx.DerivativeGP(kern, 0.05, d[:, 0], d[:, 1], d[:, 2], np.zeros(n), d)
            t1 = time.perf_counter()
            st = g.stats
            if rep == 0:
                g.close()
        qx, qy, qz = ds.query_grid(64)
        g.evaluate(qx[:256], qy[:256], qz[:256])
        t2 = time.perf_counter()
        o = g.evaluate(qx, qy, qz)
        t3 = time.perf_counter()
        st2 = g.stats
        g.close()
        print("n=%5d (4n=%5d) %-13s create %.2f ms wall (matrix %.2f, LDL^T %.2f, alpha %.2f) | 64^3 queries f+grad+var %.1f ms wall (mean %.2f, var %.2f ms on the device) = %.2e q/s" % (
            n, 4 * n, kn, (t1 - t0) * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], (t3 - t2) * 1e3, st2["t_mean_ms"], st2["t_var_ms"], len(qx) / (t3 - t2)), flush=True)
