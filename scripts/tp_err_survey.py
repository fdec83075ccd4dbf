"""Thin-plate variance error of the fp32-family modes against the fp64 pipeline, on the shapes the tests use
(mugD N = 277 R = 2 / 4, Fibonacci N = 300 ... 16384 R = 4), under both normalisations:
  e_k0  = max|v - v64| / max(max|v64|, k(0))     (tests/conftest.py:verr)
  e_v   = max|v - v64| / max|v64|
Usage: python scripts/tp_err_survey.py [max_n]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
max_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
golden = np.load(os.path.join(ROOT, "tests", "golden", "gp_golden.npz"))


def queries(x, y, z, g):
    qx, qy, qz = ds.query_grid(g)
    return (np.concatenate([qx, x[:9], [3.0]]), np.concatenate([qy, y[:9], [0.1]]), np.concatenate([qz, z[:9], [-2.0]]))


cases = []
mug = tuple(golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
cases.append(("mugD277 R=2", mug, 2.0, 7))
cases.append(("mugD277 R=4", mug, 4.0, 7))
cases.append(("mugD277 R=5.5", mug, 5.5, 6))
for n in (300, 600, 1500, 4096, 16384):
    if n <= max_n:
        cases.append(("fib%d R=4" % n, ds.fibonacci_training_set(n), 4.0, 5 if n < 4096 else 16))
cases.insert(3, ("fib300 R=2", ds.fibonacci_training_set(300), 2.0, 6))
for name, (x, y, z, lab, s2), R, g in cases:
    kern = gpx.make_kernel("thinplate", R)
    k0 = R ** 3
    q = queries(x, y, z, g) if len(x) < 16384 else ds.query_grid(g)
    g64 = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F64)
    o64 = g64.evaluate(*q, want_v=True)
    neg = g64.stats["n_negative_pivots"]
    g64.close()
    vmax = np.abs(o64["v"]).max()
    line = "%-14s neg=%d max|v|=%.3g k0=%g :" % (name, neg, vmax, k0)
    for prec, pn in ((gpx.F32, "F32"), (gpx.MIXED, "MIXED"), (gpx.F32_SPLIT, "SPLIT")):
        gm = gpx.Model(kern, x, y, z, lab, s2, precision=prec)
        o = gm.evaluate(*q, want_v=True)
        gm.close()
        e = np.abs(o["v"] - o64["v"]).max()
        line += "  %s e_k0=%.2e e_v=%.2e" % (pn, e / max(vmax, k0), e / vmax)
    print(line, flush=True)
