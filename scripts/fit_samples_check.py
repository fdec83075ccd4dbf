"""Accuracy (N = 16384, both kernels) and cost (C5-shaped model) of the per-query fit against its number of samples
(GPX_VAR_FIT_SAMPLES, read once per process: run one process per value)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
ns = os.environ.get("GPX_VAR_FIT_SAMPLES", "default")
n = 16384
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
qx, qy, qz = ds.query_grid(16)
for kn, par in (("matern52", (1.0, 1.0)), ("thinplate", (4.0,))):
    kern = gpx.make_kernel(kn, *par)
    g64 = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F64)
    o64 = g64.evaluate(qx, qy, qz, want_v=True); g64.close()
    g = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F32)
    o = g.evaluate(qx, qy, qz, want_v=True); g.close()
    e = np.abs(o["v"] - o64["v"]).max() / np.abs(o64["v"]).max()
    print("samples %-8s %-9s N=16384 F32 e_v = %.2e" % (ns, kn, e), flush=True)
data = gpx.node_training_set(gpx.pcd_read(os.path.join(ROOT, "tests", "golden", "pcd", "containerB.pcd")))
gq = ds.query_grid(64)
for kn, par in (("gaussian", (1.0, 1.0)), ("thinplate", (4.0,))):
    kern = gpx.make_kernel(kn, *par)
    m64 = gpx.Model(kern, *data, precision=gpx.F64)
    r64 = m64.evaluate(*gq, want_v=True); m64.close()
    m = gpx.Model(kern, *data, precision=gpx.F32, prepare_variance=True)
    m.evaluate(*gq, want_v=True)
    t0 = time.perf_counter()
    r = m.evaluate(*gq, want_v=True)
    dt = time.perf_counter() - t0
    st = m.stats; m.close()
    print("samples %-8s %-9s containerB N=%d 64^3 F32: e_v = %.2e   t_var %.2f ms (gemm %.2f, kqp %.2f)" % (
        ns, kn, st["n"], np.abs(r["v"] - r64["v"]).max() / np.abs(r64["v"]).max(), st["t_var_ms"], st["t_var_gemm_ms"], st["t_var_kqp_ms"]), flush=True)
