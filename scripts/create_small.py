import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
for n in (277, 724):
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpx.make_kernel("gaussian", 1.0, 1.0)
    for pv in (False, True):
        ws = []
        for rep in range(12):
            t = time.perf_counter()
            gm = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F32, prepare_variance=pv)
            w = time.perf_counter() - t
            st = gm.stats
            t2 = time.perf_counter()
            gm.close()
            w2 = time.perf_counter() - t2
            ws.append((w, w2))
        ws = sorted(ws)[:6]
        print("N=%d prepare_variance=%s: create wall %.3f ms (min), close %.3f ms; device stages kbuild %.3f LDLT %.3f alpha %.3f inverse %.3f = %.3f ms" % (
            n, pv, ws[0][0]*1e3, ws[0][1]*1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], st["t_inverse_ms"],
            st["t_kbuild_ms"] + st["t_factor_ms"] + st["t_solve_ms"] + st["t_inverse_ms"]), flush=True)
