"""update(): rank-n append vs refactorisation from scratch (GPX_UPDATE_APPEND=0), wall time through the C ABI."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
for n0, n1, prec, pname in ((16384 - 256, 256, gpx.F32, "f32"), (4096 - 64, 64, gpx.F64, "f64"), (277, 16, gpx.F64, "f64")):
    x, y, z, lab, s2 = ds.fibonacci_training_set(n0 + n1)
    kern = gpx.make_kernel("matern52", 1.0, 1.0)
    for mode in ("1", "0"):
        os.environ["GPX_UPDATE_APPEND"] = mode
        gpx.debug_reload()
        ts = []
        for rep in range(3):
            gm = gpx.Model(kern, x[:n0], y[:n0], z[:n0], lab[:n0], s2[:n0], precision=prec)
            t = time.perf_counter()
            gm.update(x[n0:], y[n0:], z[n0:], lab[n0:], s2[n0:])
            ts.append(time.perf_counter() - t)
            st = gm.stats
            gm.close()
        if mode == "1":  # with the inverse factor present before the update: it is extended by the new rows
            qx, qy, qz = ds.query_grid(4)
            best = None
            for rep in range(3):
                gm = gpx.Model(kern, x[:n0], y[:n0], z[:n0], lab[:n0], s2[:n0], precision=prec, prepare_variance=True)
                t = time.perf_counter()
                gm.update(x[n0:], y[n0:], z[n0:], lab[n0:], s2[n0:])
                t_upd = time.perf_counter() - t
                t_inv = gm.stats["t_inverse_ms"]
                t = time.perf_counter()
                gm.evaluate(qx, qy, qz, want_v=True)
                t_q = time.perf_counter() - t
                gm.close()
                if best is None or t_upd < best[0]:
                    best = (t_upd, t_inv, t_q)
            print("N %5d + %3d %s  append, inverse factor extended: update %.2f ms wall (inverse rows %.2f ms), next variance query %.2f ms" % (
                n0, n1, pname, best[0] * 1e3, best[1], best[2] * 1e3), flush=True)
        print("N %5d + %3d %s  %-8s: update %.2f ms wall (device: kbuild %.2f factor %.2f solve %.2f), alpha residual %.1e" % (
            n0, n1, pname, "append" if mode == "1" else "rebuild", min(ts) * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], st["alpha_residual"]), flush=True)
