"""Three-launch create of small models (gpx_small.hip) against its twin, the general chain (GPX_DATAFLOW=0), and the
oracle: alpha, D, R, f, v at several sizes / kernels / precisions, and the wall time of create().
Usage: python scripts/small_create_check.py [reps]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
import gp_oracle as orc
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
qx, qy, qz = ds.query_grid(9)

def rel(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

def build(kern, data, prec, small):
    os.environ.pop("GPX_DATAFLOW", None) if small else os.environ.__setitem__("GPX_DATAFLOW", "0")
    gpx.debug_reload()
    m = gpx.Model(kern, *data, precision=prec, prepare_variance=True)
    return m

worst = 0.0
for n in (17, 64, 65, 166, 277, 300, 512, 513, 724, 1000, 1024):
    for kn, par in (("gaussian", (1.0, 1.0)), ("matern52", (1.0, 1.0)), ("thinplate", (4.0,)), ("thinplate", (2.0,))):
        data = ds.fibonacci_training_set(n)
        om = orc.Model(orc.make_kernel(kn, *par), *data)
        ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        for prec, tol in ((gpx.F64, 1e-9), (gpx.F32, 1e-5)):
            ms = build(gpx.make_kernel(kn, *par), data, prec, True)
            mc = build(gpx.make_kernel(kn, *par), data, prec, False)
            os_, oc = ms.evaluate(qx, qy, qz, want_v=True, want_grad=True), mc.evaluate(qx, qy, qz, want_v=True, want_grad=True)
            st = ms.stats
            e = {"alpha": rel(ms.alpha, om.alpha), "alpha_tw": rel(ms.alpha, mc.alpha), "D_tw": rel(ms.D, mc.D),
                 "R": abs(ms.R - om.R) / om.R, "f": rel(os_["f"], ref["f"]), "v": rel(os_["v"], ref["v"]), "g": rel(os_["grad"], ref["grad"]),
                 "v_tw": rel(os_["v"], oc["v"]), "f_tw": rel(os_["f"], oc["f"])}
            bad = e["f"] > tol or e["v"] > tol or e["g"] > tol or e["R"] > 1e-13 or st["solve_fallbacks"] != 0 or \
                st["n_negative_pivots"] != mc.stats["n_negative_pivots"]
            worst = max(worst, e["v"] / tol, e["f"] / tol)
            print("n=%4d %-9s %-4s prec=%d neg=%d ir=%d fb=%d %s %s" % (n, kn, par[0], prec, st["n_negative_pivots"], st["ir_steps_done"],
                  st["solve_fallbacks"], " ".join("%s=%.1e" % kv for kv in e.items()), "BAD" if bad else ""), flush=True)
            ms.close(); mc.close()
print("worst error / tolerance: %.3f" % worst)
# wall time of create + destroy (pool warm), both paths
for n in (277, 512, 724, 1024):
    data = ds.fibonacci_training_set(n)
    for prec in (gpx.F64, gpx.F32):
        line = "create+destroy n=%4d prec=%d:" % (n, prec)
        for small in (True, False):
            kern = gpx.make_kernel("gaussian", 1.0, 1.0)
            for _ in range(3):
                build(kern, data, prec, small).close()
            t0 = time.perf_counter()
            for _ in range(reps):
                m = build(kern, data, prec, small)
                st = m.stats
                m.close()
            dt = (time.perf_counter() - t0) / reps
            line += "  %s %.3f ms (factor %.3f solve %.3f inv %.3f kbuild %.3f)" % ("small" if small else "chain", dt * 1e3, st["t_factor_ms"],
                    st["t_solve_ms"], st["t_inverse_ms"], st["t_kbuild_ms"])
        print(line, flush=True)
