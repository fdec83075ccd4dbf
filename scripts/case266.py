"""The thread-0 case of test_independent_models_from_concurrent_threads, single-threaded: N = 266 thin-plate R = 2."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
import gp_oracle as orc
for n, seed in ((266, 100), (277, 107), (447, 104)):
    x, y, z, lab, s2 = ds.fibonacci_training_set(n, seed=seed)
    qx, qy, qz = ds.query_grid(9)
    om = orc.Model(orc.make_kernel("thinplate", 2.0), x, y, z, lab, s2)
    ref = om.evaluate(qx, qy, qz, want_v=True)
    for prec, pn in ((gpx.F64, "F64"), (gpx.F32, "F32"), (gpx.MIXED, "MIXED"), (gpx.F32_SPLIT, "SPLIT")):
        gm = gpx.Model(gpx.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=prec)
        o = gm.evaluate(qx, qy, qz, want_v=True)
        e = np.abs(o["v"] - ref["v"])
        print(n, pn, "neg", gm.stats["n_negative_pivots"], "verr %.2e" % (e.max() / max(8.0, np.abs(ref["v"]).max())), "max|v| %.3g" % np.abs(ref["v"]).max(), "minD %.3g" % np.abs(gm.D).min(), flush=True)
        gm.close()
