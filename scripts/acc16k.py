import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
qx, qy, qz = ds.query_grid(16)
for kn, par, k0 in (("matern52", (1.0, 1.0), 1.0), ("gaussian", (1.0, 1.0), 1.0), ("thinplate", (4.0,), 64.0)):
    kern = gpx.make_kernel(kn, *par)
    g64 = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F64)
    o64 = g64.evaluate(qx, qy, qz, want_v=True); g64.close()
    for prec, name in ((gpx.F32, "F32"), (gpx.MIXED, "MIXED"), (gpx.F32_SPLIT, "SPLIT")):
        g = gpx.Model(kern, x, y, z, lab, s2, precision=prec)
        o = g.evaluate(qx, qy, qz, want_v=True); g.close()
        e = np.abs(o["v"] - o64["v"])
        vmax = np.abs(o64["v"]).max()
        print("%-9s N=%d %-6s e_k0=%.2e  e_v=%.2e (max|dv| / max|v_ref|, max|v_ref| = %.3g)  rms/k0 %.2e  ferr=%.1e" % (
            kn, n, name, e.max() / max(k0, vmax), e.max() / vmax, vmax, np.sqrt((e**2).mean()) / k0,
            np.abs(o["f"] - o64["f"]).max() / np.abs(o64["f"]).max()), flush=True)
