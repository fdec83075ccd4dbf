import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n = int(sys.argv[1])
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
kern = gpx.make_kernel("matern52", 1.0, 1.0)
for prec, name in ((gpx.F32, "f32"), (gpx.F64, "f64")):
    for la in ("1", "0"):
        os.environ["GPX_LOOKAHEAD"] = la
        best = None
        for rep in range(3):
            gm = gpx.Model(kern, x, y, z, lab, s2, precision=prec)
            st = gm.stats
            gm.close()
            if best is None or st["t_factor_ms"] < best["t_factor_ms"]:
                best = st
        print("N %d %s lookahead=%s: LDL^T %.2f ms (GEMM %.2f) alpha %.2f" % (n, name, la, best["t_factor_ms"], best["t_factor_gemm_ms"], best["t_solve_ms"]), flush=True)
