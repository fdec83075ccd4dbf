import importlib, os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/scripts") else os.environ.get("GRAFT_REPO_ROOT", "."))
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
for prec, name in ((gpx.F32, "fp32"), (gpx.F64, "fp64")):
    for n in (277, 724, 1024, 2048, 4096, 6144):
        data = ds.fibonacci_training_set(n)
        kern = gpx.make_kernel("matern52", 1.0, 1.0)
        tf = ts = 0.0
        for i in range(6):
            m = gpx.Model(kern, *data, precision=prec)
            st = m.stats
            m.close()
            if i:
                tf += st["t_factor_ms"] / 5; ts += st["t_solve_ms"] / 5
        print("%s N=%5d: factor %.3f ms, solve %.3f ms" % (name, n, tf, ts), flush=True)
