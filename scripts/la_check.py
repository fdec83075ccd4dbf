"""Repeated create() at N = 16384 (wall + stage times) for timelines under rocprofv3 --kernel-trace."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
prec = {"f32": gpx.F32, "f64": gpx.F64}[sys.argv[2] if len(sys.argv) > 2 else "f32"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
kern = gpx.make_kernel("matern52", 1.0, 1.0)
for rep in range(reps):
    t = time.perf_counter()
    gm = gpx.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=True)
    wall = time.perf_counter() - t
    st = gm.stats
    import hashlib
    digest = hashlib.sha1(gm.alpha.tobytes() + gm.D.tobytes()).hexdigest()[:12]
    gm.close()
    print("create %.2f ms wall; kbuild %.2f LDL^T %.2f (GEMM %.2f) alpha %.2f inverse %.2f  sha1(alpha, D) %s" % (
        wall * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_factor_gemm_ms"], st["t_solve_ms"], st["t_inverse_ms"], digest), flush=True)
