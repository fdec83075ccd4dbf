"""C3_surface under the profiler: the headline model (N = 16384 fp32 Matern-5/2) + gpx_model_sample_surface over the 128^3 lattice,
three timed calls after one of the same size (rocprofv3 --kernel-trace --stats -- python3 scripts/surface_profile.py)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n, G = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 128
data = ds.fibonacci_training_set(n)
tt = np.linspace(-1.01, 1.01, G)
gx, gy, gz = np.meshgrid(tt, tt, tt, indexing="ij")
qx, qy, qz = gx.ravel(), gy.ravel(), gz.ravel()
m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *data, precision=gpx.F32, prepare_variance=True)
m.sync()
for rep in range(4):
    t0 = time.perf_counter()
    o = m.sample_surface(qx, qy, qz, f_tol=0.01)
    dt = time.perf_counter() - t0
    st = m.stats
    print("call %d: %.2f ms, survivors %d, fp64-mean candidates %d of %d, variance %.2f ms, survivor mean %.2f ms" %
          (rep, dt * 1e3, o["n_total"], st["surface_candidates"], qx.size, st["t_var_ms"], st["t_mean_ms"]), flush=True)
m.close()
