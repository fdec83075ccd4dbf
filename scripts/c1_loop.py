"""The C1 step of bench.py (mugD node training set, Gaussian(1,1), fp64, 32^3 grid: create + evaluate_device + sync + close)
in a loop, for host-API traces: python scripts/c1_loop.py [reps] [f64|f32]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
prec = gpx.F32 if len(sys.argv) > 2 and sys.argv[2] == "f32" else gpx.F64
dev = torch.device("cuda", 0)
data = gpx.node_training_set(gpx.pcd_read(os.path.join(ROOT, "tests", "golden", "pcd", "mugD.pcd")))
g = 32
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(0, g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
kern = gpx.make_kernel("gaussian", 1.0, 1.0)
def step(parts=None):
    t0 = time.perf_counter()
    m = gpx.Model(kern, *data, precision=prec, prepare_variance=True)
    t1 = time.perf_counter()
    m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
    t2 = time.perf_counter()
    m.sync()
    t3 = time.perf_counter()
    st = m.stats
    t4 = time.perf_counter()
    m.close()
    t5 = time.perf_counter()
    if parts is not None:
        parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
    return st
for _ in range(5):
    step()
torch.cuda.synchronize()
parts = []
t0 = time.perf_counter()
for _ in range(reps):
    st = step(parts)
dt = (time.perf_counter() - t0) / reps
import numpy as np
p = np.mean(np.array(parts), axis=0) * 1e3
print("C1 step %.3f ms: create %.3f  evaluate_device (enqueue) %.3f  sync %.3f  stats %.3f  close %.3f | device: factor %.3f solve %.3f mean %.3f var %.3f" %
      (dt * 1e3, p[0], p[1], p[2], p[3], p[4], st["t_factor_ms"], st["t_solve_ms"], st["t_mean_ms"], st["t_var_ms"]))
