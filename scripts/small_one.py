"""One small model (Matern-5/2 on the Fibonacci cloud), evaluate(f, v) on 2^21 lattice queries a few times (profiling target).
Usage: python scripts/small_one.py N f64|f32 [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n = int(sys.argv[1]); prec = gpx.F64 if sys.argv[2] == "f64" else gpx.F32; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
g = 128
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=prec, prepare_variance=True)
for _ in range(reps):
    m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
    m.sync()
print(n, sys.argv[2], m.stats["t_var_gemm_ms"])
m.close()
