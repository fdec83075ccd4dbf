"""Small-model variance path probe: evaluate(f, v) on 2^19 lattice queries, Matern-5/2 fp32 models on the Fibonacci cloud.
Prints ms per evaluate, the variance kernel's time and its share of the fp32 MFMA peak on N^2 flop per query."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda:0")
g = 80
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(0, 2 ** 19, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = int(idx.numel())
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
sizes = [int(a) for a in sys.argv[1:]] or [277, 512, 724, 1024]
for n in sizes:
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), x, y, z, lab, s2, precision=gpx.F32, prepare_variance=True)
    best = None
    for _ in range(5):
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
        st = m.stats
        if best is None or st["t_var_ms"] < best["t_var_ms"]:
            best = dict(st)
    tf = float(n) ** 2 * nq / (best["t_var_gemm_ms"] * 1e-3) / 1e12
    tv = float(n) ** 2 * nq / (best["t_var_ms"] * 1e-3) / 1e12
    print("N=%5d: evaluate %.3f ms | mean %.3f var %.3f (kernel %.3f = %.1f TFLOP/s = %.1f %% of 157.3; kqp %.3f; whole variance stage %.1f %%)" % (
        n, best["t_mean_ms"] + best["t_var_ms"], best["t_mean_ms"], best["t_var_ms"], best["t_var_gemm_ms"], tf, 100 * tf / 157.3,
        best["t_var_kqp_ms"], 100 * tv / 157.3), flush=True)
    m.close()
