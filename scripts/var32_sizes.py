"""Variance stage and contraction kernel of fp32-mode small models over the model size (2^19 lattice queries, Matern-5/2, mean of 5
evaluations after 2 warm-ups); GPX_LIB selects a library variant for A/B runs, PREC=f64 the fp64 mode (against the fp64 peak).
Usage: python scripts/var32_sizes.py [label] [sizes...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
label = sys.argv[1] if len(sys.argv) > 1 else ""
sizes = [int(a) for a in sys.argv[2:]] or [166, 277, 400, 512, 724, 1024]
dev = torch.device("cuda:0")
g = 80
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(0, 2 ** 19, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = int(idx.numel())
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
for n in sizes:
    f64 = os.environ.get("PREC") == "f64"
    m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=gpx.F64 if f64 else gpx.F32, prepare_variance=True)
    tv = tg = 0.0
    for i in range(7):
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
        st = m.stats
        if i >= 2:
            tv += st["t_var_ms"] / 5; tg += st["t_var_gemm_ms"] / 5
    peak = 78.6e12 if f64 else 157.3e12
    print("%s N=%5d: mean %.3f ms, variance stage %.3f ms = %.1f %%, contraction kernel(s) %.3f ms = %.1f %% of the %s MFMA peak on N^2 flop per query" % (
        label, n, st["t_mean_ms"], tv, 100 * 2.0 * n * n / 2 * nq / (tv * 1e-3) / peak, tg, 100 * 2.0 * n * n / 2 * nq / (tg * 1e-3) / peak,
        "fp64" if f64 else "fp32"), flush=True)
    m.close()
