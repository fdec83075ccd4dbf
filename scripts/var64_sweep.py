"""Variance stage of fp64 models over the model size: the small-model kernel (csrc/gpx_varcols64.hip) against the general fp64
path (operand tile -> one-wave contraction tiles -> finish), 2^21 queries, Gaussian(1,1) on the Fibonacci cloud.
Usage: python scripts/var64_sweep.py [sizes...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda:0")
g = 128
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 200, 256, 277, 320, 400, 480, 512, 600, 724, 800, 900, 992]
kern = gpx.make_kernel("gaussian", 1.0, 1.0)
print("%6s %36s %40s" % ("N", "small-model kernel: stage ms (of fp64 peak)", "general path: stage ms (of peak; its kernel ms)"))
for n in sizes:
    m = gpx.Model(kern, *ds.fibonacci_training_set(n), precision=gpx.F64, prepare_variance=True)
    row = []
    for on in ("1", "0"):
        os.environ["GPX_VAR_COLS64"] = on
        gpx.debug_reload()
        for _ in range(3):
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
            m.sync()
        st = m.stats
        row.append((st["t_var_ms"], st["t_var_gemm_ms"]))
    F = (n + 15) // 16
    flop = 2.0 * nq * 256 * F * (F + 1) / 2  # the triangle of X per 16 x 16 fragment: the algorithmic work of the stage
    print("%6d %20.2f (%4.1f %%) %26.2f (%4.1f %%; %.2f)   ratio %.2f" % (n, row[0][0], 100 * flop / (row[0][0] * 1e-3) / 78.6e12, row[1][0],
                                                                       100 * flop / (row[1][0] * 1e-3) / 78.6e12, row[1][1], row[1][0] / row[0][0]), flush=True)
    m.close()
