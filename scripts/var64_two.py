"""Small fp64 models: the one-wave-per-SIMD kernel of round 5 (GPX_VAR_COLS64=1) against the two-waves-per-SIMD form (the default,
csrc/gpx_varcols64.hip: VC64Two), 2^21 queries, Gaussian(1,1) and Matern-5/2 on the Fibonacci cloud: kernel ms, fraction of the
fp64 MFMA peak on the algorithmic triangle, and the largest difference of v and f between the two (relative to max |v|, |f|).
Usage: python scripts/var64_two.py [sizes...]"""
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
sizes = [int(a) for a in sys.argv[1:]] or [64, 166, 277, 320, 321, 352, 512, 724, 900, 992]
print("%10s %6s %30s %30s %8s %10s %10s" % ("kernel", "N", "one wave / SIMD: ms (of peak)", "two waves / SIMD: ms (of peak)", "ratio", "dv", "df"))
for kn in ("gaussian", "matern52"):
    kern = gpx.make_kernel(kn, 1.0, 1.0)
    for n in sizes:
        m = gpx.Model(kern, *ds.fibonacci_training_set(n), precision=gpx.F64, prepare_variance=True)
        row, res = [], []
        for mode in ("1", None):
            with gpx.switches(GPX_VAR_COLS64=mode):
                ts = []
                for i in range(4):
                    m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
                    m.sync()
                    if i:
                        ts.append(m.stats["t_var_gemm_ms"])
                row.append(sum(ts) / len(ts))
                res.append((f.clone(), v.clone()))
        F = (n + 15) // 16
        flop = 2.0 * nq * 256 * F * (F + 1) / 2
        dv = float((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max())
        df = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max())
        print("%10s %6d %20.3f (%4.1f %%) %20.3f (%4.1f %%) %8.3f %10.1e %10.1e" %
              (kn, n, row[0], 100 * flop / (row[0] * 1e-3) / 78.6e12, row[1], 100 * flop / (row[1] * 1e-3) / 78.6e12, row[0] / row[1], dv, df), flush=True)
        m.close()
