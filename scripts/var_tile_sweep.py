"""Variance contraction by model size: evaluate(f, v) on 2^19 lattice queries for N = 277 .. 16384, fp32-mode Matern-5/2 models
on the Fibonacci cloud.  Up to 1024 points the small-model kernel (gpx_varcols_kernel.hpp), above the one-wave 128 x 128 tile
(GPX_VAR_TILE=3: its LDS-staged fallback; GPX_VAR_COLS=0: the tiles at every size); the switches are read once per
process, so this script re-runs itself per setting.  Prints ms per evaluate, the variance kernel's time and its share of
the fp32 MFMA peak on N^2 flop per query."""
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
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
    for n in (277, 512, 724, 1024, 1536, 2048, 3072, 4096, 8192, 16384):
        x, y, z, lab, s2 = ds.fibonacci_training_set(n)
        m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), x, y, z, lab, s2, precision=gpx.F32, prepare_variance=True)
        for _ in range(2):
            m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
        st = m.stats
        tf = float(n) ** 2 * nq / (st["t_var_gemm_ms"] * 1e-3) / 1e12
        print("%s N=%5d: evaluate %.2f ms  (variance kernel %.3f ms in %d launch(es) = %.1f TFLOP/s = %.1f %% of 157.3; whole variance stage %.2f ms)" % (
            sys.argv[1], n, st["t_mean_ms"] + st["t_var_ms"], st["t_var_gemm_ms"], st["var_gemm_launches"], tf, 100 * tf / 157.3, st["t_var_ms"]), flush=True)
        m.close()
else:
    for env_add, name in (({}, "default            "), ({"GPX_VAR_COLS": "0"}, "128 x 128 tiles    "), ({"GPX_VAR_COLS": "0", "GPX_VAR_TILE": "3"}, "LDS-staged fallback")):
        subprocess.run([sys.executable, os.path.abspath(__file__), name], env=dict(os.environ, **env_add), check=True)
