"""Small models in the split-fp16 mode: the fp16 matrix-core kernel (csrc/gpx_varcols16.hip) against its twin (GPX_VAR_COLS16=0:
the fp32 small-model kernel) and the fp64 pipeline -- variance error max|dv| / max|v_ref| and the variance stage's time on 2^19
lattice queries.  Usage: python scripts/var16_check.py [sizes...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda:0")
g = 80
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(0, 2 ** 19, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = int(idx.numel())
f = torch.empty(nq, dtype=torch.float64, device=dev)
sizes = [int(a) for a in sys.argv[1:]] or [166, 277, 512, 724, 1024]
for kn in ("matern52", "gaussian"):
    for n in sizes:
        data = ds.fibonacci_training_set(n)
        kern = gpx.make_kernel(kn, 1.0, 1.0)
        m64 = gpx.Model(kern, *data, precision=gpx.F64, prepare_variance=True)
        vref = torch.empty(nq, dtype=torch.float64, device=dev)
        m64.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), vref.data_ptr()); m64.sync(); m64.close()
        m = gpx.Model(kern, *data, precision=gpx.F32_SPLIT, prepare_variance=True)
        out = []
        for on in ("1", "0"):
            os.environ["GPX_VAR_COLS16"] = on
            gpx.debug_reload()
            v = torch.empty(nq, dtype=torch.float64, device=dev)
            for _ in range(4):
                m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
            st = m.stats
            err = float((v - vref).abs().max() / vref.abs().max())
            out.append((err, st["t_var_ms"], st["t_var_gemm_ms"]))
        m.close()
        print("%-9s N=%5d: fp16 kernel err %.2e stage %.3f ms (kernel %.3f) | fp32 kernel err %.2e stage %.3f ms (kernel %.3f) | speed-up %.2f" % (
            kn, n, out[0][0], out[0][1], out[0][2], out[1][0], out[1][1], out[1][2], out[1][1] / out[0][1]), flush=True)

# the reference's own clouds (tests/golden/pcd): node training sets, Gaussian(1,1), split mode against the fp64 pipeline
names = ["bowlA", "bowlB", "containerA", "containerB", "jug", "kettle", "pot", "mugD", "kitchenUtensilB"]
worst = 0.0
for nm in names:
    path = os.path.join(ROOT, "tests", "golden", "pcd", nm + ".pcd")
    if not os.path.exists(path):
        continue
    data = gpx.node_training_set(gpx.pcd_read(path))
    kern = gpx.make_kernel("gaussian", 1.0, 1.0)
    m64 = gpx.Model(kern, *data, precision=gpx.F64, prepare_variance=True)
    vref = torch.empty(nq, dtype=torch.float64, device=dev)
    m64.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), vref.data_ptr()); m64.sync(); m64.close()
    os.environ["GPX_VAR_COLS16"] = "1"
    gpx.debug_reload()
    m = gpx.Model(kern, *data, precision=gpx.F32_SPLIT, prepare_variance=True)
    v = torch.empty(nq, dtype=torch.float64, device=dev)
    m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
    err = float((v - vref).abs().max() / vref.abs().max())
    worst = max(worst, err)
    print("%-11s N=%4d: split-mode variance error %.2e of max v" % (nm, m.stats["n"], err), flush=True)
    m.close()
print("worst over the reference's clouds: %.2e (VERDICT r4 item 6 asks for <= 5e-6)" % worst)
