"""Stage times of one BASELINE C5 object (PCD cloud + 15 exterior points, N ~ 300-700, 128^3 grid, fp32 mode): where the
17 ms per object go.  Usage: python scripts/c5_stages.py [object] [kernel] [grid]"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
name = sys.argv[1] if len(sys.argv) > 1 else "containerB"
kn = sys.argv[2] if len(sys.argv) > 2 else "gaussian"
g = int(sys.argv[3]) if len(sys.argv) > 3 else 128
dev = torch.device("cuda:0")
data = gpx.node_training_set(gpx.pcd_read(os.path.join(ROOT, "tests", "golden", "pcd", name + ".pcd")))
kern = gpx.make_kernel(kn, 2.0) if kn == "thinplate" else gpx.make_kernel(kn, 1.0, 1.0)
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev)
v = torch.empty(nq, dtype=torch.float64, device=dev)
for prec, pn in ((gpx.F32, "F32"), (gpx.F64, "F64"), (gpx.F32_SPLIT, "SPLIT")):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m = gpx.Model(kern, *data, precision=prec, prepare_variance=True)
        t1 = time.perf_counter()
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
        m.sync()
        t2 = time.perf_counter()
        st = m.stats
        m.close()
    print("%s %s N=%d nq=%d %s: create %.2f ms, evaluate %.2f ms wall | mean %.2f var %.2f (gemm %.2f in %d launches, kqp %.2f) ms" % (
        name, kn, st["n"], nq, pn, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st["t_mean_ms"], st["t_var_ms"], st["t_var_gemm_ms"],
        st["var_gemm_launches"], st["t_var_kqp_ms"]), flush=True)
