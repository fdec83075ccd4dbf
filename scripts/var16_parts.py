import importlib, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda:0")
g = 80
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(0, 2 ** 19, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = int(idx.numel())
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
for n in (277, 724, 1024):
    m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=gpx.F32_SPLIT, prepare_variance=True)
    for _ in range(4):
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
    print("%s N=%5d kernel %.3f ms" % (sys.argv[1], n, m.stats["t_var_gemm_ms"]), flush=True)
    m.close()
