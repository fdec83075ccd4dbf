#!/bin/bash
# A/B of the diagonal-fragment skipping in the fp64 one-wave tile: parity (fp64 cases), then C2's shape (N = 4096 fp64, 64^3) and N = 16384
set -o pipefail
out=$PWD/gpurun_out/r4d; mkdir -p $out
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_dgp.py -m gpu -x -q > $out/tests64.log 2>&1 || { grep -v "^  File" $out/tests64.log | tail -30; exit 1; }
tail -n 2 $out/tests64.log
cat > /tmp/f64_sweep.py <<'PY'
import importlib, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
dev = torch.device("cuda:0")
g = 64
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
for n in (2048, 4096, 16384):
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    m = gpx.Model(gpx.make_kernel("gaussian", 1.0, 1.0), x, y, z, lab, s2, precision=gpx.F64, prepare_variance=True)
    for _ in range(2):
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr()); m.sync()
    st = m.stats
    tf = float(n) ** 2 * nq / (st["t_var_gemm_ms"] * 1e-3) / 1e12
    print("%s N=%5d fp64: variance kernel %.3f ms = %.1f TFLOP/s = %.1f %% of 78.6; variance stage %.2f ms; checksum %.17g" % (
        sys.argv[1], n, st["t_var_gemm_ms"], tf, 100 * tf / 78.6, st["t_var_ms"], float(v.sum())), flush=True)
    m.close()
PY
for s in 1 0 1 0; do GPX_VAR_DIAG_SKIP=$s timeout -k 10 200 python3 /tmp/f64_sweep.py "skip=$s" 2>&1 | grep fp64 | tee -a $out/sweep64.txt; done
