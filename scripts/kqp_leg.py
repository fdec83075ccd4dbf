"""kqp_kernel<f32> alone: 20 back-to-back launches through gpx_dev_kqp (N = 16384, 8192 queries), with and without a fit
array; GPX_LIB selects a library variant.  Prints ms per launch and TB/s."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
L = gpx.lib()
n, npad, qb = 16384, 16384, 8192
dev = torch.device("cuda:0")
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
pts = [torch.from_numpy(a).to(dev).contiguous() for a in (x, y, z)]  # fp64 points (round 3: the operand is formed in fp64)
q = [torch.linspace(-1, 1, qb, dtype=torch.float64, device=dev) for _ in range(3)]
fab = torch.full((3 * qb,), 0.1, dtype=torch.float64, device=dev)
Kq = torch.empty(qb * npad, dtype=torch.float32, device=dev)
kern = gpx.make_kernel("matern52", 1.0, 1.0)
strm = torch.cuda.current_stream()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, f in (("with fit array", fab), ("fab = NULL", None)):
    fn = lambda: gpx._check(L.gpx_dev_kqp(C.byref(kern), gpx.F32, n, npad, vp(pts[0]), vp(pts[1]), vp(pts[2]), qb, vp(q[0]), vp(q[1]),
                                         vp(q[2]), vp(f), vp(Kq), C.c_void_p(strm.cuda_stream)))
    fn(); torch.cuda.synchronize()
    e0.record(strm)
    for _ in range(20):
        fn()
    e1.record(strm); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("%s %-16s %.1f us per launch  %.2f TB/s" % (os.path.basename(gpx.LIB_PATH), name, ms * 1e3, qb * npad * 4 / ms / 1e9))
