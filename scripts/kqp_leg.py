"""The operand kernel of a variance batch alone: 20 back-to-back launches through gpx_dev_kqp (operand formed in fp64 from
fp64 points: what thin-plate models use) and gpx_dev_kqp_f32 (fp32 arithmetic on centred fp32 points: the exponential
kernels), N = 16384, 8192 queries, with the fit array and without.  GPX_LIB selects a library variant; the store pattern is
an environment switch read at load time (GPX_PAIR_WIDE=1: 1-KiB row segments per wave store; GPX_PAIR_NT=1: non-temporal
stores), so run one process per variant.  Prints us per launch and TB/s.
Usage: python scripts/kqp_leg.py [kernel]"""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
L = gpx.lib()
kn = sys.argv[1] if len(sys.argv) > 1 else "matern52"
n, npad, qb = 16384, 16384, 8192
dev = torch.device("cuda:0")
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
pts = [torch.from_numpy(a).to(dev).contiguous() for a in (x, y, z)]
cen = torch.tensor([x.mean(), y.mean(), z.mean(), 0, 0, 0, 0, 0], dtype=torch.float64, device=dev)
pts32 = [(p - cen[i]).float().contiguous() for i, p in enumerate(pts)]
q = [torch.linspace(-1, 1, qb, dtype=torch.float64, device=dev) for _ in range(3)]
fab = torch.full((3 * qb,), 0.1, dtype=torch.float64, device=dev)
Kq = torch.empty(qb * npad, dtype=torch.float32, device=dev)
kern = gpx.make_kernel(kn, 4.0) if kn == "thinplate" else gpx.make_kernel(kn, 1.0, 1.0)
strm = torch.cuda.current_stream()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tag = "wide=%s nt=%s" % (os.environ.get("GPX_PAIR_WIDE", "-"), os.environ.get("GPX_PAIR_NT", "-"))
for form in ("fp64-formed", "fp32-formed"):
    for name, f in (("with fit", fab), ("plain", None)):
        if form == "fp64-formed":
            fn = lambda: gpx._check(L.gpx_dev_kqp(C.byref(kern), gpx.F32, n, npad, vp(pts[0]), vp(pts[1]), vp(pts[2]), qb, vp(q[0]),
                                                 vp(q[1]), vp(q[2]), vp(f), vp(Kq), C.c_void_p(strm.cuda_stream)))
        else:
            fn = lambda: gpx._check(L.gpx_dev_kqp_f32(C.byref(kern), n, npad, vp(pts32[0]), vp(pts32[1]), vp(pts32[2]), vp(cen), qb,
                                                     vp(q[0]), vp(q[1]), vp(q[2]), vp(f), vp(Kq), C.c_void_p(strm.cuda_stream)))
        fn(); torch.cuda.synchronize()
        e0.record(strm)
        for _ in range(20):
            fn()
        e1.record(strm); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%-10s %-12s %-9s [%s] %.1f us per launch  %.2f TB/s" % (kn, form, name, tag, ms * 1e3, qb * npad * 4 / ms / 1e9), flush=True)
