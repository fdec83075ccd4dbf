"""Where does the variance error of the F32-family modes come from?  Exact (fp64, torch) contraction of the F32
model's own inverse factor and pivots versus what the native and the packed-fp16 kernels return."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
qx, qy, qz = ds.query_grid(12)
kern = gpx.make_kernel("matern52", 1.0, 1.0)
dev = torch.device("cuda:0")
def kfun(d):
    t = np.sqrt(5.0) * d
    return torch.exp(-t) * (1 + t + t * t / 3)
def state(prec):
    m = gpx.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=True)
    p0, b0 = m.state_blob(0); p1, b1 = m.state_blob(1)
    t0 = sh.device_blob_as_tensor(torch, p0, b0, dev); t1 = sh.device_blob_as_tensor(torch, p1, b1, dev)
    esz = 8 if prec == gpx.F64 else 4
    npad = b0 // (4 * 8 + 9 * esz)  # fp64 x y z alpha | T x y z 1/D | T 5 correction vectors
    d64 = t0[: 4 * 8 * npad].view(torch.float64).view(4, npad)
    tt = t0[8 * 4 * npad:].view(torch.float64 if esz == 8 else torch.float32).view(9, npad)[:4]
    X = t1.view(torch.float64 if esz == 8 else torch.float32).view(npad, npad)
    return m, d64.clone(), tt.double().clone(), X.double().clone(), npad
m32, p32, t32, X32, npad = state(gpx.F32)
m64, p64, t64, X64, _ = state(gpx.F64)
Q = torch.tensor(np.stack([qx, qy, qz], 1), device=dev)
def exact_var(P, X, dinv, qcast=None):
    Pm = P[:3].T  # npad x 3 (internal order)
    Qm = Q if qcast is None else Q.to(qcast).double()
    d = torch.cdist(Qm, Pm if qcast is None else Pm.to(qcast).double())
    K = kfun(d); K[:, n:] = 0
    W = K @ X.T
    return (1.0 - (W * W * dinv[None, :]).sum(1)).cpu().numpy()
v64 = m64.evaluate(qx, qy, qz, want_v=True)["v"]
print("max|X32 - X64| = %.2e, max rel |dinv32 - dinv64| = %.2e" % ((X32 - X64).abs().max().item(), ((t32[3] - t64[3]).abs() / t64[3].abs()).max().item()))
vA = exact_var(p32, X32, t32[3])
vB = exact_var(p64, X64, t64[3])
print("F64 model, torch fp64 contraction vs F64 evaluate      : %.2e" % np.abs(vB - v64).max())
print("F32 model state, exact contraction        vs F64       : %.2e   <- floor of every F32-family mode" % np.abs(vA - v64).max())
vA32 = exact_var(t32, X32, t32[3], torch.float32)
print("F32 model state, fp32-cast points/queries vs F64       : %.2e" % np.abs(vA32 - v64).max())
for prec, name in ((gpx.F32, "F32 native"), (gpx.F32_SPLIT, "F32_SPLIT")):
    g = gpx.Model(kern, x, y, z, lab, s2, precision=prec)
    v = g.evaluate(qx, qy, qz, want_v=True)["v"]; g.close()
    print("%-10s evaluate vs F64: %.2e | vs exact contraction of the F32 state: %.2e | vs fp32-cast-points contraction: %.2e" % (name, np.abs(v - v64).max(), np.abs(v - vA).max(), np.abs(v - vA32).max()))

# ---- emulate the packed operands in fp64 on the F32 model's real state -----------------------------------------
def f16(a): return a.to(torch.float16).double()
def halves(a, s):
    a = a * s; h = f16(a); r1 = (a - h) * 2048; M = f16(r1); L = f16(r1 - M); return h, M, L
amax = X32.abs().max().item()
import math
sx = 2.0 ** (-math.frexp(amax)[1]); sk = 0.5
print("max|X| = %.4g -> sx = %g ; fraction of |sx X| entries (lower triangle) below 6.1e-5: %.3f, below 6e-8: %.3f" % (
    amax, sx, ((X32.abs() * sx < 6.1e-5) & (X32 != 0)).double().sum().item() / (X32 != 0).double().sum().item(),
    ((X32.abs() * sx < 6e-8) & (X32 != 0)).double().sum().item() / (X32 != 0).double().sum().item()))
Pm = t32[:3].T; d = torch.cdist(Q.float().double(), Pm); K = kfun(d).float().double(); K[:, n:] = 0
hx, Mx, Lx = halves(X32, sx); hk, Mk, Lk = halves(K, sk)
def fin(W): return (1.0 - (W * W * t32[3][None, :]).sum(1)).cpu().numpy()
vref = fin(K @ X32.T)
for name, W in (("h h", hk @ hx.T), ("+ (hM + Mh)/2048", hk @ hx.T + (hk @ Mx.T + Mk @ hx.T) / 2048),
                ("+ L_x h_k /2048", hk @ hx.T + (hk @ Mx.T + Mk @ hx.T + hk @ Lx.T) / 2048),
                ("all nine", (hk + (Mk + Lk) / 2048) @ (hx + (Mx + Lx) / 2048).T)):
    print("exact fp64 contraction of the packed operands, terms %-18s: err vs fp32-operand contraction %.2e" % (name, np.abs(fin(W / (sx * sk)) - vref).max()))

# ---- the same with the hi halves on one quantum per group of 8 consecutive k (what split8 does) ------------------
def halves_q(a, s):
    a = a * s
    g = a.view(a.shape[0], -1, 8)
    amax = g.abs().amax(dim=2, keepdim=True).clamp_min(2.0 ** -30)
    e = torch.floor(torch.log2(amax)) + 1           # amax in [2^(e-1), 2^e)
    q = torch.pow(2.0, torch.clamp(e - 11, min=-24.0))
    h = (torch.round(g / q) * q).view_as(a)
    return f16(h), f16((a - h) * 2048)
hxq, lxq = halves_q(X32, sx); hkq, lkq = halves_q(K, sk)
W = hkq @ hxq.T + (hkq @ lxq.T + lkq @ hxq.T) / 2048
print("shared-quantum hi halves, exact contraction of hh + (hl + lh)/2048      : err vs fp32-operand contraction %.2e" % np.abs(fin(W / (sx * sk)) - vref).max())
