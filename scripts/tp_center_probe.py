"""Thin-plate variance at N = 16384: where the fp32 contraction loses its accuracy and what centring the kernel
operand buys.  Uses the F32 / MIXED model's own inverse factor X and 1/D (state blobs) and contracts them in torch:
  exact      : fp64 contraction of the stored state with fp64 kernel values         (floor of the mode)
  k32        : kernel values rounded to fp32, fp64 accumulation                      (operand rounding only)
  gemm32     : fp32 torch GEMM of fp32 operands                                      (~ what the MFMA kernel does)
  c-*        : the same with k - c_q (c_q = mid-range of the query's kernel row) and w += c_q (X 1) in fp64
  chunk-*    : fp32 GEMM over k-chunks of 2048, chunks summed in fp64
Usage: python scripts/tp_center_probe.py [n] [kernel] [prec]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
kn = sys.argv[2] if len(sys.argv) > 2 else "thinplate"
pn = sys.argv[3] if len(sys.argv) > 3 else "F32"
R = 4.0
x, y, z, lab, s2 = ds.fibonacci_training_set(n)
qx, qy, qz = ds.query_grid(16)
kern = gpx.make_kernel(kn, R) if kn == "thinplate" else gpx.make_kernel(kn, 1.0, 1.0)
k0 = R ** 3 if kn == "thinplate" else 1.0
dev = torch.device("cuda:0")


def kfun(d):
    if kn == "thinplate":
        return (d - R) ** 2 * (2 * d + R)
    t = np.sqrt(5.0) * d
    return torch.exp(-t) * (1 + t + t * t / 3)


g64 = gpx.Model(kern, x, y, z, lab, s2, precision=gpx.F64)
v64 = g64.evaluate(qx, qy, qz, want_v=True)["v"]
g64.close()
m = gpx.Model(kern, x, y, z, lab, s2, precision=getattr(gpx, pn), prepare_variance=True)
vm = m.evaluate(qx, qy, qz, want_v=True)["v"]
p0, b0 = m.state_blob(0)
p1, b1 = m.state_blob(1)
t0 = sh.device_blob_as_tensor(torch, p0, b0, dev)
t1 = sh.device_blob_as_tensor(torch, p1, b1, dev)
npad = b0 // (4 * 8 + 9 * 4)  # fp64 x y z alpha | T x y z 1/D | T 5 correction vectors
d64 = t0[: 4 * 8 * npad].view(torch.float64).view(4, npad)
tt = t0[8 * 4 * npad:].view(torch.float32).view(9, npad)[:4]
X32 = t1.view(torch.float32).view(npad, npad)
dinv = tt[3].double()
P = d64[:3].T.contiguous()
Q = torch.tensor(np.stack([qx, qy, qz], 1), device=dev)
K = kfun(torch.cdist(Q, P))
K[:, n:] = 0
Xd = X32.double()
v64t = torch.tensor(v64, device=dev)


def rep(name, W):
    v = k0 - (W * W * dinv[None, :]).sum(1)
    e = (v - v64t).abs()
    print("%-44s max %.2e  rms %.2e   (/k0)" % (name, e.max().item() / k0, (e * e).mean().sqrt().item() / k0), flush=True)


print("N=%d %s %s: library evaluate vs F64 pipeline: %.2e" % (n, kn, pn, np.abs(vm - v64).max() / k0))
rep("exact (fp64 k, fp64 acc)", K @ Xd.T)
K32 = K.float()
rep("k32 (fp32 k, fp64 acc)", K32.double() @ Xd.T)
rep("gemm32 (fp32 k, torch fp32 GEMM)", (K32 @ X32.T).double())


def chunked(Kf, c=2048):
    W = torch.zeros(Kf.shape[0], npad, dtype=torch.float64, device=dev)
    for k0_ in range(0, npad, c):
        W += (Kf[:, k0_:k0_ + c] @ X32[:, k0_:k0_ + c].T).double()
    return W


rep("chunk-2048 gemm32, chunks summed in fp64", chunked(K32))
rep("chunk-512 gemm32, chunks summed in fp64", chunked(K32, 512))
s1 = Xd[:, :n].sum(1)  # X 1 over the real points, fp64
for cname, cq in (("mid-range", 0.5 * (K[:, :n].max(1).values + K[:, :n].min(1).values)), ("mean", K[:, :n].mean(1)),
                  ("global 0.75 k0", torch.full((K.shape[0],), 0.75 * k0, dtype=torch.float64, device=dev))):
    Kc = K - cq[:, None]
    Kc[:, n:] = 0
    corr = cq[:, None] * s1[None, :]
    rep("c[%s] exact" % cname, Kc @ Xd.T + corr)
    Kc32 = Kc.float()
    rep("c[%s] k32, fp64 acc" % cname, Kc32.double() @ Xd.T + corr)
    rep("c[%s] gemm32" % cname, (Kc32 @ X32.T).double() + corr)
    rep("c[%s] gemm32, s1 and c in fp32" % cname, ((Kc32 @ X32.T) + cq.float()[:, None] * s1.float()[None, :]).double())
    rep("c[%s] chunk-2048 gemm32" % cname, chunked(Kc32) + corr)
# kernel values computed in fp32 from fp32 points (what kqp does), centred in fp32
Pf, Qf = P.float(), Q.float()
d2 = ((Qf[:, None, :] - Pf[None, :, :]) ** 2).sum(2)
df = d2.sqrt()
Kf = (df - R) ** 2 * (2 * df + R) if kn == "thinplate" else None
if Kf is not None:
    Kf[:, n:] = 0
    rep("fp32-computed k, gemm32", (Kf @ X32.T).double())
    cq = (0.5 * (K[:, :n].max(1).values + K[:, :n].min(1).values)).float()
    # k - c evaluated as d^2 (2 d - 3 R) + (R^3 - c) with the product's rounding error recovered by an fma
    t = 2 * df - 3 * R
    hi = d2 * t
    Kcf = (hi + (R ** 3 - cq[:, None]))
    Kcf[:, n:] = 0
    rep("fp32-computed k - c (Horner), gemm32", (Kcf @ X32.T).double() + cq.double()[:, None] * s1[None, :])
m.close()
