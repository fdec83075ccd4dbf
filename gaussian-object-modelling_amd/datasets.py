"""Input recipes for tests and bench (host side, NumPy only; no GPU, no oracle).

Mirrors what the reference node feeds GPRegressor::create / evaluate:
  * read_pcd                   -- pcl::io::loadPCDFile            (src/gp_node.cpp:557)
  * node_training_set          -- deMeanAndNormalizeData + prepareExtData + prepareData + computeGP's
                                  concatenation                   (src/gp_node.cpp:85-117, :793-888, :898-914)
  * exterior_points            -- prepareExtData                  (src/gp_node.cpp:821-849)
  * fibonacci_training_set     -- the synthetic recipe of SURVEY.md 8d (deterministic, seed 20151106)
  * query_grid                 -- fakeDeterministicSampling's lattice (src/gp_node.cpp:1025-1036), by linspace
"""
import math
import struct

import numpy as np

SIGMA2 = 0.1          # src/gp_node.cpp:16
OUT_SPHERE_RAD = 2.0  # src/gp_node.cpp:16


# ---------------------------------------------------------------------------------------------- PCD
def _lzf_decompress(src, out_len):
    out = bytearray(out_len)
    ip, op, n = 0, 0, len(src)
    while ip < n:
        ctrl = src[ip]
        ip += 1
        if ctrl < 32:  # literal run
            ln = ctrl + 1
            out[op:op + ln] = src[ip:ip + ln]
            ip += ln
            op += ln
        else:  # back reference
            ln = ctrl >> 5
            ref = op - ((ctrl & 0x1F) << 8) - 1
            if ln == 7:
                ln += src[ip]
                ip += 1
            ref -= src[ip]
            ip += 1
            ln += 2
            for _ in range(ln):  # may overlap
                out[op] = out[ref]
                op += 1
                ref += 1
    if op != out_len:
        raise ValueError("LZF: decoded %d bytes, expected %d" % (op, out_len))
    return bytes(out)


def read_pcd(path):
    """Returns float32 array (n,3) of x,y,z from an ascii / binary / binary_compressed PCD."""
    with open(path, "rb") as fh:
        raw = fh.read()
    hdr = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key] = val.split()
        if key == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(fields))]
    npts = int(hdr["POINTS"][0])
    mode = hdr["DATA"][0]
    np_t = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 4): "<u4", ("U", 2): "<u2", ("U", 1): "u1",
            ("I", 4): "<i4", ("I", 2): "<i2", ("I", 1): "i1"}
    cols = {}
    if mode == "ascii":
        rows = raw[pos:].decode("ascii").split()
        arr = np.array(rows, dtype=np.float64).reshape(npts, -1)
        c = 0
        for f, cnt in zip(fields, counts):
            cols[f] = arr[:, c]
            c += cnt
    elif mode == "binary":
        dt = np.dtype([(f, np_t[(t, s)], (c,)) for f, t, s, c in zip(fields, types, sizes, counts)])
        rec = np.frombuffer(raw, dtype=dt, count=npts, offset=pos)
        for f in fields:
            cols[f] = rec[f][:, 0]
    elif mode == "binary_compressed":
        csize, usize = struct.unpack_from("<II", raw, pos)
        data = _lzf_decompress(raw[pos + 8:pos + 8 + csize], usize)
        off = 0
        for f, t, s, c in zip(fields, types, sizes, counts):  # SoA: all x, then all y, ...
            cols[f] = np.frombuffer(data, dtype=np_t[(t, s)], count=npts * c, offset=off).reshape(npts, c)[:, 0]
            off += npts * c * s
    else:
        raise ValueError("unknown PCD DATA mode " + mode)
    return np.stack([cols["x"], cols["y"], cols["z"]], axis=1).astype(np.float32)


# ---------------------------------------------------------------------------- node-equivalent prep
def exterior_points(rad=OUT_SPHERE_RAD):
    """The 3x5 points of prepareExtData (src/gp_node.cpp:821-849), same accumulating double loops."""
    ang_div, lin_div = 5, 3
    ang_step = math.pi * 2 / ang_div
    lin_step = 2 * rad / lin_div
    pts = []
    lin = -rad + lin_step / 2
    while lin < rad:
        ang = 0.0
        while ang < 2 * math.pi:
            r = math.sqrt(rad ** 2 - lin * lin)
            pts.append((r * math.cos(ang), r * math.sin(ang), lin))
            ang += ang_step
        lin += lin_step
    return np.array(pts, dtype=np.float64)


def normalize_cloud(points_f32):
    """deMeanAndNormalizeData (src/gp_node.cpp:85-117) in PCL's float arithmetic.
    Returns (normalised float32 (n,3), centroid float32 (3,), scale double)."""
    p = np.asarray(points_f32, dtype=np.float32)
    c = np.zeros(3, dtype=np.float32)
    for row in p:  # pcl::compute3DCentroid<PointT,float>: sequential float accumulation
        c += row
    c = (c / np.float32(len(p))).astype(np.float32)
    t = (p - c).astype(np.float32)
    n2 = (t[:, 0] * t[:, 0] + t[:, 1] * t[:, 1]).astype(np.float32) + t[:, 2] * t[:, 2]
    scale = float(np.max(np.sqrt(n2.astype(np.float32))))
    s = np.float32(1.0 / scale)
    return (t * s).astype(np.float32), c, scale


def node_training_set(points_f32, sigma2=SIGMA2, rad=OUT_SPHERE_RAD):
    """Training Data exactly as computeGP assembles it: normalised cloud (label 0) followed by the
    exterior sphere (label 1), sigma2 everywhere (src/gp_node.cpp:853-914)."""
    surf, _, _ = normalize_cloud(points_f32)
    ext = exterior_points(rad)
    P = np.concatenate([surf.astype(np.float64), ext], axis=0)
    label = np.concatenate([np.zeros(len(surf)), np.ones(len(ext))])
    s2 = np.full(len(P), float(sigma2))
    return P[:, 0].copy(), P[:, 1].copy(), P[:, 2].copy(), label, s2


# --------------------------------------------------------------------------------- synthetic recipe
class MT19937_64:
    """std::mt19937_64 (so that C++ callers can regenerate the same clouds)."""
    NN, MM = 312, 156
    MATRIX_A, UM, LM = 0xB5026F5AA96619E9, 0xFFFFFFFF80000000, 0x7FFFFFFF
    M64 = (1 << 64) - 1

    def __init__(self, seed):
        mt = [0] * self.NN
        mt[0] = seed & self.M64
        for i in range(1, self.NN):
            mt[i] = (6364136223846793005 * (mt[i - 1] ^ (mt[i - 1] >> 62)) + i) & self.M64
        self.mt, self.idx = mt, self.NN

    def _twist(self):
        mt, NN, MM = self.mt, self.NN, self.MM
        for i in range(NN):
            x = (mt[i] & self.UM) | (mt[(i + 1) % NN] & self.LM)
            xa = x >> 1
            if x & 1:
                xa ^= self.MATRIX_A
            mt[i] = mt[(i + MM) % NN] ^ xa
        self.idx = 0

    def next(self):
        if self.idx >= self.NN:
            self._twist()
        x = self.mt[self.idx]
        self.idx += 1
        x ^= (x >> 29) & 0x5555555555555555
        x ^= (x << 17) & 0x71D67FFFEDA60000
        x ^= (x << 37) & 0xFFF7EEE000000000
        x ^= x >> 43
        return x & self.M64

    def uniform(self, a, b):
        """std::uniform_real_distribution<double>(a,b) on a 64-bit engine (libstdc++)."""
        u = float(self.next()) * (2.0 ** -64)
        if u >= 1.0:
            u = math.nextafter(1.0, 0.0)
        return a + (b - a) * u


def fibonacci_training_set(n, seed=20151106, jitter=1e-3, sigma2=SIGMA2, rad=OUT_SPHERE_RAD):
    """n-15 jittered Fibonacci-sphere surface points (label 0) + the 15 exterior points (label 1)."""
    ns = n - 15
    if ns <= 0:
        raise ValueError("n must exceed 15")
    i = np.arange(ns, dtype=np.float64)
    phi = np.arccos(1 - 2 * (i + 0.5) / ns)
    theta = math.pi * (1 + math.sqrt(5.0)) * (i + 0.5)
    P = np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)], axis=1)
    rng = MT19937_64(seed)
    J = np.array([rng.uniform(-jitter, jitter) for _ in range(3 * ns)]).reshape(ns, 3)
    P = P + J
    ext = exterior_points(rad)
    P = np.concatenate([P, ext], axis=0)
    label = np.concatenate([np.zeros(ns), np.ones(len(ext))])
    s2 = np.full(n, float(sigma2))
    return P[:, 0].copy(), P[:, 1].copy(), P[:, 2].copy(), label, s2


def random_shell_training_set(n, seed=7, r_lo=0.9, r_hi=1.1, sigma2=SIGMA2):
    """An IRREGULAR cloud: n points uniform in the shell r_lo <= |p| <= r_hi (rejection from the cube, std::mt19937_64
    stream `seed`, so the cloud is reproducible without NumPy's generators), label 1 for every tenth point else 0, plus
    0.01 * a uniform(-1, 1) draw.  The thin-plate counterpart of the regular Fibonacci cloud for the N = 16384 anchor:
    on random clouds the predictor weights K^-1 k_q are large."""
    rng = MT19937_64(seed)
    P = np.empty((n, 3))
    k = 0
    while k < n:
        p = (rng.uniform(-r_hi, r_hi), rng.uniform(-r_hi, r_hi), rng.uniform(-r_hi, r_hi))
        r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2]
        if r_lo * r_lo <= r2 <= r_hi * r_hi:
            P[k] = p
            k += 1
    label = np.array([(1.0 if i % 10 == 0 else 0.0) + 0.01 * rng.uniform(-1.0, 1.0) for i in range(n)])
    return P[:, 0].copy(), P[:, 1].copy(), P[:, 2].copy(), label, np.full(n, float(sigma2))


def query_grid(g, scale=1.01):
    """g^3 lattice on [-scale, scale]^3, x slowest / z fastest (src/gp_node.cpp:1025-1036)."""
    t = np.linspace(-scale, scale, g)
    X, Y, Z = np.meshgrid(t, t, t, indexing="ij")
    return X.ravel().copy(), Y.ravel().copy(), Z.ravel().copy()


def query_grid_slab(g, rank, world, scale=1.01):
    """x-slab `rank` of `world` of the g^3 lattice (contiguous slabs, remainder to the low ranks)."""
    base, rem = divmod(g, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    t = np.linspace(-scale, scale, g)
    X, Y, Z = np.meshgrid(t[lo:hi], t, t, indexing="ij")
    return X.ravel().copy(), Y.ravel().copy(), Z.ravel().copy(), lo * g * g
