"""Generates tests/golden/gp_golden.npz with an implementation INDEPENDENT of oracle/ and of libgpx:
plain NumPy/SciPy (numpy.linalg.solve, scipy.linalg.ldl) on the formulas of the reference
(include/gp_regression/gp_regressor.hpp, kernels/*.hpp; Matern from
matlab_src/test_gp_regression_3Dsurf.m:117-123).  Run in the build container:

    python tests/golden/make_golden.py

Inputs: tests/golden/pcd/*.pcd (the reference's resources/*.pcd, data fixtures) prepared exactly as the
node does (gaussian-object-modelling_amd/datasets.py), and a 64-point synthetic sphere.
"""
import importlib
import os
import sys

import numpy as np
import scipy.linalg as sl

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")

KERNELS = {
    "gaussian": lambda d, s=1.0, l=1.0: s * s * np.exp(-d / (l * l)),
    "laplace": lambda d, s=1.0, l=1.0: 2 * s * np.exp(-d / l),
    "thinplate2": lambda d, R=2.0: 2 * d ** 3 - 3 * R * d ** 2 + R ** 3,
    "thinplate4": lambda d, R=4.0: 2 * d ** 3 - 3 * R * d ** 2 + R ** 3,
    "matern32": lambda d, s=1.0, l=1.0: s * s * (1 + np.sqrt(3) * d / l) * np.exp(-np.sqrt(3) * d / l),
    "matern52": lambda d, s=1.0, l=1.0: s * s * (1 + np.sqrt(5) * d / l + 5 * d * d / (3 * l * l)) * np.exp(-np.sqrt(5) * d / l),
}
KDIFF = {  # the reference's computediff conventions (k'(d) for Gaussian/Laplace, k'(d)/d otherwise)
    "gaussian": lambda d: -KERNELS["gaussian"](d),
    "laplace": lambda d: -KERNELS["laplace"](d),
    "thinplate2": lambda d: -6 * (2.0 - d),
    "thinplate4": lambda d: -6 * (4.0 - d),
    "matern32": lambda d: -3 * np.exp(-np.sqrt(3) * d),
    "matern52": lambda d: -(5.0 / 3.0) * (1 + np.sqrt(5) * d) * np.exp(-np.sqrt(5) * d),
}
KPARAMS = {"gaussian": ("gaussian", (1.0, 1.0)), "laplace": ("laplace", (1.0, 1.0)),
           "thinplate2": ("thinplate", (2.0,)), "thinplate4": ("thinplate", (4.0,)),
           "matern32": ("matern32", (1.0, 1.0)), "matern52": ("matern52", (1.0, 1.0))}


def pdist(A, B):
    d = A[:, None, :] - B[None, :, :]
    return np.sqrt((d * d).sum(-1))


def queries(P):
    g = np.linspace(-1.01, 1.01, 4)
    X, Y, Z = np.meshgrid(g, g, g, indexing="ij")
    Q = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1)
    extra = np.array([[3.0, 0.1, -2.0], [0.0, 0.0, 0.0], [0.2, -0.7, 0.4]])
    return np.concatenate([Q[:56], P[:5], extra], 0)  # 64 queries, some ON training points, one far outside


def main():
    out = {}
    sets = {}
    pts = ds.read_pcd(os.path.join(HERE, "pcd", "mugD.pcd"))
    sets["mugD"] = ds.node_training_set(pts)
    rng = np.random.default_rng(20151106)
    u = rng.normal(size=(49, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    ext = ds.exterior_points()
    P64 = np.concatenate([u, ext], 0)
    sets["sphere64"] = (P64[:, 0].copy(), P64[:, 1].copy(), P64[:, 2].copy(),
                        np.concatenate([np.zeros(49), np.ones(15)]), np.full(64, 0.1))
    for sname, (x, y, z, lab, s2) in sets.items():
        P = np.stack([x, y, z], 1)
        Q = queries(P)
        out[sname + "/x"], out[sname + "/y"], out[sname + "/z"] = x, y, z
        out[sname + "/label"], out[sname + "/sigma2"], out[sname + "/Q"] = lab, s2, Q
        D, Dq = pdist(P, P), pdist(Q, P)
        out[sname + "/R"] = np.array(D.max())
        for kname, kf in KERNELS.items():
            K = kf(D) + np.diag(s2)
            alpha = np.linalg.solve(K, lab)
            Kq = kf(Dq)
            f = Kq @ alpha
            v = kf(np.zeros(1))[0] - np.einsum("ij,ji->i", Kq, np.linalg.solve(K, Kq.T))
            W = KDIFF[kname](Dq) * alpha[None, :]
            grad = (W[:, :, None] * (Q[:, None, :] - P[None, :, :])).sum(1)
            # inertia from SciPy's Bunch-Kaufman LDL (independent of any LDL^T here)
            _, Dm, _ = sl.ldl(K)
            neg = int((np.linalg.eigvalsh(Dm) < 0).sum())
            pre = "%s/%s/" % (sname, kname)
            out[pre + "alpha"], out[pre + "f"], out[pre + "v"], out[pre + "grad"] = alpha, f, v, grad
            out[pre + "n_negative"] = np.array(neg)
            print(sname, kname, "n=%d cond=%.2e neg=%d" % (len(x), np.linalg.cond(K), neg))
    # PCD decode census: point counts of every fixture
    names = sorted(fn[:-4] for fn in os.listdir(os.path.join(HERE, "pcd")) if fn.endswith(".pcd"))
    out["pcd/names"] = np.array(names)
    out["pcd/counts"] = np.array([len(ds.read_pcd(os.path.join(HERE, "pcd", n + ".pcd"))) for n in names])
    out["pcd/mugD_first3"] = pts[:3].astype(np.float64)
    np.savez_compressed(os.path.join(HERE, "gp_golden.npz"), **out)
    print("wrote gp_golden.npz with %d arrays" % len(out))


if __name__ == "__main__":
    main()
