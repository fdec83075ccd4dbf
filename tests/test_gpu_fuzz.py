"""Seeded differential fuzzing of the fp64 GPU path against the oracle: random sizes around the tile / panel
boundaries, random kernels and parameters, uniform or per-point noise (the latter permutes the pivot order),
normals, updates (append and rebuild), queries on / near / far from the training points."""
import numpy as np
import pytest

from conftest import nerr, verr, verr_v

pytestmark = pytest.mark.gpu

SIZES = [1, 2, 3, 17, 127, 128, 129, 255, 256, 257, 300, 383, 384, 385, 511, 512, 513, 640]
KERNELS = [("gaussian", lambda r: (r.uniform(0.5, 2.0), r.uniform(0.4, 1.5))),
           ("laplace", lambda r: (r.uniform(0.5, 2.0), r.uniform(0.4, 1.5))),
           ("thinplate", lambda r: (r.choice([2.0, 3.0, 4.0]),)),
           ("matern32", lambda r: (r.uniform(0.5, 2.0), r.uniform(0.4, 1.5))),
           ("matern52", lambda r: (r.uniform(0.5, 2.0), r.uniform(0.4, 1.5)))]


@pytest.mark.parametrize("seed", range(30))
def test_fuzz_fp64_against_oracle(gpu, orc, seed):
    r = np.random.default_rng(1000 + seed)
    n = int(SIZES[seed % len(SIZES)]) if seed < 2 * len(SIZES) else int(r.integers(1, 700))
    kn, kpar = KERNELS[int(r.integers(len(KERNELS)))]
    par = tuple(float(p) for p in kpar(r))
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    P = d * r.uniform(0.9, 1.1, size=(n, 1))
    lab = np.where(r.uniform(size=n) < 0.1, 1.0, 0.0) + 0.01 * r.normal(size=n)
    mode = int(r.integers(3))
    s2 = None if mode == 0 else (np.full(n, 0.1) if mode == 1 else r.uniform(0.05, 0.3, size=n))
    if s2 is None and n > 1:
        s2 = np.full(n, 1e-3)  # keep the noiseless case for n == 1 only (K is then 1 x 1)
    with_normals = bool(r.integers(2))
    om = orc.Model(orc.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2, with_normals=with_normals)
    gm = gpu.Model(gpu.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2, precision=gpu.F64,
                   with_normals=with_normals)
    k0 = float(orc.k(orc.make_kernel(kn, *par), 0.0)[0])
    cond_scale = 1e3 if kn == "thinplate" else 1.0  # thin plate: cond up to 1e7 at these sizes
    tol = 1e-10 * cond_scale

    def check(gm, om, tag):
        nq = int(r.integers(1, 90))
        Q = r.uniform(-1.3, 1.3, size=(nq, 3))
        m = min(nq, n, 3)
        Q[:m] = np.stack([om_x, om_y, om_z], 1)[:m]  # on training points
        ref = om.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
        out = gm.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
        assert nerr(gm.alpha, om.alpha) < tol, tag
        assert nerr(out["f"], ref["f"]) < tol, tag
        gscale = max(np.max(np.abs(ref["grad"])), 1e-300)
        assert np.max(np.abs(out["grad"] - ref["grad"])) / gscale < tol, tag
        assert verr(out["v"], ref["v"], k0) < tol, tag

    om_x, om_y, om_z = P[:, 0], P[:, 1], P[:, 2]
    check(gm, om, "create")
    if with_normals and n > 1:
        assert nerr(gm.normals, om.normals) < 1e-7 * cond_scale
    if r.integers(2):  # update: append when the noise is uniform, rebuild otherwise
        n1 = int(r.integers(1, 200))
        d1 = r.normal(size=(n1, 3))
        d1 /= np.linalg.norm(d1, axis=1)[:, None]
        P1 = d1 * r.uniform(0.9, 1.1, size=(n1, 1))
        lab1 = 0.01 * r.normal(size=n1)
        s21 = np.full(n1, 0.1) if mode == 1 else (r.uniform(0.05, 0.3, size=n1) if mode == 2 else np.full(n1, 1e-3))
        gm.update(P1[:, 0], P1[:, 1], P1[:, 2], lab1, s21)
        om.update(P1[:, 0], P1[:, 1], P1[:, 2], lab1, s21)
        om_x, om_y, om_z = np.concatenate([P[:, 0], P1[:, 0]]), np.concatenate([P[:, 1], P1[:, 1]]), np.concatenate([P[:, 2], P1[:, 2]])
        n = n + n1
        check(gm, om, "update")
    gm.close()


@pytest.mark.parametrize("prec", [0, 2, 3])  # F32, MIXED, F32_SPLIT
@pytest.mark.parametrize("seed", range(16))
def test_fuzz_fp32_family_against_oracle(gpu, orc, seed, prec):
    r = np.random.default_rng(5000 + seed)
    n = int(r.choice([5, 129, 256, 300, 511, 700, 1100, 1500, 2305]))  # 2305: above the fp64-training threshold of F32 (exp kernels)
    kn, kpar = KERNELS[int(r.choice([0, 1, 2, 3, 4]))]
    par = tuple(float(p) for p in kpar(r))
    if seed in (12, 13):  # two cases pinned above the fp64-training threshold: an fp32 kernel matrix and LDL^T feed the variance
        n, (kn, kpar) = 2305, KERNELS[4 if seed == 12 else 3]
        par = tuple(float(p) for p in kpar(r))
    if seed in (14, 15):
        # thin plate at the same size.  The variance sees the backward error E of an fp32 LDL^T as a^T E a with a = K^-1 k_q,
        # and thin-plate predictor weights are large (|a|_1 = 11 at the centre of this cloud, 20-70 outside it; Matern: 1-2):
        # fp32-trained this case measured 4.4e-5 k(0), which is why F32 models with this kernel
        # train in fp64 at every size the device holds (round 3).  R is drawn from {2, 3, 4}: R = 2 is INDEFINITE on
        # this cloud (diameter 2.2), as the node's own setting is on its clouds.
        n, (kn, kpar) = 2305, KERNELS[2]
        par = tuple(float(p) for p in kpar(r))
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    P = d * r.uniform(0.9, 1.1, size=(n, 1))
    lab = np.where(r.uniform(size=n) < 0.1, 1.0, 0.0) + 0.01 * r.normal(size=n)
    s2 = np.full(n, 0.1) if seed % 2 else r.uniform(0.05, 0.3, size=n)
    om = orc.Model(orc.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2)
    gm = gpu.Model(gpu.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2, precision=prec)
    k0 = float(orc.k(orc.make_kernel(kn, *par), 0.0)[0])
    for nq in (1, 64, 65, 700):  # one-launch path, its limit, general path
        Q = r.uniform(-1.3, 1.3, size=(nq, 3))
        Q[0] = P[0]
        ref = om.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
        out = gm.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
        assert nerr(out["f"], ref["f"]) < 1e-5, nq
        assert np.max(np.abs(out["grad"] - ref["grad"])) / np.max(np.abs(ref["grad"])) < 1e-5, nq
        assert verr(out["v"], ref["v"], k0) < 1e-5, nq
        if nq >= 64:  # SURVEY 8d's norm-wise metric needs a set of queries (for one query it is element-wise)
            assert verr_v(out["v"], ref["v"]) < 1e-5, nq
    assert nerr(gm.alpha, om.alpha) < 1e-5
    gm.close()
