"""GPU parity suite: the HIP path, called through the C ABI of libgpx.so, against the CPU oracle on the
same inputs and against the committed golden vectors.  Tolerances (norm-wise, max|a-b| / max|b|, SURVEY
8d): 1e-10 for fp64 compute, 1e-5 for fp32 compute (north star: "within 1e-5 rel. of the CPU
reference")."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

from conftest import GOLDEN_DIR, KERNEL_CASES, nerr, verr, verr_v

pytestmark = pytest.mark.gpu
TOL = {0: 1e-5, 1: 1e-10, 2: 1e-5, 3: 1e-5}  # gpx.F32, gpx.F64, gpx.MIXED, gpx.F32_SPLIT


def _queries(ds, x, y, z, g=7):
    qx, qy, qz = ds.query_grid(g)
    # plus points ON training points and one far outside
    return (np.concatenate([qx, x[:9], [3.0]]), np.concatenate([qy, y[:9], [0.1]]),
            np.concatenate([qz, z[:9], [-2.0]]))


def _k0(om):
    import gp_oracle
    return float(gp_oracle.k(om.kern, 0.0)[0])


def _check(gm, om, q, prec, basis=True):
    qx, qy, qz = q
    ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True, want_basis=True)
    out = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True, want_basis=True)
    tol = TOL[prec]
    # alpha, the mean and the gradient are fp64 work in every mode; with an fp32 factor alpha comes from two
    # fp64-residual refinement steps (1e-8 typical, 1e-6 for the worst-conditioned thin-plate matrices)
    mtol = 1e-10 if prec == 1 else 1e-9 if prec == 2 else 1e-5
    assert nerr(gm.alpha, om.alpha) < mtol
    for key in ("f", "grad"):
        assert nerr(out[key], ref[key]) < mtol, key
    # every fp32 mode is held to the north-star 1e-5, thin-plate included, in both normalisations of the variance error
    # (SURVEY 8d's max|dv| / max|v_ref| and the k(0)-scaled one): the contraction runs on the centred kernel operand
    # with an fp64 epilogue; small models and thin-plate models are trained in fp64
    assert verr(out["v"], ref["v"], _k0(om)) < tol and verr_v(out["v"], ref["v"]) < tol, "v"
    if basis:
        # the tangent basis normalises the gradient: compare where the gradient is not tiny
        gn = np.linalg.norm(ref["grad"], axis=1)
        ok = gn > 1e-2 * gn.max()
        for key in ("tx", "ty"):
            assert np.max(np.abs(out[key][ok] - ref[key][ok])) < mtol * 1e3, key
    # the three evaluate overloads agree with each other: to the bit where the mean kernel answers all of them, to rounding for
    # small fp64 models, whose evaluate(f, v) takes the mean from the variance kernel (another summation order, gpx_varcols64.hip)
    f_only = gm.evaluate(qx, qy, qz)["f"]
    f_with_v = gm.evaluate(qx, qy, qz, want_v=True)["f"]
    if gm.n <= 992:  # (fp64 state: asked for, or kept by the promotion rule)
        assert nerr(f_with_v, f_only) < 1e-12
    else:
        np.testing.assert_array_equal(f_only, f_with_v)
    return out


@pytest.mark.parametrize("prec", [1, 0, 2, 3])
@pytest.mark.parametrize("kkey", list(KERNEL_CASES))
def test_mugd_node_training_set(gpu, orc, ds, golden, kkey, prec):
    """C1: resources/mugD.pcd prepared as the node does (N = 277), all six kernel settings, incl. the
    node's own indefinite ThinPlate(2.0) (src/gp_node.cpp:919)."""
    kn, par = KERNEL_CASES[kkey]
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
    _check(gm, om, _queries(ds, x, y, z), prec)
    assert gm.R == pytest.approx(om.R, rel=1e-14)
    assert gm.stats["n_negative_pivots"] == int(golden["mugD/%s/n_negative" % kkey])
    # golden vectors (independent NumPy/SciPy)
    Q = golden["mugD/Q"]
    out = gm.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
    pre = "mugD/%s/" % kkey
    tol = TOL[prec]
    assert nerr(gm.alpha, golden[pre + "alpha"]) < max(tol, 1e-9)
    for key in ("f", "grad"):
        assert nerr(out[key], golden[pre + key]) < max(tol, 1e-9), key
    assert verr(out["v"], golden[pre + "v"], _k0(om)) < max(tol, 1e-9)
    assert verr_v(out["v"], golden[pre + "v"]) < max(tol, 1e-9)
    gm.close()


@pytest.mark.parametrize("prec", [1, 0, 2, 3])
@pytest.mark.parametrize("n", [16, 128, 129, 256, 257, 600, 1500])
def test_ragged_sizes(gpu, orc, ds, n, prec):
    """Sizes around the 128 / 256 tile and panel edges (padding with an identity block)."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    for kn, par in (("matern52", (1, 1)), ("thinplate", (4.0,))):
        om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
        gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
        _check(gm, om, _queries(ds, x, y, z, g=5), prec, basis=False)
        gm.close()


@pytest.mark.parametrize("kn,par", [("gaussian", (1.7, 0.6)), ("laplace", (0.4, 1.9)), ("thinplate", (5.5,)),
                                    ("matern32", (2.0, 0.35)), ("matern52", (0.7, 2.5))])
def test_non_default_hyper_parameters(gpu, orc, ds, golden, kn, par):
    """setCovFunction with non-default kernels (gp_regressor.hpp:488-491): sigma, length, R away from 1."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2, with_normals=True)
    for prec in (1, 0, 2, 3):
        gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec, with_normals=True)
        _check(gm, om, _queries(ds, x, y, z, g=6), prec)
        assert nerr(gm.normals, om.normals) < (1e-9 if prec in (1, 2) else 1e-5)
        gm.close()


@pytest.mark.parametrize("name", ["bowlA", "bowlB", "containerA", "containerB", "jug", "kettle", "pot", "mugD",
                                  "kitchenUtensilB"])
def test_c5_objects_from_pcd(gpu, orc, ds, name):
    """BASELINE config 5: one GP per object, fp32 Gaussian(1,1); PCD -> node training set on the HOST C++ path
    (gpx_pcd_read + gpx_node_training_set), 128^3 lattice (oracle on a 700-point sub-sample)."""
    xyz = gpu.pcd_read(os.path.join(GOLDEN_DIR, "pcd", name + ".pcd"))
    x, y, z, lab, s2 = gpu.node_training_set(xyz)
    assert len(x) == len(xyz) + 15
    gm = gpu.Model(gpu.make_kernel("gaussian", 1, 1), x, y, z, lab, s2, precision=gpu.F32)
    om = orc.Model(orc.make_kernel("gaussian", 1, 1), x, y, z, lab, s2)
    qx, qy, qz = ds.query_grid(128)
    out = gm.evaluate(qx, qy, qz, want_v=True)
    sel = np.arange(0, 128 ** 3, 2999)
    ref = om.evaluate(qx[sel], qy[sel], qz[sel], want_v=True)
    assert nerr(gm.alpha, om.alpha) < 1e-5
    assert nerr(out["f"][sel], ref["f"]) < 1e-5
    assert verr(out["v"][sel], ref["v"], 1.0) < 1e-5
    # the surface the node would extract from this grid
    surf = gm.sample_surface(qx, qy, qz, f_tol=0.01)
    keep = np.nonzero(np.abs(out["f"]) <= 0.01)[0]
    np.testing.assert_array_equal(surf["idx"], keep)
    gm.close()


@pytest.mark.parametrize("name", ["bowlA", "containerA", "kettle"])
def test_c5_objects_in_the_fp64_and_the_split_fp16_mode(gpu, orc, ds, name):
    """BASELINE config 5 in its two other forms (bench.py: configs.C5_f64, configs.C5_split): the reference's own arithmetic --
    small fp64 models answer evaluate(f, v) with one kernel (gpx_varcols64.hip) -- and the opt-in split-fp16 mode on the fp16
    matrix cores (gpx_varcols16.hip), 128^3 lattice, oracle on a sub-sample; split mode: 5e-6 of max v."""
    x, y, z, lab, s2 = gpu.node_training_set(gpu.pcd_read(os.path.join(GOLDEN_DIR, "pcd", name + ".pcd")))
    om = orc.Model(orc.make_kernel("gaussian", 1, 1), x, y, z, lab, s2)
    qx, qy, qz = ds.query_grid(128)
    sel = np.arange(0, 128 ** 3, 2999)
    ref = om.evaluate(qx[sel], qy[sel], qz[sel], want_v=True)
    for prec, tol_f, tol_v in ((gpu.F64, 1e-9, 1e-9), (gpu.F32_SPLIT, 1e-5, 5e-6)):
        gm = gpu.Model(gpu.make_kernel("gaussian", 1, 1), x, y, z, lab, s2, precision=prec)
        out = gm.evaluate(qx, qy, qz, want_v=True)
        assert nerr(out["f"][sel], ref["f"]) < tol_f, (name, prec)
        assert verr_v(out["v"][sel], ref["v"]) < tol_v, (name, prec)
        assert float(out["v"].min()) > -1e-6  # (the whole lattice, not only the sub-sample)
        gm.close()


def test_single_training_point(gpu, orc):
    for prec in (1, 0):
        gm = gpu.Model(gpu.make_kernel("gaussian", 1, 1), [0.2], [0.0], [-0.1], [1.0], [0.1], precision=prec)
        om = orc.Model(orc.make_kernel("gaussian", 1, 1), [0.2], [0.0], [-0.1], [1.0], [0.1])
        q = (np.array([0.2, 1.0]), np.array([0.0, 0.5]), np.array([-0.1, 0.0]))
        ref = om.evaluate(*q, want_v=True)
        out = gm.evaluate(*q, want_v=True)
        assert nerr(out["f"], ref["f"]) < TOL[prec] and verr(out["v"], ref["v"], 1.0) < TOL[prec]
        assert gm.R == 0.0 and gm.n == 1
        gm.close()


def test_empty_sigma2_interpolates(gpu, orc, golden):
    """Data::sigma2 empty => no diagonal noise (gp_regressor.hpp:154): f(p_i) = y_i, v(p_i) = 0."""
    x, y, z, lab = (golden["sphere64/" + k] for k in ("x", "y", "z", "label"))
    for kn, par in (("gaussian", (1, 1)), ("thinplate", (4.0,))):
        gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, None, precision=gpu.F64)
        om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, None)
        out = gm.evaluate(x, y, z, want_v=True)
        assert np.max(np.abs(out["f"] - lab)) < 1e-8
        assert np.max(np.abs(out["v"])) < 1e-7 * float(orc.k(orc.make_kernel(kn, *par), 0.0)[0])
        assert nerr(gm.alpha, om.alpha) < 1e-9
        np.testing.assert_array_equal(gm.S2, np.zeros(64))
        gm.close()


def test_nonuniform_noise_uses_eigen_pivot_order(gpu, orc, golden):
    """Per-point sigma2 makes the diagonal non-uniform: the Eigen pivot rule permutes the points."""
    x, y, z, lab, s2 = (golden["sphere64/" + k].copy() for k in ("x", "y", "z", "label", "sigma2"))
    s2[::3] = 0.5
    s2[5] = 2.0
    om = orc.Model(orc.make_kernel("matern32", 1, 1), x, y, z, lab, s2)
    for prec in (1, 0):
        gm = gpu.Model(gpu.make_kernel("matern32", 1, 1), x, y, z, lab, s2, precision=prec)
        perm = gm.perm
        assert perm[0] == 5 and sorted(perm) == list(range(64))
        # D of P K P^T = L D L^T in Eigen's order
        F, _ = om.ldlt()
        assert nerr(gm.D, np.diag(F)) < (1e-9 if prec else 1e-4)
        Q = golden["sphere64/Q"]
        ref = om.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True)
        out = gm.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True)
        assert nerr(out["f"], ref["f"]) < TOL[prec] and nerr(out["v"], ref["v"]) < TOL[prec]
        gm.close()


def test_normals_create_true(gpu, orc, golden):
    """create<true>: Model::N, gp_regressor.hpp:166-181 (zero-initialised accumulation)."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    for kn, par in (("gaussian", (1, 1)), ("thinplate", (2.0,))):
        om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2, with_normals=True)
        for prec in (1, 0):
            gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec, with_normals=True)
            assert nerr(gm.normals, om.normals) < TOL[prec] * 10
            gm.close()
    gm = gpu.Model(gpu.make_kernel("gaussian", 1, 1), x, y, z, lab, s2)
    with pytest.raises(gpu.GpxError):
        _ = gm.normals
    gm.close()


def test_update_appends_and_refactors(gpu, orc, ds, golden):
    """GPRegressor::update, gp_regressor.hpp:367-479."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    kern = ("thinplate", (2.0,))
    for prec in (1, 0):
        gm = gpu.Model(gpu.make_kernel(*kern[:1], *kern[1]), x[:250], y[:250], z[:250], lab[:250], s2[:250],
                       precision=prec)
        R0 = gm.R
        gm.evaluate(x[:4], y[:4], z[:4], want_v=True)  # builds the inverse factor; update must drop it
        gm.update(x[250:], y[250:], z[250:], lab[250:], s2[250:])
        om = orc.Model(orc.make_kernel(*kern[:1], *kern[1]), x, y, z, lab, s2)
        assert gm.n == 277 and gm.R == R0
        _check(gm, om, _queries(ds, x, y, z, g=5), prec, basis=False)
        np.testing.assert_array_equal(gm.P, np.stack([x, y, z], 1))
        np.testing.assert_array_equal(gm.Y, lab)
        gm.close()


@pytest.mark.parametrize("inv_first", [False, True])
@pytest.mark.parametrize("prec,kname", [(1, "thinplate"), (0, "thinplate"), (0, "matern52")])
@pytest.mark.parametrize("n0,n1", [(1500, 40), (1408, 700), (300, 1300), (1000, 24)])
def test_update_rank_n_append_equals_rebuild(gpu, orc, ds, prec, kname, n0, n1, inv_first, monkeypatch):
    """update() appends to the existing factor (new kernel rows, left-looking row update against the old
    column blocks, factorisation of the new trailing block) instead of refactoring from scratch.  Same alpha, D,
    inertia and predictions as a rebuild (GPX_UPDATE_APPEND=0) and as a fresh model; cases: inside the last padded
    tile, across the padding (larger matrix), old N a multiple of 128, first tile partially filled.  inv_first: the
    inverse factor exists before the update (a variance query) and is extended by the new rows instead of rebuilt.
    Thin-plate R = 2 is indefinite (negative pivots on both sides of the split, cond ~1e7): in fp64 the paths agree to
    1e-9.  The fp32 append arithmetic is forced (GPX_TRAIN_F64_MAX=0: F32 models of this size train in fp64 by default)
    on the Matern-5/2 system; an F32 thin-plate model runs with the product rule -- it trains in fp64 and holds no
    factor, so update() rebuilds it -- and is held to the same 1e-5 on everything, the variance included."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n0 + n1)
    tp = kname == "thinplate"
    kern = gpu.make_kernel("thinplate", 2.0) if tp else gpu.make_kernel("matern52", 1.0, 1.0)
    k0 = 8.0 if tp else 1.0
    qx, qy, qz = ds.query_grid(5)
    res = {}
    forced32 = prec == 0 and not tp
    if forced32:
        monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")
        gpu.debug_reload()
    for mode in ("1", "0"):
        monkeypatch.setenv("GPX_UPDATE_APPEND", mode)
        gpu.debug_reload()
        gm = gpu.Model(kern, x[:n0], y[:n0], z[:n0], lab[:n0], s2[:n0], precision=prec)
        if inv_first:
            gm.evaluate(qx, qy, qz, want_v=True)
        gm.update(x[n0:], y[n0:], z[n0:], lab[n0:], s2[n0:])
        if inv_first and mode == "1" and (prec == 1 or forced32) and n0 + n1 > 1024:
            assert gm.stats["t_inverse_ms"] > 0  # extended inside update(), not left to the next query
        # (a grown model of at most 1024 padded rows is rebuilt by the three-launch create in either mode -- faster than any
        # append, scripts/update_bench.py -- and gets its inverse factor from the factorisation launch itself)
        o = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        res[mode] = (gm.alpha.copy(), gm.D.copy(), o, gm.stats["n_negative_pivots"], gm.stats["alpha_residual"])
        gm.close()
    fresh = gpu.Model(kern, x, y, z, lab, s2, precision=prec)
    of = fresh.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    assert res["1"][3] == res["0"][3] == fresh.stats["n_negative_pivots"]
    assert (res["1"][3] > 0) == tp
    tol = 1e-9 if prec == 1 else 1e-5
    for other_alpha, other_D, other_o in ((res["0"][0], res["0"][1], res["0"][2]), (fresh.alpha, fresh.D, of)):
        assert nerr(res["1"][0], other_alpha) < tol
        assert nerr(res["1"][1], other_D) < (1e-9 if prec == 1 else 1e-3)
        for key in ("f", "grad"):
            assert nerr(res["1"][2][key], other_o[key]) < tol, key
        assert verr(res["1"][2]["v"], other_o["v"], k0) < tol
        assert verr_v(res["1"][2]["v"], other_o["v"]) < tol
    assert res["1"][4] < (1e-9 if prec == 1 else 1e-6)
    fresh.close()


@pytest.mark.parametrize("n", [16, 129, 300, 600, 1500])
def test_forced_fp32_training_of_small_models(gpu, orc, ds, n, monkeypatch):
    """GPX_PREC_F32 models of up to 2048 padded rows are trained in fp64 (free at that size).  With the switch off the
    fp32 kernel matrix / LDL^T / substitution run at these ragged sizes too, for the kernels that ARE fp32-trained at
    larger sizes (the thin plate never is while fp64 fits the device: its fp32 factorisation cost the variance up to
    9e-6 of max|v| here and 4.4e-5 on random clouds -- set_training_precision): everything at 1e-5."""
    monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")
    gpu.debug_reload()
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    q = _queries(ds, x, y, z, g=5)
    for kn, par in (("matern52", (1, 1)), ("gaussian", (1, 1))):
        om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
        ref = om.evaluate(*q, want_v=True, want_grad=True)
        for prec in (gpu.F32, gpu.F32_SPLIT):
            gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
            out = gm.evaluate(*q, want_v=True, want_grad=True)
            assert nerr(gm.alpha, om.alpha) < 1e-5
            assert nerr(out["f"], ref["f"]) < 1e-5 and nerr(out["grad"], ref["grad"]) < 1e-5
            assert verr(out["v"], ref["v"], _k0(om)) < 1e-5 and verr_v(out["v"], ref["v"]) < 1e-5
            gm.close()


@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("n", [1500, 2305])
def test_substitution_give_up_falls_back_to_step_launches(gpu, ds, prec, n, monkeypatch):
    """The one-launch block substitution hands results from lower to higher block rows through polled entries; a poll whose
    time budget is spent (GPX_WAIT_BUDGET_US=0 forces that -- and the same for the dataflow factorisation in front of it, which
    the launch chain then redoes) voids the solve, and create() recomputes alpha in-process with the launch-per-step kernels:
    success, gpx_stats.solve_fallbacks >= 1, the same alpha on every such run bit for bit, and the alpha of the default path to
    the working precision."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")
    with gpu.switches(GPX_WAIT_BUDGET_US="0"):
        gf = gpu.Model(kern, x, y, z, lab, s2, precision=prec)
        assert gf.stats["solve_fallbacks"] >= 1 and gf.stats["factor_gemm_launches"] > 0
        a_steps, res_steps = gf.alpha.copy(), gf.stats["alpha_residual"]
        g2 = gpu.Model(kern, x, y, z, lab, s2, precision=prec)
        np.testing.assert_array_equal(g2.alpha, a_steps)
        assert g2.stats["alpha_residual"] == res_steps
        g2.close()
        gf.update(x[:5] * 0.5, y[:5] * 0.5, z[:5] * 0.5, lab[:5], s2[:5])  # the update path takes the same fallback
        assert gf.stats["solve_fallbacks"] >= 1
        gf.close()
    gn = gpu.Model(kern, x, y, z, lab, s2, precision=prec)
    assert gn.stats["solve_fallbacks"] == 0
    assert nerr(gn.alpha, a_steps) < (1e-10 if prec == 1 else 1e-5)
    gn.close()


@pytest.mark.parametrize("prec", [1, 0, 2, 3])
def test_replicas_are_bit_identical_to_their_source(gpu, ds, prec):
    """gpx_model_replicate: read-only copies of a trained model on the listed devices (this box has one GPU, so the
    replicas land on device 0; across devices the same two blobs travel by hipMemcpyPeer).  Replicas answer
    evaluate / sample_surface / accessors exactly like the source and outlive it."""
    n = 2305
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel("thinplate", 4.0)
    src = gpu.Model(kern, x, y, z, lab, s2, precision=prec, device=0)
    qx, qy, qz = ds.query_grid(9)
    ref = src.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    reps = src.replicate([0, 0])
    assert len(reps) == 2
    a, D, R = src.alpha.copy(), src.D.copy(), src.R
    surf = src.sample_surface(qx, qy, qz, f_tol=0.05)
    src.close()  # replicas own their state
    for r in reps:
        out = r.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        for key in ("f", "v", "grad"):
            np.testing.assert_array_equal(out[key], ref[key])
        one = r.evaluate(qx[:3], qy[:3], qz[:3], want_v=True)
        np.testing.assert_array_equal(one["f"], ref["f"][:3])
        np.testing.assert_array_equal(r.alpha, a)
        np.testing.assert_array_equal(r.D, D)
        assert r.R == R and r.n == n
        s2_ = r.sample_surface(qx, qy, qz, f_tol=0.05)
        np.testing.assert_array_equal(s2_["idx"], surf["idx"])
        np.testing.assert_array_equal(s2_["v"], surf["v"])
    # a replica can still be updated: it rebuilds from its host copy of the data
    reps[0].update(x[:3] * 0.7, y[:3] * 0.7, z[:3] * 0.7, lab[:3], s2[:3])
    assert reps[0].n == n + 3
    for r in reps:
        r.close()
    with pytest.raises(gpu.GpxError):
        gpu.Model(kern, x[:50], y[:50], z[:50], lab[:50], s2[:50], precision=prec).replicate([7])


@pytest.mark.parametrize("prec", [1, 0])
@pytest.mark.parametrize("kname,kpar", [("matern52", (1.0, 1.0)), ("thinplate", (2.0,))])
@pytest.mark.parametrize("n", [100, 277, 1500, 2305])
def test_one_launch_substitution_equals_step_launches(gpu, ds, prec, kname, kpar, n, monkeypatch):
    """LDLT::solve (gp_regressor.hpp:163): the substitution in one launch per direction (workgroup per block row,
    self-validating hand-over) gives the alpha of the launch-per-block-step path (its fallback, forced by a wait budget of zero);
    both behind the launch chain's factor (GPX_DATAFLOW=0: no other wait gives up); 1 .. 19 block rows, a well-conditioned
    and an indefinite, ill-conditioned system.  No refinement: the raw solve."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel(kname, *kpar)
    res = {}
    monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")  # prec 0: the fp32 substitution kernels at every size
    for mode in ("0", "1"):
        with gpu.switches(GPX_DATAFLOW="0", GPX_WAIT_BUDGET_US="0" if mode == "1" else None):
            gm = gpu.Model(kern, x, y, z, lab, s2, precision=prec, ir_steps=0)
        # (with one or two block rows nothing may have had to wait: then there is nothing to give up)
        assert gm.stats["solve_fallbacks"] == int(mode) or (mode == "1" and n < 1500 and gm.stats["solve_fallbacks"] == 0)
        res[mode] = (gm.alpha.copy(), gm.stats["alpha_residual"])
        gm.close()
    # the two paths sum in different orders: equal up to the conditioning of the system times the working precision
    tol = {("matern52", 1): 1e-11, ("thinplate", 1): 1e-6, ("matern52", 0): 1e-3}.get((kname, prec))
    if tol is not None:
        assert nerr(res["0"][0], res["1"][0]) < tol
    assert res["0"][1] < 20 * res["1"][1] + 1e-12  # and the residual is no worse


def test_update_with_a_different_noise_level_falls_back_to_rebuild(gpu, orc, ds):
    """Appended points with a larger sigma2 come FIRST in Eigen's pivot order: not an append, the model is rebuilt."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(400)
    s2b = s2.copy()
    s2b[300:] = 0.5
    kern = ("matern52", (1.0, 1.0))
    gm = gpu.Model(gpu.make_kernel(kern[0], *kern[1]), x[:300], y[:300], z[:300], lab[:300], s2b[:300], precision=gpu.F64)
    gm.update(x[300:], y[300:], z[300:], lab[300:], s2b[300:])
    om = orc.Model(orc.make_kernel(kern[0], *kern[1]), x, y, z, lab, s2b)
    assert nerr(gm.alpha, om.alpha) < 1e-10
    qx, qy, qz = ds.query_grid(4)
    assert nerr(gm.evaluate(qx, qy, qz)["f"], om.evaluate(qx, qy, qz)["f"]) < 1e-10
    gm.close()


def test_kpp_accessor(gpu, orc, golden):
    x, y, z, lab, s2 = (golden["sphere64/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    om = orc.Model(orc.make_kernel("laplace", 1, 1), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel("laplace", 1, 1), x, y, z, lab, s2, precision=gpu.F64)
    assert nerr(gm.Kpp, om.Kpp) < 1e-14
    gm.close()


def test_variance_batching_and_large_query_sets(gpu, orc, ds):
    """Queries that straddle variance batches (query_batch) and the 128-query tile, nq >> N."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(300)
    om = orc.Model(orc.make_kernel("matern52", 1, 1), x, y, z, lab, s2)
    rng = np.random.default_rng(5)
    nq = 3 * 256 + 37
    q = rng.uniform(-1.2, 1.2, size=(3, nq))
    ref = om.evaluate(q[0], q[1], q[2], want_v=True, want_grad=True)
    for prec in (1, 0):
        for qb in (0, 256, 128):
            gm = gpu.Model(gpu.make_kernel("matern52", 1, 1), x, y, z, lab, s2, precision=prec, query_batch=qb)
            out = gm.evaluate(q[0], q[1], q[2], want_v=True, want_grad=True)
            for key in ("f", "v", "grad"):
                assert nerr(out[key], ref[key]) < TOL[prec], (key, qb)
            # a single query, the node's call pattern (src/gp_node.cpp:1069-1074)
            one = gm.evaluate(q[0][:1], q[1][:1], q[2][:1], want_v=True)
            assert abs(one["f"][0] - out["f"][0]) <= 1e-12 * (1 + abs(out["f"][0])) if prec else True
            assert abs(one["v"][0] - ref["v"][0]) < TOL[prec] * np.max(np.abs(ref["v"]))
            gm.close()


@pytest.mark.parametrize("fused", ["1", "0"])
@pytest.mark.parametrize("prec", [1, 0])
def test_project_matches_atlas_restatement(gpu, orc, ds, golden, prec, fused, monkeypatch):
    """the batched device-side AtlasBase::project (atlas.hpp:201-276) against the oracle's per-point
    restatement on the node's own model (mugD, ThinPlate(2.0)): same exits, same iteration counts, same points.
    fused = the whole loop in one launch (models that fit the LDS); otherwise one mean+gradient pass per iteration."""
    monkeypatch.setenv("GPX_PROJECT_FUSED", fused)
    gpu.debug_reload()
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    om = orc.Model(orc.make_kernel("thinplate", 2.0), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=prec)
    rng = np.random.default_rng(17)
    surf = np.stack([x, y, z], 1)[lab == 0]
    P = surf[rng.integers(0, len(surf), 300)] * rng.uniform(0.8, 1.3, size=(300, 1)) + rng.normal(0, 0.02, (300, 3))
    g0 = om.evaluate(P[:, 0], P[:, 1], P[:, 2], want_grad=True)["grad"]
    g0[:5] = 0.0                      # "wrong" start directions (atlas.hpp:246-249)
    tol_pos = 1e-9 if prec == 1 else 1e-4
    for kw in (dict(step_mul=0.5, max_iter=80), dict(max_iter=24), dict(step_mul=0.5, max_iter=0)):
        ref = om.project(P[:, 0], P[:, 1], P[:, 2], g0, **kw)
        out = gm.project(P[:, 0], P[:, 1], P[:, 2], g0, **kw)
        same = (out["status"] == ref["status"]) & (out["iter"] == ref["iter"])
        # a tolerance test can flip where |f| sits within rounding of f_tol; in fp64 that does not happen here
        assert same.mean() >= (1.0 if prec == 1 else 0.97), kw
        assert np.max(np.abs(out["xyz"][same] - ref["xyz"][same])) < tol_pos, kw
        assert np.max(np.abs(out["f"][same] - ref["f"][same])) < (1e-9 if prec == 1 else 1e-4), kw
        if kw["max_iter"] > 0 and kw.get("step_mul", 0) == 0.5:
            assert (ref["status"] == 1).mean() > 0.9
    gm.close()


@pytest.mark.parametrize("prec", [1, 0, 2])
def test_small_batch_path_agrees_with_general_path(gpu, orc, ds, golden, prec):
    """Up to 64 queries on a model of N <= 1024 take the one-launch path (a workgroup per query, pinned I/O);
    the same queries inside a larger batch take the general path.  Both agree with each other and the oracle."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    kn, par = "thinplate", (2.0,)
    om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
    rng = np.random.default_rng(5)
    Q = rng.uniform(-1.1, 1.1, size=(300, 3))
    Q[:4] = np.stack([x, y, z], 1)[:4]                       # on training points
    big = gm.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True, want_basis=True)
    ref = om.evaluate(Q[:40, 0], Q[:40, 1], Q[:40, 2], want_v=True, want_grad=True, want_basis=True)
    tol = 1e-10 if prec == 1 else 1e-5
    for lo, hi in ((0, 1), (1, 40), (40, 104)):
        small = gm.evaluate(Q[lo:hi, 0], Q[lo:hi, 1], Q[lo:hi, 2], want_v=True, want_grad=True, want_basis=True)
        for key in ("f", "grad"):
            assert nerr(small[key], big[key][lo:hi]) < 1e-12 * (1 if prec == 1 else 1e4), key
        assert verr(small["v"], big["v"][lo:hi], 8.0) < tol
        gn = np.linalg.norm(big["grad"][lo:hi], axis=1)
        ok = gn > 1e-2 * gn.max()
        for key in ("tx", "ty"):
            assert np.max(np.abs(small[key][ok] - big[key][lo:hi][ok])) < 1e-6, key
        if hi <= 40:
            assert nerr(small["f"], ref["f"][lo:hi]) < (1e-10 if prec == 1 else 1e-5)
            assert verr(small["v"], ref["v"][lo:hi], 8.0) < tol
    f_only = gm.evaluate(Q[:3, 0], Q[:3, 1], Q[:3, 2])["f"]      # mean only: no inverse factor needed
    assert nerr(f_only, big["f"][:3]) < 1e-12 * (1 if prec == 1 else 1e4)
    gm.close()


def test_host_batches_larger_than_one_slice(gpu, orc, ds):
    """Host evaluate slices very large batches (2^20 queries per slice); results do not depend on the slicing."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(64)
    gm = gpu.Model(gpu.make_kernel("laplace", 1, 1), x, y, z, lab, s2, precision=gpu.F64)
    om = orc.Model(orc.make_kernel("laplace", 1, 1), x, y, z, lab, s2)
    rng = np.random.default_rng(11)
    nq = (1 << 20) + 12345
    q = rng.uniform(-1.1, 1.1, size=(3, nq))
    out = gm.evaluate(q[0], q[1], q[2], want_v=True, want_grad=True)
    sel = np.concatenate([np.arange(0, nq, 40009), [(1 << 20) - 1, 1 << 20, nq - 1]])
    ref = om.evaluate(q[0][sel], q[1][sel], q[2][sel], want_v=True, want_grad=True)
    for key in ("f", "v", "grad"):
        assert nerr(out[key][sel], ref[key]) < 1e-10, key
    gm.close()


def test_concurrent_single_point_evaluate(gpu, orc, ds):
    """fakeDeterministicSampling: 841 host threads evaluate ONE point each on the same const model
    (src/gp_node.cpp:1027-1038); the call must be re-entrant."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(200)
    gm = gpu.Model(gpu.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=gpu.F64)
    om = orc.Model(orc.make_kernel("thinplate", 2.0), x, y, z, lab, s2)
    # 29 x 29 = 841 threads at once: one plane of the node's 29^3 lattice, the number it starts per pass
    # (src/gp_node.cpp:1027-1038)
    t29 = np.linspace(-1.01, 1.01, 29)
    gx, gy = np.meshgrid(t29, t29, indexing="ij")
    qx, qy, qz = gx.ravel().copy(), gy.ravel().copy(), np.full(841, 0.07)
    ref = om.evaluate(qx, qy, qz, want_v=True)
    f = np.zeros(len(qx))
    v = np.zeros(len(qx))
    errs = []

    def work(i):
        try:
            o = gm.evaluate(qx[i:i + 1], qy[i:i + 1], qz[i:i + 1], want_v=True)
            f[i], v[i] = o["f"][0], o["v"][0]
        except Exception as e:  # noqa
            errs.append(e)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(qx))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs
    assert nerr(f, ref["f"]) < 1e-10 and nerr(v, ref["v"]) < 1e-10
    gm.close()


def test_independent_models_from_concurrent_threads(gpu, orc, ds):
    """C5 shape on one GPU: eight independent models created, evaluated and destroyed from eight host threads at the
    same time (each model has its own stream; the launchers' one-time function attributes are process-wide)."""
    import threading
    sizes = [266, 481, 466, 724, 447, 712, 354, 277]  # the eight reference objects + 15 exterior points each
    kerns = [("thinplate", (2.0,)), ("matern52", (1.0, 1.0)), ("gaussian", (1.0, 1.0)), ("laplace", (1.0, 1.0))]
    qx, qy, qz = ds.query_grid(9)
    errs, lock = [], threading.Lock()

    def work(i):
        try:
            n = sizes[i]
            x, y, z, lab, s2 = ds.fibonacci_training_set(n, seed=100 + i)
            kn, par = kerns[i % len(kerns)]
            for rep in range(3):
                prec = gpu.F64 if (i + rep) % 2 == 0 else gpu.F32
                gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
                out = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True)
                one = gm.evaluate(qx[:3], qy[:3], qz[:3], want_v=True)     # the one-launch path as well
                gm.close()
                om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
                ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
                k0 = float(orc.k(orc.make_kernel(kn, *par), 0.0)[0])
                tol = 1e-9 if prec == gpu.F64 else 1e-5
                assert nerr(out["f"], ref["f"]) < tol and verr(out["v"], ref["v"], k0) < tol
                assert nerr(out["grad"], ref["grad"]) < tol
                assert nerr(one["f"], ref["f"][:3]) < tol and verr(one["v"], ref["v"][:3], k0) < tol
        except Exception as e:  # noqa: BLE001 -- reported below
            with lock:
                errs.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs


def test_no_device_memory_leak_over_model_lifecycles(gpu, ds):
    """create / evaluate (general + one-launch paths) / sample_surface / project / update / destroy, 60 times in every
    precision mode: the free device memory returns to where it started (hipFree of every workspace, temporary and
    kept factor)."""
    torch = pytest.importorskip("torch")
    x, y, z, lab, s2 = ds.fibonacci_training_set(420)
    qx, qy, qz = ds.query_grid(7)

    def cycle(prec):
        gm = gpu.Model(gpu.make_kernel("matern52", 1, 1), x[:300], y[:300], z[:300], lab[:300], s2[:300], precision=prec)
        gm.evaluate(qx, qy, qz, want_v=True, want_grad=True, want_basis=True)
        gm.evaluate(qx[:2], qy[:2], qz[:2], want_v=True)
        gm.sample_surface(qx, qy, qz, f_tol=0.05)
        g = gm.evaluate(qx[:8], qy[:8], qz[:8], want_grad=True)["grad"]
        gm.project(qx[:8], qy[:8], qz[:8], g, step_mul=0.5, max_iter=20)
        gm.update(x[300:], y[300:], z[300:], lab[300:], s2[300:])
        gm.evaluate(qx, qy, qz, want_v=True)
        gm.close()

    for prec in (gpu.F64, gpu.F32, gpu.MIXED, gpu.F32_SPLIT):
        cycle(prec)  # first use: one-time allocations of the runtime
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for rep in range(15):
        for prec in (gpu.F64, gpu.F32, gpu.MIXED, gpu.F32_SPLIT):
            cycle(prec)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, "device memory shrank by %.1f MiB over 60 model lifecycles" % ((free0 - free1) / 2**20)


def test_device_resident_evaluate(gpu, orc, ds):
    """gpx_model_evaluate_device: inputs/outputs stay in HBM (torch only provides the memory)."""
    torch = pytest.importorskip("torch")
    x, y, z, lab, s2 = ds.fibonacci_training_set(400)
    gm = gpu.Model(gpu.make_kernel("gaussian", 1, 1), x, y, z, lab, s2, precision=gpu.F64, prepare_variance=True)
    om = orc.Model(orc.make_kernel("gaussian", 1, 1), x, y, z, lab, s2)
    qx, qy, qz = ds.query_grid(8)
    dq = [torch.from_numpy(a).cuda() for a in (qx, qy, qz)]
    nq = len(qx)
    df, dv = torch.empty(nq, dtype=torch.float64, device="cuda"), torch.empty(nq, dtype=torch.float64, device="cuda")
    dg = torch.empty(nq, 3, dtype=torch.float64, device="cuda")
    gm.evaluate_device(nq, dq[0].data_ptr(), dq[1].data_ptr(), dq[2].data_ptr(), df.data_ptr(), dv.data_ptr(),
                       dg.data_ptr())
    gm.sync()
    ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    assert nerr(df.cpu().numpy(), ref["f"]) < 1e-10
    assert nerr(dv.cpu().numpy(), ref["v"]) < 1e-10
    assert nerr(dg.cpu().numpy(), ref["grad"]) < 1e-10
    st = gm.stats
    assert st["t_mean_ms"] > 0 and st["t_var_ms"] > 0 and st["var_gemm_launches"] == 1
    gm.close()


def test_device_evaluate_on_two_caller_streams(gpu, ds):
    """Evaluations enqueued on different caller streams share the model's workspaces; the library orders them
    (an event after each evaluation), so overlapping submissions give the same results as sequential ones."""
    torch = pytest.importorskip("torch")
    x, y, z, lab, s2 = ds.fibonacci_training_set(1500)
    gm = gpu.Model(gpu.make_kernel("matern52", 1, 1), x, y, z, lab, s2, precision=gpu.F32, prepare_variance=True)
    rng = np.random.default_rng(23)
    nq = 40000
    qs = [torch.from_numpy(rng.uniform(-1.1, 1.1, size=(3, nq))).cuda() for _ in range(2)]
    outs = [(torch.empty(nq, dtype=torch.float64, device="cuda"), torch.empty(nq, dtype=torch.float64, device="cuda"))
            for _ in range(2)]
    refs = []
    for q in qs:  # sequential reference on the model's own stream
        f, v = torch.empty(nq, dtype=torch.float64, device="cuda"), torch.empty(nq, dtype=torch.float64, device="cuda")
        gm.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
        gm.sync()
        refs.append((f.cpu().numpy(), v.cpu().numpy()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for _ in range(3):  # back-to-back submissions, no synchronisation in between
        for q, (f, v), st in zip(qs, outs, streams):
            gm.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr(),
                               stream=st.cuda_stream)
    torch.cuda.synchronize()
    for (f, v), (rf, rv) in zip(outs, refs):
        np.testing.assert_array_equal(f.cpu().numpy(), rf)
        np.testing.assert_array_equal(v.cpu().numpy(), rv)
    gm.close()


def test_shell_broadcast_commit_roundtrip(gpu, orc, ds):
    """Sharded-grid path on one GPU: copy the state blobs of a factorised model into a shell
    (stand-in for the RCCL broadcast), commit, and evaluate."""
    torch = pytest.importorskip("torch")
    import importlib
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    x, y, z, lab, s2 = ds.fibonacci_training_set(300)
    kern = gpu.make_kernel("matern52", 1, 1)
    for prec in (1, 0, 3):
        src = gpu.Model(kern, x, y, z, lab, s2, precision=prec, prepare_variance=True)
        dst = gpu.Model.shell(kern, 300, precision=prec)
        with pytest.raises(gpu.GpxError):
            dst.evaluate([0.0], [0.0], [0.0])  # not committed yet
        for part in (0, 1):
            a = sh.device_blob_as_tensor(torch, *src.state_blob(part), "cuda")
            b = sh.device_blob_as_tensor(torch, *dst.state_blob(part), "cuda")
            assert a.numel() == b.numel()
            b.copy_(a)
        torch.cuda.synchronize()
        dst.commit(with_variance=True)
        qx, qy, qz = ds.query_grid(6)
        o1 = src.evaluate(qx, qy, qz, want_v=True)
        o2 = dst.evaluate(qx, qy, qz, want_v=True)
        np.testing.assert_array_equal(o1["f"], o2["f"])
        np.testing.assert_array_equal(o1["v"], o2["v"])
        src.close()
        dst.close()


@pytest.mark.parametrize("prec", [1, 0])
def test_march_surface_follows_the_reference_walk(gpu, orc, ds, prec):
    """gpx_model_march_surface == marchingSampling + marchingCubes (src/gp_node.cpp:1102-1291) as restated in
    oracle/gp_oracle.c:orc_march_surface: same start point (lattice search), same cubes, the same kept points in the
    same order with float-computed coordinates bit-identical, f and v at the fp64 / fp32 tolerances; an explicit
    start point, the max_cubes cut and the capacity overflow behave alike."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(300)
    kn, par = "thinplate", (2.0,)
    om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
    tol = 1e-10 if prec == 1 else 1e-5
    for leaf, step, start in ((0.15, 0.05, None), (0.2, 0.05, None), (0.15, 0.05, (0.95, 0.15, 0.15))):
        ref = om.march_surface(leaf, step, start=start)
        out = gm.march_surface(leaf, step, start=start)
        assert ref["n_total"] > 1000 and ref["n_cubes"] > 100
        assert out["n_total"] == ref["n_total"] and out["n_cubes"] == ref["n_cubes"] and not out["truncated"]
        np.testing.assert_array_equal(out["xyz"], ref["xyz"])
        assert np.max(np.abs(out["f"] - ref["f"])) < tol * 1e-2 + 1e-12  # |f| <= 0.01 on every kept point
        assert verr(out["v"], ref["v"], 8.0) < tol
        assert np.max(np.abs(out["f"])) <= 0.01
    # no kept lattice point sits on the threshold (the walk would then depend on the last bit of f)
    ref = om.march_surface(0.15, 0.05, f_tol=0.0105)
    assert ref["n_total"] != om.march_surface(0.15, 0.05)["n_total"]
    # cut after 40 cubes, and a capacity smaller than the result
    r40, o40 = om.march_surface(0.15, 0.05, max_cubes=40), gm.march_surface(0.15, 0.05, max_cubes=40)
    assert o40["n_cubes"] == r40["n_cubes"] == 40 and o40["n_total"] == r40["n_total"]
    np.testing.assert_array_equal(o40["xyz"], r40["xyz"])
    few = gm.march_surface(0.15, 0.05, capacity=100)
    assert few["truncated"] and few["n_total"] == om.march_surface(0.15, 0.05)["n_total"] and len(few["f"]) == 100
    np.testing.assert_array_equal(few["xyz"], om.march_surface(0.15, 0.05)["xyz"][:100])
    # a model whose surface misses the search lattice: the reference's error
    far = gpu.Model(gpu.make_kernel("gaussian", 1, 3), [0.0, 0.1], [0.0, 0.0], [0.0, 0.0], [1.0, 1.0], [0.1, 0.1], precision=prec)
    with pytest.raises(gpu.GpxError, match="No starting point found"):
        far.march_surface(0.15, 0.05)
    far.close()
    gm.close()


@pytest.mark.parametrize("prec", [1, 0, 3])
@pytest.mark.parametrize("n", [700, 1500])
def test_variance_batches_are_bit_identical(gpu, ds, prec, n):
    """The variance is evaluated per batch of queries (operand buffer / coefficient array of bounded size): many small
    batches with a ragged last one give bit-identical values to one batch, call after call.  700 points: the small-model
    kernel (gpx_varcols_kernel.hpp) for the fp32 mode; 1500: the 128 x 128 tiles."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    qx, qy, qz = ds.query_grid(11)  # 1331 queries
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    res = {}
    for qb in (256, 0):
        gm = gpu.Model(kern, x, y, z, lab, s2, precision=prec, query_batch=qb)
        res[qb] = [gm.evaluate(qx, qy, qz, want_v=True)["v"] for _ in range(2)]
        gm.close()
    for a in res[256] + res[0]:
        np.testing.assert_array_equal(a, res[0][0])


def test_sample_surface_matches_filtered_evaluate(gpu, orc, ds):
    """gpx_model_sample_surface == evaluate everywhere, keep |f| <= tol (src/gp_node.cpp:1066-1100),
    but the variance is only computed for the survivors."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(500)
    om = orc.Model(orc.make_kernel("thinplate", 2.0), x, y, z, lab, s2)
    qx, qy, qz = ds.query_grid(29)  # the node's 29^3 lattice (sample_res 0.07 at scale 1.01)
    ref = om.evaluate(qx, qy, qz, want_v=False)
    tol = 0.01
    near = np.abs(np.abs(ref["f"]) - tol)
    assert near.min() > 1e-9  # no lattice point sits on the threshold
    keep = np.nonzero(np.abs(ref["f"]) <= tol)[0]
    assert 50 < len(keep) < len(qx) // 10
    refv = om.evaluate(qx[keep], qy[keep], qz[keep], want_v=True)["v"]
    for prec in (1, 0, 2):
        gm = gpu.Model(gpu.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=prec)
        out = gm.sample_surface(qx, qy, qz, f_tol=tol)
        np.testing.assert_array_equal(out["idx"], keep)
        assert out["n_total"] == len(keep) and not out["truncated"]
        assert nerr(out["f"], ref["f"][keep]) < 1e-6 if prec == 0 else nerr(out["f"], ref["f"][keep]) < 1e-9
        assert verr(out["v"], refv, 8.0) < TOL[prec]
        # identical to the unfused path on the same model
        full = gm.evaluate(qx, qy, qz, want_v=True)
        np.testing.assert_allclose(out["v"], full["v"][keep], rtol=0, atol=1e-5 * 8.0 if prec != 1 else 1e-12)
        # truncation: capacity smaller than the number of survivors
        small = gm.sample_surface(qx, qy, qz, f_tol=tol, capacity=10)
        assert small["truncated"] and small["n_total"] == len(keep)
        np.testing.assert_array_equal(small["idx"], keep[:10])
        # nothing survives
        none = gm.sample_surface(qx[:100], qy[:100], qz[:100], f_tol=0.0)
        assert none["n_total"] == 0 and len(none["idx"]) == 0
        gm.close()


@pytest.mark.parametrize("kn,par,n", [("matern52", (1.0, 1.0), 2300), ("gaussian", (1.3, 0.7), 600), ("laplace", (0.8, 1.2), 277),
                                      ("matern32", (1.0, 0.5), 1500)])
def test_sample_surface_screen_selects_exactly_the_fp64_set(gpu, ds, kn, par, n):
    """Round 6: on grids of >= 32768 queries gpx_model_sample_surface screens the lattice with an fp32 mean
    and a proved bound on its error (csrc/gpx_predict.hip, screen_kernel), evaluates the fp64 mean of the candidates only and
    applies the exact test to those.  The selected indices, f and v must be those of the fp64 filter BIT FOR BIT: against
    evaluate() on the whole grid, in every precision mode, on a cloud far from the origin, with a few non-finite queries in the
    middle of the grid (which neither survive nor disturb anything else), and the screen must actually have run (candidates <
    queries) and be worth it (candidates within a small multiple of the survivors)."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    off = (40.0, -25.0, 7.5)
    x, y, z = x + off[0], y + off[1], z + off[2]
    qx, qy, qz = (c.copy() for c in ds.query_grid(40))  # 64000 lattice points
    qx += off[0]; qy += off[1]; qz += off[2]
    bad = [1234, 30000, 63999]
    qx[bad[0]], qy[bad[1]], qz[bad[2]] = np.nan, np.inf, -np.inf
    tol = 0.01
    for prec in (gpu.F64, gpu.F32, gpu.MIXED):
        gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec, prepare_variance=True)
        full = gm.evaluate(qx, qy, qz, want_v=True)
        fmean = gm.evaluate(qx, qy, qz)["f"]  # (the mean kernel; small fp64 models carry f on the variance kernel when v is asked)
        keep = np.nonzero(np.abs(fmean) <= tol)[0]
        assert 100 < len(keep) < len(qx) // 4, len(keep)
        out = gm.sample_surface(qx, qy, qz, f_tol=tol)
        cand = gm.stats["surface_candidates"]
        np.testing.assert_array_equal(out["idx"], keep)
        np.testing.assert_array_equal(out["f"], fmean[keep])
        if prec == gpu.F64:  # (the fp32 contractions batch their queries: the last bits of v depend on the batch, as before)
            np.testing.assert_array_equal(out["v"], full["v"][keep])
        else:  # (an infinite query has f = 0 -- it survives -- and no finite variance, in either path)
            fin = np.isfinite(full["v"][keep])
            assert np.array_equal(fin, np.isfinite(out["v"])) and fin.sum() >= len(keep) - 3
            assert verr_v(out["v"][fin], full["v"][keep][fin]) < 1e-6
        assert len(keep) <= cand < len(qx) and cand < 6 * len(keep) + 2000, (cand, len(keep))
        # a small grid takes the plain path: the same answer
        sl = slice(20000, 50000)
        small = gm.sample_surface(qx[sl], qy[sl], qz[sl], f_tol=tol)
        k2 = keep[(keep >= 20000) & (keep < 50000)] - 20000
        np.testing.assert_array_equal(small["idx"], k2)
        assert gm.stats["surface_candidates"] == 30000
        gm.close()


def test_error_codes_on_device(gpu):
    k = gpu.make_kernel("gaussian", 1, 1)
    m = gpu.Model(k, [0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [0.0, 0.0, 0.0], [0.0, 1.0, 1.0], [0.1, 0.1, 0.1])
    with pytest.raises(gpu.GpxError) as ei:
        m.evaluate([], [], [])
    assert ei.value.code == gpu.E_EMPTY and ei.value.message == "All input data is empty!"
    with pytest.raises(gpu.GpxError) as ei:
        m.evaluate([0.0], [0.0], [0.0], label=[1.0])
    assert ei.value.code == gpu.E_LABELED_QUERY
    m.close()
    # exactly singular: two coincident points without noise
    with pytest.raises(gpu.GpxError) as ei:
        gpu.Model(gpu.make_kernel("thinplate", 1.0), [0.0, 0.0], [0.0, 0.0], [0.0, 0.0], [1.0, 1.0], None,
                  precision=gpu.F64)
    assert ei.value.code == gpu.E_SINGULAR


def test_small_model_variance_paths_agree(gpu, orc, ds, tmp_path):
    """Models of up to 1024 points take the variance contraction of gpx_varcols_kernel.hpp (every row fragment resident in one
    wave, the triangle of X skipped per 16-row fragment, fp64 add-back inside the triangle, operand formed in the wave).
    Against the general path (GPX_VAR_COLS=0: kqp_kernel -> 128 x 128 tiles -> var_finish) the fp32 contraction is the same
    k-ordered sum: 5e-7 of max|v|; each within 1e-5 of the fp64 oracle.  Sizes: one pass (<= 192 rows), 2 .. 6 passes, a last fragment with 1 and with 15
    padding rows, a cloud translated by (10, -7, 3); every covariance function (the thin plate's operand is formed in fp64
    and always read from the buffer)."""
    import subprocess, sys
    child = (
        "import sys, importlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "gpx = importlib.import_module('gaussian-object-modelling_amd.gpx')\n"
        "ds = importlib.import_module('gaussian-object-modelling_amd.datasets')\n"
        "out = {}\n"
        "qx, qy, qz = ds.query_grid(9)\n"
        "for n, off in ((17, 0), (100, 0), (193, 0), (277, 0), (300, 1), (511, 0), (724, 0), (1024, 0)):\n"
        "    x, y, z, lab, s2 = ds.fibonacci_training_set(n)\n"
        "    o = (10.0, -7.0, 3.0) if off else (0.0, 0.0, 0.0)\n"
        "    for kn, par in (('matern52', (1.0, 1.0)), ('matern32', (1.0, 0.7)), ('gaussian', (1.0, 1.0)), ('laplace', (0.8, 1.3)), ('thinplate', (4.0,))):\n"
        "        gm = gpx.Model(gpx.make_kernel(kn, *par), x + o[0], y + o[1], z + o[2], lab, s2, precision=gpx.F32)\n"
        "        out['%%d/%%s' %% (n, kn)] = gm.evaluate(qx + o[0], qy + o[1], qz + o[2], want_v=True)['v']\n"
        "        gm.close()\n"
        "np.savez(sys.argv[1], **out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base_env = {k: v for k, v in os.environ.items() if not k.startswith("GPX_VAR_")}
    res = {}
    for name, extra in (("cols", {}), ("general", {"GPX_VAR_COLS": "0"})):
        path = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", child, path], env=dict(base_env, **extra), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = np.load(path)
    qx, qy, qz = ds.query_grid(9)
    assert len(res["cols"].files) == 40
    for key in res["cols"].files:
        n, kn = key.split("/")
        vmax = np.max(np.abs(res["general"][key]))
        assert np.max(np.abs(res["cols"][key] - res["general"][key])) / vmax < 5e-7, key
    # the fp64 oracle on a subset (it is the slow side)
    for n, kn, par in ((277, "matern52", (1.0, 1.0)), (724, "gaussian", (1.0, 1.0)), (193, "matern32", (1.0, 0.7)), (511, "thinplate", (4.0,))):
        x, y, z, lab, s2 = ds.fibonacci_training_set(n)
        ref = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2).evaluate(qx, qy, qz, want_v=True)["v"]
        assert np.max(np.abs(res["cols"]["%d/%s" % (n, kn)] - ref)) / np.max(np.abs(ref)) < 1e-5, (n, kn)


def test_small_model_variance_over_more_than_one_whole_call_batch(gpu, ds):
    """The small-model kernel takes a whole evaluate call as ONE launch up to 2^21 queries (its only per-query workspace is the
    fit's coefficient array); a longer call is cut there.  2^21 + 77 device-resident queries on a 150-point model: the two
    pieces (second one: 77 queries in a single 128-query tile, columns past the last query computed and discarded) equal
    separate calls on the same points bit for bit, on the model's own stream and on a caller's stream."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    x, y, z, lab, s2 = ds.fibonacci_training_set(150)
    gm = gpu.Model(gpu.make_kernel("matern52", 1.0, 1.0), x, y, z, lab, s2, precision=gpu.F32, prepare_variance=True)
    nq = (1 << 21) + 77
    gen = torch.Generator(device="cpu").manual_seed(7)
    q = [(torch.rand(nq, generator=gen, dtype=torch.float64) * 2.4 - 1.2).to(dev) for _ in range(3)]
    f = torch.empty(nq, dtype=torch.float64, device=dev)
    v = torch.empty_like(f)
    gm.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
    gm.sync()
    assert gm.stats["var_gemm_launches"] == 2
    side = torch.cuda.Stream(device=dev)
    for lo, hi in ((0, 1000), ((1 << 21) - 50, (1 << 21) + 77), (nq - 77, nq), (12345, 12345 + 4097)):
        n1 = hi - lo
        f1 = torch.empty(n1, dtype=torch.float64, device=dev)
        v1 = torch.empty_like(f1)
        qq = [t[lo:hi].contiguous() for t in q]
        torch.cuda.synchronize()
        gm.evaluate_device(n1, qq[0].data_ptr(), qq[1].data_ptr(), qq[2].data_ptr(), f1.data_ptr(), v1.data_ptr(),
                           stream=side.cuda_stream)
        side.synchronize()
        assert torch.equal(f1, f[lo:hi]) and torch.equal(v1, v[lo:hi]), (lo, hi)
    assert float(v.min()) > -1e-6 and float(v.max()) <= 1.0 + 1e-9
    gm.close()


def test_variance_tiles_agree(gpu, ds, tmp_path):
    """The one-wave variance tile (gpx_vargemm.hip, default) against its documented fallback, the LDS-staged tile of
    gpx_gemm.hip (GPX_VAR_TILE=3; the switch is read once per process, hence the children): the same contraction
    (gp_regressor.hpp:316-319) with the same k-to-lane assignment, so the fp32 accumulators agree to rounding and the
    variances to 5e-7 of max|v| (3e-6 with the plain fp32 epilogue); each tile within 1e-5 of the fp64 pipeline in the survey's metric.  F32 with the fit
    (fp64 epilogue on the fp64 matrix pipe in tile 6), F32 without it (GPX_VAR_FIT=0: plain fp32 epilogue), MIXED, and F64
    (one-wave 128 x 64 fp64 tile against the LDS-staged fp64 tile: 1e-12);
    thin-plate included; 1100 rows (9 row tiles, the last partly padding) and 2305 (19 row tiles).  (Models of up to 1024
    points never reach these tiles in the fp32 modes: test_small_model_variance_paths_agree.)"""
    import subprocess, sys
    child = (
        "import sys, importlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "gpx = importlib.import_module('gaussian-object-modelling_amd.gpx')\n"
        "ds = importlib.import_module('gaussian-object-modelling_amd.datasets')\n"
        "out = {}\n"
        "for n in (1100, 2305):\n"
        "    x, y, z, lab, s2 = ds.fibonacci_training_set(n)\n"
        "    qx, qy, qz = ds.query_grid(7)\n"
        "    for kn, par in (('matern52', (1.0, 1.0)), ('thinplate', (4.0,))):\n"
        "        for prec in (gpx.F64, gpx.F32, gpx.MIXED):\n"
        "            gm = gpx.Model(gpx.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)\n"
        "            out['%%d/%%s/%%d' %% (n, kn, prec)] = gm.evaluate(qx, qy, qz, want_v=True)['v']\n"
        "            gm.close()\n"
        "np.savez(sys.argv[1], **out)\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    base_env = {k: v for k, v in os.environ.items() if k not in ("GPX_VAR_TILE", "GPX_VAR_FIT")}
    for tile, fit in (("6", "1"), ("3", "1"), ("6", "0"), ("3", "0")):
        path = str(tmp_path / ("tile%s_%s.npz" % (tile, fit)))
        env = dict(base_env, GPX_VAR_TILE=tile, GPX_VAR_FIT=fit)
        r = subprocess.run([sys.executable, "-c", child, path], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[tile, fit] = np.load(path)
    keys = sorted(res["6", "1"].files)
    assert len(keys) == 12
    for key in keys:
        n, kn, prec = key.split("/")
        ref = res["6", "1"]["%s/%s/%d" % (n, kn, gpu.F64)]
        vmax = np.max(np.abs(ref))
        for (tile, fit), r in res.items():
            if int(prec) == gpu.F64:  # fp64 models: tile 6 = the one-wave fp64 tile, every other value = the LDS-staged one
                assert np.max(np.abs(r[key] - res["6", "1"][key])) / vmax < 1e-12, (key, tile, fit)
                continue
            if fit == "1":  # (without the fit the operand keeps k's full magnitude: the round-2 accuracy, not asserted here)
                assert np.max(np.abs(r[key] - ref)) / vmax < 1e-5, (key, tile, fit)
            # (the plain epilogue sums the squares in fp32, each tile in its own order)
            # (and without the fit the thin plate's operand and quadratic form are of size k(0) = R^3 = 64, not of size v)
            scale = vmax if fit == "1" or kn != "thinplate" else 64.0
            # Without the fit (GPX_VAR_FIT=0) the operand keeps k's full magnitude and the plain epilogue sums w^2 / D in fp32:
            # the result then depends on the ORDER of the k sum at the 4e-6 level (round 3: 3.95e-6 between tile 3 and tile 6 with
            # tile 6's walk of k turned around, 2.6e-6 in the default order) -- 5e-6 covers both, the fit path stays at 5e-7.
            assert np.max(np.abs(r[key] - res["6", fit][key])) / scale < (5e-7 if fit == "1" else 5e-6), (key, tile, fit)


@pytest.mark.parametrize("kn,par", [("gaussian", (1.3, float("inf"))), ("laplace", (0.7, 1e200)), ("matern52", (1.0, float("inf")))])
def test_degenerate_length_scales(gpu, orc, ds, kn, par):
    """kernels/gaussian.hpp:15-20 etc. put no condition on the length scale.  An infinite one makes the decay parameter 0
    (k = const): the mean / gradient kernel's table-based exponential is selected on the host only for a positive finite
    decay, anything else takes the general path; a huge finite one (decay 1e-200) stays on the table path."""
    n = 200
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    om = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
    gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=gpu.F64)
    qx, qy, qz = ds.query_grid(5)
    ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    out = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    assert np.all(np.isfinite(out["f"])) and np.all(np.isfinite(out["grad"]))
    assert nerr(gm.alpha, om.alpha) < 1e-9
    assert nerr(out["f"], ref["f"]) < 1e-9
    assert np.max(np.abs(out["grad"] - ref["grad"])) <= 1e-9 * max(np.max(np.abs(ref["grad"])), 1e-300) + 1e-300
    gm.close()


@pytest.mark.parametrize("offset", [10.0, 100.0])
@pytest.mark.parametrize("kn,par,n", [("matern52", (1.0, 1.0), 768), ("thinplate", (4.0,), 768), ("matern52", (1.0, 1.0), 2305)])
def test_cloud_far_from_the_origin(gpu, orc, ds, kn, par, n, offset):
    """the fp32 variance contraction expanded d^2 = |q|^2 - 2 q.p + |p|^2 in RAW coordinates, so a
    cloud of radius 0.5 sitting at offset 100 (metre coordinates in a camera frame) lost the variance to cancellation
    (6e-3 k(0)).  Now every working-precision quantity is relative to the model's centre -- the fp32 points (kernel
    matrix of an fp32-trained model: the n = 2305 Matern case), the basis of the low-rank fit and its query-side
    coefficients -- while the fp64 paths use differences of the given coordinates.  Held to 1e-5 in both metrics."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    qx, qy, qz = ds.query_grid(7)
    sh = np.array([offset, -0.5 * offset, 0.25 * offset])
    x, y, z = 0.5 * x + sh[0], 0.5 * y + sh[1], 0.5 * z + sh[2]
    qx, qy, qz = 0.5 * qx + sh[0], 0.5 * qy + sh[1], 0.5 * qz + sh[2]
    kern_o = orc.make_kernel(kn, *par)
    om = orc.Model(kern_o, x, y, z, lab, s2)
    ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    k0 = float(orc.k(kern_o, 0.0)[0])
    for prec in (gpu.F32, gpu.MIXED, gpu.F32_SPLIT, gpu.F64):
        gm = gpu.Model(gpu.make_kernel(kn, *par), x, y, z, lab, s2, precision=prec)
        out = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        tol = 1e-9 if prec == gpu.F64 else 1e-5  # (fp64: the oracle's own distances lose 1e-12 at offset 100)
        assert nerr(out["f"], ref["f"]) < tol and nerr(out["grad"], ref["grad"]) < tol, (prec, offset)
        assert verr(out["v"], ref["v"], k0) < tol and verr_v(out["v"], ref["v"]) < tol, (prec, offset)
        gm.close()


def test_mean_only_shell_has_no_variance_and_no_replicas(gpu, ds):
    """a shell committed WITHOUT its inverse factor holds no LDL^T either; asking it for a variance
    or for replicas (which carry the inverse factor) must be a status, not a GPU fault on a null factor."""
    torch = pytest.importorskip("torch")
    import importlib
    sh = importlib.import_module("gaussian-object-modelling_amd.sharding")
    n = 300
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    src = gpu.Model(kern, x, y, z, lab, s2, precision=gpu.F64)
    dst = gpu.Model.shell(kern, n, precision=gpu.F64)
    a = sh.device_blob_as_tensor(torch, *src.state_blob(0), "cuda")
    b = sh.device_blob_as_tensor(torch, *dst.state_blob(0), "cuda")
    b.copy_(a)
    torch.cuda.synchronize()
    dst.commit(with_variance=False)
    qx, qy, qz = ds.query_grid(5)
    np.testing.assert_array_equal(dst.evaluate(qx, qy, qz)["f"], src.evaluate(qx, qy, qz)["f"])
    with pytest.raises(gpu.GpxError) as ei:
        dst.evaluate(qx, qy, qz, want_v=True)
    assert ei.value.code == gpu.E_STATE
    with pytest.raises(gpu.GpxError) as ei:
        dst.replicate([0])
    assert ei.value.code == gpu.E_STATE
    with pytest.raises(gpu.GpxError) as ei:
        dst.prepare_variance()
    assert ei.value.code == gpu.E_STATE
    src.close()
    dst.close()


@pytest.mark.parametrize("prec", [1, 0, 3])
def test_replicas_on_a_second_device(gpu, ds, prec):
    """The cross-device half of gpx_model_replicate (peer enable, hipMemcpyPeerAsync between ordinals, per-device
    kernel attributes, the pool's per-device matching) and of gpx_options.device: runs where the box has more than
    one GPU, skips on the one-GPU test box (where it has never run: README / gpx.h say so)."""
    if gpu.device_count() < 2:
        pytest.skip("one HIP device visible: the cross-device path cannot run here")
    n = 1500
    x, y, z, lab, s2 = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel("thinplate", 4.0)
    src = gpu.Model(kern, x, y, z, lab, s2, precision=prec, device=0)
    qx, qy, qz = ds.query_grid(9)
    ref = src.evaluate(qx, qy, qz, want_v=True, want_grad=True)
    last = gpu.device_count() - 1
    reps = src.replicate([1, last, 0])
    for r in reps:
        out = r.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        for key in ("f", "v", "grad"):
            np.testing.assert_array_equal(out[key], ref[key])
        r.close()
    # a model created directly on the other device
    m1 = gpu.Model(kern, x, y, z, lab, s2, precision=prec, device=1)
    out = m1.evaluate(qx, qy, qz, want_v=True)
    np.testing.assert_array_equal(out["f"], ref["f"])
    np.testing.assert_array_equal(out["v"], ref["v"])
    m1.close()
    src.close()
