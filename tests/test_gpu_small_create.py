"""GPU suite: GPRegressor::create for the reference's own model sizes in three launches (csrc/gpx_small.hip) -- against the
oracle, against its twin (the general launch chain, GPX_DATAFLOW=0) and through its give-up path."""
import os

import numpy as np
import pytest

from conftest import nerr, verr, verr_v

pytestmark = pytest.mark.gpu


def _twin(gpu, kern, data, prec, small, **kw):
    with gpu.switches(GPX_DATAFLOW=None if small else "0"):
        return gpu.Model(kern, *data, precision=prec, **kw)


@pytest.mark.parametrize("n", [17, 63, 64, 65, 129, 277, 320, 321, 513, 724, 1024])
def test_three_launch_create_matches_the_chain_and_the_oracle(gpu, orc, ds, n):
    """Every tile-count edge of the 64 x 64 dataflow (one partial tile, exactly one tile, the 128 / 256 padding edges, the
    largest size), six kernels incl. the indefinite ThinPlate(2.0), fp64 and fp32 mode: alpha, D, pivots, R, f, v, grad."""
    data = ds.fibonacci_training_set(n)
    qx, qy, qz = ds.query_grid(6, scale=1.3)
    for kn, par in (("gaussian", (1.0, 1.0)), ("laplace", (1.0, 1.0)), ("matern32", (1.0, 1.0)), ("matern52", (1.0, 1.0)),
                    ("thinplate", (4.0,)), ("thinplate", (2.0,))):
        om = orc.Model(orc.make_kernel(kn, *par), *data)
        ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        oD = np.diag(om.ldlt()[0]).copy()  # Eigen's vectorD (uniform sigma2: no transpositions)
        for prec, tol, tol_tw in ((gpu.F64, 1e-10, 1e-10), (gpu.F32, 1e-5, 2e-7)):
            ms = _twin(gpu, gpu.make_kernel(kn, *par), data, prec, True, prepare_variance=True)
            mc = _twin(gpu, gpu.make_kernel(kn, *par), data, prec, False, prepare_variance=True)
            ss, sc = ms.stats, mc.stats
            assert ss["solve_fallbacks"] == 0 and ss["t_kbuild_ms"] == 0.0 and sc["t_kbuild_ms"] > 0.0  # each took its own path
            assert ss["n_negative_pivots"] == sc["n_negative_pivots"] == int(np.sum(oD < 0))
            assert abs(ms.R - om.R) <= 1e-14 * om.R and abs(ms.R - mc.R) <= 1e-14 * om.R  # (float arg-max of the chain: a tie may pick another pair)
            assert nerr(ms.alpha, om.alpha) < 1e-9 and nerr(ms.alpha, mc.alpha) < 1e-10
            assert nerr(ms.D, oD) < 1e-10 and nerr(ms.D, mc.D) < 1e-11
            assert np.array_equal(ms.perm, mc.perm)
            a, b = (m.evaluate(qx, qy, qz, want_v=True, want_grad=True) for m in (ms, mc))
            for key in ("f", "grad"):
                assert nerr(a[key], ref[key]) < 1e-9 and nerr(a[key], b[key]) < 1e-10
            assert verr_v(a["v"], ref["v"]) < tol, (n, kn, par, prec)
            assert verr_v(a["v"], b["v"]) < tol_tw, (n, kn, par, prec)  # same state up to the rounding of the factorisation order
            ms.close(), mc.close()


def test_three_launch_create_with_normals_and_noise_order(gpu, orc, ds, golden):
    """create<true> (gp_regressor.hpp:166-181) and Eigen's pivot order for a non-uniform sigma2 on the node's own cloud."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    s2 = s2 * (1.0 + 0.5 * np.cos(np.arange(len(s2))))
    kern = ("matern52", (1.0, 1.0))
    om = orc.Model(orc.make_kernel(kern[0], *kern[1]), x, y, z, lab, s2, with_normals=True)
    for prec in (gpu.F64, gpu.F32, gpu.MIXED, gpu.F32_SPLIT):
        gm = gpu.Model(gpu.make_kernel(kern[0], *kern[1]), x, y, z, lab, s2, precision=prec, with_normals=True, prepare_variance=True)
        assert gm.stats["t_kbuild_ms"] == 0.0  # the three-launch path
        assert nerr(gm.alpha, om.alpha) < 1e-9 and nerr(gm.normals, om.normals) < 1e-9
        assert not np.array_equal(gm.perm, np.arange(len(x)))  # the order is Eigen's, not the caller's
        qx, qy, qz = ds.query_grid(5)
        out, ref = gm.evaluate(qx, qy, qz, want_v=True), om.evaluate(qx, qy, qz, want_v=True)
        assert nerr(out["f"], ref["f"]) < 1e-9 and verr_v(out["v"], ref["v"]) < (1e-10 if prec == gpu.F64 else 1e-5)
        gm.close()


def test_a_wait_that_gives_up_falls_back_to_the_chain(gpu, orc, ds):
    """Every wait of the dataflow launches has a time budget; with the budget forced to zero some workgroup gives up, the
    host sees the flag, and the create is redone by the general chain (whose one-launch substitution gives up as well and is
    redone by the step launches): same model, gpx_stats.solve_fallbacks >= 1."""
    data = ds.fibonacci_training_set(300)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    om = orc.Model(orc.make_kernel("matern52", 1.0, 1.0), *data)
    import time
    with gpu.switches(GPX_WAIT_BUDGET_US="0"):
        for prec in (gpu.F64, gpu.F32):
            t0 = time.perf_counter()
            gm = gpu.Model(kern, *data, precision=prec, prepare_variance=True)
            assert time.perf_counter() - t0 < 1.0  # (a give-up costs milliseconds: the budget is clock time, not a poll count)
            st = gm.stats
            assert st["solve_fallbacks"] >= 1 and st["t_kbuild_ms"] > 0.0
            assert nerr(gm.alpha, om.alpha) < 1e-9
            qx, qy, qz = ds.query_grid(5)
            out, ref = gm.evaluate(qx, qy, qz, want_v=True), om.evaluate(qx, qy, qz, want_v=True)
            assert nerr(out["f"], ref["f"]) < 1e-9 and verr_v(out["v"], ref["v"]) < (1e-10 if prec == gpu.F64 else 1e-5)
            gm.close()
    gm = gpu.Model(kern, *data, precision=gpu.F64)
    assert gm.stats["solve_fallbacks"] == 0
    gm.close()


def test_three_launch_create_then_update(gpu, orc, ds):
    """What follows a create keeps working on a model built by the three launches: the rank-n append of update() on its factor
    and inverse factor (the state blobs of such a model: test_shell_broadcast_commit_roundtrip, N = 300)."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(420)
    kern = gpu.make_kernel("gaussian", 1.0, 1.0)
    k = 300
    gm = gpu.Model(kern, x[:k], y[:k], z[:k], lab[:k], s2[:k], precision=gpu.F64, prepare_variance=True)
    assert gm.stats["t_kbuild_ms"] == 0.0
    gm.update(x[k:], y[k:], z[k:], lab[k:], s2[k:])
    om = orc.Model(orc.make_kernel("gaussian", 1.0, 1.0), x, y, z, lab, s2)
    qx, qy, qz = ds.query_grid(5)
    out, ref = gm.evaluate(qx, qy, qz, want_v=True), om.evaluate(qx, qy, qz, want_v=True)
    assert nerr(gm.alpha, om.alpha) < 1e-9 and nerr(out["f"], ref["f"]) < 1e-9 and verr_v(out["v"], ref["v"]) < 1e-9
    gm.close()


def _mid_twin(gpu, kern, data, prec, dataflow, tiles=None, **kw):
    """dataflow False: the launch chain (GPX_DATAFLOW=0); True: the dataflow factorisation, tiles = 64 | 128 forces that form"""
    with gpu.switches(GPX_DATAFLOW=(str(tiles) if tiles else None) if dataflow else "0"):
        return gpu.Model(kern, *data, precision=prec, **kw)


@pytest.mark.parametrize("n", [1025, 1100, 2305, 4096])
def test_dataflow_factorisation_of_mid_size_models_matches_the_chain_and_the_oracle(gpu, orc, ds, n, monkeypatch):
    """Above the small-model path the kernel matrix and the LDL^T are one dataflow launch (csrc/gpx_dataflow.hpp) where the
    blocked launch chain (GPX_DATAFLOW=0, its twin) is bound by its chain, not its flops.  Sizes: the first padded size above
    1024, tile rows that end inside a 128-block, BASELINE's C2 size; fp64, and the fp32 factorisation at every size
    (GPX_TRAIN_F64_MAX=0: chunked fp32 sums); the indefinite ThinPlate(2.0) keeps its inertia."""
    monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")  # (_mid_twin reloads the switches)
    gpu.debug_reload()
    data = ds.fibonacci_training_set(n)
    qx, qy, qz = ds.query_grid(5, scale=1.2)
    cases = [("matern52", (1.0, 1.0))] + ([("gaussian", (1.0, 1.0)), ("thinplate", (2.0,))] if n <= 2305 else [])
    for kn, par in cases:
        om = orc.Model(orc.make_kernel(kn, *par), *data) if n <= 2305 else None
        for prec in (gpu.F64, gpu.F32):
            if kn == "thinplate" and prec == gpu.F32:
                continue  # (an fp32 LDL^T of the thin plate is outside every precision rule of the library)
            md = _mid_twin(gpu, gpu.make_kernel(kn, *par), data, prec, True, prepare_variance=True)
            mc = _mid_twin(gpu, gpu.make_kernel(kn, *par), data, prec, False, prepare_variance=True)
            sd, sc = md.stats, mc.stats
            assert sd["solve_fallbacks"] == 0 and sd["factor_gemm_launches"] == 0 and sc["factor_gemm_launches"] > 0  # each its own path
            assert sd["n_negative_pivots"] == sc["n_negative_pivots"]
            assert abs(md.R - mc.R) <= 1e-14 * mc.R
            tol = 1e-10 if prec == gpu.F64 else 1e-5
            assert nerr(md.alpha, mc.alpha) < (1e-9 if prec == gpu.F64 else 1e-5), (n, kn, prec)
            assert nerr(md.D, mc.D) < (1e-10 if prec == gpu.F64 else 2e-4), (n, kn, prec)
            a, b = (m.evaluate(qx, qy, qz, want_v=True) for m in (md, mc))
            assert nerr(a["f"], b["f"]) < (1e-9 if prec == gpu.F64 else 1e-5)
            assert verr_v(a["v"], b["v"]) < tol, (n, kn, prec)
            if om is not None:
                ref = om.evaluate(qx, qy, qz, want_v=True)
                assert nerr(md.alpha, om.alpha) < (1e-9 if prec == gpu.F64 else 1e-5)
                assert nerr(a["f"], ref["f"]) < (1e-9 if prec == gpu.F64 else 1e-5) and verr_v(a["v"], ref["v"]) < tol
            md.close(), mc.close()


def test_dataflow_factorisation_that_gives_up_is_redone_by_the_chain(gpu, orc, ds):
    """Mid-size form of the give-up path: the factor of the dataflow launch is void (info[6]), the whole create is redone by
    the launch chain -- same model, gpx_stats.solve_fallbacks = 1."""
    data = ds.fibonacci_training_set(1500)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    om = orc.Model(orc.make_kernel("matern52", 1.0, 1.0), *data)
    with gpu.switches(GPX_WAIT_BUDGET_US="0"):
        gm = gpu.Model(kern, *data, precision=gpu.F64, prepare_variance=True)
    st = gm.stats
    assert st["solve_fallbacks"] >= 1 and st["factor_gemm_launches"] > 0
    assert nerr(gm.alpha, om.alpha) < 1e-9
    qx, qy, qz = ds.query_grid(5)
    out, ref = gm.evaluate(qx, qy, qz, want_v=True), om.evaluate(qx, qy, qz, want_v=True)
    assert nerr(out["f"], ref["f"]) < 1e-9 and verr_v(out["v"], ref["v"]) < 1e-10
    gm.close()


def test_dataflow_launches_are_deterministic_and_survive_concurrent_creates(gpu, ds):
    """The dataflow launches hand tiles from workgroup to workgroup through memory behind flags; every sum has a fixed order,
    so repeated creates must agree BIT FOR BIT (a consumer that ever read a tile before -- or a stale copy after -- its flag
    would show up here), also when four host threads create small, mid-size and large models at the same time on one device
    (their grids compete for the CUs; a wait that gave up would only cost a fallback, never a different model)."""
    import hashlib
    import threading
    cases = [(277, gpu.F64, "gaussian"), (724, gpu.F32, "matern52"), (1500, gpu.F64, "matern52"), (3000, gpu.F32, "gaussian"),
             (5000, gpu.F64, "matern32"), (9000, gpu.F32, "matern52")]  # (the last one: 128 x 128 tiles)
    sets = {n: ds.fibonacci_training_set(n) for n, _, _ in cases}

    def digest(n, prec, kn):
        m = gpu.Model(gpu.make_kernel(kn, 1.0, 1.0), *sets[n], precision=prec, prepare_variance=True)
        out = m.evaluate(*ds.query_grid(4), want_v=True)
        h = hashlib.sha1(m.alpha.tobytes() + m.D.tobytes() + out["f"].tobytes() + out["v"].tobytes()).hexdigest()
        fb = m.stats["solve_fallbacks"]
        m.close()
        return h, fb

    ref = {c: digest(*c) for c in cases}
    assert all(fb == 0 for _, fb in ref.values())
    for c in cases:
        for _ in range(4):
            assert digest(*c)[0] == ref[c][0], c
    errors = []

    def worker(k):
        try:
            for rep in range(6):
                c = cases[(k + rep) % len(cases)]
                h, _ = digest(*c)
                if h != ref[c][0]:
                    errors.append((k, rep, c))
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("n", [1100, 2305])
def test_wide_tile_dataflow_factorisation_matches_the_chain_and_the_oracle(gpu, orc, ds, n, monkeypatch):
    """From 8192 padded rows on the dataflow factorisation runs on 128 x 128 tiles (csrc/gpx_dataflow_wide.hpp: eight waves per
    tile, the launch chain's diagonal-block routine on the tile's sums, the panel solve as a slice loop against the inverse
    block).  Forced here at test sizes (GPX_DATAFLOW=128): a last tile row that is mostly padding and one that is full, fp64
    and the chunked fp32 sums, the indefinite ThinPlate(2.0)'s inertia; the product sizes run it in test_gpu_scale.py."""
    monkeypatch.setenv("GPX_TRAIN_F64_MAX", "0")  # (_mid_twin reloads the switches)
    gpu.debug_reload()
    data = ds.fibonacci_training_set(n)
    qx, qy, qz = ds.query_grid(5, scale=1.2)
    for kn, par in (("matern52", (1.0, 1.0)), ("gaussian", (1.0, 1.0)), ("thinplate", (2.0,))):
        om = orc.Model(orc.make_kernel(kn, *par), *data)
        ref = om.evaluate(qx, qy, qz, want_v=True)
        for prec in (gpu.F64, gpu.F32):
            if kn == "thinplate" and prec == gpu.F32:
                continue
            md = _mid_twin(gpu, gpu.make_kernel(kn, *par), data, prec, True, tiles=128, prepare_variance=True)
            mc = _mid_twin(gpu, gpu.make_kernel(kn, *par), data, prec, False, prepare_variance=True)
            sd, sc = md.stats, mc.stats
            assert sd["solve_fallbacks"] == 0 and sd["factor_gemm_launches"] == 0 and sc["factor_gemm_launches"] > 0
            assert sd["n_negative_pivots"] == sc["n_negative_pivots"] and abs(md.R - mc.R) <= 1e-14 * mc.R
            t64 = prec == gpu.F64
            assert nerr(md.alpha, om.alpha) < (1e-9 if t64 else 1e-5) and nerr(md.D, mc.D) < (1e-10 if t64 else 2e-4)
            a, b = (m.evaluate(qx, qy, qz, want_v=True) for m in (md, mc))
            assert nerr(a["f"], ref["f"]) < (1e-9 if t64 else 1e-5) and nerr(a["f"], b["f"]) < (1e-9 if t64 else 1e-5)
            assert verr_v(a["v"], ref["v"]) < (1e-10 if t64 else 1e-5) and verr_v(a["v"], b["v"]) < (1e-10 if t64 else 1e-5)
            md.close(), mc.close()


@pytest.mark.parametrize("n", [1, 2, 5, 31, 33])
def test_three_launch_create_of_a_handful_of_points(gpu, orc, n):
    """The smallest models: one training point, fewer points than a 32 x 32 sub-block, one row into the second sub-block --
    almost the whole 64 x 64 tile is identity padding."""
    rng = np.random.default_rng(100 + n)
    P = rng.normal(size=(n, 3))
    lab = rng.normal(size=n)
    s2 = np.full(n, 0.05)
    qx, qy, qz = (rng.normal(size=7) for _ in range(3))
    for kn, par in (("gaussian", (1.0, 1.0)), ("thinplate", (6.0,))):
        om = orc.Model(orc.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2)
        ref = om.evaluate(qx, qy, qz, want_v=True, want_grad=True)
        for prec in (gpu.F64, gpu.F32):
            gm = gpu.Model(gpu.make_kernel(kn, *par), P[:, 0], P[:, 1], P[:, 2], lab, s2, precision=prec, prepare_variance=True)
            st = gm.stats
            assert st["solve_fallbacks"] == 0 and st["t_kbuild_ms"] == 0.0 and gm.n == n
            if n > 1:
                assert abs(gm.R - om.R) <= 1e-14 * om.R
            assert nerr(gm.alpha, om.alpha) < 1e-10
            out = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True)
            assert nerr(out["f"], ref["f"]) < 1e-10 and nerr(out["grad"], ref["grad"]) < 1e-10
            # (random points up to 6 apart under ThinPlate(6): k(0) = 216 against variances of 1-3 -- the fp32 mode is held to the
            # k(0)-scaled metric there, as in test_gpu_parity.py; fp64 to the strict one)
            k0 = 216.0 if kn == "thinplate" else 1.0
            assert (verr_v(out["v"], ref["v"]) < 1e-10) if prec == gpu.F64 else (verr(out["v"], ref["v"], k0) < 1e-5)
            gm.close()


@pytest.mark.parametrize("tiles", [128, 64])
def test_dataflow_factorisation_above_16384_rows_matches_the_chain(gpu, ds, tiles, monkeypatch):
    """by default the dataflow factorisation stops at 16384 padded rows, but its flags, per-tile results and
    tile enumeration scale; forced (GPX_DATAFLOW=128 | 64) at N = 20480 in fp32 -- 160 tile rows of 128, 320 of 64: 12880 /
    51360 workgroups -- it must agree with the launch chain (pivot signs, D to the fp32 factorisation's rounding, alpha after
    the fp64 refinement, mean and variance at a few queries)."""
    n = 20480 - 15
    data = ds.fibonacci_training_set(n)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)
    qx, qy, qz = ds.query_grid(4, scale=1.1)
    md = _mid_twin(gpu, kern, data, gpu.F32, True, tiles=tiles, prepare_variance=True)
    sd = md.stats
    assert sd["n_padded"] == 20480 and sd["solve_fallbacks"] == 0 and sd["factor_gemm_launches"] == 0
    a = md.evaluate(qx, qy, qz, want_v=True)
    aD, aal = md.D.copy(), md.alpha.copy()
    md.close()
    mc = _mid_twin(gpu, kern, data, gpu.F32, False, prepare_variance=True)
    sc = mc.stats
    assert sc["factor_gemm_launches"] > 0 and sc["n_negative_pivots"] == sd["n_negative_pivots"] == 0
    b = mc.evaluate(qx, qy, qz, want_v=True)
    assert nerr(aD, mc.D) < 2e-4 and nerr(aal, mc.alpha) < 1e-5
    assert nerr(a["f"], b["f"]) < 1e-5 and verr_v(a["v"], b["v"]) < 1e-5
    mc.close()
    gpu.trim()
