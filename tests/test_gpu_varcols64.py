"""GPU suite: the variance of small fp64 models as one kernel (csrc/gpx_varcols64.hip) -- against the oracle and against its
twin, the general fp64 path (operand tile -> one-wave contraction tiles -> finish; GPX_VAR_COLS64=0), in one process."""
import os

import numpy as np
import pytest

from conftest import nerr, verr_v

pytestmark = pytest.mark.gpu

KERNELS = (("gaussian", (1.0, 1.0)), ("laplace", (1.0, 1.0)), ("matern32", (1.0, 1.0)), ("matern52", (1.0, 1.0)),
           ("thinplate", (4.0,)), ("thinplate", (2.0,)))


def _eval(m, q, cols64, **kw):
    """evaluate() with the kernel on or off (GPX_VAR_COLS64; the library re-reads its switches on gpx_debug_reload)"""
    import importlib
    gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
    with gpx.switches(GPX_VAR_COLS64=None if cols64 is True else ("1" if cols64 == "one-wave" else "0")):
        return m.evaluate(*q, want_v=True, **kw)


def _same_as_one_wave_form(m, q, a):
    """Round 6: the default is the two-waves-per-SIMD shape (16 queries and 22 row-fragment slots per wave); the one-wave shape of
    round 5 (32 queries, the same 22 slots: GPX_VAR_COLS64=1) sums everything in the same order -- equal bit for bit."""
    w = _eval(m, q, "one-wave")
    assert np.array_equal(w["v"], a["v"]) and np.array_equal(w["f"], a["f"])


@pytest.mark.parametrize("n", [16, 17, 33, 166, 277, 352, 353, 448, 512])
def test_small_fp64_variance_matches_the_oracle_and_the_general_path(gpu, orc, ds, n):
    """One row fragment, the fragment edges, one and two passes over the row fragments (352 / 353 rows), a larger size;
    six kernels incl. the indefinite ThinPlate(2.0) (negative 1/D); a query count that is no multiple of 16."""
    data = ds.fibonacci_training_set(n)
    q = ds.query_grid(11, scale=1.3)  # 1331 queries: the last wave holds 3
    for kn, par in KERNELS:
        om = orc.Model(orc.make_kernel(kn, *par), *data)
        ref = om.evaluate(*q, want_v=True)
        m = gpu.Model(gpu.make_kernel(kn, *par), *data, precision=gpu.F64, prepare_variance=True)
        a, b = _eval(m, q, True), _eval(m, q, False)
        assert verr_v(a["v"], ref["v"]) < 1e-10, (n, kn, par)
        assert verr_v(a["v"], b["v"]) < 1e-12, (n, kn, par)
        # the mean: carried by the variance kernel (a), from the mean kernel (b; and d: with a gradient always from the mean kernel)
        assert nerr(a["f"], ref["f"]) < 1e-10 and nerr(a["f"], b["f"]) < 1e-12, (n, kn, par)
        _same_as_one_wave_form(m, q, a)
        d = _eval(m, q, True, want_grad=True)
        assert nerr(d["f"], b["f"]) < 1e-13 and np.array_equal(d["v"], a["v"])
        assert nerr(d["grad"], om.evaluate(*q, want_grad=True)["grad"]) < 1e-9
        m.close()


def test_small_fp64_variance_is_taken_by_promoted_models_and_after_update(gpu, orc, ds):
    """A model asked for in fp32 but kept in fp64 by the promotion rule (indefinite thin plate) and a model grown by update()
    go through the same kernel."""
    data = ds.fibonacci_training_set(300)
    q = ds.query_grid(9, scale=1.2)
    om = orc.Model(orc.make_kernel("thinplate", 2.0), *data)
    ref = om.evaluate(*q, want_v=True)
    m = gpu.Model(gpu.make_kernel("thinplate", 2.0), *data, precision=gpu.F32, prepare_variance=True)
    a, b = _eval(m, q, True), _eval(m, q, False)
    assert m.stats["n_negative_pivots"] > 0  # (the promotion rule's condition: the state is fp64)
    assert verr_v(a["v"], ref["v"]) < 1e-10 and verr_v(a["v"], b["v"]) < 1e-12
    m.close()
    rng = np.random.default_rng(7)  # a handful of points: one row fragment, almost all of it padding
    for n in (1, 5):
        P, lab = rng.normal(size=(n, 3)), rng.normal(size=n)
        cols = (P[:, 0].copy(), P[:, 1].copy(), P[:, 2].copy(), lab, np.full(n, 0.05))
        qs = tuple(rng.normal(size=7) for _ in range(3))
        ref = orc.Model(orc.make_kernel("gaussian", 1.0, 1.0), *cols).evaluate(*qs, want_v=True)
        m = gpu.Model(gpu.make_kernel("gaussian", 1.0, 1.0), *cols, precision=gpu.F64, prepare_variance=True)
        assert verr_v(_eval(m, qs, True)["v"], ref["v"]) < 1e-10 and verr_v(_eval(m, qs, False)["v"], ref["v"]) < 1e-10
        m.close()
    full = ds.fibonacci_training_set(520)
    head = tuple(np.ascontiguousarray(c[:500]) for c in full)
    tail = tuple(np.ascontiguousarray(c[500:]) for c in full)
    om = orc.Model(orc.make_kernel("matern52", 1.0, 1.0), *full)
    ref = om.evaluate(*q, want_v=True)
    m = gpu.Model(gpu.make_kernel("matern52", 1.0, 1.0), *head, precision=gpu.F64, prepare_variance=True)
    v500 = _eval(m, q, True)["v"]
    assert verr_v(v500, _eval(m, q, False)["v"]) < 1e-12
    m.update(*tail)  # 520 points: one more pass over the row fragments
    assert verr_v(_eval(m, q, True)["v"], ref["v"]) < 1e-10
    m.close()


@pytest.mark.parametrize("n", [600, 724, 992])
def test_small_fp64_variance_kernel_up_to_its_routing_limit(gpu, orc, ds, n):
    """Two and three passes over the row fragments, up to the 992 points the kernel is routed (above, the general path is faster)."""
    data = ds.fibonacci_training_set(n)
    q = ds.query_grid(8, scale=1.3)
    for kn, par in (("matern32", (1.0, 1.0)), ("thinplate", (4.0,))):
        ref = orc.Model(orc.make_kernel(kn, *par), *data).evaluate(*q, want_v=True)
        m = gpu.Model(gpu.make_kernel(kn, *par), *data, precision=gpu.F64, prepare_variance=True)
        a, b = _eval(m, q, True), _eval(m, q, False)
        assert verr_v(a["v"], ref["v"]) < 1e-10 and verr_v(a["v"], b["v"]) < 1e-12, (n, kn)
        _same_as_one_wave_form(m, q, a)
        m.close()


def test_small_fp64_variance_on_random_clouds_kernels_and_query_counts(gpu, orc):
    """Forty seeded cases: anisotropic, uncentred clouds of 16 .. 992 points, every kernel with hyper-parameters away from 1,
    query counts from 1 to a few thousand (partial waves, partial workgroups): the kernel against the general fp64 path (v: 1e-11)
    and -- for the smaller models -- the oracle."""
    rng = np.random.default_rng(20151106)
    kinds = ("gaussian", "laplace", "matern32", "matern52", "thinplate")
    for case in range(40):
        n = int(rng.integers(16, 993))
        P = rng.normal(size=(n, 3)) * rng.uniform(0.2, 1.5, size=3) + rng.uniform(-2.0, 2.0, size=3)
        lab = np.where(rng.uniform(size=n) < 0.1, 1.0, 0.0) + 0.01 * rng.normal(size=n)
        s2 = np.full(n, float(rng.uniform(1e-3, 5e-2)))
        kn = kinds[case % len(kinds)]
        span = float(np.max(np.linalg.norm(P - P.mean(axis=0), axis=1)))
        par = (2.5 * span,) if kn == "thinplate" else (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.3, 1.5)))
        nq = int(rng.choice([1, 7, 31, 33, 127, 129, 1000, 4099]))
        q = tuple(P[rng.integers(0, n, size=nq), k] + 0.3 * rng.normal(size=nq) for k in range(3))
        cols = tuple(np.ascontiguousarray(P[:, k]) for k in range(3)) + (lab, s2)
        m = gpu.Model(gpu.make_kernel(kn, *par), *cols, precision=gpu.F64, prepare_variance=True)
        a, b = _eval(m, q, True), _eval(m, q, False)
        # (the mean of a thin-plate model is an alternating sum of terms ~R^3: two summation orders differ by 1e-16 of THAT)
        assert verr_v(a["v"], b["v"]) < 1e-11 and nerr(a["f"], b["f"]) < 1e-9, (case, n, kn, par, nq)
        _same_as_one_wave_form(m, q, a)
        if n <= 300:
            ref = orc.Model(orc.make_kernel(kn, *par), *cols).evaluate(*q, want_v=True)
            assert verr_v(a["v"], ref["v"]) < 1e-9 and nerr(a["f"], ref["f"]) < 1e-9, (case, n, kn, par, nq)
        m.close()
