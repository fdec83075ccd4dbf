"""GPU suite: small models of the opt-in split-fp16 mode (GPX_PREC_F32_SPLIT) on the fp16 matrix cores (csrc/gpx_varcols16.hip)
-- against the fp64 pipeline, the oracle and its twin, the fp32 small-model kernel (GPX_VAR_COLS16=0)."""
import os

import numpy as np
import pytest

from conftest import nerr, verr_v

pytestmark = pytest.mark.gpu


def _eval(m, q, cols16, **kw):
    import importlib
    gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
    with gpx.switches(GPX_VAR_COLS16=None if cols16 else "0"):
        return m.evaluate(*q, want_v=True, **kw)


@pytest.mark.parametrize("n", [16, 33, 166, 277, 352, 353, 512, 704, 705, 1024])
def test_split_fp16_small_model_variance_matches_fp64_and_its_fp32_twin(gpu, orc, ds, n):
    """One chunk, fragment and chunk edges, one / two / three passes over the row fragments (352 | 353, 704 | 705 rows), the
    largest size; the four exponential kernels; a query count that fills neither the last wave nor the last workgroup.
    The north star's tolerance for an fp32 mode: 1e-5 of max|v|; measured 5e-7 .. 2.5e-6 (scripts/var16_check.py)."""
    data = ds.fibonacci_training_set(n)
    q = ds.query_grid(11, scale=1.3)
    for kn, par in (("gaussian", (1.0, 1.0)), ("laplace", (1.0, 1.0)), ("matern32", (1.0, 1.0)), ("matern52", (1.0, 1.0))):
        m64 = gpu.Model(gpu.make_kernel(kn, *par), *data, precision=gpu.F64, prepare_variance=True)
        ref = m64.evaluate(*q, want_v=True)
        m64.close()
        m = gpu.Model(gpu.make_kernel(kn, *par), *data, precision=gpu.F32_SPLIT, prepare_variance=True)
        a, b = _eval(m, q, True), _eval(m, q, False)
        assert verr_v(a["v"], ref["v"]) < 1e-5 and verr_v(b["v"], ref["v"]) < 1e-5, (n, kn)
        assert not np.array_equal(a["v"], b["v"])  # (each took its own kernel)
        assert np.array_equal(a["f"], b["f"]) and nerr(a["f"], ref["f"]) < 1e-5
        # the ORACLE itself (not the repo's own fp64 pipeline): every kernel up to 300 points, one exponential kernel each at 512
        # and at the largest size (the oracle's N = 1024 create + 1331 per-query solves take seconds)
        if n <= 300 or (n, kn) in ((512, "gaussian"), (512, "laplace"), (1024, "matern52"), (1024, "matern32")):
            o = orc.Model(orc.make_kernel(kn, *par), *data).evaluate(*q, want_v=True)
            assert verr_v(a["v"], o["v"]) < 1e-5 and nerr(a["f"], o["f"]) < 1e-5, (n, kn)
        m.close()


def test_split_fp16_small_models_thin_plate_and_growth(gpu, orc, ds):
    """The thin plate forms its operand in fp64: those models keep the fp32 kernel (the switch changes nothing); a model grown by
    update() past 1024 points packs its inverse factor and moves to the 128 x 128 split contraction."""
    data = ds.fibonacci_training_set(300)
    q = ds.query_grid(9, scale=1.2)
    m = gpu.Model(gpu.make_kernel("thinplate", 4.0), *data, precision=gpu.F32_SPLIT, prepare_variance=True)
    a, b = _eval(m, q, True), _eval(m, q, False)
    assert np.array_equal(a["v"], b["v"])
    m.close()
    full = ds.fibonacci_training_set(1100)
    head = tuple(np.ascontiguousarray(c[:1000]) for c in full)
    tail = tuple(np.ascontiguousarray(c[1000:]) for c in full)
    m64 = gpu.Model(gpu.make_kernel("matern52", 1.0, 1.0), *full, precision=gpu.F64, prepare_variance=True)
    ref = m64.evaluate(*q, want_v=True)
    m64.close()
    m = gpu.Model(gpu.make_kernel("matern52", 1.0, 1.0), *head, precision=gpu.F32_SPLIT, prepare_variance=True)
    _eval(m, q, True)
    m.update(*tail)
    assert verr_v(_eval(m, q, True)["v"], ref["v"]) < 1e-5
    m.close()


def test_split_fp16_small_models_on_random_clouds_kernels_and_query_counts(gpu):
    """Thirty seeded cases: anisotropic, uncentred clouds of 16 .. 1024 points, the exponential kernels with hyper-parameters away
    from 1, query counts from 1 to a few thousand (partial waves and workgroups): 1e-5 of max|v| against the fp64 pipeline, for the
    fp16 matrix-core kernel and for its fp32 twin."""
    rng = np.random.default_rng(20151107)
    kinds = ("gaussian", "laplace", "matern32", "matern52")
    worst = 0.0
    for case in range(30):
        n = int(rng.integers(16, 1025))
        P = rng.normal(size=(n, 3)) * rng.uniform(0.3, 1.5, size=3) + rng.uniform(-2.0, 2.0, size=3)
        lab = np.where(rng.uniform(size=n) < 0.1, 1.0, 0.0) + 0.01 * rng.normal(size=n)
        s2 = np.full(n, float(rng.uniform(1e-3, 5e-2)))
        kn = kinds[case % len(kinds)]
        par = (float(rng.uniform(0.5, 2.0)), float(rng.uniform(0.3, 1.5)))
        nq = int(rng.choice([1, 7, 31, 33, 127, 129, 1000, 4099]))
        q = tuple(P[rng.integers(0, n, size=nq), k] + 0.3 * rng.normal(size=nq) for k in range(3))
        cols = tuple(np.ascontiguousarray(P[:, k]) for k in range(3)) + (lab, s2)
        m64 = gpu.Model(gpu.make_kernel(kn, *par), *cols, precision=gpu.F64, prepare_variance=True)
        ref = m64.evaluate(*q, want_v=True)
        m64.close()
        m = gpu.Model(gpu.make_kernel(kn, *par), *cols, precision=gpu.F32_SPLIT, prepare_variance=True)
        a, b = _eval(m, q, True), _eval(m, q, False)
        ea, eb = verr_v(a["v"], ref["v"]), verr_v(b["v"], ref["v"])
        worst = max(worst, ea)
        assert ea < 1e-5 and eb < 1e-5, (case, n, kn, par, nq, ea, eb)
        m.close()
    print("worst split-fp16 error over the random cases: %.2e" % worst)


@pytest.mark.parametrize("amp", [1e-6, 1e5])
def test_split_fp16_small_models_scale_the_kernel_amplitude(gpu, ds, amp):
    """small split-mode models keep their inverse factor unpacked, and the power-of-two scale sk of the kernel values
    used to be computed only on the packing path -- with sk = 1 an amplitude k(0) >= 65504 overflowed the fp16 halves (NaN
    variances) and one around 1e-6 fell into their subnormals.  Every F32_SPLIT model now carries sk = 2^-e, k(0) * sk in
    [0.5, 1): amplitudes of 1e-6 and 1e5 (noise in proportion) hold the mode's 1e-5 of max|v| against the fp64 pipeline, for the
    model itself, a replica and a shell committed from its state blobs."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(400)
    data = (x, y, z, lab * np.sqrt(amp), s2 * amp)
    q = ds.query_grid(9, scale=1.2)
    for kn, par in (("gaussian", (np.sqrt(amp), 1.0)), ("matern52", (np.sqrt(amp), 0.8)), ("laplace", (amp / 2.0, 1.0))):
        kern = gpu.make_kernel(kn, *par)
        m64 = gpu.Model(kern, *data, precision=gpu.F64, prepare_variance=True)
        ref = m64.evaluate(*q, want_v=True)
        m64.close()
        assert 0.2 * amp < ref["v"].max() < 5.0 * amp
        m = gpu.Model(kern, *data, precision=gpu.F32_SPLIT, prepare_variance=True)
        a = _eval(m, q, True)
        assert np.all(np.isfinite(a["v"])) and verr_v(a["v"], ref["v"]) < 1e-5, (kn, amp, verr_v(a["v"], ref["v"]))
        r = m.replicate([0])[0]
        assert np.array_equal(_eval(r, q, True)["v"], a["v"])
        r.close()
        m.close()


@pytest.mark.parametrize("n,kn", [(512, "matern32"), (1024, "gaussian")])
def test_split_fp16_small_model_kernel_against_the_oracle_at_its_largest_sizes(gpu, orc, ds, n, kn):
    """above 300 points the fp16 matrix-core kernel was only ever held to the repo's own fp64 pipeline.  Two
    and three passes over the row fragments against the CPU oracle (f, v; 1e-5 of max|v|)."""
    data = ds.fibonacci_training_set(n)
    q = ds.query_grid(8, scale=1.25)
    o = orc.Model(orc.make_kernel(kn, 1.0, 1.0), *data).evaluate(*q, want_v=True)
    m = gpu.Model(gpu.make_kernel(kn, 1.0, 1.0), *data, precision=gpu.F32_SPLIT, prepare_variance=True)
    a = _eval(m, q, True)
    assert m.stats["n"] == n
    assert verr_v(a["v"], o["v"]) < 1e-5 and nerr(a["f"], o["f"]) < 1e-5
    m.close()
