"""GPU parity of the first slice of the reference's second library, gp::GaussianProcess (include/gp/GaussianProcess.h:
a GP trained on values and gradients), through the C ABI (gpx_dgp_*) against the oracle restatement.  fp64: 1e-10
norm-wise for alpha, the mean, its gradient and the variance (the systems are 4n x 4n with cond up to ~1e7 for the thin
plate: 1e-8 there)."""
import ctypes as C

import numpy as np
import pytest

from conftest import nerr

pytestmark = pytest.mark.gpu


def _cloud(n, seed):
    r = np.random.default_rng(seed)
    d = r.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    P = d * r.uniform(0.95, 1.05, size=(n, 1))
    return P, np.zeros(n) + 0.01 * r.normal(size=n), d  # points, targets (~0 on the surface), outward normals


@pytest.mark.parametrize("n", [1, 7, 33, 200, 300, 700])
@pytest.mark.parametrize("kern", [("se", 1.2, 0.8), ("thinplate", 4.0)])  # R above every query-to-point distance
def test_derivative_gp_against_oracle(gpu, orc, kern, n):
    P, t, nr = _cloud(n, 100 + n)
    noise = 0.05
    og = orc.DerivativeGP(kern, noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    assert og.info == 0
    gk = gpu.make_kernel("se", kern[1], kern[2]) if kern[0] == "se" else gpu.make_kernel("thinplate", kern[1])
    gg = gpu.DerivativeGP(gk, noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    tol = 1e-10 if kern[0] == "se" else 1e-8
    assert nerr(gg.alpha, og.alpha) < tol
    assert abs(gg.loglik - og.loglik) < 1e-7 * max(1.0, abs(og.loglik))
    st = gg.stats
    assert st["n"] == 4 * n and st["n_padded"] % 256 == 0 and st["n_negative_pivots"] == 0
    r = np.random.default_rng(n)
    for nq in (1, 130, 1000):
        Q = r.uniform(-1.3, 1.3, size=(nq, 3))
        m = min(nq, n, 3)
        Q[:m] = P[:m]  # on training points
        ref = og.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])
        out = gg.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])
        # norm-wise; a single query on the surface has f ~ 1e-4, so the scale is at least the size of the field (~0.1 .. 1)
        assert np.max(np.abs(out["f"] - ref["f"])) / max(np.max(np.abs(ref["f"])), 0.1) < tol, nq
        assert np.max(np.abs(out["grad"] - ref["grad"])) / max(np.max(np.abs(ref["grad"])), 1e-300) < tol, nq
        k0 = kern[1] ** 2 if kern[0] == "se" else kern[1] ** 3
        assert np.max(np.abs(out["v"] - ref["v"])) / k0 < tol, nq
        assert out["v"].min() > -1e-9 * k0 and out["v"].max() <= k0 * (1 + 1e-12)
        mean_only = gg.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=False)
        np.testing.assert_array_equal(mean_only["f"], out["f"])
    gg.close()


def test_derivative_gp_without_normals_and_errors(gpu, orc):
    P, t, nr = _cloud(40, 9)
    t = np.where(np.arange(40) % 5 == 0, 1.0, 0.0)
    gk = gpu.make_kernel("se", 1.0, 0.6)
    og = orc.DerivativeGP(("se", 1.0, 0.6), 0.1, P[:, 0], P[:, 1], P[:, 2], t, None)
    gg = gpu.DerivativeGP(gk, 0.1, P[:, 0], P[:, 1], P[:, 2], t, None)
    assert nerr(gg.alpha, og.alpha) < 1e-10
    gg.close()
    # a matrix that is not positive definite (two identical points, no noise) is a status, as llt() has no answer for it
    P2 = np.concatenate([P[:5], P[:1]])
    with pytest.raises(gpu.GpxError) as ei:
        gpu.DerivativeGP(gk, 0.0, P2[:, 0], P2[:, 1], P2[:, 2], np.zeros(6), None)
    assert ei.value.code == gpu.E_SINGULAR
    with pytest.raises(gpu.GpxError) as ei:  # the first library's kernels have no derivative blocks here
        gpu.DerivativeGP(gpu.make_kernel("matern52", 1, 1), 0.1, P[:, 0], P[:, 1], P[:, 2], t, None)
    assert ei.value.code == gpu.E_BAD_ARG
    with pytest.raises(gpu.GpxError) as ei:
        gpu.DerivativeGP(gk, float("nan"), P[:, 0], P[:, 1], P[:, 2], t, None)
    assert ei.value.code == gpu.E_BAD_ARG
    lib = gpu.lib()
    assert lib.gpx_dgp_evaluate(None, 1, None, None, None, None, None) == gpu.E_NULL
    assert lib.gpx_last_error() == b"Empty Model pointer"


def test_derivative_gp_reconstructs_a_sphere_from_points_and_normals(gpu):
    """What the library is for (the reference's tests/test_gp.cpp shape): surface points with label 0 and their normals;
    the zero level of the mean is the surface, its gradient there the normal."""
    P, _, nr = _cloud(400, 1)
    P = nr.copy()  # exactly on the unit sphere
    gg = gpu.DerivativeGP(gpu.make_kernel("se", 1.0, 0.7), 0.01, P[:, 0], P[:, 1], P[:, 2], np.zeros(400), nr)
    r = np.random.default_rng(5)
    d = r.normal(size=(200, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    on = gg.evaluate(d[:, 0], d[:, 1], d[:, 2])
    assert np.abs(on["f"]).max() < 5e-3  # new points of the sphere lie on the zero level
    g = on["grad"] / np.linalg.norm(on["grad"], axis=1)[:, None]
    assert np.min(np.einsum("ij,ij->i", g, d)) > 0.999  # and the gradient there is the outward normal
    out = gg.evaluate(1.2 * d[:, 0], 1.2 * d[:, 1], 1.2 * d[:, 2], want_v=False)
    inn = gg.evaluate(0.8 * d[:, 0], 0.8 * d[:, 1], 0.8 * d[:, 2], want_v=False)
    assert out["f"].min() > 0.1 and inn["f"].max() < -0.1  # signed: positive outside, negative inside
    gg.close()


@pytest.mark.parametrize("append", ["1", "0"])
@pytest.mark.parametrize("kern", [("se", 1.2, 0.8), ("thinplate", 4.0)])
def test_derivative_gp_add_patterns_equals_create_on_the_union(gpu, orc, kern, append, monkeypatch):
    """gpx_dgp_add (add_patterns, GaussianProcess.h:340-374): after two appends -- one with normals, one without -- alpha,
    the log-likelihood and evaluate equal a model created on the concatenated samples and the oracle on the union to the
    usual tolerance: with the rows appended to the old factor (default; the first append carries 384 of 480 rows over, the
    second 896 of 920, and the model says so) and, bit for bit, with GPX_DGP_APPEND=0 (rebuild on the union); a failing
    append (non-finite sample) leaves the model as it was; a model with fewer than 128 rows is rebuilt."""
    monkeypatch.setenv("GPX_DGP_APPEND", append)
    gpu.debug_reload()
    P, t, nr = _cloud(300, 77)
    gk = gpu.make_kernel("se", kern[1], kern[2]) if kern[0] == "se" else gpu.make_kernel("thinplate", kern[1])
    a, b = 120, 230
    nr_u = nr.copy()
    nr_u[b:] = 0.0  # the last append passes no normals
    gg = gpu.DerivativeGP(gk, 0.05, P[:a, 0], P[:a, 1], P[:a, 2], t[:a], nr[:a])
    gg.add(P[a:b, 0], P[a:b, 1], P[a:b, 2], t[a:b], nr[a:b])
    assert gg.appended_from == (384 if append == "1" else 0)
    before = gg.alpha.copy()
    bad = P[b:, 0].copy()
    bad[3] = np.nan
    with pytest.raises(gpu.GpxError) as ei:
        gg.add(bad, P[b:, 1], P[b:, 2], t[b:], None)
    assert ei.value.code == gpu.E_NAN_INPUT
    assert gg.n == b
    np.testing.assert_array_equal(gg.alpha, before)
    gg.add(P[b:, 0], P[b:, 1], P[b:, 2], t[b:], None)
    assert gg.n == 300 and gg.stats["n"] == 1200
    assert gg.appended_from == (896 if append == "1" else 0)
    fresh = gpu.DerivativeGP(gk, 0.05, P[:, 0], P[:, 1], P[:, 2], t, nr_u)
    tol = 1e-10 if kern[0] == "se" else 1e-8
    Q = np.random.default_rng(5).uniform(-1.3, 1.3, size=(500, 3))
    o1, o2 = gg.evaluate(Q[:, 0], Q[:, 1], Q[:, 2]), fresh.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])
    k0 = kern[1] ** 2 if kern[0] == "se" else kern[1] ** 3
    if append == "0":
        np.testing.assert_array_equal(gg.alpha, fresh.alpha)
        assert gg.loglik == fresh.loglik
        for key in ("f", "grad", "v"):
            np.testing.assert_array_equal(o1[key], o2[key])
    else:
        assert nerr(gg.alpha, fresh.alpha) < tol
        assert abs(gg.loglik - fresh.loglik) < 1e-9 * abs(fresh.loglik)
        assert np.max(np.abs(o1["f"] - o2["f"])) / max(np.max(np.abs(o2["f"])), 0.1) < tol
        assert np.max(np.abs(o1["grad"] - o2["grad"])) / np.max(np.abs(o2["grad"])) < tol
        assert np.max(np.abs(o1["v"] - o2["v"])) / k0 < tol
    og = orc.DerivativeGP(kern, 0.05, P[:, 0], P[:, 1], P[:, 2], t, nr_u)
    assert nerr(gg.alpha, og.alpha) < tol
    ref = og.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])
    assert np.max(np.abs(o1["f"] - ref["f"])) / max(np.max(np.abs(ref["f"])), 0.1) < tol
    assert np.max(np.abs(o1["v"] - ref["v"])) / k0 < tol
    if kern[0] == "se":  # the likelihood gradient reads the appended row order too
        assert np.abs(gg.loglik_gradient() - og.loglik_gradient()).max() < 1e-8 * np.abs(og.loglik_gradient()).max()
    gg.close()
    fresh.close()
    small = gpu.DerivativeGP(gk, 0.05, P[:20, 0], P[:20, 1], P[:20, 2], t[:20], nr[:20])  # 80 rows: nothing to carry over
    small.add(P[20:50, 0], P[20:50, 1], P[20:50, 2], t[20:50], nr[20:50])
    assert small.appended_from == 0
    og2 = orc.DerivativeGP(kern, 0.05, P[:50, 0], P[:50, 1], P[:50, 2], t[:50], nr[:50])
    assert nerr(small.alpha, og2.alpha) < tol
    small.close()


@pytest.mark.parametrize("n", [5, 40, 150])
def test_likelihood_gradient_against_oracle(gpu, orc, n):
    """gpx_dgp_loglik_gradient (include/gp/GaussianProcess.h:387-410) against the oracle's NumPy restatement (itself held to
    central differences of the likelihood in tests/test_oracle.py), and directly against central differences of the device
    likelihood; models without hyper-parameter derivatives refuse."""
    P, t, nr = _cloud(n, 500 + n)
    sf, l, noise = 1.1, 0.7, 0.05
    og = orc.DerivativeGP(("se", sf, l), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    gg = gpu.DerivativeGP(gpu.make_kernel("se", sf, l), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    ref, got = og.loglik_gradient(), gg.loglik_gradient()
    assert np.abs(got - ref).max() < 1e-8 * max(1.0, np.abs(ref).max()), (got, ref)
    np.testing.assert_array_equal(got, gg.loglik_gradient())  # fixed summation order: repeatable
    eps = 1e-5
    fd = np.zeros(2)
    for j in range(2):
        lik = []
        for sgn in (+1, -1):
            p = np.array([np.log(l), np.log(sf)])
            p[j] += sgn * eps
            gp = gpu.DerivativeGP(gpu.make_kernel("se", float(np.exp(p[1])), float(np.exp(p[0]))), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
            lik.append(gp.loglik)
            gp.close()
        fd[j] = (lik[0] - lik[1]) / (2 * eps)
    assert np.abs(got - fd).max() < 1e-5 * max(1.0, np.abs(fd).max()), (got, fd)
    gg.evaluate(P[:3, 0], P[:3, 1], P[:3, 2])  # the model is still usable
    gg.close()
    tp = gpu.DerivativeGP(gpu.make_kernel("thinplate", 4.0), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    with pytest.raises(gpu.GpxError) as e:
        tp.loglik_gradient()
    assert e.value.code == gpu.E_BAD_ARG
    with pytest.raises(gpu.GpxError):
        tp.optimise()
    tp.close()


def test_rprop_optimiser_follows_the_reference_loop(gpu, orc):
    """gpx_dgp_optimise = Optimisation::find (include/gp/GaussianProcess.h:86-122) against the oracle's statement-by-statement
    restatement: same best parameters, likelihood and number of applied steps (the step sizes are powers of 0.5 / 1.2 times
    delta0 and only the SIGN of the gradient enters, so the two searches take identical steps unless a gradient component
    is within rounding of zero), and the model ends refitted on the best parameters."""
    P, t, nr = _cloud(60, 77)
    sf, l, noise = 0.9, 0.5, 0.05
    for kw in ({"max_iter": 12}, {"max_iter": 30, "delta0": 0.05, "eta_plus": 1.3}, {"max_iter": 0}):
        ref = orc.rprop_find(sf, l, noise, P[:, 0], P[:, 1], P[:, 2], t, nr, **kw)
        gg = gpu.DerivativeGP(gpu.make_kernel("se", sf, l), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
        start = gg.loglik
        got = gg.optimise(**kw)
        assert got["iterations"] == ref["iterations"], (kw, got, ref)
        assert np.abs(got["loghyper"] - ref["loghyper"]).max() < 1e-12, (kw, got, ref)
        assert abs(got["loglik"] - ref["loglik"]) < 1e-7 * max(1.0, abs(ref["loglik"]))
        assert got["loglik"] >= start - 1e-9
        assert abs(gg.loglik - got["loglik"]) < 1e-9 * max(1.0, abs(got["loglik"]))  # the model sits on the best parameters
        best = orc.DerivativeGP(("se", float(np.exp(ref["loghyper"][1])), float(np.exp(ref["loghyper"][0]))), noise,
                                P[:, 0], P[:, 1], P[:, 2], t, nr)
        assert nerr(gg.alpha, best.alpha) < 1e-9
        gg.close()


def test_rprop_descriptor_is_validated(gpu):
    """gpx_dgp_optimise rejects step sizes that would send NaN or a growing step into exp(): non-finite or non-positive
    delta / eta, delta_min > delta_max, eta_minus > 1, eta_plus < 1 -- GPX_E_BAD_ARG, the model untouched."""
    P, t, nr = _cloud(40, 5)
    gg = gpu.DerivativeGP(gpu.make_kernel("se", 0.9, 0.5), 0.05, P[:, 0], P[:, 1], P[:, 2], t, nr)
    before = gg.loglik
    for kw in ({"delta0": float("nan")}, {"delta0": -0.1}, {"eta_minus": 1.5}, {"eta_plus": 0.9}, {"delta_min": 1.0, "delta_max": 0.5},
               {"delta_max": float("inf")}, {"eps_stop": -1.0}):
        with pytest.raises(gpu.GpxError) as ei:
            gg.optimise(**kw)
        assert ei.value.code == gpu.E_BAD_ARG, kw
    assert gg.loglik == before
    assert gg.optimise(max_iter=2)["iterations"] <= 2
    gg.close()
