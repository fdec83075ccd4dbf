"""CPU suite, part 1: pin the oracle (oracle/gp_oracle.c) with analytic known answers, GP identities
and the independent NumPy/SciPy golden vectors (tests/golden/make_golden.py).  PARITY UNPINNED against
the reference itself: its tests hold no values."""
import numpy as np
import pytest

from conftest import KERNEL_CASES, nerr


def test_kernel_known_answers(orc):
    # thin_plate.hpp:12-20: k(0) = R^3, k(R) = 0, computediff(R) = 0
    tp = orc.make_kernel("thinplate", 2.0)
    assert orc.k(tp, 0.0)[0] == 8.0
    assert orc.k(tp, 2.0)[0] == 0.0
    assert orc.kdiff(tp, 2.0)[0] == 0.0
    assert orc.kdiff(tp, 0.5)[0] == -6 * 1.5
    # gaussian.hpp:15-27: sigma^2 exp(-d/l^2) on the UN-squared distance
    g = orc.make_kernel("gaussian", 1.5, 0.5)
    assert orc.k(g, 0.0)[0] == 2.25
    assert orc.k(g, 0.25)[0] == pytest.approx(2.25 * np.exp(-1.0), rel=1e-15)
    assert orc.kdiff(g, 0.25)[0] == pytest.approx(-4 * 2.25 * np.exp(-1.0), rel=1e-15)
    # laplace.hpp:37-49: 2 sigma exp(-d/l)
    lp = orc.make_kernel("laplace", 1.5, 0.5)
    assert orc.k(lp, 0.0)[0] == 3.0
    assert orc.kdiff(lp, 0.5)[0] == pytest.approx(-2 * 3.0 * np.exp(-1.0), rel=1e-15)
    # Matern closed forms (matlab_src/test_gp_regression_3Dsurf.m:117-123)
    d = np.array([0.0, 0.1, 0.7, 2.5])
    m32 = orc.make_kernel("matern32", 1.2, 0.8)
    m52 = orc.make_kernel("matern52", 1.2, 0.8)
    s3, s5 = np.sqrt(3) * d / 0.8, np.sqrt(5) * d / 0.8
    np.testing.assert_allclose(orc.k(m32, d), 1.44 * (1 + s3) * np.exp(-s3), rtol=1e-15)
    np.testing.assert_allclose(orc.k(m52, d), 1.44 * (1 + s5 + 5 * d * d / (3 * 0.64)) * np.exp(-s5), rtol=1e-15)
    # computediff of the Matern kernels is k'(d)/d: check against a central difference
    for kern in (m32, m52):
        for dd in (0.3, 1.1):
            h = 1e-6
            num = (orc.k(kern, dd + h)[0] - orc.k(kern, dd - h)[0]) / (2 * h) / dd
            assert orc.kdiff(kern, dd)[0] == pytest.approx(num, rel=1e-7)
    for kern in (tp, g, lp, m32, m52):  # computediffdiff is a stub returning 0 in every reference kernel
        assert orc.kdiffdiff(kern, 0.3)[0] == 0.0


@pytest.mark.parametrize("sname", ["mugD", "sphere64"])
@pytest.mark.parametrize("kkey", list(KERNEL_CASES))
def test_oracle_matches_independent_numpy(orc, golden, sname, kkey):
    kn, par = KERNEL_CASES[kkey]
    x, y, z = golden[sname + "/x"], golden[sname + "/y"], golden[sname + "/z"]
    lab, s2, Q = golden[sname + "/label"], golden[sname + "/sigma2"], golden[sname + "/Q"]
    m = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, s2)
    pre = "%s/%s/" % (sname, kkey)
    assert m.ldlt_info == 0
    assert m.R == pytest.approx(float(golden[sname + "/R"]), rel=1e-14)
    assert nerr(m.alpha, golden[pre + "alpha"]) < 1e-9
    out = m.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True, want_grad=True)
    assert nerr(out["f"], golden[pre + "f"]) < 1e-10
    assert nerr(out["v"], golden[pre + "v"]) < 1e-10
    assert nerr(out["grad"], golden[pre + "grad"]) < 1e-10
    # inertia: D of the LDL^T has as many negative entries as K has negative eigenvalues
    F, _ = m.ldlt()
    assert int((np.diag(F) < 0).sum()) == int(golden[pre + "n_negative"])


def test_gp_identities_without_noise(orc, golden):
    """Empty sigma2 (gp_regressor.hpp:154): the GP interpolates, f(p_i) = y_i and v(p_i) = 0."""
    x, y, z, lab = (golden["sphere64/" + k] for k in ("x", "y", "z", "label"))
    for kn, par in (("gaussian", (1, 1)), ("matern32", (1, 1)), ("thinplate", (4.0,))):
        m = orc.Model(orc.make_kernel(kn, *par), x, y, z, lab, None)
        out = m.evaluate(x, y, z, want_v=True)
        scale = float(orc.k(orc.make_kernel(kn, *par), 0.0)[0])
        assert np.max(np.abs(out["f"] - lab)) < 1e-8
        assert np.max(np.abs(out["v"])) < 1e-8 * scale
        K = m.Kpp
        assert np.max(np.abs(K @ m.alpha - lab)) < 1e-9


def test_noise_identity(orc, golden):
    """With sigma2: f(p_i) = y_i - sigma2_i alpha_i (k_i = K e_i - sigma2_i e_i)."""
    x, y, z, lab, s2 = (golden["mugD/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    m = orc.Model(orc.make_kernel("matern52", 1, 1), x, y, z, lab, s2)
    out = m.evaluate(x, y, z)
    assert np.max(np.abs(out["f"] - (lab - s2 * m.alpha))) < 1e-10


def test_distance_expansion_vs_direct(orc, golden):
    """gp_regressor.hpp:548-557 literally vs the build's direct differences: they agree
    wherever the expansion is finite; the expansion may produce NaN on (near-)coincident points."""
    P = np.stack([golden["mugD/" + k] for k in ("x", "y", "z")], 1)
    De = orc.dist_matrix(P, P, orc.DIST_EXPANSION)
    Dd = orc.dist_matrix(P, P, orc.DIST_DIRECT)
    assert np.all(np.diag(Dd) == 0.0)
    fin = np.isfinite(De)
    assert np.max(np.abs(De[fin] - Dd[fin])) < 1e-7  # sqrt of an O(1e-16) cancellation error near d = 0
    off = fin & (Dd > 1e-3)
    assert np.max(np.abs(De[off] - Dd[off])) < 1e-12
    # full pipeline in the reference's own formulation agrees away from the training points
    x, y, z, lab, s2 = (golden["sphere64/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    kern = orc.make_kernel("thinplate", 4.0)
    me = orc.Model(kern, x, y, z, lab, s2, dist_mode=orc.DIST_EXPANSION)
    md = orc.Model(kern, x, y, z, lab, s2, dist_mode=orc.DIST_DIRECT)
    if np.all(np.isfinite(me.alpha)):
        Q = golden["sphere64/Q"][:56]
        a = me.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])["f"]
        b = md.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])["f"]
        assert nerr(a, b) < 1e-8


def test_variance_diagonal_equals_full_covariance(orc, golden):
    """gp_regressor.hpp:307-319 builds the Nq x Nq matrix; its diagonal is what the build computes."""
    x, y, z, lab, s2, Q = (golden["sphere64/" + k] for k in ("x", "y", "z", "label", "sigma2", "Q"))
    m = orc.Model(orc.make_kernel("gaussian", 1, 1), x, y, z, lab, s2)
    Q = Q[:20]
    v_full, V = m.evaluate_fullcov(Q[:, 0], Q[:, 1], Q[:, 2])
    v = m.evaluate(Q[:, 0], Q[:, 1], Q[:, 2], want_v=True)["v"]
    assert nerr(v, v_full) < 1e-12
    assert np.max(np.abs(V - V.T)) < 1e-12


def test_update_equals_create_on_concatenation(orc, golden):
    """gp_regressor.hpp:367-479: update appends and refactors from scratch; R is not refreshed."""
    x, y, z, lab, s2 = (golden["sphere64/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    kern = orc.make_kernel("laplace", 1, 1)
    m = orc.Model(kern, x[:50], y[:50], z[:50], lab[:50], s2[:50])
    R0 = m.R
    m.update(x[50:], y[50:], z[50:], lab[50:], s2[50:])
    full = orc.Model(kern, x, y, z, lab, s2)
    assert m.n == 64
    assert nerr(m.alpha, full.alpha) < 1e-12
    assert m.R == R0 and full.R >= R0


def test_ldlt_against_scipy(orc):
    rng = np.random.default_rng(3)
    for n, shift in ((40, 5.0), (65, 0.0)):  # SPD and indefinite
        A = rng.normal(size=(n, n))
        A = A + A.T + shift * n * np.eye(n) * (1 if shift else 0) + np.diag(rng.uniform(1, 2, n))
        F, t, info = orc.ldlt(A)
        assert info == 0
        # reconstruct P A P^T = L D L^T
        L = np.tril(F, -1) + np.eye(n)
        D = np.diag(np.diag(F))
        perm = np.arange(n)
        for k in range(n):
            perm[[k, t[k]]] = perm[[t[k], k]]
        assert np.max(np.abs(L @ D @ L.T - A[np.ix_(perm, perm)])) < 1e-9 * np.abs(A).max()
        b = rng.normal(size=n)
        assert nerr(orc.ldlt_solve(F, t, b), np.linalg.solve(A, b)) < 1e-9
        assert int((np.diag(F) < 0).sum()) == int((np.linalg.eigvalsh(A) < 0).sum())
    # pivot rule: first largest |diagonal| of the not-yet-eliminated block (non-uniform diagonal)
    A = np.diag([1.0, 5.0, 3.0, 5.0]) + 0.1
    _, t, _ = orc.ldlt(A)
    assert list(t[:2]) == [1, 3]


def test_tangent_basis(orc):
    """computeTangentBasis, gp_regressor.hpp:29-44."""
    N, Tx, Ty = orc.tangent_basis([0.0, 0.0, 2.0])
    np.testing.assert_allclose(N, [0, 0, 1])
    np.testing.assert_allclose(Tx, [1, 0, 0])
    np.testing.assert_allclose(Ty, [0, 1, 0])
    N, Tx, Ty = orc.tangent_basis([3.0, 0.0, 0.0])  # along UnitX -> falls back to UnitY
    np.testing.assert_allclose(Tx, [0, 1, 0])
    np.testing.assert_allclose(Ty, [0, 0, 1])
    g = np.array([0.3, -1.2, 0.5])
    N, Tx, Ty = orc.tangent_basis(g)
    for a, b in ((N, Tx), (N, Ty), (Tx, Ty)):
        assert abs(np.dot(a, b)) < 1e-14
    for a in (N, Tx, Ty):
        assert abs(np.linalg.norm(a) - 1) < 1e-14


def test_oracle_normals(orc, golden):
    """create<true>, gp_regressor.hpp:166-181: normalised gradient at the training points."""
    x, y, z, lab, s2 = (golden["sphere64/" + k] for k in ("x", "y", "z", "label", "sigma2"))
    m = orc.Model(orc.make_kernel("gaussian", 1, 1), x, y, z, lab, s2, with_normals=True)
    g = m.evaluate(x, y, z, want_grad=True)["grad"]
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    assert nerr(m.normals, g) < 1e-12


def test_project_restatement_properties(orc, ds):
    """orc_project follows AtlasBase::project (atlas.hpp:201-276): exits and their invariants."""
    x, y, z, lab, s2 = ds.fibonacci_training_set(150)
    om = orc.Model(orc.make_kernel("thinplate", 2.0), x, y, z, lab, s2)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(24, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    P = d * rng.uniform(0.85, 1.25, size=(24, 1))
    g = om.evaluate(P[:, 0], P[:, 1], P[:, 2], want_grad=True)["grad"]
    r = om.project(P[:, 0], P[:, 1], P[:, 2], g, step_mul=0.5, max_iter=60)
    f_at = om.evaluate(r["xyz"][:, 0], r["xyz"][:, 1], r["xyz"][:, 2])["f"]
    np.testing.assert_allclose(r["f"], f_at, rtol=0, atol=1e-12)          # out_f is the mean at the returned point
    assert np.all(np.abs(r["f"][r["status"] == 1]) < 1e-2)                 # first criterion
    assert np.all(r["iter"][r["status"] == 3] == 60)
    assert np.all(r["status"] == 1) and np.all(r["iter"] < 60)             # this start set converges
    # the reference's defaults (step_mul 0.001) barely move: most points run into max_iter, as in the node's logs
    r0 = om.project(P[:, 0], P[:, 1], P[:, 2], g, max_iter=40)
    far = np.abs(om.evaluate(P[:, 0], P[:, 1], P[:, 2])["f"]) >= 2e-2
    assert np.all(r0["status"][far] == 3)
    assert np.max(np.linalg.norm(r0["xyz"] - P, axis=1)) < 40 * 0.001 * np.max(np.abs(r["f"]) + 1) * np.max(np.linalg.norm(g, axis=1)) + 1
    # max_iter = 0: nothing happens, f is the mean at the start point
    rz = om.project(P[:, 0], P[:, 1], P[:, 2], g, max_iter=0)
    np.testing.assert_array_equal(rz["xyz"], P)
    assert np.all(rz["status"] == 3) and np.all(rz["iter"] == 0)
    # a zero start direction is a "wrong step" (atlas.hpp:246-249): the point stays until a gradient is adopted
    r1 = om.project(P[:1, 0], P[:1, 1], P[:1, 2], np.zeros((1, 3)), step_mul=0.5, max_iter=60)
    assert r1["status"][0] == 1


# ---- second library of the reference, gp::GaussianProcess (values + gradients): the oracle's own consistency ----------
@pytest.mark.parametrize("kern", [("se", 1.3, 0.7), ("thinplate", 3.0)])
def test_derivative_gp_oracle_blocks_and_identities(orc, kern):
    """oracle/gp_oracle.c, second half (include/gp/GaussianProcess.h:532-583 with exact derivative blocks; the
    reference's own blocks are inconsistent, see the header there).  Pins the restatement WITHOUT the reference: the
    derivative blocks of the 4n x 4n covariance against central differences of k, alpha against numpy.linalg.solve,
    the gradient of the mean against differences of the mean, interpolation of values and normals at small noise."""
    rng = np.random.default_rng(3)
    n = 14
    P = rng.normal(size=(n, 3))
    P /= np.linalg.norm(P, axis=1)[:, None]
    t, nr = np.zeros(n), P.copy()
    g = orc.DerivativeGP(kern, 0.02, P[:, 0], P[:, 1], P[:, 2], t, nr)
    assert g.info == 0
    K = g.K
    assert np.abs(K - K.T).max() < 1e-14 and np.linalg.eigvalsh(K).min() > 0

    def k(a, b):
        r = np.linalg.norm(a - b)
        return kern[1] ** 2 * np.exp(-0.5 * r * r / kern[2] ** 2) if kern[0] == "se" else 2 * r ** 3 - 3 * kern[1] * r * r + kern[1] ** 3

    eps, e1, e2 = 1e-5, 0.0, 0.0
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            for d in range(3):
                a = np.zeros(3)
                a[d] = eps
                e1 = max(e1, abs((k(P[i] + a, P[j]) - k(P[i] - a, P[j])) / (2 * eps) - K[n + 3 * i + d, j]))
                for d2 in range(3):
                    b = np.zeros(3)
                    b[d2] = eps
                    fd = (k(P[i] + a, P[j] + b) - k(P[i] + a, P[j] - b) - k(P[i] - a, P[j] + b) + k(P[i] - a, P[j] - b)) / (4 * eps * eps)
                    e2 = max(e2, abs(fd - K[n + 3 * i + d, n + 3 * j + d2]))
    assert e1 < 1e-8 and e2 < 2e-4  # (second differences at eps = 1e-5 carry ~1e-5 of rounding)
    yv = np.concatenate([t, nr.reshape(-1)])
    a_np = np.linalg.solve(K, yv)
    assert nerr(g.alpha, a_np) < 1e-10
    sign, logdet = np.linalg.slogdet(K)
    assert abs(g.loglik - (-0.5 * yv @ a_np - 0.5 * logdet - 0.5 * 4 * n * np.log(2 * np.pi))) < 1e-8
    Q = rng.uniform(-1.2, 1.2, size=(6, 3))
    o = g.evaluate(Q[:, 0], Q[:, 1], Q[:, 2])
    fd = np.zeros((6, 3))
    for d in range(3):
        e = np.zeros(3)
        e[d] = 1e-6
        fp = g.evaluate(Q[:, 0] + e[0], Q[:, 1] + e[1], Q[:, 2] + e[2], want_v=False)["f"]
        fm = g.evaluate(Q[:, 0] - e[0], Q[:, 1] - e[1], Q[:, 2] - e[2], want_v=False)["f"]
        fd[:, d] = (fp - fm) / 2e-6
    assert np.abs(fd - o["grad"]).max() < 1e-7
    # variance against the dense formula, 0 <= v <= k(0)
    ks = np.zeros((6, 4 * n))
    for q in range(6):
        for j in range(n):
            u = Q[q] - P[j]
            r = np.linalg.norm(u)
            gg = (-k(Q[q], P[j]) / kern[2] ** 2) if kern[0] == "se" else 6 * r - 6 * kern[1]
            ks[q, j] = k(Q[q], P[j])
            ks[q, n + 3 * j:n + 3 * j + 3] = -gg * u
    v_np = k(P[0], P[0]) - np.einsum("qi,qi->q", ks, np.linalg.solve(K, ks.T).T)
    assert nerr(o["v"], v_np) < 1e-9 and o["v"].min() > -1e-12
    # at the training points: the mean reproduces the targets and the gradient the normals (to the noise level)
    at = g.evaluate(P[:, 0], P[:, 1], P[:, 2], want_v=False)
    assert np.abs(at["f"] - t).max() < 5e-3 and np.abs(at["grad"] - nr).max() < 5e-2


def test_derivative_gp_oracle_likelihood_gradient_and_rprop(orc):
    """oracle/gp_oracle.py DerivativeGP.loglik_gradient / rprop_find (include/gp/GaussianProcess.h:387-410, :86-122) pinned
    WITHOUT the reference (that class does not build): d K / d log l against central differences of K, the gradient
    against central differences of the likelihood in (log l, log sf), and the RProp search against its own invariants --
    the result is the best likelihood of the trace, never below the start, and repeatable."""
    rng = np.random.default_rng(11)
    n = 12
    P = rng.normal(size=(n, 3))
    P /= np.linalg.norm(P, axis=1)[:, None]
    t, nr = 0.01 * rng.normal(size=n), P.copy()
    sf, l, noise = 0.9, 0.6, 0.05
    mk = lambda pl, ps: orc.DerivativeGP(("se", float(np.exp(ps)), float(np.exp(pl))), noise, P[:, 0], P[:, 1], P[:, 2], t, nr)
    p0 = np.array([np.log(l), np.log(sf)])
    g = mk(*p0)
    eps = 1e-6
    fdK = (mk(p0[0] + eps, p0[1]).K - mk(p0[0] - eps, p0[1]).K) / (2 * eps)
    assert np.abs(fdK - g.dK_dlogl()).max() < 1e-8 * np.abs(fdK).max()
    grad = g.loglik_gradient()
    fd = np.array([(mk(p0[0] + eps, p0[1]).loglik - mk(p0[0] - eps, p0[1]).loglik) / (2 * eps),
                   (mk(p0[0], p0[1] + eps).loglik - mk(p0[0], p0[1] - eps).loglik) / (2 * eps)])
    assert np.abs(grad - fd).max() < 1e-6 * max(1.0, np.abs(fd).max()), (grad, fd)
    r = orc.rprop_find(sf, l, noise, P[:, 0], P[:, 1], P[:, 2], t, nr, max_iter=25)
    assert r["iterations"] == len(r["trace"]) and 1 <= r["iterations"] <= 25
    liks = [lk for _, lk in r["trace"]]
    assert r["loglik"] == max(liks) and r["loglik"] > g.loglik
    i_best = int(np.argmax(liks))
    np.testing.assert_array_equal(r["loghyper"], r["trace"][i_best][0])
    r2 = orc.rprop_find(sf, l, noise, P[:, 0], P[:, 1], P[:, 2], t, nr, max_iter=25)
    np.testing.assert_array_equal(r["loghyper"], r2["loghyper"])
    # a long search ends where the likelihood is flatter than at the start
    r3 = orc.rprop_find(sf, l, noise, P[:, 0], P[:, 1], P[:, 2], t, nr, max_iter=100)
    gb = mk(*r3["loghyper"]).loglik_gradient()
    assert r3["loglik"] >= r["loglik"] and np.linalg.norm(gb) < 0.5 * np.linalg.norm(grad), (gb, grad)
