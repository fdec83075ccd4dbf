"""GPU suite, corners the parity suites left open: an allocation the device cannot satisfy, and non-finite
queries of ONE caller inside work shared with others (a flat-combined batch of concurrent host calls, a compacted sample_surface
grid) -- every other caller's output must stay what it is without them, bit for bit."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_out_of_memory_is_a_status_and_leaves_pool_and_device_intact(gpu, ds):
    """A shell whose inverse factor alone (fp64, 200 192^2 entries = 320 GB) exceeds the 288 GB of the device: GPX_E_OOM with the
    runtime's message, no model handle; the caller's current device is unchanged, a normal model created next works and gives the
    same values as before, and after gpx_trim the free device memory is back where it was."""
    torch = pytest.importorskip("torch")
    data = ds.fibonacci_training_set(300)
    q = ds.query_grid(7)
    kern = gpu.make_kernel("matern52", 1.0, 1.0)

    def ordinary():
        m = gpu.Model(kern, *data, precision=gpu.F32, prepare_variance=True)
        o = m.evaluate(*q, want_v=True)
        m.close()
        return o

    before = ordinary()
    gpu.trim()
    torch.cuda.synchronize()
    dev0 = torch.cuda.current_device()
    free0 = torch.cuda.mem_get_info()[0]
    for prec in (gpu.F64, gpu.F32):
        with pytest.raises(gpu.GpxError) as ei:
            gpu.Model.shell(kern, 200000 if prec == gpu.F64 else 290000, precision=prec)
        assert ei.value.code == gpu.E_OOM, ei.value
        assert "memory" in ei.value.message.lower()
    assert torch.cuda.current_device() == dev0
    after = ordinary()
    np.testing.assert_array_equal(before["f"], after["f"])
    np.testing.assert_array_equal(before["v"], after["v"])
    gpu.trim()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert abs(free0 - free1) < 8 << 20, "free device memory moved by %.1f MiB across the failed allocations" % ((free0 - free1) / 2**20)


@pytest.mark.parametrize("n, prec", [(277, "F64"), (277, "F32"), (1500, "F32")])
def test_non_finite_query_of_one_caller_touches_no_other_caller(gpu, ds, n, prec):
    """64 host threads, one query point each, on one const model (src/gp_node.cpp:1027-1038): the calls are flat-combined into
    shared device batches.  Two of the callers pass NaN / inf coordinates: their own outputs are non-finite, every other caller
    receives exactly what it receives when it is alone.  (The poisoned callers receive NaN, as from the reference's arithmetic --
    the device's fast kernel evaluators clamp their arguments; gpx_predict.hip writes the NaN.)"""
    m = gpu.Model(gpu.make_kernel("gaussian", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=getattr(gpu, prec),
                  prepare_variance=True)
    rng = np.random.default_rng(n)
    q = rng.uniform(-1.1, 1.1, size=(64, 3))
    alone = [m.evaluate(q[i:i + 1, 0].copy(), q[i:i + 1, 1].copy(), q[i:i + 1, 2].copy(), want_v=True, want_grad=True) for i in range(64)]
    bad = q.copy()
    bad[17, 1] = np.nan
    bad[40, 0] = np.inf
    for rep in range(3):
        out, errs = [None] * 64, []
        gate = threading.Barrier(64)

        def work(i):
            try:
                gate.wait()
                out[i] = m.evaluate(bad[i:i + 1, 0].copy(), bad[i:i + 1, 1].copy(), bad[i:i + 1, 2].copy(), want_v=True, want_grad=True)
            except Exception as e:  # noqa
                errs.append(e)

        th = [threading.Thread(target=work, args=(i,)) for i in range(64)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(64):
            if i in (17, 40):
                assert np.isnan(out[i]["f"][0]) and np.isnan(out[i]["v"][0]) and np.isnan(out[i]["grad"]).all()
                continue
            for key in ("f", "v", "grad"):
                np.testing.assert_array_equal(out[i][key], alone[i][key], err_msg="caller %d %s" % (i, key))
    m.close()


@pytest.mark.parametrize("g, n, prec", [(40, 724, "F32"), (20, 300, "F64")])
def test_non_finite_queries_inside_a_sample_surface_grid(gpu, ds, g, n, prec):
    """NaN / +-inf coordinates in the middle of the lattice (64000 points: the fp32 screen and its compaction run; 8000: they do
    not): the poisoned points are not selected, everything else is selected and valued exactly as on the clean lattice."""
    m = gpu.Model(gpu.make_kernel("matern52", 1.0, 1.0), *ds.fibonacci_training_set(n), precision=getattr(gpu, prec),
                  prepare_variance=True)
    t = np.linspace(-1.01, 1.01, g)
    qx, qy, qz = (a.ravel().copy() for a in np.meshgrid(t, t, t, indexing="ij"))
    clean = m.sample_surface(qx, qy, qz, f_tol=0.03)
    assert clean["n_total"] > 50
    # poison survivors (and their neighbours in the array) and a few arbitrary points
    poison = sorted(set([int(clean["idx"][k]) for k in (0, 7, clean["n_total"] // 2, clean["n_total"] - 1)] + [1, g ** 3 // 2, g ** 3 - 1]))
    bx, by, bz = qx.copy(), qy.copy(), qz.copy()
    for k, p in enumerate(poison):
        (bx, by, bz)[k % 3][p] = (np.nan, np.inf, -np.inf)[(k // 3) % 3]
    got = m.sample_surface(bx, by, bz, f_tol=0.03)
    keep = ~np.isin(clean["idx"], poison)
    assert got["n_total"] == int(keep.sum())
    for key in ("idx", "f", "v"):
        np.testing.assert_array_equal(got[key], clean[key][keep], err_msg=key)
    # and through evaluate: the poisoned outputs are non-finite, all others bit-identical
    a, b = m.evaluate(qx, qy, qz, want_v=True), m.evaluate(bx, by, bz, want_v=True)
    ok = np.ones(g ** 3, bool)
    ok[poison] = False
    np.testing.assert_array_equal(a["f"][ok], b["f"][ok])
    np.testing.assert_array_equal(a["v"][ok], b["v"][ok])
    assert np.isnan(b["f"][poison]).all() and np.isnan(b["v"][poison]).all()
    m.close()
