"""Latency of the node's call pattern: one query point per evaluate(f, v) call, many host threads."""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
x, y, z, lab, s2 = ds.node_training_set(ds.read_pcd(os.path.join(ROOT, "tests/golden/pcd/mugD.pcd")))
m = gpx.Model(gpx.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=gpx.F64)
qx, qy, qz = ds.query_grid(29)
n = len(qx)
m.evaluate(qx[:1], qy[:1], qz[:1], want_v=True)
t = time.perf_counter()
for i in range(2000):
    m.evaluate(qx[i:i + 1], qy[i:i + 1], qz[i:i + 1], want_v=True)
dt = time.perf_counter() - t
print("serial: %.1f us per single-point evaluate(f,v)" % (dt / 2000 * 1e6), flush=True)
for nth in (8, 64, 256):
    f = np.zeros(n); v = np.zeros(n)
    def work(lo, hi):
        for i in range(lo, hi):
            o = m.evaluate(qx[i:i + 1], qy[i:i + 1], qz[i:i + 1], want_v=True)
            f[i] = o["f"][0]; v[i] = o["v"][0]
    chunk = (n + nth - 1) // nth
    th = [threading.Thread(target=work, args=(k * chunk, min(n, (k + 1) * chunk))) for k in range(nth)]
    t = time.perf_counter()
    for a in th: a.start()
    for a in th: a.join()
    dt = time.perf_counter() - t
    print("%d threads: %d single-point calls in %.3f s = %.1f us per call" % (nth, n, dt, dt / n * 1e6), flush=True)
t = time.perf_counter(); o = m.evaluate(qx, qy, qz, want_v=True); dt = time.perf_counter() - t
print("one batched call of %d points: %.3f ms; max |f diff| vs threaded = %.2e" % (n, dt * 1e3, np.max(np.abs(o["f"] - f))))
for prec, name in ((gpx.F64, "f64"), (gpx.F32, "f32")):
    t = time.perf_counter()
    for _ in range(20):
        mm = gpx.Model(gpx.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=prec)
        mm.close()
    dt = (time.perf_counter() - t) / 20
    mm = gpx.Model(gpx.make_kernel("thinplate", 2.0), x, y, z, lab, s2, precision=prec)
    st = mm.stats
    print("create+destroy N=%d %s: %.2f ms wall (device: kbuild %.3f factor %.3f solve %.3f ms)" % (len(x), name, dt * 1e3, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"]))
    t = time.perf_counter(); mm.evaluate(qx[:1], qy[:1], qz[:1], want_v=True); print("  first variance query (builds the inverse factor): %.2f ms" % ((time.perf_counter() - t) * 1e3))
    mm.close()
