"""Per-workgroup time stamps of small_factor_kernel (library built with EXTRA=-DSM_TIMING into lib_t):
GPX_LIB=gaussian-object-modelling_amd/lib_t/libgpx.so python scripts/small_timing.py N"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPX_LIB", os.path.join(ROOT, "gaussian-object-modelling_amd", "lib_t", "libgpx.so"))
dump = os.path.join(ROOT, "gpurun_out", "small_stamps.txt")
os.environ["GPX_SMALL_TIMING_DUMP"] = dump
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 277
data = ds.fibonacci_training_set(n)
for _ in range(3):
    m = gpx.Model(gpx.make_kernel("gaussian", 1.0, 1.0), *data, precision=gpx.F64, prepare_variance=True)
    st = m.stats
    m.close()
print("n=%d factor %.3f ms solve %.3f ms" % (n, st["t_factor_ms"], st["t_solve_ms"]))
rows = [list(map(int, l.split())) for l in open(dump)]
npad = (n + 255) // 256 * 256
nbt = npad // 64
t0 = min(v for r in rows for v in r[1:] if v > 0)
names = {0: "built", 1: "upd_done", 2: "A_staged", 3: "diag_k_seen", 4: "Xd_staged", 5: "own_upd", 6: "ldl_start", 7: "ldl_done", 8: "published",
         9: "diag_end", 10: "wait_diag", 11: "diag_seen", 12: "L_ready", 13: "wait_own_diag", 14: "own_diag_seen", 15: "X_published", 16: "sub0", 17: "W21", 18: "c22", 19: "sub1"}
for r in rows:
    t = r[0]
    j, rem = 0, t
    while rem >= nbt - j:
        rem -= nbt - j
        j += 1
    i = j + rem
    if i >= (n + 63) // 64:
        continue
    if not (i == j or i == j + 1):
        continue
    print("tile (%2d,%2d): " % (i, j) + "  ".join("%s=%.2f" % (names[k], (v - t0) / 100.0) for k, v in sorted(enumerate(r[1:25]), key=lambda kv: kv[1]) if v > 0))
