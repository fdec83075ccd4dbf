"""Development check: GPU path vs the CPU oracle on small cases (prints norm-wise errors)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gp_oracle as orc
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")

def nerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))

def run(name, tr, kn, par, prec, nq_grid=9, normals=False):
    x, y, z, lab, s2 = tr
    ok = orc.make_kernel(kn, *par); gk = gpx.make_kernel(kn, *par)
    t = time.time(); om = orc.Model(ok, x, y, z, lab, s2, with_normals=normals, omp=False); to = time.time() - t
    t = time.time(); gm = gpx.Model(gk, x, y, z, lab, s2, precision=prec, with_normals=normals); tg = time.time() - t
    qx, qy, qz = ds.query_grid(nq_grid)
    qx = np.concatenate([qx, x[:7], [3.0]]); qy = np.concatenate([qy, y[:7], [0.1]]); qz = np.concatenate([qz, z[:7], [-2.0]])
    o = om.evaluate(qx, qy, qz, want_v=True, want_grad=True, want_basis=True)
    g = gm.evaluate(qx, qy, qz, want_v=True, want_grad=True, want_basis=True)
    st = gm.stats
    print("%-10s %-9s %s n=%d | alpha %.2e f %.2e v %.2e grad %.2e tx %.2e R %.2e | neg %d res %.1e | cpu %.2fs gpu %.2fs kb %.3f fac %.3f sol %.3f inv %.3f mean %.3f var %.3f ms" % (
        name, kn, "f64" if prec else "f32", len(x), nerr(gm.alpha, om.alpha), nerr(g["f"], o["f"]), nerr(g["v"], o["v"]),
        nerr(g["grad"], o["grad"]), nerr(g["tx"], o["tx"]), abs(gm.R - om.R), st["n_negative_pivots"], st["alpha_residual"],
        to, tg, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], st["t_inverse_ms"], st["t_mean_ms"], st["t_var_ms"]), flush=True)
    if normals:
        print("    normals err %.2e" % nerr(gm.normals, om.normals))
    gm.close()

if __name__ == "__main__":
    print("devices:", gpx.device_count(), flush=True)
    mug = ds.node_training_set(ds.read_pcd(os.path.join(ROOT, "tests/golden/pcd/mugD.pcd")))
    big = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    syn = ds.fibonacci_training_set(big)
    precs = [gpx.F64, gpx.F32] if len(sys.argv) < 3 else [int(a) for a in sys.argv[2].split(",")]
    for prec in precs:
        for kn, par in [("gaussian", (1, 1)), ("laplace", (1, 1)), ("thinplate", (2.0,)), ("thinplate", (4.0,)),
                        ("matern32", (1, 1)), ("matern52", (1, 1))]:
            run("mugD", mug, kn, par, prec, normals=(kn == "gaussian"))
            run("fib%d" % big, syn, kn, par, prec)
