"""Where the time of BASELINE config C5 goes besides the variance kernel: per object wall time of create / evaluate / close
and the device stage times (fp32 mode, Gaussian(1,1), 128^3 lattice, second pass over the eight objects)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
dev = torch.device("cuda:0")
g = 128
t = torch.linspace(-1.01, 1.01, g, dtype=torch.float64, device=dev)
idx = torch.arange(g ** 3, device=dev)
q = [t[(idx // (g * g)) % g].contiguous(), t[(idx // g) % g].contiguous(), t[idx % g].contiguous()]
nq = g ** 3
f = torch.empty(nq, dtype=torch.float64, device=dev); v = torch.empty_like(f)
names = ["bowlA", "bowlB", "containerA", "containerB", "jug", "kettle", "pot", "mugD"]
sets = [gpx.node_training_set(gpx.pcd_read(os.path.join(ROOT, "tests", "golden", "pcd", nm + ".pcd"))) for nm in names]
kern = gpx.make_kernel("gaussian", 1.0, 1.0)
tot = [0.0] * 6
for rep in range(3):
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    for nm, d_ in zip(names, sets):
        t0 = time.perf_counter()
        m = gpx.Model(kern, *d_, precision={"split": gpx.F32_SPLIT, "f64": gpx.F64}.get(sys.argv[1] if len(sys.argv) > 1 else "", gpx.F32), prepare_variance=True)
        t1 = time.perf_counter()
        m.evaluate_device(nq, q[0].data_ptr(), q[1].data_ptr(), q[2].data_ptr(), f.data_ptr(), v.data_ptr())
        m.sync()
        t2 = time.perf_counter()
        st = m.stats
        m.close()
        t3 = time.perf_counter()
        if rep == 2:
            dev_create = st["t_kbuild_ms"] + st["t_factor_ms"] + st["t_solve_ms"] + st["t_inverse_ms"]
            print("%-11s N=%4d  create %.2f ms wall (device stages %.2f: kbuild %.2f LDL^T %.2f alpha %.2f inverse %.2f) | evaluate %.2f ms wall "
                  "(mean %.2f, variance stage %.2f of which kernel %.2f) | close %.2f ms" % (
                      nm, st["n"], (t1 - t0) * 1e3, dev_create, st["t_kbuild_ms"], st["t_factor_ms"], st["t_solve_ms"], st["t_inverse_ms"],
                      (t2 - t1) * 1e3, st["t_mean_ms"], st["t_var_ms"], st["t_var_gemm_ms"], (t3 - t2) * 1e3), flush=True)
            for i, x in enumerate(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, st["t_mean_ms"], st["t_var_ms"], st["t_var_gemm_ms"])):
                tot[i] += x
    if rep == 2:
        print("pass of 8 objects: %.2f ms wall = create %.2f + evaluate %.2f + close %.2f; inside evaluate: mean %.2f, variance stage %.2f (kernel %.2f)" % (
            (time.perf_counter() - T0) * 1e3, tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]))
