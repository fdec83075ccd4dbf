"""LDL^T above the dataflow's upper bound (N = 24576, 32768; GPX_DATAFLOW=128 extends the 128 x 128-tile dataflow there):
where the launch chain's big GEMM updates take over again -- profiles/r05_ldlt_sweep.txt, last block."""
import importlib, os, sys
sys.path.insert(0, '/root/repo')
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
for prec, name in ((gpx.F32, "fp32"), (gpx.F64, "fp64")):
    for n in (24576, 32768):
        data = ds.fibonacci_training_set(n)
        kern = gpx.make_kernel("matern52", 1.0, 1.0)
        for i in range(3):
            m = gpx.Model(kern, *data, precision=prec)
            st = m.stats
            m.close()
        print(name, n, "factor %.2f kbuild %.2f solve %.2f residual %.1e fb %d" % (st["t_factor_ms"], st["t_kbuild_ms"], st["t_solve_ms"], st["alpha_residual"], st["solve_fallbacks"]), flush=True)
