"""Whole LDL^T by size and precision: create() (no inverse factor) of Matern-5/2 models on the Fibonacci cloud, mean of `reps`
creates after one warm-up: t_factor_ms, N^3/3 flop rate and share of the MFMA peak, the event-timed trailing updates beside it.
GPX_TRAIN_F64_MAX=0 in the environment makes the fp32 mode factorise in fp32 at every size (default: fp64 up to 2048 rows).
Usage: python scripts/ldlt_sweep.py [reps]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
PEAK = {gpx.F32: 157.3, gpx.F64: 78.6}
for prec, name in ((gpx.F32, "fp32"), (gpx.F64, "fp64")):
    for n in (1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
        data = ds.fibonacci_training_set(n)
        kern = gpx.make_kernel("matern52", 1.0, 1.0)
        tf = tg = fl = ts = tk = 0.0
        for i in range(reps + 1):
            m = gpx.Model(kern, *data, precision=prec)
            st = m.stats
            m.close()
            if i:
                tf += st["t_factor_ms"] / reps; tg += st["t_factor_gemm_ms"] / reps; fl = st["factor_gemm_flops"]
                ts += st["t_solve_ms"] / reps; tk += st["t_kbuild_ms"] / reps
        whole = n ** 3 / 3.0 / (tf * 1e-3) / 1e12
        upd = fl / (tg * 1e-3) / 1e12 if tg > 0 else 0.0
        print("%s N=%5d: LDL^T %8.3f ms = %6.2f TFLOP/s = %5.1f %% of %5.1f | trailing updates %8.3f ms (%5.1f %% of peak on their own flops), "
              "everything else %7.3f ms | kbuild %.3f solve %.3f ms" % (name, n, tf, whole, 100 * whole / PEAK[prec], PEAK[prec], tg,
              100 * upd / PEAK[prec], tf - tg, tk, ts), flush=True)
