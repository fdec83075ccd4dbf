"""The shader-clock split of the wide dataflow factorisation (make EXTRA=-DWIDE_TIMING OUTDIR=../lib_t OBJDIR=../build_t; run with
GPX_LIB=.../lib_t/libgpx.so): creates of Matern-5/2 models, the library prints one `wide_timing` line per launch on stderr.
Usage: GPX_LIB=... python scripts/wide_timing.py [sizes...]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
gpx = importlib.import_module("gaussian-object-modelling_amd.gpx")
ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
sizes = [int(a) for a in sys.argv[1:]] or [8192, 12288, 16384]
for prec, name in ((gpx.F32, "fp32"), (gpx.F64, "fp64")):
    for n in sizes:
        data = ds.fibonacci_training_set(n)
        for i in range(3):
            m = gpx.Model(gpx.make_kernel("matern52", 1.0, 1.0), *data, precision=prec)
            st = m.stats
            m.close()
        print("%s N=%d t_factor_ms %.3f fallbacks %d" % (name, n, st["t_factor_ms"], st["solve_fallbacks"]), flush=True)
