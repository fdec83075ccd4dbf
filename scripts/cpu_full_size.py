#!/usr/bin/env python3
"""Times the reference-shaped CPU path ONCE at the headline size:
one-core oracle create (materialised K, unblocked pivoted LDL^T of gp_regressor.hpp:161-163) at N_train = 16384 and
256 per-query mean+variance solves (gp_regressor.hpp:307-319 with Nq = 1 each, as the node calls it).

No GPU needed; ~10-25 min on one core.  Also times the N = 3072 sample bench.py's cpu_baseline leg extrapolates
from, on the same core, so that the measured / extrapolated ratio can be quoted.

    python3 scripts/cpu_full_size.py [N_TRAIN] > profiles/r04_cpu_full_size.txt
"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
    nqs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    orc = importlib.import_module("gp_oracle")
    ds = importlib.import_module("gaussian-object-modelling_amd.datasets")
    kern = orc.make_kernel("matern52", 1.0, 1.0)
    print("host: %s, cores visible %d, 1 thread used" % (os.uname().nodename, os.cpu_count()), flush=True)

    def leg(ns, nq):
        x, y, z, lab, s2 = ds.fibonacci_training_set(ns)
        qx, qy, qz = ds.query_grid(11)
        qx, qy, qz = qx[:nq], qy[:nq], qz[:nq]
        t0 = time.perf_counter()
        m = orc.Model(kern, x, y, z, lab, s2, omp=False)
        t1 = time.perf_counter()
        m.evaluate(qx, qy, qz, want_v=True)
        t2 = time.perf_counter()
        return t1 - t0, (t2 - t1) / nq

    c_s, q_s = leg(3072, 1024)
    print("sample  N_train=3072 : create %.3f s, %.4f ms/query (1024 queries)" % (c_s, q_s * 1e3), flush=True)
    r = n_train / 3072.0
    c_x, q_x = c_s * r ** 3, q_s * r ** 2
    print("extrapolated to N_train=%d by N^3 / N^2: create %.1f s, %.3f ms/query" % (n_train, c_x, q_x * 1e3), flush=True)
    c_f, q_f = leg(n_train, nqs)
    print("measured N_train=%d : create %.1f s, %.3f ms/query (%d queries)" % (n_train, c_f, q_f * 1e3, nqs), flush=True)
    print("measured / extrapolated: create %.3f, per-query %.3f" % (c_f / c_x, q_f / q_x), flush=True)
    nq = 1 << 20
    full_m, full_x = c_f + q_f * nq, c_x + q_x * nq
    print("train + predict of 2^20 queries: measured-rate %.0f s (%.3f query-points/s), extrapolated %.0f s (%.3f); ratio %.3f"
          % (full_m, nq / full_m, full_x, nq / full_x, full_m / full_x), flush=True)


if __name__ == "__main__":
    main()
