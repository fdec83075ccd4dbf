"""Golden vectors from the REFERENCE's own covariance classes (kernels/gaussian.hpp, laplace.hpp, thin_plate.hpp),
compiled from /root/reference by oracle/Makefile into oracle/_ref/libref_kernels.so (the wrapper
oracle/ref_kernels_wrap.cpp is ours).  Writes tests/golden/ref_kernels.npz:

    d            distances (incl. 0, tiny, the node's scale, large)
    cases        rows (kernel id, default_ctor, p0, p1)
    k, kdiff, kdiffdiff   [case][d] = compute / computediff / computediffdiff of the reference class

Run in the build container (the reference is not present on the GPU box):  python tests/golden/make_ref_kernel_golden.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load():
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libref_kernels.so"))
    dp = C.POINTER(C.c_double)
    lib.ref_kernel_eval_n.restype = None
    lib.ref_kernel_eval_n.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, dp, dp]
    return lib


def evaluate(lib, kid, which, default_ctor, p0, p1, d):
    d = np.ascontiguousarray(d, dtype=np.float64)
    out = np.empty_like(d)
    dp = C.POINTER(C.c_double)
    lib.ref_kernel_eval_n(int(kid), int(which), int(default_ctor), float(p0), float(p1), len(d), d.ctypes.data_as(dp),
                          out.ctypes.data_as(dp))
    return out


def main():
    lib = load()
    rng = np.random.default_rng(20151106)
    d = np.concatenate([[0.0, 1e-300, 1e-12, 1e-6, 0.07, 0.5, 1.0, 1.01 * np.sqrt(3.0), 2.0, 4.0, 10.0, 50.0],
                        rng.uniform(0.0, 4.5, size=200)])
    cases = [(0, 1, 0, 0), (1, 1, 0, 0), (2, 1, 0, 0),                       # default-constructed
             (0, 0, 1.0, 1.0), (0, 0, 0.7, 1.9), (0, 0, 2.5, 0.3),           # Gaussian(sigma, length)
             (1, 0, 1.0, 1.0), (1, 0, 0.7, 1.9), (1, 0, 2.5, 0.3),           # Laplace(sigma, length)
             (2, 0, 2.0, 0), (2, 0, 4.0, 0), (2, 0, 0.5, 0), (2, 0, 3.3, 0)]  # ThinPlate(R): the node's 2.0, the bench's 4.0
    k = np.stack([evaluate(lib, c[0], 0, c[1], c[2], c[3], d) for c in cases])
    kd = np.stack([evaluate(lib, c[0], 1, c[1], c[2], c[3], d) for c in cases])
    kdd = np.stack([evaluate(lib, c[0], 2, c[1], c[2], c[3], d) for c in cases])
    np.savez_compressed(os.path.join(HERE, "ref_kernels.npz"), d=d, cases=np.array(cases, dtype=np.float64), k=k,
                        kdiff=kd, kdiffdiff=kdd)
    print("wrote ref_kernels.npz:", k.shape, "finite:", np.isfinite(k).all())


if __name__ == "__main__":
    main()
