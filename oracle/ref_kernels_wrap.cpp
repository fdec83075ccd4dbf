// ref_kernels_wrap.cpp -- TEST INFRASTRUCTURE.  A C wrapper (this file is ours) around the reference's OWN covariance
// classes, compiled from the headers where they lie: /root/reference/include/gp_regression/kernels/{gaussian,laplace,
// thin_plate}.hpp are plain C++ over <cmath> (no Eigen), the one part of the hot path that builds in this image.
// oracle/Makefile compiles it into oracle/_ref/libref_kernels.so (git-ignored; never copied into the repo); it pins
// orc_k / orc_kdiff / orc_kdiffdiff of gp_oracle.c and generates tests/golden/ref_kernels.npz
// (tests/golden/make_ref_kernel_golden.py).  Kernel ids as in gp_oracle.c: 0 Gaussian, 1 Laplace, 2 ThinPlate.
#include <cmath>
#include <gp_regression/kernels/gaussian.hpp>
#include <gp_regression/kernels/laplace.hpp>
#include <gp_regression/kernels/thin_plate.hpp>

extern "C" {

// which: 0 compute, 1 computediff, 2 computediffdiff; default_ctor != 0 uses the class's default constructor
double ref_kernel_eval(int id, int which, int default_ctor, double p0, double p1, double d)
{
    double v = d;  // the Gaussian / Laplace members take a non-const reference
    switch (id) {
    case 0: {
        gp_regression::Gaussian g = default_ctor ? gp_regression::Gaussian() : gp_regression::Gaussian(p0, p1);
        return which == 0 ? g.compute(v) : which == 1 ? g.computediff(v) : g.computediffdiff(v);
    }
    case 1: {
        gp_regression::Laplace g = default_ctor ? gp_regression::Laplace() : gp_regression::Laplace(p0, p1);
        return which == 0 ? g.compute(v) : which == 1 ? g.computediff(v) : g.computediffdiff(v);
    }
    case 2: {
        gp_regression::ThinPlate g = default_ctor ? gp_regression::ThinPlate() : gp_regression::ThinPlate(p0);
        return which == 0 ? g.compute(v) : which == 1 ? g.computediff(v) : g.computediffdiff(v);
    }
    }
    return NAN;
}

void ref_kernel_eval_n(int id, int which, int default_ctor, double p0, double p1, int n, const double *d, double *out)
{
    for (int i = 0; i < n; ++i)
        out[i] = ref_kernel_eval(id, which, default_ctor, p0, p1, d[i]);
}
}
