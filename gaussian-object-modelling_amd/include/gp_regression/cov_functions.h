// Header shim, not the reference's file: pulls in every covariance class a GPRegressor<Cov> can be instantiated
// with.  The three classes of the reference keep their names, constructor shapes and the reference's semantics
// (Gaussian = sigma^2 exp(-d / l^2) on the UN-squared distance, Laplace amplitude 2 sigma, ThinPlate
// 2 d^3 - 3 R d^2 + R^3 with computediff = k'(d) / d); Matern32 / Matern52 are additions that follow the closed
// forms of the reference's MATLAB prototype.  On the device the same functions live in csrc/gpx_cov.hpp; the
// classes here only carry the parameters to gpx_model_create and answer compute() / computediff() on the host.
#ifndef GPX_SHIM_COV_FUNCTIONS_H
#define GPX_SHIM_COV_FUNCTIONS_H
#include <gp_regression/kernels/matern.hpp>      // Matern32, Matern52 (new)
#include <gp_regression/kernels/thin_plate.hpp>  // ThinPlate
#include <gp_regression/kernels/laplace.hpp>     // Laplace
#include <gp_regression/kernels/gaussian.hpp>    // Gaussian
#endif
