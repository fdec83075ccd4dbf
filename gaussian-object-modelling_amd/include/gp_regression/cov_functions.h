// Same role as the reference's include/gp_regression/cov_functions.h:4-6, plus the Matern kernels.
#ifndef GPX_SHIM_COV_FUNCTIONS_H
#define GPX_SHIM_COV_FUNCTIONS_H
#include <gp_regression/kernels/gaussian.hpp>
#include <gp_regression/kernels/laplace.hpp>
#include <gp_regression/kernels/matern.hpp>
#include <gp_regression/kernels/thin_plate.hpp>
#endif
