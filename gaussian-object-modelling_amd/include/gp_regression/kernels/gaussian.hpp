// gp_regression::Gaussian -- interface of the reference's kernels/gaussian.hpp:9-57.
// Semantics kept verbatim: sigma^2 * exp(-d / length^2) on the UN-squared distance (:15-20),
// computediff = -(1/length^2) * compute (:22-27), computediffdiff = 0 (:29-34).
// The host methods exist for callers that evaluate the kernel themselves; GPRegressor hands the
// parameters to the GPU library.
#ifndef GPX_SHIM_GAUSSIAN_HPP
#define GPX_SHIM_GAUSSIAN_HPP
#include <cmath>
namespace gp_regression
{
class Gaussian
{
public:
    const double sigma_;
    const double length_;
    Gaussian() : sigma_(1.0), length_(1.0) {}
    Gaussian(double sigma, double length) : sigma_(sigma), length_(length) {}
    // the operation order of the reference (products with the reciprocal 1 / length^2), so that host-side
    // evaluations agree with it bit for bit (tests/test_reference_pin.py)
    double compute(double &d) const { return (sigma_ * sigma_) * std::exp(-1 * d * (1.0 / (length_ * length_))); }
    double computediff(double &d) const { return -1 * (1.0 / (length_ * length_)) * compute(d); }
    double computediffdiff(double &) const { return 0.0; }
};
}  // namespace gp_regression
#endif
