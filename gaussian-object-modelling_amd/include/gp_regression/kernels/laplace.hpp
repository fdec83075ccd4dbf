// gp_regression::Laplace -- interface of the reference's kernels/laplace.hpp:31-76.
// 2*sigma * exp(-d / length) (:37-42; the amplitude 2*sigma is the reference's),
// computediff = -(1/length) * compute (:44-49), computediffdiff = 0.
#ifndef GPX_SHIM_LAPLACE_HPP
#define GPX_SHIM_LAPLACE_HPP
#include <cmath>
namespace gp_regression
{
class Laplace
{
public:
    const double sigma_;
    const double length_;
    Laplace() : sigma_(1.0), length_(1.0) {}
    Laplace(double sigma, double length) : sigma_(sigma), length_(length) {}
    // the operation order of the reference (products with the reciprocal 1 / length): bit-identical host results
    double compute(double &d) const { return 2 * sigma_ * std::exp(-1 * d * (1.0 / length_)); }
    double computediff(double &d) const { return -1 * (1.0 / length_) * compute(d); }
    double computediffdiff(double &) const { return 0.0; }
};
}  // namespace gp_regression
#endif
