// gp_regression::Matern32 / Matern52 -- NEW kernels (the reference has them only as MATLAB closed
// forms, matlab_src/test_gp_regression_3Dsurf.m:117-123); Gaussian-shaped constructor.
//   Matern32: sigma^2 (1 + s) exp(-s),            s = sqrt(3) d / length
//   Matern52: sigma^2 (1 + s + s^2/3) exp(-s),    s = sqrt(5) d / length
// computediff is k'(d)/d (finite at d = 0), the thin-plate convention.
#ifndef GPX_SHIM_MATERN_HPP
#define GPX_SHIM_MATERN_HPP
#include <cmath>
namespace gp_regression
{
class Matern32
{
public:
    const double sigma_;
    const double length_;
    Matern32() : sigma_(1.0), length_(1.0) {}
    Matern32(double sigma, double length) : sigma_(sigma), length_(length) {}
    double compute(double &d) const
    {
        const double s = std::sqrt(3.0) * d / length_;
        return sigma_ * sigma_ * (1.0 + s) * std::exp(-s);
    }
    double computediff(double &d) const
    {
        const double s = std::sqrt(3.0) * d / length_;
        return -3.0 * sigma_ * sigma_ / (length_ * length_) * std::exp(-s);
    }
    double computediffdiff(double &) const { return 0.0; }
};
class Matern52
{
public:
    const double sigma_;
    const double length_;
    Matern52() : sigma_(1.0), length_(1.0) {}
    Matern52(double sigma, double length) : sigma_(sigma), length_(length) {}
    double compute(double &d) const
    {
        const double s = std::sqrt(5.0) * d / length_;
        return sigma_ * sigma_ * (1.0 + s + s * s / 3.0) * std::exp(-s);
    }
    double computediff(double &d) const
    {
        const double s = std::sqrt(5.0) * d / length_;
        return -(5.0 * sigma_ * sigma_ / (3.0 * length_ * length_)) * (1.0 + s) * std::exp(-s);
    }
    double computediffdiff(double &) const { return 0.0; }
};
}  // namespace gp_regression
#endif
