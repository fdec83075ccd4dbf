// gp_regression::ThinPlate -- interface of the reference's kernels/thin_plate.hpp:9-43.
// 2d^3 - 3R d^2 + R^3 (:12-15); computediff = -6 (R - d), i.e. k'(d)/d (:17-20); computediffdiff = 0.
#ifndef GPX_SHIM_THIN_PLATE_HPP
#define GPX_SHIM_THIN_PLATE_HPP
#include <cmath>
namespace gp_regression
{
class ThinPlate
{
public:
    ThinPlate() : R_(1.0) {}
    explicit ThinPlate(double R) : R_(R) {}
    // the monomial form and operation order of the reference: bit-identical host results (the device evaluates
    // the factored form (d - R)^2 (2d + R), csrc/gpx_cov.hpp)
    double compute(double d) const { return 2 * d * d * d - 3 * R_ * d * d + R_ * R_ * R_; }
    double computediff(double d) const { return -6 * (R_ - d); }
    double computediffdiff(double) const { return 0.0; }
    double R() const { return R_; }  // accessor added for the GPU hand-off (the reference keeps R_ private)

private:
    double R_;
};
}  // namespace gp_regression
#endif
