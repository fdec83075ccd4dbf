// Drop-in for the reference's include/gp_regression/gp_regression_exception.h:9-17
// (same class name, constructor and what()); thrown by the header shim with the reference's
// message strings (gp_regressor.hpp:198, :231, :374, :566, :570).
#ifndef GPX_SHIM_GP_REGRESSION_EXCEPTION_H
#define GPX_SHIM_GP_REGRESSION_EXCEPTION_H

#include <exception>
#include <string>

namespace gp_regression
{
class GPRegressionException : public std::exception
{
public:
    explicit GPRegressionException(const std::string &message) : text_(message) {}
    ~GPRegressionException() noexcept override {}
    const char *what() const noexcept override { return text_.c_str(); }

private:
    std::string text_;
};
}  // namespace gp_regression
#endif
