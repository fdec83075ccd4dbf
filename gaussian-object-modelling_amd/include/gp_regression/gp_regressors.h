// Convenience regressors, as the reference's include/gp_regression/gp_regressors.h:11-18
// (default-constructed kernels), plus the two Matern ones.
#ifndef GPX_SHIM_GP_REGRESSORS_H
#define GPX_SHIM_GP_REGRESSORS_H
#include <gp_regression/gp_regressor.hpp>
namespace gp_regression
{
class GaussianRegressor : public GPRegressor<Gaussian>
{
};
class LaplaceRegressor : public GPRegressor<Laplace>
{
};
class ThinPlateRegressor : public GPRegressor<ThinPlate>
{
public:
    typedef std::shared_ptr<ThinPlateRegressor> Ptr;
    typedef std::shared_ptr<const ThinPlateRegressor> ConstPtr;
};
class Matern32Regressor : public GPRegressor<Matern32>
{
};
class Matern52Regressor : public GPRegressor<Matern52>
{
};
}  // namespace gp_regression
#endif
