// Compile-only check of the Eigen-typed overloads of the header shim: the calls made by
// include/atlas/atlas_variance.hpp:78, include/atlas/atlas.hpp:259 and include/gp_regression/gp_projector.hpp:144 of the
// reference -- evaluate(gp, query, f, v, N [, Tx, Ty]) with Eigen::MatrixXd outputs and the free function
// computeTangentBasis on Eigen::Vector3d.  Built by tests/test_host.py only where <Eigen/Core> exists.
#include <Eigen/Core>
#include <Eigen/Dense>
#include <gp_regression/gp_regressors.h>

int eigen_overloads_compile(gp_regression::ThinPlateRegressor &reg, gp_regression::Model::ConstPtr gp,
                            gp_regression::Data::ConstPtr query)
{
    std::vector<double> f, v;
    Eigen::MatrixXd N, Tx, Ty;
    reg.evaluate(gp, query, f, v, N);          // gp_regressor.hpp:222-273
    reg.evaluate(gp, query, f, v, N, Tx, Ty);  // gp_regressor.hpp:194-212
    Eigen::Vector3d g = N.row(0), n, tx, ty;
    gp_regression::computeTangentBasis(g, n, tx, ty);  // gp_regressor.hpp:29-44
    return (int)(N.rows() + Tx.rows() + Ty.rows());
}
