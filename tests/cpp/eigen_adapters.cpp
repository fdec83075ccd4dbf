// Runs the Eigen-typed overloads of the header shim (reference call sites include/atlas/atlas_variance.hpp:78,
// include/atlas/atlas.hpp:259, include/gp_regression/gp_projector.hpp:144) against their std::vector twins: same f, v;
// N / Tx / Ty as n x 3 matrices holding the row-major vectors; computeTangentBasis on Vector3d == on double[3].
// Built by tests/test_gpu_shim.py against the real Eigen where it is installed, else against the interface stand-in
// in tests/cpp/eigen_iface (which is not Eigen and says so).
#include <cmath>
#include <cstdio>
#include <gp_regression/gp_regressors.h>
#ifndef GPX_SHIM_HAVE_EIGEN
#error "the shim did not see <Eigen/Core>"
#endif
using namespace gp_regression;

int main()
{
    Data::Ptr d = std::make_shared<Data>();
    for (int i = 0; i < 150; ++i) {
        const double t = 0.61803398875 * i, h = 1.0 - 2.0 * (i + 0.5) / 150.0, r = std::sqrt(1.0 - h * h);
        d->coord_x.push_back(r * std::cos(6.283185307 * t));
        d->coord_y.push_back(r * std::sin(6.283185307 * t));
        d->coord_z.push_back(h);
        d->label.push_back(0.0);
        d->sigma2.push_back(0.1);
    }
    d->coord_x.push_back(0), d->coord_y.push_back(0), d->coord_z.push_back(2.0), d->label.push_back(1.0), d->sigma2.push_back(0.1);
    ThinPlateRegressor reg;
    reg.setCovFunction(std::make_shared<ThinPlate>(4.0));
    Model::Ptr gp;
    reg.create<false>(d, gp);
    Data::Ptr q = std::make_shared<Data>();
    q->coord_x = {0.3, -0.8, 0.1}, q->coord_y = {0.2, 0.1, 0.9}, q->coord_z = {0.5, -0.4, 0.2};
    std::vector<double> f, v, fg, vg, g, tx, ty;
    Eigen::MatrixXd N, N2, Tx, Ty;
    reg.evaluate(gp, q, f, v, N);                  // gp_regressor.hpp:222-273
    reg.evaluate(gp, q, fg, vg, N2, Tx, Ty);       // :194-212
    reg.evaluate(gp, q, fg, vg, g, tx, ty);        // vector twins
    int bad = 0;
    bad += N.rows() != 3 || N.cols() != 3 || Tx.rows() != 3 || Ty.cols() != 3 || f.size() != 3 || f != fg || v != vg;
    for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c)
            bad += N(i, c) != g[3 * i + c] || N2(i, c) != g[3 * i + c] || Tx(i, c) != tx[3 * i + c] || Ty(i, c) != ty[3 * i + c];
    Eigen::Vector3d grad(g[0], g[1], g[2]), n, t1, t2;
    computeTangentBasis(grad, n, t1, t2);          // :29-44
    double n3[3], a3[3], b3[3];
    computeTangentBasis(&g[0], n3, a3, b3);
    for (int c = 0; c < 3; ++c)
        bad += n(c) != n3[c] || t1(c) != a3[c] || t2(c) != b3[c] || std::fabs(t1(c) - tx[c]) > 1e-9;
    std::printf(bad ? "eigen_adapters: %d MISMATCHES\n" : "eigen_adapters: OK\n", bad);
    return bad ? 1 : 0;
}
