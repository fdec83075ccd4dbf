// caller_shape.cpp -- a translation unit shaped like the reference's callers of gp_regression::
// (src/gp_node.cpp:898-922 computeGP, :1025-1038 + :1066-1074 one std::thread per grid point,
// include/atlas/atlas_variance.hpp:72-78 five-argument evaluate, src/gp_node.cpp:680-751 update),
// compiled against the header shim and linked with libgpx.so.  Writes what it computed to a text
// file that tests/test_gpu_shim.py compares with the CPU oracle.
//
// usage: caller_shape <file.pcd> <out.txt> <grid_per_axis>
#include <cmath>
#include <cstdio>
#include <mutex>
#include <thread>
#include <vector>

#include <gp_regression/gp_regressors.h>

using namespace gp_regression;

static int failures = 0;
#define EXPECT(cond, what)                                   \
    do {                                                     \
        if (!(cond)) {                                       \
            std::printf("FAIL: %s\n", what);                 \
            ++failures;                                      \
        }                                                    \
    } while (0)

template <typename Fn>
static std::string thrown(Fn fn)
{
    try {
        fn();
    } catch (const GPRegressionException &e) {
        return e.what();
    } catch (const std::exception &e) {
        return std::string("std::exception: ") + e.what();
    }
    return "";
}

int main(int argc, char **argv)
{
    if (argc < 4)
        return 2;
    const int grid = std::atoi(argv[3]);
    // ---- load + prepare data as the node does (loadPCDFile, deMeanAndNormalizeData, prepareExtData) ----
    long npts = gpx_pcd_read(argv[1], nullptr, 0);
    if (npts <= 0)
        return 3;
    std::vector<float> xyz(3 * (size_t)npts);
    gpx_pcd_read(argv[1], xyz.data(), (size_t)npts);
    const double sigma2 = 0.1, out_sphere_rad = 2.0;  // src/gp_node.cpp:16
    Data::Ptr data_gp = std::make_shared<Data>();
    const size_t n = (size_t)npts + 15;
    data_gp->coord_x.resize(n);
    data_gp->coord_y.resize(n);
    data_gp->coord_z.resize(n);
    data_gp->label.resize(n);
    data_gp->sigma2.resize(n);
    int next = gpx_node_training_set(xyz.data(), (size_t)npts, sigma2, out_sphere_rad, data_gp->coord_x.data(),
                                     data_gp->coord_y.data(), data_gp->coord_z.data(), data_gp->label.data(),
                                     data_gp->sigma2.data());
    EXPECT(next == 15, "15 exterior points");

    // ---- computeGP (src/gp_node.cpp:916-922) ----
    Model::Ptr obj_gp = std::make_shared<Model>();
    ThinPlateRegressor::Ptr reg_ = std::make_shared<ThinPlateRegressor>();
    std::shared_ptr<ThinPlate> my_kernel = std::make_shared<ThinPlate>(2.0);
    reg_->setCovFunction(my_kernel);
    const bool withoutNormals = false;
    reg_->create<withoutNormals>(data_gp, obj_gp);
    EXPECT(obj_gp->size() == n, "model size");
    EXPECT(obj_gp->R > 3.0 && obj_gp->R < 4.0, "R = max pairwise distance");

    // ---- fakeDeterministicSampling: one std::thread per grid point, joined per x-slab ----
    const double scale = 1.01;
    std::vector<double> gx, gy, gz, gf, gv;
    std::mutex mtx_samp;
    for (int ix = 0; ix < grid; ++ix) {
        std::vector<std::thread> threads;
        for (int iy = 0; iy < grid; ++iy)
            for (int iz = 0; iz < grid; ++iz) {
                const double x = -scale + 2 * scale * ix / (grid - 1), y = -scale + 2 * scale * iy / (grid - 1),
                             z = -scale + 2 * scale * iz / (grid - 1);
                threads.emplace_back([&, x, y, z]() {
                    Data::Ptr qq = std::make_shared<Data>();  // samplePoint, :1069-1074
                    qq->coord_x.push_back(x);
                    qq->coord_y.push_back(y);
                    qq->coord_z.push_back(z);
                    std::vector<double> ff, vv;
                    reg_->evaluate(obj_gp, qq, ff, vv);
                    std::lock_guard<std::mutex> lk(mtx_samp);
                    gx.push_back(x);
                    gy.push_back(y);
                    gz.push_back(z);
                    gf.push_back(ff.at(0));
                    gv.push_back(vv.at(0));
                });
            }
        for (auto &t : threads)
            t.join();
    }
    EXPECT((int)gf.size() == grid * grid * grid, "all grid points evaluated");

    // ---- the same sampling as ONE batched call: survivors of |f| <= tol, variance only for them ----
    {
        Data::Ptr all = std::make_shared<Data>();
        all->coord_x = gx;
        all->coord_y = gy;
        all->coord_z = gz;
        const double tol = 0.25;
        std::vector<size_t> idx;
        std::vector<double> sf, sv;
        reg_->sampleSurface(obj_gp, all, tol, idx, sf, sv);
        size_t expect = 0;
        for (size_t i = 0; i < gf.size(); ++i)
            if (std::fabs(gf[i]) <= tol)
                ++expect;
        EXPECT(idx.size() == expect && expect > 0, "sampleSurface keeps exactly the |f| <= tol points");
        bool same = idx.size() == sf.size() && idx.size() == sv.size();
        for (size_t k = 0; same && k < idx.size(); ++k)
            same = std::fabs(sf[k] - gf[idx[k]]) <= 1e-12 * (1 + std::fabs(gf[idx[k]])) &&
                   std::fabs(sv[k] - gv[idx[k]]) <= 1e-9 * (1 + std::fabs(gv[idx[k]])) && (k == 0 || idx[k] > idx[k - 1]);
        EXPECT(same, "sampleSurface values == per-point evaluate, ascending order");
    }

    // ---- atlas_variance.hpp:72-78 : f, v and the gradient at a chart centre ----
    Data::Ptr c = std::make_shared<Data>();
    c->coord_x = {0.3, data_gp->coord_x[5]};
    c->coord_y = {-0.2, data_gp->coord_y[5]};
    c->coord_z = {0.6, data_gp->coord_z[5]};
    std::vector<double> f, v, gg, tx, ty;
    reg_->evaluate(obj_gp, c, f, v, gg, tx, ty);
    EXPECT(f.size() == 2 && v.size() == 2 && gg.size() == 6 && tx.size() == 6 && ty.size() == 6, "output sizes");
    double N3[3], T1[3], T2[3];
    computeTangentBasis(&gg[0], N3, T1, T2);
    EXPECT(std::fabs(T1[0] - tx[0]) < 1e-9 && std::fabs(T2[2] - ty[2]) < 1e-9, "device basis == computeTangentBasis");
    std::vector<double> f1;
    reg_->evaluate(obj_gp, c, f1);
    EXPECT(f1.size() == 2 && std::fabs(f1[0] - f[0]) < 1e-12 * (1 + std::fabs(f[0])), "evaluate(f) == evaluate(f,v,...)");

    // ---- atlas_variance.hpp:122-124 : project the chosen sample along the chart gradient (batched form) ----
    {
        Data::Ptr proj = std::make_shared<Data>();
        std::vector<int> st;
        reg_->project(obj_gp, c, gg, proj, st, 1e-2, 1e-7, 80, 0.5);
        EXPECT(proj->coord_x.size() == 2 && st.size() == 2, "project output sizes");
        std::vector<double> fp;
        reg_->evaluate(obj_gp, proj, fp);
        EXPECT(st[0] == 1 && std::fabs(fp[0]) < 1e-2, "project reaches |f| < f_tol from a point off the surface");
        EXPECT(thrown([&] { reg_->project(obj_gp, c, std::vector<double>(3), proj, st); }) ==
                   "Input data vectors have different lengths", "project: normals length");
    }

    // ---- src/gp_node.cpp:258 : marchingSampling(false, 0.06, 0.02) -- the surface-following sampler, batched ----
    {
        Data::Ptr surf = std::make_shared<Data>();
        std::vector<double> mf, mv;
        const size_t cubes = reg_->marchSurface(obj_gp, nullptr, 0.15f, 0.05f, surf, mf, mv);
        bool ok = cubes > 10 && surf->coord_x.size() == mf.size() && mf.size() == mv.size() && mf.size() > 100;
        for (size_t i = 0; ok && i < mf.size(); ++i)
            ok = std::fabs(mf[i]) <= 0.01;
        EXPECT(ok, "marchSurface: every kept point has |f| <= 0.01");
        // a kept point, re-evaluated alone, gives the same mean and variance
        Data::Ptr one = std::make_shared<Data>();
        one->coord_x = {surf->coord_x[7]}, one->coord_y = {surf->coord_y[7]}, one->coord_z = {surf->coord_z[7]};
        std::vector<double> of, ov;
        reg_->evaluate(obj_gp, one, of, ov);
        EXPECT(std::fabs(of[0] - mf[7]) < 1e-12 && std::fabs(ov[0] - mv[7]) < 1e-9 * (1 + std::fabs(mv[7])), "marchSurface values");
    }

    // ---- read-only replicas on other devices (here: the same one), the in-process form of the sharded query grid ----
    {
        std::vector<Model::Ptr> reps = reg_->replicate(obj_gp, std::vector<int>{0});
        std::vector<double> rf, rv;
        reg_->evaluate(reps[0], c, rf, rv);
        EXPECT(reps.size() == 1 && rf == f && rv == v, "replica answers exactly like its source");
        EXPECT(thrown([&] { reg_->replicate(obj_gp, std::vector<int>{99}); }) == "device ordinal out of range", "replicate: bad device");
    }

    // ---- the four reference exceptions, verbatim (gp_regressor.hpp:198/:231/:374/:566/:570) ----
    Data::Ptr labelled = std::make_shared<Data>(*c);
    labelled->label = {0.0, 0.0};
    EXPECT(thrown([&] { reg_->evaluate(obj_gp, labelled, f); }) == "Query is already labeled!", "labelled query");
    EXPECT(thrown([&] { reg_->evaluate(Model::ConstPtr(), c, f); }) == "Empty Model pointer", "null model");
    EXPECT(thrown([&] { reg_->evaluate(obj_gp, Data::ConstPtr(), f); }) == "Empty data pointer", "null data");
    EXPECT(thrown([&] { reg_->evaluate(obj_gp, std::make_shared<Data>(), f); }) == "All input data is empty!",
           "empty data");
    EXPECT(thrown([&] { reg_->update<false>(c, Model::Ptr()); }) == "Empty model pointer", "update null model");
    {
        Model::Ptr tmp;
        EXPECT(thrown([&] { reg_->create<false>(std::make_shared<Data>(), tmp); }) == "All input data is empty!",
               "create on empty data");
    }

    // ---- update (gp_regressor.hpp:367-479): three touched points, label 0 ----
    const double R_before = obj_gp->R;
    Data::Ptr fresh = std::make_shared<Data>();
    fresh->coord_x = {0.05, -0.4, 0.7};
    fresh->coord_y = {0.9, 0.1, -0.3};
    fresh->coord_z = {-0.2, 0.85, 0.55};
    fresh->label = {0.0, 0.0, 0.0};
    fresh->sigma2 = {sigma2, sigma2, sigma2};
    reg_->update<false>(fresh, obj_gp);
    EXPECT(obj_gp->size() == n + 3, "update appended 3 points");
    {
        double Rnow = 0;
        gpx_model_get(obj_gp->handle(), GPX_FIELD_R, &Rnow, sizeof(Rnow));
        EXPECT(Rnow == R_before, "update keeps R (gp_regressor.hpp:454-455)");
    }
    std::vector<double> fu, vu;
    reg_->evaluate(obj_gp, c, fu, vu);

    FILE *out = std::fopen(argv[2], "w");
    if (!out)
        return 4;
    std::fprintf(out, "%zu %zu\n", n, gf.size());
    for (size_t i = 0; i < n; ++i)
        std::fprintf(out, "%.17g %.17g %.17g %.17g %.17g\n", data_gp->coord_x[i], data_gp->coord_y[i],
                     data_gp->coord_z[i], data_gp->label[i], data_gp->sigma2[i]);
    for (size_t i = 0; i < gf.size(); ++i)
        std::fprintf(out, "%.17g %.17g %.17g %.17g %.17g\n", gx[i], gy[i], gz[i], gf[i], gv[i]);
    std::fprintf(out, "%.17g %.17g %.17g %.17g\n", f[0], v[0], f[1], v[1]);
    std::fprintf(out, "%.17g %.17g %.17g %.17g %.17g %.17g\n", gg[0], gg[1], gg[2], gg[3], gg[4], gg[5]);
    std::fprintf(out, "%.17g %.17g %.17g %.17g\n", fu[0], vu[0], fu[1], vu[1]);
    std::fclose(out);
    std::printf(failures ? "caller_shape: %d FAILURES\n" : "caller_shape: OK\n", failures);
    return failures ? 1 : 0;
}
