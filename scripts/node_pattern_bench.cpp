// node_pattern_bench.cpp -- the reference node's sampling loop, verbatim in shape (src/gp_node.cpp:1025-1038 +
// :1066-1074): for every x slice of the 29^3 grid one std::thread PER GRID POINT, each calling
// reg_->evaluate(obj_gp, qq, ff, vv) with ONE query point; joined per slice.  Times that loop against the header
// shim + libgpx.so and compares with the two batched forms (one evaluate call, one sampleSurface call).
//   g++ -std=c++17 -O2 scripts/node_pattern_bench.cpp -I gaussian-object-modelling_amd/include -I include \
//       -L gaussian-object-modelling_amd/lib -lgpx -Wl,-rpath,$PWD/gaussian-object-modelling_amd/lib -Wl,-rpath-link,/opt/rocm/lib -lpthread
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <gp_regression/gp_regressors.h>
using namespace gp_regression;
typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); }

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    const bool json = argc > 2 && std::strcmp(argv[2], "--json") == 0;  // one JSON line on stdout (bench.py: configs.C1_node)
    long npts = gpx_pcd_read(argv[1], nullptr, 0);
    if (npts <= 0)
        return 3;
    std::vector<float> xyz(3 * (size_t)npts);
    gpx_pcd_read(argv[1], xyz.data(), (size_t)npts);
    Data::Ptr data_gp = std::make_shared<Data>();
    const size_t n = (size_t)npts + 15;
    data_gp->coord_x.resize(n), data_gp->coord_y.resize(n), data_gp->coord_z.resize(n), data_gp->label.resize(n), data_gp->sigma2.resize(n);
    gpx_node_training_set(xyz.data(), (size_t)npts, 0.1, 2.0, data_gp->coord_x.data(), data_gp->coord_y.data(),
                          data_gp->coord_z.data(), data_gp->label.data(), data_gp->sigma2.data());
    ThinPlateRegressor::Ptr reg_ = std::make_shared<ThinPlateRegressor>();
    reg_->setCovFunction(std::make_shared<ThinPlate>(2.0));
    Model::Ptr obj_gp;
    auto t0 = Clock::now();
    reg_->create<false>(data_gp, obj_gp);
    const double t_create_first = ms_since(t0);
    t0 = Clock::now();
    reg_->create<false>(data_gp, obj_gp);
    const double t_create = ms_since(t0);
    if (!json) {
        std::printf("create N=%zu: %.2f ms (first call: includes device initialisation)\n", n, t_create_first);
        std::printf("create N=%zu again: %.2f ms\n", n, t_create);
    }
    const double scale = 1.01, pass = 0.07;  // src/gp_node.cpp:635, sample_res default
    std::mutex mtx;
    size_t kept = 0, count = 0;
    double sum_v = 0;
    {   // warm-up: builds the inverse factor
        Data::Ptr qq = std::make_shared<Data>();
        qq->coord_x.push_back(0), qq->coord_y.push_back(0), qq->coord_z.push_back(0);
        std::vector<double> ff, vv;
        reg_->evaluate(obj_gp, qq, ff, vv);
    }
    t0 = Clock::now();
    for (double x = -scale; x <= scale; x += pass) {
        std::vector<std::thread> threads;
        for (double y = -scale; y <= scale; y += pass)
            for (double z = -scale; z <= scale; z += pass) {
                ++count;
                threads.emplace_back([&, x, y, z] {
                    Data::Ptr qq = std::make_shared<Data>();
                    qq->coord_x.push_back(x), qq->coord_y.push_back(y), qq->coord_z.push_back(z);
                    std::vector<double> ff, vv;
                    reg_->evaluate(obj_gp, qq, ff, vv);
                    if (std::fabs(ff.at(0)) <= 0.01) {
                        std::lock_guard<std::mutex> lk(mtx);
                        ++kept, sum_v += vv[0];
                    }
                });
            }
        for (auto &t : threads)
            t.join();
    }
    const double t_node = ms_since(t0);
    if (!json)
        std::printf("node pattern: %zu single-point evaluate(f,v) calls, one thread each: %.1f ms = %.1f us per call; %zu points with |f| <= 0.01 (mean v %.6f)\n",
                    count, t_node, t_node * 1e3 / count, kept, kept ? sum_v / kept : 0.0);
    // thread creation alone, for scale
    t0 = Clock::now();
    for (int s = 0; s < 29; ++s) {
        std::vector<std::thread> threads;
        for (int i = 0; i < 841; ++i)
            threads.emplace_back([] {});
        for (auto &t : threads)
            t.join();
    }
    const double t_threads = ms_since(t0);
    if (!json)
        std::printf("  (creating and joining the same %d empty threads: %.1f ms)\n", 29 * 841, t_threads);
    // the batched forms
    Data::Ptr all = std::make_shared<Data>();
    for (double x = -scale; x <= scale; x += pass)
        for (double y = -scale; y <= scale; y += pass)
            for (double z = -scale; z <= scale; z += pass)
                all->coord_x.push_back(x), all->coord_y.push_back(y), all->coord_z.push_back(z);
    std::vector<double> f, v;
    reg_->evaluate(obj_gp, all, f, v);
    t0 = Clock::now();
    reg_->evaluate(obj_gp, all, f, v);
    const double t_batched = ms_since(t0);
    if (!json)
        std::printf("one evaluate(f,v) call over all %zu points: %.2f ms\n", f.size(), t_batched);
    std::vector<size_t> idx;
    reg_->sampleSurface(obj_gp, all, 0.01, idx, f, v);
    t0 = Clock::now();
    reg_->sampleSurface(obj_gp, all, 0.01, idx, f, v);
    const double t_surface = ms_since(t0);
    if (!json)
        std::printf("one sampleSurface call (variance only for the %zu survivors): %.2f ms\n", idx.size(), t_surface);
    else
        std::printf("{\"n_train\": %zu, \"create_ms\": %.4f, \"calls\": %zu, \"node_ms\": %.3f, \"us_per_call\": %.3f, "
                    "\"threads_only_ms\": %.3f, \"batched_evaluate_ms\": %.4f, \"sample_surface_ms\": %.4f, \"survivors\": %zu, "
                    "\"kept_by_node_loop\": %zu, \"mean_v_kept\": %.9g}\n",
                    n, t_create, count, t_node, t_node * 1e3 / count, t_threads, t_batched, t_surface, idx.size(), kept,
                    kept ? sum_v / kept : 0.0);
    return idx.size() == kept ? 0 : 1;
}
