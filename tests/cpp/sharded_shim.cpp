// A caller of the reference's shape against the header shim: create, the evaluate overloads, sampleSurface, update.  It knows
// nothing of devices; with GPX_DEVICES=0,0,0 (and GPX_SHARD_MIN_NQ) in the environment the shim cuts its large calls over three
// replicas.  Every result is printed as an exact checksum (the bit patterns summed as integers), so that two runs can be
// compared bit for bit by their text.  (tests/test_gpu_sharded_call.py::test_unchanged_caller_with_gpx_devices)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <gp_regression/gp_regressors.h>

using namespace gp_regression;

static uint64_t bits(const std::vector<double> &v)
{
    uint64_t s = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        uint64_t b;
        std::memcpy(&b, &v[i], 8);
        s += b * (uint64_t)(2 * i + 1);
    }
    return s;
}

int main()
{
    const size_t n = 420;
    Data::Ptr data = std::make_shared<Data>();
    for (size_t i = 0; i < n; ++i) {  // points of the unit sphere (label 0) and a few of the sphere of radius 2 (label 1)
        const bool ext = i % 28 == 27;
        const double t = std::acos(1.0 - 2.0 * (i + 0.5) / n), p = 2.399963229728653 * i, r = ext ? 2.0 : 1.0;
        data->coord_x.push_back(r * std::sin(t) * std::cos(p));
        data->coord_y.push_back(r * std::sin(t) * std::sin(p));
        data->coord_z.push_back(r * std::cos(t));
        data->label.push_back(ext ? 1.0 : 0.0);
        data->sigma2.push_back(0.05);
    }
    GaussianRegressor reg;  // GPRegressor<Gaussian>, gp_regressors.h
    reg.setCovFunction(std::make_shared<Gaussian>(1.0, 1.0));
    Model::Ptr gp;
    reg.create<false>(data, gp);
    const int g = 81;  // 531441 queries = 3 slices of the evaluate pipeline (two of 2^18, one of 7153): one per replica
    Data::Ptr q = std::make_shared<Data>();
    for (int i = 0; i < g; ++i)
        for (int j = 0; j < g; ++j)
            for (int k = 0; k < g; ++k) {
                q->coord_x.push_back(-1.2 + 2.4 * i / (g - 1));
                q->coord_y.push_back(-1.2 + 2.4 * j / (g - 1));
                q->coord_z.push_back(-1.2 + 2.4 * k / (g - 1));
            }
    std::vector<double> f, v, grad, tx, ty;
    reg.evaluate(gp, q, f);
    std::printf("sum f %llu\n", (unsigned long long)bits(f));
    reg.evaluate(gp, q, f, v);
    std::printf("sum fv %llu %llu\n", (unsigned long long)bits(f), (unsigned long long)bits(v));
    reg.evaluate(gp, q, f, v, grad, tx, ty);
    std::printf("sum all %llu %llu %llu %llu %llu\n", (unsigned long long)bits(f), (unsigned long long)bits(v),
                (unsigned long long)bits(grad), (unsigned long long)bits(tx), (unsigned long long)bits(ty));
    std::vector<size_t> idx;
    reg.sampleSurface(gp, q, 0.05, idx, f, v);
    uint64_t si = 0;
    for (size_t i = 0; i < idx.size(); ++i)
        si += idx[i] * (2 * i + 1);
    std::printf("sum surface %zu %llu %llu %llu\n", idx.size(), (unsigned long long)si, (unsigned long long)bits(f),
                (unsigned long long)bits(v));
    std::printf("shards %zu\n", gp->shards());
    // a small call stays on the model itself
    Data::Ptr one = std::make_shared<Data>();
    one->coord_x.push_back(0.3), one->coord_y.push_back(-0.2), one->coord_z.push_back(0.6);
    reg.evaluate(gp, one, f, v);
    std::printf("sum one %llu %llu\n", (unsigned long long)bits(f), (unsigned long long)bits(v));
    // update: the replicas of the old model are dropped and made again at the next large call
    Data::Ptr more = std::make_shared<Data>();
    more->coord_x = {0.0, 0.6}, more->coord_y = {0.0, 0.0}, more->coord_z = {1.0, 0.8};
    more->label = {0.0, 0.0}, more->sigma2 = {0.05, 0.05};
    reg.update<false>(more, gp);
    reg.evaluate(gp, q, f, v);
    std::printf("sum updated %llu %llu\n", (unsigned long long)bits(f), (unsigned long long)bits(v));
    return gp->size() == n + 2 ? 0 : 1;
}
