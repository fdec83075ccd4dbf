// gpx_host.cpp -- see gpx_host.hpp.  No HIP in this translation unit: it is also compiled by g++ with
// -fsanitize=thread / address,undefined for tests/cpp/host_concurrency.cpp.
#include "gpx_host.hpp"

#include <atomic>
#include <cmath>
#include <cstdlib>
#include <utility>

namespace gpxh {

namespace {
std::atomic<const DeviceBackend *> g_backend{nullptr};
}

void set_device_backend(const DeviceBackend *b) { g_backend.store(b, std::memory_order_release); }
const DeviceBackend *device_backend() { return g_backend.load(std::memory_order_acquire); }

int BigPool::alloc(void **p, size_t bytes)
{
    *p = nullptr;
    const DeviceBackend *be = device_backend();
    if (!be)
        return 2;
    const int dev = be->get_device();
    if (bytes >= BIG_POOL_MIN) {
        std::lock_guard<std::mutex> lk(mtx_);
        int best = -1;
        for (int i = 0; i < (int)free_.size(); ++i) {
            const Buf &b = free_[i];
            if (b.dev == dev && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 &&
                (best < 0 || b.bytes < free_[best].bytes))
                best = i;
        }
        if (best >= 0) {
            const Buf b = free_[best];
            free_.erase(free_.begin() + best);
            parked_ -= b.bytes;
            live_.push_back(b);
            *p = b.p;
            return 0;
        }
    }
    int rc = be->dev_malloc(p, bytes);
    if (rc == 1 && bytes >= BIG_POOL_MIN) {  // out of memory with buffers parked: release them and retry once
        trim();
        rc = be->dev_malloc(p, bytes);
    }
    if (rc != 0) {
        *p = nullptr;
        return rc;
    }
    if (bytes >= BIG_POOL_MIN) {
        std::lock_guard<std::mutex> lk(mtx_);
        live_.push_back(Buf{*p, bytes, dev});
    }
    return 0;
}

void BigPool::release(void *p)
{
    if (!p)
        return;
    const DeviceBackend *be = device_backend();
    bool park = false;
    {
        // decide and park under ONE lock: no window in which the cap can be overshot or the counter read torn
        std::lock_guard<std::mutex> lk(mtx_);
        Buf b{nullptr, 0, 0};
        for (size_t i = 0; i < live_.size(); ++i)
            if (live_[i].p == p) {
                b = live_[i];
                live_.erase(live_.begin() + i);
                break;
            }
        if (b.p && parked_ + b.bytes <= cap_ && free_.size() < BIG_POOL_MAX_PARKED) {
            free_.push_back(b);
            parked_ += b.bytes;
            park = true;
        }
    }
    if (!park && be)
        be->dev_free(p);
}

void BigPool::trim()
{
    std::vector<Buf> drop;
    {
        std::lock_guard<std::mutex> lk(mtx_);
        drop.swap(free_);
        parked_ = 0;
    }
    const DeviceBackend *be = device_backend();
    if (!be)
        return;
    const int prev = be->get_device();
    for (const Buf &b : drop) {
        be->set_device(b.dev);
        be->dev_free(b.p);
    }
    if (prev >= 0)
        be->set_device(prev);
}

size_t BigPool::parked_bytes()
{
    std::lock_guard<std::mutex> lk(mtx_);
    return parked_;
}

size_t BigPool::live_buffers()
{
    std::lock_guard<std::mutex> lk(mtx_);
    return live_.size();
}

void eigen_pivot_order(const std::vector<double> &diag, std::vector<int> &perm)
{
    const int n = (int)diag.size();
    perm.resize(n);
    for (int i = 0; i < n; ++i)
        perm[i] = i;
    bool uniform = true;
    for (int i = 1; i < n && uniform; ++i)
        uniform = std::fabs(diag[i]) == std::fabs(diag[0]);
    if (uniform)
        return;
    std::vector<double> d(diag);
    for (int k = 0; k < n; ++k) {
        int big = k;
        double bv = std::fabs(d[k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(d[i]) > bv) {
                bv = std::fabs(d[i]);
                big = i;
            }
        if (big != k) {
            std::swap(d[k], d[big]);
            std::swap(perm[k], perm[big]);
        }
    }
}

}  // namespace gpxh

// ---- switches ---------------------------------------------------------------------------------------------------------------
namespace gpxh {
namespace {
std::atomic<const Switches *> g_switches{nullptr};
std::mutex g_switches_mtx;

const Switches *parse_switches()
{
    Switches *s = new Switches();
    auto num = [](const char *name, long unset) {
        const char *e = std::getenv(name);
        return (e && *e) ? std::atol(e) : unset;
    };
    s->pool_mb = num("GPX_POOL_MB", -1);
    s->train_f64_max = (int)num("GPX_TRAIN_F64_MAX", -1);
    s->dataflow = (int)num("GPX_DATAFLOW", -1);
    s->wait_budget_us = num("GPX_WAIT_BUDGET_US", -1);
    s->update_append = (int)num("GPX_UPDATE_APPEND", -1);
    s->dgp_append = (int)num("GPX_DGP_APPEND", -1);
    s->var_cols = (int)num("GPX_VAR_COLS", -1);
    s->var_cols16 = (int)num("GPX_VAR_COLS16", -1);
    s->var_cols64 = (int)num("GPX_VAR_COLS64", -1);
    s->var_tile = (int)num("GPX_VAR_TILE", -1);
    s->var_fit = (int)num("GPX_VAR_FIT", -1);
    s->no_promote = (int)num("GPX_NO_PROMOTE", -1);
    s->small_eval = (int)num("GPX_SMALL_EVAL", -1);
    s->project_fused = (int)num("GPX_PROJECT_FUSED", -1);
    s->inv64 = (int)num("GPX_INV64", -1);
    return s;
}
}  // namespace

const Switches &switches()
{
    const Switches *s = g_switches.load(std::memory_order_acquire);
    if (!s) {
        std::lock_guard<std::mutex> lk(g_switches_mtx);
        s = g_switches.load(std::memory_order_acquire);
        if (!s) {
            s = parse_switches();
            g_switches.store(s, std::memory_order_release);
        }
    }
    return *s;
}

// (the previous block is leaked on purpose: a caller that fetched the reference a moment ago may still read it; a test hook,
// a few hundred bytes per call)
void switches_reload()
{
    std::lock_guard<std::mutex> lk(g_switches_mtx);
    g_switches.store(parse_switches(), std::memory_order_release);
}
}  // namespace gpxh
