// gpx_api.hip -- host side of libgpx.so: the C ABI of include/gpx.h over the HIP stages.
//
// Mirrors gp_regression::GPRegressor<Cov>::{create, evaluate x4, update} of the reference
// (include/gp_regression/gp_regressor.hpp:110-182, :194-357, :367-479) with the data on the GPU:
//
//   create   : permute (Eigen's diagonal-pivot rule, decided by the original diagonal) -> kbuild
//              -> blocked right-looking LDL^T (diag block + inverse | panel solve | MFMA trailing
//              update) -> alpha by block substitution + fp64 matrix-free residual refinement
//              [-> normals] [-> inverse factor X = L^-1 by recursive doubling on the GEMM core]
//   evaluate : mean/gradient kernel; variance v = k(0) - sum_j (X k)_j^2 / D_j per query batch
//              as one fused GEMM (Kqp tile built on the device, never the Nq x Nq matrix).
//
// There is no CPU compute path: without a HIP device every compute entry point fails.
//
// This file: error state, model life cycle (create / update / destroy / shell / state blobs / commit), accessors.
// gpx_build.hip: kernel matrix, factorisation, solves, inverse factor.  gpx_eval.hip: every prediction path.
#include <atomic>

#include "gpx_model.hpp"
#include "gpx_small.hpp"

// ------------------------------------------------------------------------------------------------
namespace gpxh {
thread_local std::string g_err;
int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
}  // namespace gpxh

// ---- pool of large device buffers: gpxh::BigPool (gpx_host.cpp) over the HIP backend ------------------------------
namespace gpxh {
namespace {
int hip_get_device()
{
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess) {
        (void)hipGetLastError();
        return -1;
    }
    return d;
}
void hip_set_device(int d) { (void)hipSetDevice(d); }
int hip_dev_malloc(void **p, size_t bytes)
{
    const hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess)
        return 0;
    (void)hipGetLastError();
    return e == hipErrorOutOfMemory ? 1 : 2;
}
void hip_dev_free(void *p) { (void)hipFree(p); }
const DeviceBackend g_hip_backend{hip_get_device, hip_set_device, hip_dev_malloc, hip_dev_free};
const bool g_backend_installed = (set_device_backend(&g_hip_backend), true);

BigPool &pool()
{
    static BigPool p([] {
        const long mb = gpxh::switches().pool_mb;
        return (size_t)(mb >= 0 ? mb : 16384) << 20;
    }());
    return p;
}
}  // namespace
struct KbuildScratch {  // gpx_dev_kbuild's per-device arg-max scratch
    float *tm = nullptr;
    int *tj = nullptr;
    int cap = 0;
};
std::mutex g_kb_mtx;
KbuildScratch g_kb_scratch[MAX_DEVICES];

hipError_t big_alloc(void **p, size_t bytes)
{
    (void)g_backend_installed;
    const int rc = pool().alloc(p, bytes);
    return rc == 0 ? hipSuccess : (rc == 1 ? hipErrorOutOfMemory : hipErrorUnknown);
}

void big_free(void *p) { pool().release(p); }

static std::mutex g_stream_mtx;
static std::vector<hipStream_t> g_stream_pool[MAX_DEVICES];
constexpr size_t STREAM_POOL_MAX = 64;  // per device

hipError_t stream_acquire(int device, hipStream_t *s)
{
    if (device >= 0 && device < MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_stream_mtx);
        std::vector<hipStream_t> &v = g_stream_pool[device];
        if (!v.empty()) {
            *s = v.back();
            v.pop_back();
            return hipSuccess;
        }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}

void stream_release(int device, hipStream_t s)
{
    if (!s)
        return;
    (void)hipStreamSynchronize(s);
    if (device >= 0 && device < MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_stream_mtx);
        std::vector<hipStream_t> &v = g_stream_pool[device];
        if (v.size() < STREAM_POOL_MAX) {
            v.push_back(s);
            return;
        }
    }
    (void)hipStreamDestroy(s);
}

// Pinned host blocks for the staging of a small model's create (one host-to-device and one device-to-host copy per
// model): hipHostMalloc / hipHostFree cost more than the device work of such a create, so the blocks are recycled.
static std::mutex g_pin_mtx;
static std::vector<std::pair<void *, size_t>> g_pin_pool;
constexpr size_t PIN_POOL_MAX = 64;
static std::vector<std::pair<void *, size_t>> g_pin_live;

hipError_t pinned_acquire(size_t bytes, void **p)
{
    const size_t want = (bytes + 65535) / 65536 * 65536;
    {
        std::lock_guard<std::mutex> lk(g_pin_mtx);
        for (size_t i = 0; i < g_pin_pool.size(); ++i)
            if (g_pin_pool[i].second >= want) {
                *p = g_pin_pool[i].first;
                g_pin_live.push_back(g_pin_pool[i]);
                g_pin_pool.erase(g_pin_pool.begin() + i);
                return hipSuccess;
            }
    }
    const hipError_t e = hipHostMalloc(p, want, hipHostMallocDefault);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pin_mtx);
        g_pin_live.push_back({*p, want});
    }
    return e;
}

void pinned_release(void *p)
{
    if (!p)
        return;
    std::pair<void *, size_t> blk{nullptr, 0};
    {
        std::lock_guard<std::mutex> lk(g_pin_mtx);
        for (size_t i = 0; i < g_pin_live.size(); ++i)
            if (g_pin_live[i].first == p) {
                blk = g_pin_live[i];
                g_pin_live.erase(g_pin_live.begin() + i);
                break;
            }
        if (blk.first && g_pin_pool.size() < PIN_POOL_MAX) {
            g_pin_pool.push_back(blk);
            return;
        }
    }
    (void)hipHostFree(p);
}

static void pinned_trim()
{
    std::vector<std::pair<void *, size_t>> v;
    {
        std::lock_guard<std::mutex> lk(g_pin_mtx);
        v.swap(g_pin_pool);
    }
    for (auto &b : v)
        (void)hipHostFree(b.first);
}
}  // namespace gpxh

namespace gpxh {
long long wait_budget_ticks(int npad)
{
    const long us = gpxh::switches().wait_budget_us;
    return 100LL * (us >= 0 ? us : (npad <= 8192 ? 20000L : 200000L));
}
int device_cu_count(int dev)
{
    static std::atomic<int> cus[MAX_DEVICES];
    if (dev < 0 || dev >= MAX_DEVICES)
        return 0;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) {
            (void)hipGetLastError();
            return 0;
        }
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}
}  // namespace gpxh

extern "C" void gpx_debug_reload(void) { gpxh::switches_reload(); }

extern "C" void gpx_trim(void)
{
    gpxh::pool().trim();
    gpxh::pinned_trim();
    int prev = -1;
    (void)hipGetDevice(&prev);
    {
        std::lock_guard<std::mutex> lk(gpxh::g_stream_mtx);
        for (int d = 0; d < MAX_DEVICES; ++d) {
            if (gpxh::g_stream_pool[d].empty())
                continue;
            (void)hipSetDevice(d);
            for (hipStream_t s : gpxh::g_stream_pool[d])
                (void)hipStreamDestroy(s);
            gpxh::g_stream_pool[d].clear();
        }
    }
    {
        std::lock_guard<std::mutex> lk(gpxh::g_kb_mtx);
        for (int d = 0; d < MAX_DEVICES; ++d) {
            gpxh::KbuildScratch &ks = gpxh::g_kb_scratch[d];
            if (!ks.tm && !ks.tj)
                continue;
            (void)hipSetDevice(d);
            (void)hipDeviceSynchronize();
            (void)hipFree(ks.tm);
            (void)hipFree(ks.tj);
            ks = gpxh::KbuildScratch{};
        }
    }
    if (prev >= 0)
        (void)hipSetDevice(prev);
}

extern "C" const char *gpx_last_error(void) { return g_err.c_str(); }
extern "C" const char *gpx_version(void) { return "gpx 0.1 (gfx950)"; }
extern "C" int gpx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}
extern "C" size_t gpx_padded_n(size_t n) { return ((n + PANEL - 1) / PANEL) * PANEL; }

namespace gpx {
CovHost make_cov(const gpx_kernel &k)
{
    CovHost c{};
    c.id = k.id;
    const double p0 = k.p[0], p1 = k.p[1];
    switch (k.id) {
    case GPX_KERNEL_GAUSSIAN:  // sigma^2 exp(-d / l^2)   kernels/gaussian.hpp:15-20,:36-42
        c.a = p0 * p0;
        c.s = 1.0 / (p1 * p1);
        c.k0 = c.a;
        break;
    case GPX_KERNEL_LAPLACE:  // 2 sigma exp(-d / l)     kernels/laplace.hpp:37-42
        c.a = 2 * p0;
        c.s = 1.0 / p1;
        c.k0 = c.a;
        break;
    case GPX_KERNEL_THINPLATE:  // 2d^3 - 3R d^2 + R^3   kernels/thin_plate.hpp:12-15
        c.R = p0;
        c.R3 = p0 * p0 * p0;
        c.k0 = c.R3;
        break;
    case GPX_KERNEL_MATERN32:
        c.a = p0 * p0;
        c.s = std::sqrt(3.0) / p1;
        c.k0 = c.a;
        break;
    case GPX_KERNEL_MATERN52:
        c.a = p0 * p0;
        c.s = std::sqrt(5.0) / p1;
        c.k0 = c.a;
        break;
    }
    return c;
}
}  // namespace gpx

extern "C" void gpx_model_destroy(gpx_model *m)
{
    if (!m)
        return;
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(m->device);
    if (m->stream)
        (void)hipStreamSynchronize(m->stream);
    free_dev(m);
    for (auto &e : m->ev)
        if (e)
            (void)hipEventDestroy(e);
    for (auto &e : m->la_ev)
        (void)hipEventDestroy(e);
    gpxh::stream_release(m->device, m->stream2);
    gpxh::stream_release(m->device, m->stream);
    if (prev >= 0)
        (void)hipSetDevice(prev);
    delete m;
}

static int check_opts(const gpx_options *opt, gpx_options &o)
{
    std::memset(&o, 0, sizeof(o));
    o.device = -1;
    o.ir_steps = -1;
    if (opt)
        o = *opt;
    if (o.precision < GPX_PREC_F32 || o.precision > GPX_PREC_F32_SPLIT)
        return fail(GPX_E_BAD_ARG, "options.precision must be GPX_PREC_F32, _F64, _MIXED or _F32_SPLIT");
    if (o.query_batch < 0 || (o.query_batch % TILE) != 0)
        return fail(GPX_E_BAD_ARG, "options.query_batch must be a non-negative multiple of 128");
    return GPX_OK;
}

static int new_model(const gpx_kernel *kernel, size_t n, const gpx_options &o, gpx_model **out)
{
    if (kernel->id < GPX_KERNEL_GAUSSIAN || kernel->id > GPX_KERNEL_MATERN52)
        return fail(GPX_E_BAD_ARG, "unknown kernel id");
    int ndev = gpx_device_count();
    if (ndev <= 0)
        return fail(GPX_E_NO_DEVICE, "no HIP device available (libgpx has no CPU path)");
    int dev = o.device;
    if (dev < 0)
        HIPCHK(hipGetDevice(&dev));
    if (dev >= ndev)
        return fail(GPX_E_BAD_ARG, "options.device out of range");
    HIPCHK(hipSetDevice(dev));
    gpx_model *m = new gpx_model();
    m->device = dev;
    m->kern = *kernel;
    m->cov = make_cov(*kernel);
    m->opt = o;
    m->n = (int)n;
    m->npad = (int)gpx_padded_n(n);
    m->nblk = m->npad / TILE;
    set_training_precision(m);  // MIXED and small F32 models train in fp64
    set_query_batch(m);
    if (gpxh::switches().inv64 >= 0)
        m->inv64 = gpxh::switches().inv64 != 0;
    // fp32 variance contraction: take the per-query fit out of the kernel operand (GPX_VAR_FIT=0: plain kernel values)
    m->var_fit = o.precision != GPX_PREC_F64;
    if (gpxh::switches().var_fit >= 0)
        m->var_fit = m->var_fit && gpxh::switches().var_fit != 0;
    // ... and form that operand, k - fit, in fp64 from the fp64 points, rounding once -- for the thin plate, whose
    // values are all of the size of k(0) = R^3 while the variance is ~k(0)/60 at N = 16384: forming k and the fit
    // separately in fp32 costs 8e-6 of max|v| there, against 4e-7 for Matern-5/2 (profiles/r03_fit_variants_*.txt), and the
    // exponential kernels would pay ~40 % more operand time for the fp64 exp.
    m->op64 = kernel->id == GPX_KERNEL_THINPLATE;
    m->var_fit_opt = m->var_fit;
    if (gpxh::stream_acquire(m->device, &m->stream) != hipSuccess) {
        delete m;
        return fail(GPX_E_HIP, "hipStreamCreate failed");
    }
    for (auto &e : m->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            gpx_model_destroy(m);
            return fail(GPX_E_HIP, "hipEventCreate failed");
        }
    *out = m;
    return GPX_OK;
}

static int validate_train(size_t n, const double *x, const double *y, const double *z, const double *label)
{
    if (!x || !y || !z || !label)
        return n == 0 ? fail(GPX_E_EMPTY, "All input data is empty!") : fail(GPX_E_NULL, "Empty data pointer");
    if (n == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    return GPX_OK;
}

static int check_finite(size_t n, const double *v, const char *what)
{
    if (!v)
        return GPX_OK;
    for (size_t i = 0; i < n; ++i)
        if (!std::isfinite(v[i]))
            return fail(GPX_E_NAN_INPUT, std::string("non-finite value in ") + what);
    return GPX_OK;
}

extern "C" int gpx_model_create(const gpx_kernel *kernel, size_t n, const double *x, const double *y,
                                const double *z, const double *label, const double *sigma2,
                                const gpx_options *opt, gpx_model **out)
{
    if (!out)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!kernel)
        return fail(GPX_E_NULL, "Empty kernel pointer");
    int rc = validate_train(n, x, y, z, label);
    if (rc)
        return rc;
    if (n > (size_t)1 << 20)
        return fail(GPX_E_BAD_ARG, "n too large");
    if ((rc = check_finite(n, x, "coord_x")) || (rc = check_finite(n, y, "coord_y")) ||
        (rc = check_finite(n, z, "coord_z")) || (rc = check_finite(n, label, "label")) ||
        (rc = check_finite(n, sigma2, "sigma2")))
        return rc;
    gpx_options o;
    if ((rc = check_opts(opt, o)))
        return rc;
    gpx_model *m = nullptr;
    if ((rc = new_model(kernel, n, o, &m)))
        return rc;
    m->hx.assign(x, x + n);
    m->hy.assign(y, y + n);
    m->hz.assign(z, z + n);
    m->hlabel.assign(label, label + n);
    m->has_s2 = sigma2 != nullptr;
    if (sigma2)
        m->hs2.assign(sigma2, sigma2 + n);
    else
        m->hs2.assign(n, 0.0);
    rc = build_model(m);
    if (rc) {
        std::string keep = g_err;
        gpx_model_destroy(m);
        g_err = keep;
        return rc;
    }
    *out = m;  // the C++ shim's shared_ptr reset releases any previous model (gp_regressor.hpp:116-117)
    return GPX_OK;
}

extern "C" int gpx_model_update(gpx_model *m, size_t n_new, const double *x, const double *y, const double *z,
                                const double *label, const double *sigma2)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty model pointer");
    int rc = validate_train(n_new, x, y, z, label);
    if (rc)
        return rc;
    if ((rc = check_finite(n_new, x, "coord_x")) || (rc = check_finite(n_new, y, "coord_y")) ||
        (rc = check_finite(n_new, z, "coord_z")) || (rc = check_finite(n_new, label, "label")) ||
        (rc = check_finite(n_new, sigma2, "sigma2")))
        return rc;
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    const double keepR = m->R;  // update() does not refresh R (gp_regressor.hpp:454-455)
    const int n_old = m->n;
    m->hx.insert(m->hx.end(), x, x + n_new);
    m->hy.insert(m->hy.end(), y, y + n_new);
    m->hz.insert(m->hz.end(), z, z + n_new);
    m->hlabel.insert(m->hlabel.end(), label, label + n_new);
    if (sigma2) {
        m->hs2.insert(m->hs2.end(), sigma2, sigma2 + n_new);
        m->has_s2 = true;
    } else {
        m->hs2.insert(m->hs2.end(), n_new, 0.0);
    }
    // Rank-n append instead of the reference's refactorisation from scratch (:457-459) when the old
    // factor is still there in the training precision and the pivot order of the grown set (Eigen's rule on the
    // diagonal k(0) + sigma2) keeps the old points first, in their old order.  Results agree with a rebuild to
    // rounding; anything else falls back to the rebuild.
    kept_factor keep;
    bool append_on = gpxh::switches().update_append != 0;  // GPX_UPDATE_APPEND=0: always rebuild (tests compare the two)
    {
        // A grown model that is still in the small-model range is rebuilt by the three dataflow launches (gpx_small.hip):
        // 0.25 ms at N = 277 + 16 against 0.62 ms for the append's launch chain (scripts/update_bench.py) -- and a rebuild is
        // what the reference does (:457-459)
        const bool small_on = gpxh::switches().dataflow != 0;
        if (small_on && gpx_padded_n(m->hx.size()) <= (size_t)SMALL_CREATE_MAX_NP)
            append_on = false;
    }
    if (append_on && m->ready && m->Kmat && m->linv && !m->x_packed && !m->train64 && n_old >= TILE) {
        const int n_tot = (int)m->hx.size();
        std::vector<double> diag(n_tot);
        for (int i = 0; i < n_tot; ++i)
            diag[i] = m->cov.k0 + (m->has_s2 ? m->hs2[i] : 0.0);
        std::vector<int> perm_new;
        eigen_pivot_order(diag, perm_new);
        bool prefix = true;
        for (int k = 0; k < n_old && prefix; ++k)
            prefix = perm_new[k] == m->perm[k];
        if (prefix) {
            const int t0 = n_old / TILE * TILE;
            const size_t e = m->esz;
            keep.t0 = t0;
            keep.np_old = m->npad;
            keep.K = m->Kmat, keep.linv = m->linv;  // detached: free_dev must not release them
            m->Kmat = m->linv = nullptr;
            if (hipMalloc(&keep.d, e * t0) != hipSuccess || hipMalloc(&keep.dinv, e * t0) != hipSuccess ||
                hipMemcpy(keep.d, m->t_d, e * t0, hipMemcpyDeviceToDevice) != hipSuccess ||
                hipMemcpy(keep.dinv, m->t_dinv, e * t0, hipMemcpyDeviceToDevice) != hipSuccess) {
                (void)hipGetLastError();
                keep.release();  // out of memory for the carry-over: rebuild instead
                keep = kept_factor{};
            } else {
                std::vector<char> hd(e * t0);
                HIPCHK(hipMemcpy(hd.data(), keep.d, e * t0, hipMemcpyDeviceToHost));
                for (int i = 0; i < t0; ++i)
                    keep.n_neg += (e == 8 ? ((const double *)hd.data())[i] : (double)((const float *)hd.data())[i]) < 0.0;
                if (m->has_inverse && m->X) {  // the inverse factor grows by the new rows too (gpx_build.hip)
                    keep.X = m->X;
                    m->X = nullptr;
                }
            }
        }
    }
    free_dev(m);
    m->ready = m->has_inverse = m->has_normals = false;
    m->hD.clear();
    m->x_packed = false;
    m->n = (int)m->hx.size();
    m->npad = (int)gpx_padded_n(m->n);
    m->nblk = m->npad / TILE;
    set_training_precision(m);
    set_query_batch(m);
    rc = build_model(m, keep.t0 > 0 ? &keep : nullptr);  // :457-459 refactors from scratch; same results
    keep.release();
    m->R = keepR;
    return rc;
}

extern "C" int gpx_model_get(const gpx_model *cm, int field, void *dst, size_t bytes)
{
    if (!cm)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!dst)
        return fail(GPX_E_NULL, "Empty output pointer");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    const size_t n = (size_t)m->n;
    auto need = [&](size_t b) { return bytes >= b ? GPX_OK : fail(GPX_E_SIZE_MISMATCH, "destination too small"); };
    int rc;
    switch (field) {
    case GPX_FIELD_N: {
        if ((rc = need(sizeof(int64_t))))
            return rc;
        *(int64_t *)dst = (int64_t)n;
        return GPX_OK;
    }
    case GPX_FIELD_R:
        if ((rc = need(sizeof(double))))
            return rc;
        *(double *)dst = m->R;
        return GPX_OK;
    case GPX_FIELD_P: {
        if ((rc = need(sizeof(double) * 3 * n)))
            return rc;
        double *o = (double *)dst;
        for (size_t i = 0; i < n; ++i) {
            o[3 * i] = m->hx[i];
            o[3 * i + 1] = m->hy[i];
            o[3 * i + 2] = m->hz[i];
        }
        return GPX_OK;
    }
    case GPX_FIELD_Y:
        if ((rc = need(sizeof(double) * n)))
            return rc;
        std::memcpy(dst, m->hlabel.data(), sizeof(double) * n);
        return GPX_OK;
    case GPX_FIELD_S2:
        if ((rc = need(sizeof(double) * n)))
            return rc;
        std::memcpy(dst, m->hs2.data(), sizeof(double) * n);
        return GPX_OK;
    case GPX_FIELD_PERM:
        if ((rc = need(sizeof(int32_t) * n)))
            return rc;
        std::memcpy(dst, m->perm.data(), sizeof(int32_t) * n);
        return GPX_OK;
    case GPX_FIELD_STATS:
        if ((rc = need(sizeof(gpx_stats))))
            return rc;
        HIPCHK(hipSetDevice(m->device));
        resolve_eval_stats(m);
        std::memcpy(dst, &m->stats, sizeof(gpx_stats));
        return GPX_OK;
    default:
        break;
    }
    if (!m->ready)
        return fail(GPX_E_STATE, "model is not ready");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (field == GPX_FIELD_ALPHA) {
        if ((rc = need(sizeof(double) * n)))
            return rc;
        std::vector<double> a(n);
        HIPCHK(hipMemcpy(a.data(), m->d_alpha, sizeof(double) * n, hipMemcpyDeviceToHost));
        double *o = (double *)dst;
        for (size_t k = 0; k < n; ++k)
            o[m->perm[k]] = a[k];
        return GPX_OK;
    }
    if (field == GPX_FIELD_D) {
        if ((rc = need(sizeof(double) * n)))
            return rc;
        double *o = (double *)dst;
        if (!m->hD.empty()) {
            std::memcpy(o, m->hD.data(), sizeof(double) * n);
        } else if (m->prec == GPX_PREC_F64) {
            HIPCHK(hipMemcpy(o, m->t_d, sizeof(double) * n, hipMemcpyDeviceToHost));
        } else {
            std::vector<float> t(n);
            HIPCHK(hipMemcpy(t.data(), m->t_d, sizeof(float) * n, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; ++k)
                o[k] = t[k];
        }
        return GPX_OK;
    }
    if (field == GPX_FIELD_NORMALS) {
        if (!m->has_normals)
            return fail(GPX_E_STATE, "model was created without normals");
        if ((rc = need(sizeof(double) * 3 * n)))
            return rc;
        std::vector<double> g(3 * n);
        HIPCHK(hipMemcpy(g.data(), m->d_normals, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
        double *o = (double *)dst;
        for (size_t k = 0; k < n; ++k)
            for (int c = 0; c < 3; ++c)
                o[3 * (size_t)m->perm[k] + c] = g[3 * k + c];
        return GPX_OK;
    }
    if (field == GPX_FIELD_KPP) {
        // rebuilt on demand in caller order: kbuild on the un-permuted points, mirrored on the host
        if ((rc = need(sizeof(double) * n * n)))
            return rc;
        const int np = m->npad;
        const size_t e = m->esz;
        std::vector<double> st((size_t)np * 4, 0.0);
        for (size_t i = 0; i < n; ++i) {  // relative to the centre, like the model's own working-precision points
            st[i] = m->hx[i] - m->cen[0];
            st[np + i] = m->hy[i] - m->cen[1];
            st[2 * (size_t)np + i] = m->hz[i] - m->cen[2];
            st[3 * (size_t)np + i] = m->hs2[i];
        }
        const int nt = np / TILE, ntiles = nt * (nt + 1) / 2;
        DevGuard gdd, gtt, gKt, gtm, gtj;  // freed on every exit (HIPCHK returns early)
        HIPCHK(hipMalloc(&gdd.p, sizeof(double) * 4 * np));
        HIPCHK(hipMalloc(&gtt.p, e * 4 * np));
        HIPCHK(hipMalloc(&gKt.p, e * (size_t)np * np));
        HIPCHK(hipMalloc(&gtm.p, sizeof(float) * ntiles));
        HIPCHK(hipMalloc(&gtj.p, sizeof(int) * 2 * ntiles));
        double *dd = (double *)gdd.p;
        void *tt = gtt.p, *Kt = gKt.p;
        float *tm = (float *)gtm.p;
        int *tj = (int *)gtj.p;
        HIPCHK(hipMemcpy(dd, st.data(), sizeof(double) * 4 * np, hipMemcpyHostToDevice));
        char *tb = (char *)tt;
        for (int c = 0; c < 4; ++c)
            launch_cast_vec(m->prec, np, np, dd + (size_t)c * np, tb + (size_t)c * np * e, m->stream);
        launch_kbuild(m->prec, m->cov, (int)n, np, tb, tb + np * e, tb + 2 * np * e, tb + 3 * np * e, Kt, tm, tj,
                      m->stream);
        HIPCHK(hipStreamSynchronize(m->stream));
        std::vector<char> hk(e * (size_t)np * np);
        HIPCHK(hipMemcpy(hk.data(), Kt, hk.size(), hipMemcpyDeviceToHost));
        double *o = (double *)dst;
        for (size_t i = 0; i < n; ++i)
            for (size_t j = 0; j <= i; ++j) {
                // tiles strictly above the block diagonal are not written: read the lower element
                double val = m->prec == GPX_PREC_F64 ? ((double *)hk.data())[i * np + j]
                                                     : (double)((float *)hk.data())[i * np + j];
                o[i * n + j] = val;
                o[j * n + i] = val;
            }
        return GPX_OK;
    }
    return fail(GPX_E_BAD_ARG, "unknown field");
}

// ---- sharded query grid: shell / blob / commit ---------------------------------------------------
extern "C" int gpx_model_create_shell(const gpx_kernel *kernel, size_t n, const gpx_options *opt, gpx_model **out)
{
    if (!out)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!kernel)
        return fail(GPX_E_NULL, "Empty kernel pointer");
    if (n == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    gpx_options o;
    int rc = check_opts(opt, o);
    if (rc)
        return rc;
    gpx_model *m = nullptr;
    if ((rc = new_model(kernel, n, o, &m)))
        return rc;
    if (m->train64) {  // the committed state of a model trained in fp64 (MIXED, small F32) is the fp32 layout
        m->prec = GPX_PREC_F32;
        m->esz = 4;
    }
    rc = alloc_model(m);
    if (rc == GPX_OK) {
        hipError_t e = big_alloc(&m->X, m->esz * (size_t)m->npad * m->npad);
        if (e != hipSuccess)
            rc = fail(e == hipErrorOutOfMemory ? GPX_E_OOM : GPX_E_HIP, hipGetErrorString(e));
    }
    if (rc) {
        std::string keep = g_err;
        gpx_model_destroy(m);
        g_err = keep;
        return rc;
    }
    m->perm.resize(n);
    for (size_t i = 0; i < n; ++i)
        m->perm[i] = (int)i;
    m->hx.assign(n, 0.0);
    m->hy.assign(n, 0.0);
    m->hz.assign(n, 0.0);
    m->hlabel.assign(n, 0.0);
    m->hs2.assign(n, 0.0);
    m->stats.n = (int64_t)n;
    m->stats.n_padded = m->npad;
    *out = m;
    return GPX_OK;
}

extern "C" int gpx_model_state_blob(gpx_model *m, int part, void **d_ptr, size_t *bytes)
{
    if (!m || !d_ptr || !bytes)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (m->promoted)
        return fail(GPX_E_STATE, "the kernel matrix is indefinite and the model kept its fp64 state: its blobs have the "
                                 "GPX_PREC_F64 layout -- create the source (and the shells) with GPX_PREC_F64");
    if (part == 0) {
        *d_ptr = m->blob0;
        *bytes = m->blob0_bytes;
        return GPX_OK;
    }
    if (part == 1) {
        if (!m->X)
            return fail(GPX_E_STATE, "inverse factor not built (call gpx_model_prepare_variance)");
        *d_ptr = m->X;
        *bytes = m->esz * (size_t)m->npad * m->npad;
        return GPX_OK;
    }
    return fail(GPX_E_BAD_ARG, "part must be 0 (vectors) or 1 (inverse factor)");
}

extern "C" int gpx_model_commit(gpx_model *m, int with_variance)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!m->blob0)
        return fail(GPX_E_STATE, "model has no device state");
    {  // the centre of the cloud travels in the meta block of state part 0
        int prev = -1;
        (void)hipGetDevice(&prev);
        HIPCHK(hipSetDevice(m->device));
        const hipError_t e = hipMemcpy(m->cen, m->d_meta, sizeof(double) * 3, hipMemcpyDeviceToHost);
        if (prev >= 0)
            (void)hipSetDevice(prev);
        if (e != hipSuccess)
            return fail(GPX_E_HIP, std::string("commit: ") + hipGetErrorString(e));
    }
    m->ready = true;
    if (with_variance) {
        if (!m->X)
            return fail(GPX_E_STATE, "inverse factor buffer missing");
        m->has_inverse = true;
        if (m->opt.precision == GPX_PREC_F32_SPLIT) {
            set_split_scale(m);
            if (split_packs(m))  // the received blobs are already packed / scaled
                m->x_packed = true;
        }
    }
    return GPX_OK;
}

// ---- in-library multi-device placement (SURVEY 8b / north star: "host code stays C++") ---------------------------
// A C++ caller shaped like src/gp_node.cpp (one process, many host threads sharing one model, :1025-1038) puts
// read-only replicas of a trained model on other GPUs of the node and shards its query grid over them: the source's
// state (points, alpha, 1/D, cloud moments, correction vectors; the inverse factor) is copied device to device --
// hipMemcpyPeerAsync, i.e. xGMI when peer access is available, staged through the host otherwise -- into shells and
// committed.  No RCCL communicator is needed inside one process; the one-process-per-GPU path (bench.py --mode
// shard) moves the same two blobs with a torch.distributed broadcast.
namespace {
struct DeviceRestore {  // the caller's current device survives every exit path
    int prev = -1;
    DeviceRestore() { (void)hipGetDevice(&prev); }
    ~DeviceRestore()
    {
        if (prev >= 0)
            (void)hipSetDevice(prev);
    }
};
}  // namespace

extern "C" int gpx_model_replicate(const gpx_model *csrc, int ndev, const int *devs, gpx_model **out)
{
    if (!csrc)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!devs || !out)
        return fail(GPX_E_NULL, "Empty output pointer");
    if (ndev <= 0)
        return fail(GPX_E_BAD_ARG, "ndev must be positive");
    gpx_model *src = const_cast<gpx_model *>(csrc);
    if (!src->ready)
        return fail(GPX_E_STATE, "model is not ready");
    const int have = gpx_device_count();
    for (int i = 0; i < ndev; ++i) {
        out[i] = nullptr;
        if (devs[i] < 0 || devs[i] >= have)
            return fail(GPX_E_BAD_ARG, "device ordinal out of range");
    }
    DeviceRestore restore;
    std::vector<hipEvent_t> done((size_t)ndev, nullptr);
    int rc = GPX_OK;
    auto cleanup = [&](int code) {
        const std::string keep = g_err;
        for (int i = 0; i < ndev; ++i) {
            if (done[i]) {
                (void)hipEventSynchronize(done[i]);  // no copy may still be writing into a shell that is about to go
                (void)hipEventDestroy(done[i]);
            }
            if (out[i])
                gpx_model_destroy(out[i]);
            out[i] = nullptr;
        }
        g_err = keep;
        return code;
    };
    {
        // Under the source's lock: make sure its inverse factor exists, create all shells and ISSUE all copies -- one
        // pair of peer copies per replica on that replica's own stream, so that the transfers to different devices run
        // side by side on their own xGMI links (round 2 copied one replica after the other on the source's stream and
        // waited for each).  The source's stream is made to wait for every copy, so that anything that later
        // synchronises it (update, destroy) also waits for them; the lock is released while the copies fly.
        std::lock_guard<std::mutex> lk(src->mtx);
        HIPCHK(hipSetDevice(src->device));
        if ((rc = build_inverse(src)))  // replicas carry the inverse factor: they hold no LDL^T to build it from
            return rc;
        HIPCHK(hipStreamSynchronize(src->stream));
        const size_t xbytes = src->esz * (size_t)src->npad * src->npad;
        std::vector<double> hD = src->hD;
        if (hD.empty() && src->t_d) {  // replicas hold no factor: keep D readable (GPX_FIELD_D)
            hD.resize((size_t)src->n);
            hipError_t e;
            if (src->prec == GPX_PREC_F64) {
                e = hipMemcpy(hD.data(), src->t_d, sizeof(double) * (size_t)src->n, hipMemcpyDeviceToHost);
            } else {
                std::vector<float> t((size_t)src->n);
                e = hipMemcpy(t.data(), src->t_d, sizeof(float) * (size_t)src->n, hipMemcpyDeviceToHost);
                hD.assign(t.begin(), t.end());
            }
            if (e != hipSuccess)
                return fail(GPX_E_HIP, std::string("replicate: ") + hipGetErrorString(e));
        }
        for (int i = 0; i < ndev && rc == GPX_OK; ++i) {
            gpx_options o = src->opt;
            o.device = devs[i];
            if (src->promoted)  // the source kept its fp64 state (indefinite kernel matrix): so do its replicas
                o.precision = GPX_PREC_F64;
            gpx_model *r = nullptr;
            if ((rc = gpx_model_create_shell(&src->kern, (size_t)src->n, &o, &r)))
                break;
            out[i] = r;
            hipError_t e = hipSetDevice(devs[i]);
            if (e == hipSuccess && devs[i] != src->device) {  // best effort: direct xGMI copies instead of staging through the host
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[i], src->device) == hipSuccess && can) {
                    (void)hipDeviceEnablePeerAccess(src->device, 0);
                    (void)hipGetLastError();  // "already enabled" is fine
                }
            }
            if (e == hipSuccess)
                e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
            if (e == hipSuccess)
                e = hipMemcpyPeerAsync(r->blob0, devs[i], src->blob0, src->device, src->blob0_bytes, r->stream);
            if (e == hipSuccess)
                e = hipMemcpyPeerAsync(r->X, devs[i], src->X, src->device, xbytes, r->stream);
            if (e == hipSuccess)
                e = hipEventRecord(done[i], r->stream);
            if (e == hipSuccess)
                e = hipStreamWaitEvent(src->stream, done[i], 0);
            if (e != hipSuccess) {
                rc = fail(e == hipErrorOutOfMemory ? GPX_E_OOM : GPX_E_HIP, std::string("replicate: ") + hipGetErrorString(e));
                break;
            }
            // host-side fields (accessors, a later update() by rebuild)
            r->hx = src->hx, r->hy = src->hy, r->hz = src->hz, r->hlabel = src->hlabel, r->hs2 = src->hs2;
            r->has_s2 = src->has_s2;
            r->perm = src->perm;
            r->R = src->R;
            r->hD = hD;
            r->stats = src->stats;
            r->sk = src->sk;
            r->var_fit = src->var_fit;
            r->op64 = src->op64;
        }
        if (rc)
            return cleanup(rc);
    }
    // one wait per replica, outside the lock; then commit (reads the centre back from the received blob)
    const bool packed = src->x_packed;  // (fixed once the inverse factor exists)
    for (int i = 0; i < ndev; ++i) {
        const hipError_t e = hipEventSynchronize(done[i]);
        (void)hipEventDestroy(done[i]);
        done[i] = nullptr;
        if (e != hipSuccess)
            return cleanup(fail(GPX_E_HIP, std::string("replicate: ") + hipGetErrorString(e)));
        if ((rc = gpx_model_commit(out[i], 1)))
            return cleanup(rc);
        out[i]->x_packed = packed;
    }
    return GPX_OK;
}

// ---- stand-alone kbuild (tests / roofline leg) ---------------------------------------------------
extern "C" int gpx_dev_kbuild(const gpx_kernel *kernel, int precision, size_t n, size_t n_padded, const void *d_x,
                              const void *d_y, const void *d_z, const void *d_s2, void *d_K, void *d_rmax,
                              void *stream)
{
    if (!kernel || !d_x || !d_y || !d_z || !d_s2 || !d_K)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (n == 0 || n_padded % PANEL != 0 || n_padded < n)
        return fail(GPX_E_BAD_ARG, "n_padded must be gpx_padded_n(n)");
    const int nt = (int)(n_padded / TILE), ntiles = nt * (nt + 1) / 2;
    // per-tile arg-max scratch of the kernel: one buffer per DEVICE (round 2 kept a thread-local one that stayed on
    // whichever device had been current first), grown on demand, released by gpx_trim().  The stage entry is a bench /
    // test hook: calls on one device are serialised by the lock for as long as they are being enqueued.
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES)
        return fail(GPX_E_BAD_ARG, "device ordinal out of range");
    std::lock_guard<std::mutex> lk(gpxh::g_kb_mtx);
    gpxh::KbuildScratch &ks = gpxh::g_kb_scratch[dev];
    if (ks.cap < ntiles) {
        (void)hipDeviceSynchronize();  // an earlier launch on another stream may still write the old scratch
        if (ks.tm)
            (void)hipFree(ks.tm);
        if (ks.tj)
            (void)hipFree(ks.tj);
        ks.tm = nullptr, ks.tj = nullptr, ks.cap = 0;
        HIPCHK(hipMalloc((void **)&ks.tm, sizeof(float) * ntiles));
        HIPCHK(hipMalloc((void **)&ks.tj, sizeof(int) * (2 * ntiles + 2)));
        ks.cap = ntiles;
    }
    CovHost c = make_cov(*kernel);
    launch_kbuild(precision, c, (int)n, (int)n_padded, d_x, d_y, d_z, d_s2, d_K, ks.tm, ks.tj, (hipStream_t)stream);
    (void)d_rmax;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, hipGetErrorString(le));
    return GPX_OK;
}

// ---- stand-alone kqp (bench roofline leg) --------------------------------------------------------
extern "C" int gpx_dev_kqp(const gpx_kernel *kernel, int precision, size_t n, size_t n_padded, const void *d_px,
                           const void *d_py, const void *d_pz, size_t nq, const void *d_qx, const void *d_qy,
                           const void *d_qz, const void *d_fab, void *d_Kqp, void *stream)
{
    if (!kernel || !d_px || !d_py || !d_pz || !d_qx || !d_qy || !d_qz || !d_Kqp)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (n == 0 || n_padded % PANEL != 0 || n_padded < n || nq == 0 || nq % TILE != 0)
        return fail(GPX_E_BAD_ARG, "n_padded must be gpx_padded_n(n) and nq a multiple of 128");
    if (precision != GPX_PREC_F32 && precision != GPX_PREC_F64)
        return fail(GPX_E_BAD_ARG, "precision must be GPX_PREC_F32 or GPX_PREC_F64");
    CovHost c = make_cov(*kernel);
    const int np_rows = (int)std::min<size_t>(n_padded, (n + TILE - 1) / TILE * TILE);
    // the points are fp64; distances, kernel and fit are formed in fp64 and rounded once to `precision` (what
    // evaluate does for every model that takes the fit out of its operand); no centre is needed for fp64 differences
    launch_kqp(true, precision, precision == GPX_PREC_F64, c, (int)n, (int)n_padded, d_px, d_py, d_pz, nullptr, (long)nq,
               (long)nq, (const double *)d_qx, (const double *)d_qy, (const double *)d_qz, d_Kqp, (hipStream_t)stream,
               np_rows, (const double *)d_fab, (long)nq);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, hipGetErrorString(le));
    return GPX_OK;
}

// the fp32-formed operand (exponential kernels): fp32 arithmetic on fp32 points given relative to the centre d_cen
// (3 doubles on the device), queries centred before rounding; d_fab as in gpx_dev_kqp
extern "C" int gpx_dev_kqp_f32(const gpx_kernel *kernel, size_t n, size_t n_padded, const void *d_px, const void *d_py,
                               const void *d_pz, const void *d_cen, size_t nq, const void *d_qx, const void *d_qy,
                               const void *d_qz, const void *d_fab, void *d_Kqp, void *stream)
{
    if (!kernel || !d_px || !d_py || !d_pz || !d_cen || !d_qx || !d_qy || !d_qz || !d_Kqp)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (n == 0 || n_padded % PANEL != 0 || n_padded < n || nq == 0 || nq % TILE != 0)
        return fail(GPX_E_BAD_ARG, "n_padded must be gpx_padded_n(n) and nq a multiple of 128");
    CovHost c = make_cov(*kernel);
    const int np_rows = (int)std::min<size_t>(n_padded, (n + TILE - 1) / TILE * TILE);
    launch_kqp(false, GPX_PREC_F32, false, c, (int)n, (int)n_padded, d_px, d_py, d_pz, (const double *)d_cen, (long)nq,
               (long)nq, (const double *)d_qx, (const double *)d_qy, (const double *)d_qz, d_Kqp, (hipStream_t)stream,
               np_rows, (const double *)d_fab, (long)nq);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, hipGetErrorString(le));
    return GPX_OK;
}
