// gpx_model.hpp -- host-side internals of libgpx.so shared by gpx_api.hip (C entry points, model life cycle),
// gpx_build.hip (kernel matrix, blocked LDL^T, solves, inverse factor) and gpx_eval.hip (prediction paths).
#pragma once
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "gpx_internal.hpp"


#include "gpx_internal.hpp"

using namespace gpx;

// Host internals live in namespace gpxh so that libgpx.so exports nothing un-prefixed besides the gpx_* C entries.
namespace gpxh {
// ---- error state (thread-local message behind gpx_last_error) --------------------------------------
extern thread_local std::string g_err;
int fail(int code, const std::string &msg);
}  // namespace gpxh
using namespace gpxh;
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            int code__ = (e__ == hipErrorOutOfMemory) ? GPX_E_OOM : GPX_E_HIP;                         \
            return fail(code__, std::string(#expr) + ": " + hipGetErrorString(e__));                   \
        }                                                                                              \
    } while (0)


// ------------------------------------------------------------------------------------------------
enum { EV_T0 = 0, EV_KBUILD, EV_FACTOR, EV_SOLVE, EV_NORMALS, EV_INV0, EV_INV1, EV_M0, EV_M1, EV_V1, EV_WS, EV_COUNT };

struct gpx_pending {
    size_t nq;
    const double *qx, *qy, *qz;
    double *f, *v, *grad, *tx, *ty;
    int rc = GPX_OK;
    bool done = false;
    std::string err;
};

struct gpx_model {
    int device = 0, prec = 0;
    size_t esz = 4;
    gpx_kernel kern{};
    CovHost cov{};
    gpx_options opt{};
    int n = 0, npad = 0, nblk = 0;
    bool ready = false, has_s2 = false, has_inverse = false, has_normals = false;
    bool train64 = false;   // MIXED, and F32 / F32_SPLIT models small enough that fp64 training is free (set_training_precision)
    bool inv64 = true;      // F32 modes: assemble the inverse factor in fp64 from the fp32 factor (GPX_INV64=0 disables)
    bool x_packed = false;  // F32_SPLIT: X holds packed hi/lo halves, the 1/D slot holds the scaled weights
    float sk = 1.0f;        // power-of-two scale of the kernel values in the split contraction
    std::vector<double> hx, hy, hz, hlabel, hs2;  // caller order
    std::vector<int> perm;                        // internal position -> caller index
    double R = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;         // look-ahead of the factorisation: the next panel beside the trailing update
    std::vector<hipEvent_t> la_ev;         // its cross-stream events (no timing), reused
    hipEvent_t ev[EV_COUNT] = {};
    std::vector<hipEvent_t> gemm_ev;  // pairs bracketing GEMM launches (stats)
    size_t gemm_ev_used_factor = 0, gemm_ev_used_var = 0;
    std::vector<hipEvent_t> kqp_ev;   // pairs bracketing the Kqp launches of the last evaluate (stats)
    size_t kqp_ev_used = 0;
    double factor_gemm_flops = 0;     // algorithmic flops of the event-timed trailing-update launches

    // state blob part 0 = everything evaluate() reads besides X, internal order, npad each:
    //   fp64 x y z alpha (the mean / gradient are always evaluated in fp64) | fp64 1/D | fp64 [VAR_NCORR][npad]: the
    //   row-correction vectors X b_c of the variance contraction (gpx_internal.hpp) | T x' y' z' (points RELATIVE TO THE
    //   CENTRE, so that fp32 arithmetic never sees where the cloud sits) | T 1/D | BLOB_META doubles: centre x y z,
    //   1 / (sx sk) of the split operands, reserved
    void *blob0 = nullptr;
    size_t blob0_bytes = 0;
    double *d_x = nullptr, *d_y = nullptr, *d_z = nullptr, *d_alpha = nullptr, *d_dinv64 = nullptr, *d_corr = nullptr;
    void *t_x = nullptr, *t_y = nullptr, *t_z = nullptr, *t_dinv = nullptr;
    double *d_meta = nullptr;  // [0..2] centre (also the device-side `cen` argument), [3] 1 / (sx sk)
    double cen[3] = {0, 0, 0};  // centroid of the training points (fp64), fixed at create / update
    bool var_fit = false;  // the low-rank fit is taken out of the kernel operand of the variance GEMM (fp32 modes)
    bool var_fit_opt = false;  // ... as the options / environment asked for it (var_fit: after the promotion rule)
    bool promoted = false;     // indefinite kernel matrix: the model kept its fp64 state (build_model)
    bool op64 = true;      // ... and that operand, k - fit, is formed in fp64 and rounded once (the thin plate), else in fp32
    // other fp64 vectors (npad each): label s2 r f, then one double for max|r|
    double *dvecs = nullptr;
    double *d_lab = nullptr, *d_s2 = nullptr, *d_r = nullptr, *d_f = nullptr, *d_rmax = nullptr, *d_normals = nullptr;
    // other T vectors: s2 d b y xs alpha
    void *tvecs = nullptr;
    void *t_s2 = nullptr, *t_d = nullptr, *t_b = nullptr, *t_yv = nullptr, *t_xs = nullptr, *t_alpha = nullptr;
    std::vector<double> hD;  // D kept on the host once the factor has been released (mixed precision)
    void *Kmat = nullptr;  // npad x npad, L D L^T in place
    void *linv = nullptr;  // nblk x 128 x 128
    void *Wp = nullptr;    // npad x 512 panel workspace
    void *X = nullptr;     // npad x npad inverse factor (state blob part 1)
    int *d_info = nullptr; // [0] first bad pivot (1-based), [1] negative pivots, [2..3] argmax pair; [5] substitution gave up
    float *d_tmax = nullptr;
    int *d_tij = nullptr;
    // evaluation workspaces (grown on demand, guarded by mtx)
    double *ws_pred = nullptr;
    size_t ws_pred_doubles = 0;
    void *ws_kqp = nullptr;
    size_t ws_kqp_bytes = 0;
    void *ws_partial = nullptr;
    size_t ws_partial_bytes = 0;
    void *ws_coef = nullptr;  // doubles [VAR_NCOEF][qbatch]: query-side coefficients of the fit, then a_q, b_q, c_q
    size_t ws_coef_bytes = 0;
    double *ws_grad = nullptr;
    size_t ws_grad_doubles = 0;
    void *ws_small = nullptr;      // partial sums + counters of the one-launch path for a handful of queries
    double *ws_host_io = nullptr;  // device staging for the host-pointer evaluate
    size_t ws_host_io_doubles = 0;
    int qbatch = 8192;
    // flat combining of concurrent small evaluate() calls (the node issues one call per grid point from
    // hundreds of threads, src/gp_node.cpp:1027-1038): whoever finds no leader takes every pending request
    // and runs them as ONE device batch
    gpxh::FlatCombiner<gpx_pending> combiner;  // gpx_host.hpp
    double *pin = nullptr;  // pinned host staging of the combiner
    size_t pin_doubles = 0;
    double *pin2[2] = {nullptr, nullptr};  // pinned double buffer of the pipelined large-batch path
    size_t pin2_doubles = 0;
    hipEvent_t pin2_done[2] = {nullptr, nullptr};
    std::mutex mtx;
    gpx_stats stats{};
    bool stats_eval_pending = false;
    bool eval_had_var = false;
    bool ws_in_flight = false;  // ev[EV_WS] marks the end of the last evaluation that used the shared workspaces
};

namespace gpxh {
// ---- large device buffers (gpx_api.hip) ---------------------------------------------------------------
// hipMalloc / hipFree of the multi-GiB buffers of a big model (kernel matrix, inverse factor, the three fp64
// temporaries of the inverse assembly) cost 100-700 ms per create() on this stack, and erratically so (measured:
// scripts/la_check.py, create() wall 49 ms or 300-710 ms for 25 ms of device work).  Buffers of at least
// BIG_POOL_MIN bytes released by one model are therefore parked in a per-process pool (per device, best fit within
// 25 %) and handed to the next; GPX_POOL_MB caps the parked bytes (default 16384, 0 disables), gpx_trim() empties
// the pool.  Since round 3 BIG_POOL_MIN is 4 KiB, so the ~15 buffers of a model of a few hundred points are recycled too:
// create + destroy of an N = 277 model 1.84 -> 1.19 ms wall, N = 724 2.5 -> 1.66 ms (their device work is 0.6 / 1.07 ms;
// 0.80 / 1.26 ms with the stream pool above).
// A recycled buffer holds its previous owner's data: nothing in the library may assume a fresh allocation reads as zeros
// (the whole GPU suite runs hundreds of models through the pool in one process).  big_free() does NOT synchronise: the caller has waited for the work that used the buffer (quiesce(model):
// the model's own streams and workspace event -- not the whole device, which would stall every other model and thread).
// hipStreamCreateWithFlags costs 0.37 ms and hipStreamDestroy 0.44 ms on this stack (rocprofv3 --hip-trace of
// scripts/create_small.py): more than the device work of a 277-point model.  A model's streams therefore come from, and
// go back (synchronised) to, a per-device pool of non-blocking streams; gpx_trim() destroys the pooled ones.
hipError_t stream_acquire(int device, hipStream_t *s);  // the current device must be `device`
void stream_release(int device, hipStream_t s);
hipError_t pinned_acquire(size_t bytes, void **p);  // recycled pinned host blocks (gpx_api.hip)
void pinned_release(void *p);
hipError_t big_alloc(void **p, size_t bytes);  // BigPool of gpx_host.hpp over the HIP backend
void big_free(void *p);
// releases a device allocation when the scope is left on an error path (HIPCHK returns early); release() hands it on
struct DevGuard {
    void *p = nullptr;
    bool big = false;  // from big_alloc (parked on release) or hipMalloc
    DevGuard() = default;
    DevGuard(void *q, bool b) : p(q), big(b) {}
    DevGuard(const DevGuard &) = delete;
    DevGuard &operator=(const DevGuard &) = delete;
    ~DevGuard() { reset(); }
    void reset()
    {
        if (p) {
            if (big)
                big_free(p);
            else
                (void)hipFree(p);
        }
        p = nullptr;
    }
    void *release()
    {
        void *q = p;
        p = nullptr;
        return q;
    }
};

// What a rank-n update carries over from the previous factorisation (device buffers of the OLD padded size)
struct kept_factor {
    int t0 = 0;        // rows / columns [0, t0) of L, D and the inverse diagonal blocks stay valid
    int np_old = 0;
    int n_neg = 0;     // negative pivots among the kept ones
    void *K = nullptr, *linv = nullptr, *d = nullptr, *dinv = nullptr;  // d, dinv: t0 entries each
    void *X = nullptr;  // the old inverse factor (leading dimension np_old) when it had been built, else null
    void release()
    {
        for (void *p : {d, dinv})  // hipMalloc'ed by gpx_model_update
            if (p)
                (void)hipFree(p);
        big_free(K);  // from big_alloc, like every kernel matrix / inverse factor / diagonal-block inverse
        big_free(linv);
        big_free(X);
        K = linv = d = dinv = X = nullptr;
    }
};

constexpr size_t SMALL_EVAL_MAX_NQ = 64;  // a handful of queries on a small model: one launch (gpx_predict.hip)
constexpr size_t COMBINE_MAX_NQ = 4096;   // larger host calls fill the device on their own

// ---- gpx_build.hip ----------------------------------------------------------------------------------
void free_dev(gpx_model *m);
void quiesce(gpx_model *m);
int ensure(gpx_model *m, void **p, size_t *have, size_t need);
int alloc_blob0(gpx_model *m, size_t esz, void **blob, size_t *bytes);
void carve_blob0(gpx_model *m);
int alloc_model(gpx_model *m);
hipEvent_t *gemm_events(gpx_model *m, size_t idx);
int build_inverse(gpx_model *m);
int build_model(gpx_model *m, kept_factor *keep = nullptr, bool no_dataflow = false);  // no_dataflow: the launch chain only
int alloc_factor_buffers(gpx_model *m);  // Kmat, linv, Wp for a model whose matrix is filled by the caller (gpx_dgp.hip)
void factorize_matrix_append(gpx_model *m, int t0);
void factorize_matrix(gpx_model *m);     // blocked LDL^T of m->Kmat in place (t_d, t_dinv, linv, d_info)
void solve_factored(gpx_model *m, void *b /*consumed*/, void *ytmp, void *x);  // x = (L D L^T)^-1 b on T vectors
void set_query_batch(gpx_model *m);
// Time after which a wait inside a dataflow launch / the one-launch substitution gives up, in ticks of the constant 100 MHz
// clock: 20 ms up to 8192 padded rows (such a launch takes 0.1 - 6 ms), 200 ms above (N = 16384 fp64: 30 ms); GPX_WAIT_BUDGET_US
// overrides (tests: 0 = give up at once).  A fallback costs the caller this much, not the second of round 5's poll counts.
long long wait_budget_ticks(int npad);
int device_cu_count(int dev);  // hipDeviceAttributeMultiprocessorCount, cached per ordinal (0 when it cannot be told)
void set_split_scale(gpx_model *m);  // F32_SPLIT: m->sk from the kernel's amplitude (every model of the mode, packed or not)
bool split_packs(const gpx_model *m);  // F32_SPLIT: does a model of this size hold packed fp16 operands (small ones keep the fp32 kernel)
void set_training_precision(gpx_model *m);
// ---- gpx_eval.hip -----------------------------------------------------------------------------------
// fixed_order: the order of every sum (the mean's split of the point range, the walk of the variance tiles) is made independent
// of nq -- the mean takes the plan of a FIXED_ORDER_PLAN_NQ-query call, the variance tiles are never launched in pairs -- so that
// a query receives the same bits in whatever batch it is evaluated: iso-surface sampling (whole grid, candidate set, survivors,
// the slab of a sharded call)
constexpr long FIXED_ORDER_PLAN_NQ = 32768;
int evaluate_locked(gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz, double *f, double *v,
                    double *grad, double *tx, double *ty, hipStream_t s, bool fixed_order = false);
int check_query(const gpx_model *m, size_t nq, const void *qx, const void *qy, const void *qz, const void *f);
void resolve_eval_stats(gpx_model *m);
}  // namespace gpxh
