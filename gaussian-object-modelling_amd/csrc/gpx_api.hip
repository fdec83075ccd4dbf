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

using namespace gpx;

// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess) {                                                                       \
            int code__ = (e__ == hipErrorOutOfMemory) ? GPX_E_OOM : GPX_E_HIP;                         \
            return fail(code__, std::string(#expr) + ": " + hipGetErrorString(e__));                   \
        }                                                                                              \
    } while (0)

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
    bool inv64 = true;      // F32 modes: assemble the inverse factor in fp64 from the fp32 factor (GPX_INV64=0 disables)
    bool x_packed = false;  // F32_SPLIT: X holds packed hi/lo halves, the 1/D slot holds the scaled weights
    float sk = 1.0f;        // power-of-two scale of the kernel values in the split contraction
    std::vector<double> hx, hy, hz, hlabel, hs2;  // caller order
    std::vector<int> perm;                        // internal position -> caller index
    double R = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT] = {};
    std::vector<hipEvent_t> gemm_ev;  // pairs bracketing GEMM launches (stats)
    size_t gemm_ev_used_factor = 0, gemm_ev_used_var = 0;

    // state blob part 0 = everything evaluate() reads besides X, internal order, npad each:
    //   fp64 x y z alpha (the mean / gradient are always evaluated in fp64) | T x y z 1/D
    void *blob0 = nullptr;
    size_t blob0_bytes = 0;
    double *d_x = nullptr, *d_y = nullptr, *d_z = nullptr, *d_alpha = nullptr;
    void *t_x = nullptr, *t_y = nullptr, *t_z = nullptr, *t_dinv = nullptr;
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
    int *d_info = nullptr; // [0] first bad pivot (1-based), [1] negative pivots, [2..3] argmax pair
    float *d_tmax = nullptr;
    int *d_tij = nullptr;
    // evaluation workspaces (grown on demand, guarded by mtx)
    double *ws_pred = nullptr;
    size_t ws_pred_doubles = 0;
    void *ws_kqp = nullptr;
    size_t ws_kqp_bytes = 0;
    void *ws_partial = nullptr;
    size_t ws_partial_bytes = 0;
    double *ws_grad = nullptr;
    size_t ws_grad_doubles = 0;
    void *ws_small = nullptr;      // partial sums + counters of the one-launch path for a handful of queries
    double *ws_host_io = nullptr;  // device staging for the host-pointer evaluate
    size_t ws_host_io_doubles = 0;
    int qbatch = 8192;
    // flat combining of concurrent small evaluate() calls (the node issues one call per grid point from
    // hundreds of threads, src/gp_node.cpp:1027-1038): whoever finds no leader takes every pending request
    // and runs them as ONE device batch
    std::mutex qmtx;
    std::condition_variable qcv;
    std::vector<struct gpx_pending *> pending;
    bool leader_active = false;
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

static void free_dev(gpx_model *m)
{
    auto F = [](void *p) {
        if (p)
            (void)hipFree(p);
    };
    F(m->dvecs);
    F(m->blob0);
    F(m->tvecs);
    F(m->Kmat);
    F(m->linv);
    F(m->Wp);
    F(m->X);
    F(m->d_info);
    F(m->d_tmax);
    F(m->d_tij);
    F(m->ws_pred);
    F(m->ws_kqp);
    F(m->ws_partial);
    F(m->ws_grad);
    F(m->ws_host_io);
    F(m->ws_small);
    m->ws_small = nullptr;
    F(m->d_normals);
    if (m->pin)
        (void)hipHostFree(m->pin);
    m->pin = nullptr;
    m->pin_doubles = 0;
    for (int b = 0; b < 2; ++b) {
        if (m->pin2[b])
            (void)hipHostFree(m->pin2[b]);
        m->pin2[b] = nullptr;
        if (m->pin2_done[b])
            (void)hipEventDestroy(m->pin2_done[b]);
        m->pin2_done[b] = nullptr;
    }
    m->pin2_doubles = 0;
    m->dvecs = nullptr;
    m->blob0 = m->tvecs = m->Kmat = m->linv = m->Wp = m->X = nullptr;
    m->d_info = nullptr;
    m->d_tmax = nullptr;
    m->d_tij = nullptr;
    m->ws_pred = nullptr;
    m->ws_kqp = m->ws_partial = nullptr;
    m->ws_grad = nullptr;
    m->ws_host_io = nullptr;
    m->d_normals = nullptr;
    m->ws_pred_doubles = m->ws_kqp_bytes = m->ws_partial_bytes = m->ws_grad_doubles = m->ws_host_io_doubles = 0;
    for (auto &e : m->gemm_ev)
        (void)hipEventDestroy(e);
    m->gemm_ev.clear();
}

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
    if (m->stream)
        (void)hipStreamDestroy(m->stream);
    if (prev >= 0)
        (void)hipSetDevice(prev);
    delete m;
}

static int ensure(void **p, size_t *have, size_t need)
{
    if (*have >= need && *p)
        return GPX_OK;
    if (*p)
        HIPCHK(hipFree(*p));
    *p = nullptr;
    *have = 0;
    HIPCHK(hipMalloc(p, need));
    *have = need;
    return GPX_OK;
}

// Eigen 3.2 LDLT pivot rule restated: at step k pick the FIRST largest |diagonal| among the
// not-yet-eliminated rows and swap it to k.  The left-looking algorithm never updates the
// trailing diagonal before it is chosen, so the sequence depends on diag(K) only.
static void eigen_pivot_order(const std::vector<double> &diag, std::vector<int> &perm)
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

static int alloc_blob0(gpx_model *m, size_t esz, void **blob, size_t *bytes)
{
    const size_t np = (size_t)m->npad;
    *bytes = sizeof(double) * np * 4 + esz * np * 4;
    HIPCHK(hipMalloc(blob, *bytes));
    return GPX_OK;
}

static void carve_blob0(gpx_model *m)
{
    const size_t np = (size_t)m->npad, e = m->esz;
    m->d_x = (double *)m->blob0;
    m->d_y = m->d_x + np;
    m->d_z = m->d_y + np;
    m->d_alpha = m->d_z + np;
    char *b = (char *)(m->d_alpha + np);
    m->t_x = b;
    m->t_y = b + e * np;
    m->t_z = b + 2 * e * np;
    m->t_dinv = b + 3 * e * np;
}

static int alloc_model(gpx_model *m)
{
    const size_t np = (size_t)m->npad, e = m->esz;
    int rc = alloc_blob0(m, e, &m->blob0, &m->blob0_bytes);
    if (rc)
        return rc;
    carve_blob0(m);
    HIPCHK(hipMalloc((void **)&m->dvecs, sizeof(double) * (np * 4 + 8)));
    m->d_lab = m->dvecs;
    m->d_s2 = m->d_lab + np;
    m->d_r = m->d_s2 + np;
    m->d_f = m->d_r + np;
    m->d_rmax = m->d_f + np;
    HIPCHK(hipMalloc(&m->tvecs, e * np * 6));
    char *b = (char *)m->tvecs;
    m->t_s2 = b;
    m->t_d = b + e * np;
    m->t_b = b + 2 * e * np;
    m->t_yv = b + 3 * e * np;
    m->t_xs = b + 4 * e * np;
    m->t_alpha = b + 5 * e * np;
    HIPCHK(hipMalloc((void **)&m->d_info, sizeof(int) * 8));
    return GPX_OK;
}

static hipEvent_t *gemm_events(gpx_model *m, size_t idx)
{
    while (m->gemm_ev.size() < 2 * (idx + 1)) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess)
            return nullptr;
        m->gemm_ev.push_back(e);
    }
    return &m->gemm_ev[2 * idx];
}

// ---- L y = b ; y *= 1/D ; L^T x = y on T vectors (block substitution with inverse blocks) ------
static void solve_ldl(gpx_model *m, void *b /*consumed*/, void *ytmp, void *x)
{
    for (int kb = 0; kb < m->nblk; ++kb)
        launch_fwd_step(m->prec, kb, m->nblk, m->Kmat, m->npad, m->linv, b, ytmp, m->stream);
    launch_scale_vec(m->prec, m->npad, ytmp, m->t_dinv, m->stream);
    for (int kb = m->nblk - 1; kb >= 0; --kb)
        launch_bwd_step(m->prec, kb, m->Kmat, m->npad, m->linv, ytmp, x, m->stream);
}

// ---- blocked right-looking LDL^T -----------------------------------------------------------------
// Outer panels of 256 columns = 2 diagonal blocks of 128 (GPX_PANEL=512: 4 blocks): per block the diagonal LDL^T
// (+ inverse), the panel solve as a GEMM with the inverse block (W = A21 Linv^T to the workspace, L21 = W D^-1 in
// place) and the update of the remaining columns of the panel; then ONE trailing update with K = panel width.
// Measured at N = 16384 fp32: 512-wide panels give a trailing tile 16 k-tiles instead of 8 (100.6 -> 112 TFLOP/s,
// LDL^T 31.8 -> 30.8 ms), but the longer fp32 accumulations cost the ill-conditioned thin-plate system accuracy
// (alpha after two refinement steps 2.9e-5 instead of < 1e-5 of the fp64 result), so 256 stays the default.
// Columns before c_start (a multiple of 128) are taken as already factorised and applied (rank-n update).
static void factorize(gpx_model *m, int c_start = 0)
{
    // The kernel matrix is the identity on the padding (N is padded to a multiple of 256): 128-blocks that lie
    // entirely in it are already factorised (L = I, D = 1) and are only given their identity inverse, so the loops
    // below stop at the last block that holds a training point -- N = 277 factorises 3 diagonal blocks, not 4.
    const int np_full = m->npad;
    const int np = std::min(np_full, (m->n + TILE - 1) / TILE * TILE);
    launch_identity_blocks(m->prec, np / TILE, m->nblk, m->linv, m->t_d, m->t_dinv, m->stream);
    const int ldk = np_full;
    const size_t e = m->esz;
    char *K = (char *)m->Kmat;
    char *W = (char *)m->Wp;
    auto Kp = [&](size_t r, size_t c) { return (void *)(K + (r * ldk + c) * e); };
    auto Wpp = [&](size_t r, size_t c) { return (void *)(W + (r * WIDE_PANEL + c) * e); };
    size_t gemm_idx = 0;
    // one 128-wide step: diagonal block, panel solve (W to column `wcol` of the workspace, L21 in place)
    auto block_step = [&](int cc, int wcol) {
        const int r0 = cc + TILE;
        launch_diag_ldl(m->prec, Kp(cc, cc), ldk, m->linv, m->t_d, m->t_dinv, m->d_info, cc / TILE, m->stream);
        if (r0 >= np)
            return;
        GemmArgs t;  // W = A21 * Linv^T ; L21 = W * D^-1 (in place)
        t.A = Kp(r0, cc), t.lda = ldk;
        t.B = (char *)m->linv + (size_t)(cc / TILE) * TILE * TILE * e, t.ldb = TILE;
        t.C = Kp(r0, cc), t.ldc = ldk;
        t.M = np - r0, t.N = TILE, t.K = TILE;
        t.b_lower = 1;
        t.epi = EPI_TRSM;
        t.W = Wpp(r0, wcol), t.ldw = WIDE_PANEL;
        t.colscale = (char *)m->t_dinv + (size_t)cc * e;
        launch_gemm(m->prec, t, m->stream);
    };
    // trailing matrix from row / column r0 on -= W[:, 0:kw] * L[:, c0:c0+kw]^T, lower tiles only
    auto trailing = [&](int c0, int r0, int kw) {
        GemmArgs s;
        s.A = Wpp(r0, 0), s.lda = WIDE_PANEL;
        s.B = Kp(r0, c0), s.ldb = ldk;
        s.C = Kp(r0, r0), s.ldc = ldk;
        s.M = np - r0, s.N = np - r0, s.K = kw;
        s.alpha = -1.0, s.beta = 1;
        s.lower_only = 1;
        hipEvent_t *ev = gemm_events(m, gemm_idx);
        if (ev)
            (void)hipEventRecord(ev[0], m->stream);
        launch_gemm(m->prec, s, m->stream);
        if (ev) {
            (void)hipEventRecord(ev[1], m->stream);
            ++gemm_idx;
        }
    };
    int c0 = c_start;
    if (c0 % PANEL) {  // start in the middle of a 256-column unit: a lone 128-wide step
        block_step(c0, 0);
        if (c0 + TILE < np)
            trailing(c0, c0 + TILE, TILE);
        c0 += TILE;
    }
    static const int wide = [] {
        const char *e = std::getenv("GPX_PANEL");
        return e && std::atoi(e) == WIDE_PANEL ? WIDE_PANEL : PANEL;
    }();
    while (c0 < np) {
        const int pw = std::min(wide, np - c0), nb = pw / TILE;
        for (int h = 0; h < nb; ++h) {
            const int cc = c0 + h * TILE, r0 = cc + TILE;
            block_step(cc, h * TILE);
            if (h + 1 < nb && r0 < np) {
                GemmArgs s;  // the remaining columns of the panel (incl. the next diagonal block) -= W_h * L_h^T
                s.A = Wpp(r0, h * TILE), s.lda = WIDE_PANEL;
                s.B = Kp(r0, cc), s.ldb = ldk;
                s.C = Kp(r0, r0), s.ldc = ldk;
                s.M = np - r0, s.N = c0 + pw - r0, s.K = TILE;
                s.alpha = -1.0, s.beta = 1;
                launch_gemm(m->prec, s, m->stream);
            }
        }
        if (c0 + pw < np)
            trailing(c0, c0 + pw, pw);
        c0 += pw;
    }
    m->gemm_ev_used_factor = gemm_idx;
}

// Rank-n update, rows [t0, npad): the kernel rows have just been built; the columns [0, t0) hold the old factor.
// Column block by column block: W = A_rows,j Linv_j^T (to the workspace), L_rows,j = W D_j^-1 (in place), then
// every later column of these rows -= W L_{later rows, j}^T.  2 launches per old column block.
static void factor_append_rows(gpx_model *m, int t0)
{
    const int np = m->npad;
    const size_t e = m->esz;
    char *K = (char *)m->Kmat;
    auto Kp = [&](size_t r, size_t c) { return (void *)(K + (r * np + c) * e); };
    void *Wrows = (char *)m->Wp + ((size_t)t0 * WIDE_PANEL) * e;
    for (int cc = 0; cc < t0; cc += TILE) {
        GemmArgs t;
        t.A = Kp(t0, cc), t.lda = np;
        t.B = (char *)m->linv + (size_t)(cc / TILE) * TILE * TILE * e, t.ldb = TILE;
        t.C = Kp(t0, cc), t.ldc = np;
        t.M = np - t0, t.N = TILE, t.K = TILE;
        t.b_lower = 1;
        t.epi = EPI_TRSM;
        t.W = Wrows, t.ldw = WIDE_PANEL;
        t.colscale = (char *)m->t_dinv + (size_t)cc * e;
        launch_gemm(m->prec, t, m->stream);
        GemmArgs s;  // columns [cc + 128, np) of the new rows
        s.A = Wrows, s.lda = WIDE_PANEL;
        s.B = Kp(cc + TILE, cc), s.ldb = np;
        s.C = Kp(t0, cc + TILE), s.ldc = np;
        s.M = np - t0, s.N = np - (cc + TILE), s.K = TILE;
        s.alpha = -1.0, s.beta = 1;
        launch_gemm(m->prec, s, m->stream);
    }
}

// ---- X = L^-1 by recursive doubling: X21 = -X22 * (L21 * X11) ------------------------------------
static void trtri_levels(int prec, size_t e, char *L, char *X, char *Tw, int np, hipStream_t st)
{
    auto off = [&](size_t r, size_t c) { return (r * np + c) * e; };
    for (long b = TILE; b < np; b *= 2) {
        // nodes p: left = [p*2b, p*2b+b), right = [p*2b+b, min(p*2b+2b, np))
        int P = 0;
        for (long base = 0; base + b < np; base += 2 * b)
            ++P;
        if (P == 0)
            break;
        const long last_base = (long)(P - 1) * 2 * b;
        const int m_last = (int)std::min<long>(b, np - (last_base + b));
        const long stride = 2 * b * (long)np + 2 * b;
        GemmArgs g1;  // T = L21 * X11   (B lower, [k][n])
        g1.A = L + off(b, 0), g1.lda = np;
        g1.B = X + off(0, 0), g1.ldb = np;
        g1.C = Tw + off(b, 0), g1.ldc = np;
        g1.M = (int)b, g1.N = (int)b, g1.K = (int)b;
        g1.sA = g1.sB = g1.sC = stride;
        g1.batch = P, g1.M_last = m_last;
        g1.nn = 1, g1.b_lower = 1;
        launch_gemm(prec, g1, st);
        GemmArgs g2;  // X21 = -X22 * T  (A lower)
        g2.A = X + off(b, b), g2.lda = np;
        g2.B = Tw + off(b, 0), g2.ldb = np;
        g2.C = X + off(b, 0), g2.ldc = np;
        g2.M = (int)b, g2.N = (int)b, g2.K = (int)b;
        g2.sA = g2.sB = g2.sC = stride;
        g2.batch = P, g2.M_last = m_last, g2.k_eq_m = 1;
        g2.nn = 1, g2.a_lower = 1;
        g2.alpha = -1.0;
        launch_gemm(prec, g2, st);
    }
}

static int build_inverse(gpx_model *m)
{
    if (m->has_inverse)
        return GPX_OK;
    const int np = m->npad;
    const size_t e = m->esz;
    if (!m->X)
        HIPCHK(hipMalloc(&m->X, e * (size_t)np * np));
    (void)hipEventRecord(m->ev[EV_INV0], m->stream);
    void *Tws = nullptr, *L64 = nullptr, *X64 = nullptr, *linv64 = nullptr;
    bool assemble64 = m->prec == GPX_PREC_F32 && m->inv64;
    if (assemble64) {
        // three N x N fp64 temporaries: at very large N they may not fit next to K and X -- assemble in fp32 then
        const size_t nn = (size_t)np * np;
        if (hipMalloc(&L64, sizeof(double) * nn) != hipSuccess || hipMalloc(&X64, sizeof(double) * nn) != hipSuccess ||
            hipMalloc(&Tws, sizeof(double) * nn) != hipSuccess ||
            hipMalloc(&linv64, sizeof(double) * (size_t)m->nblk * TILE * TILE) != hipSuccess) {
            (void)hipGetLastError();
            for (void **q : {&L64, &X64, &Tws, &linv64}) {
                if (*q)
                    (void)hipFree(*q);
                *q = nullptr;
            }
            assemble64 = false;
        }
    }
    if (assemble64) {
        // The fp32 factor is kept (that is what runs on the fp32 MFMA), but its inverse is assembled in fp64 and
        // rounded once.  Measured at N = 16384 (variance error / k(0) vs the fp64 pipeline): Matern-5/2 1.0e-5 ->
        // 4.5e-6, Gaussian 1.1e-5 -> 2.3e-6, thin-plate R=4 1.05e-4 -> 2.1e-5, i.e. the level of an fp64 factor:
        // the log2(N/128) levels of products of inverses, not the LDL^T, are where fp32 loses the accuracy.
        const size_t nn = (size_t)np * np;
        launch_cast_f2d(nn, (const float *)m->Kmat, (double *)L64, m->stream);
        launch_cast_f2d((size_t)m->nblk * TILE * TILE, (const float *)m->linv, (double *)linv64, m->stream);
        HIPCHK(hipMemsetAsync(X64, 0, sizeof(double) * nn, m->stream));
        launch_place_diag(GPX_PREC_F64, m->nblk, linv64, X64, np, m->stream);
        trtri_levels(GPX_PREC_F64, 8, (char *)L64, (char *)X64, (char *)Tws, np, m->stream);
        launch_cast_d2f(nn, (const double *)X64, (float *)m->X, m->stream);
    } else {
        HIPCHK(hipMalloc(&Tws, e * (size_t)np * np));
        // blocks above the diagonal are structural zeros: the 256-row variance tiles read the upper-right
        // 128-block of every 256-diagonal block
        HIPCHK(hipMemsetAsync(m->X, 0, e * (size_t)np * np, m->stream));
        launch_place_diag(m->prec, m->nblk, m->linv, m->X, np, m->stream);
        trtri_levels(m->prec, e, (char *)m->Kmat, (char *)m->X, (char *)Tws, np, m->stream);
    }
    (void)hipEventRecord(m->ev[EV_INV1], m->stream);
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipFree(Tws));
    if (L64)
        HIPCHK(hipFree(L64));
    if (X64)
        HIPCHK(hipFree(X64));
    if (linv64)
        HIPCHK(hipFree(linv64));
    float ms = 0;
    if (hipEventElapsedTime(&ms, m->ev[EV_INV0], m->ev[EV_INV1]) == hipSuccess)
        m->stats.t_inverse_ms = ms;
    if (m->opt.precision == GPX_PREC_F32_SPLIT && !m->x_packed) {
        if (!m->hD.size()) {  // keep D readable (GPX_FIELD_D) -- the 1/D slot is about to hold the weights
            std::vector<float> t((size_t)m->n);
            HIPCHK(hipMemcpy(t.data(), m->t_d, sizeof(float) * (size_t)m->n, hipMemcpyDeviceToHost));
            m->hD.assign(t.begin(), t.end());
        }
        int e2 = 0;
        (void)std::frexp(m->cov.k0 > 0 ? m->cov.k0 : 1.0, &e2);
        m->sk = (float)std::ldexp(1.0, -e2);  // k(0) * sk in [0.5, 1)
        launch_split_prepare((float *)m->X, np, (float *)m->t_dinv, m->sk, (unsigned *)(m->d_info + 4), m->stream);
        HIPCHK(hipStreamSynchronize(m->stream));
        m->x_packed = true;
    }
    m->has_inverse = true;
    return GPX_OK;
}

// ---- MIXED precision: round the fp64 state once to fp32 and release the fp64 factor -----------------
static int demote_to_f32(gpx_model *m)
{
    const size_t np = (size_t)m->npad;
    hipStream_t s = m->stream;
    m->hD.resize((size_t)m->n);
    HIPCHK(hipMemcpy(m->hD.data(), m->t_d, sizeof(double) * (size_t)m->n, hipMemcpyDeviceToHost));
    void *nb = nullptr, *nX = nullptr;
    size_t nbytes = 0;
    int rc = alloc_blob0(m, 4, &nb, &nbytes);
    if (rc)
        return rc;
    HIPCHK(hipMalloc(&nX, sizeof(float) * np * np));
    HIPCHK(hipMemcpyAsync(nb, m->blob0, sizeof(double) * np * 4, hipMemcpyDeviceToDevice, s));
    float *tf = (float *)((char *)nb + sizeof(double) * np * 4);
    launch_cast_d2f(np, (const double *)m->t_x, tf, s);
    launch_cast_d2f(np, (const double *)m->t_y, tf + np, s);
    launch_cast_d2f(np, (const double *)m->t_z, tf + 2 * np, s);
    launch_cast_d2f(np, (const double *)m->t_dinv, tf + 3 * np, s);
    launch_cast_d2f(np * np, (const double *)m->X, (float *)nX, s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipFree(m->blob0));
    HIPCHK(hipFree(m->X));
    HIPCHK(hipFree(m->Kmat));
    HIPCHK(hipFree(m->linv));
    HIPCHK(hipFree(m->Wp));
    HIPCHK(hipFree(m->tvecs));
    m->Kmat = m->linv = m->Wp = m->tvecs = nullptr;
    m->t_s2 = m->t_d = m->t_b = m->t_yv = m->t_xs = m->t_alpha = nullptr;
    m->blob0 = nb;
    m->blob0_bytes = nbytes;
    m->X = nX;
    m->prec = GPX_PREC_F32;
    m->esz = 4;
    carve_blob0(m);
    return GPX_OK;
}

// What a rank-n update carries over from the previous factorisation (device buffers of the OLD padded size)
struct kept_factor {
    int t0 = 0;        // rows / columns [0, t0) of L, D and the inverse diagonal blocks stay valid
    int np_old = 0;
    int n_neg = 0;     // negative pivots among the kept ones
    void *K = nullptr, *linv = nullptr, *d = nullptr, *dinv = nullptr;  // d, dinv: t0 entries each
    void release()
    {
        for (void *p : {K, linv, d, dinv})
            if (p)
                (void)hipFree(p);
        K = linv = d = dinv = nullptr;
    }
};

// ---- create: everything after the host arrays are in place ---------------------------------------
static int build_model(gpx_model *m, kept_factor *keep = nullptr)
{
    const int n = m->n, np = m->npad;
    const size_t e = m->esz;
    HIPCHK(hipSetDevice(m->device));
    factor_init(m->prec);
    // Eigen's pivot order from the original diagonal k(0) + sigma2_i
    std::vector<double> diag(n);
    for (int i = 0; i < n; ++i)
        diag[i] = m->cov.k0 + (m->has_s2 ? m->hs2[i] : 0.0);
    eigen_pivot_order(diag, m->perm);
    // host staging (internal order, zero padded)
    std::vector<double> st((size_t)np * 5, 0.0);
    for (int k = 0; k < n; ++k) {
        const int i = m->perm[k];
        st[k] = m->hx[i];
        st[np + k] = m->hy[i];
        st[2 * (size_t)np + k] = m->hz[i];
        st[3 * (size_t)np + k] = m->hlabel[i];
        st[4 * (size_t)np + k] = m->has_s2 ? m->hs2[i] : 0.0;
    }
    if (!m->dvecs) {
        int rc = alloc_model(m);
        if (rc)
            return rc;
        HIPCHK(hipMalloc(&m->Kmat, e * (size_t)np * np));
        HIPCHK(hipMalloc(&m->linv, e * (size_t)m->nblk * TILE * TILE));
        HIPCHK(hipMalloc(&m->Wp, e * (size_t)np * WIDE_PANEL));
        const int nt = np / TILE, ntiles = nt * (nt + 1) / 2;
        HIPCHK(hipMalloc((void **)&m->d_tmax, sizeof(float) * ntiles));
        HIPCHK(hipMalloc((void **)&m->d_tij, sizeof(int) * 2 * ntiles));
    }
    {  // workspace of the matrix-free residual / normals passes (n queries against npad points)
        size_t need = predict_ws_doubles(n, np, m->opt.with_normals != 0) * sizeof(double);
        if (need) {
            int rc = ensure((void **)&m->ws_pred, &m->ws_pred_doubles, need);
            if (rc)
                return rc;
        }
    }
    hipStream_t s = m->stream;
    HIPCHK(hipMemcpyAsync(m->d_x, st.data(), sizeof(double) * (size_t)np * 3, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(m->d_lab, st.data() + 3 * (size_t)np, sizeof(double) * (size_t)np * 2,
                          hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(m->d_alpha, 0, sizeof(double) * (size_t)np, s));
    HIPCHK(hipMemsetAsync(m->d_r, 0, sizeof(double) * (size_t)np * 2 + 64, s));
    HIPCHK(hipMemsetAsync(m->d_info, 0, sizeof(int) * 8, s));
    launch_cast_vec(m->prec, np, np, m->d_x, m->t_x, s);
    launch_cast_vec(m->prec, np, np, m->d_y, m->t_y, s);
    launch_cast_vec(m->prec, np, np, m->d_z, m->t_z, s);
    launch_cast_vec(m->prec, np, np, m->d_s2, m->t_s2, s);
    // ---- kernel matrix ----
    (void)hipEventRecord(m->ev[EV_T0], s);
    const int nt = np / TILE, ntiles = nt * (nt + 1) / 2;
    if (keep && keep->t0 > 0) {
        // rank-n update: the old factor goes back into the (possibly larger) matrix, only the new rows are built
        const size_t t0 = (size_t)keep->t0;
        HIPCHK(hipMemcpy2DAsync(m->Kmat, e * np, keep->K, e * keep->np_old, e * t0, t0, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->linv, keep->linv, e * (t0 / TILE) * TILE * TILE, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->t_d, keep->d, e * t0, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipMemcpyAsync(m->t_dinv, keep->dinv, e * t0, hipMemcpyDeviceToDevice, s));
        launch_kbuild(m->prec, m->cov, n, np, m->t_x, m->t_y, m->t_z, m->t_s2, m->Kmat, m->d_tmax, m->d_tij, s,
                      keep->t0 / TILE);
        (void)hipEventRecord(m->ev[EV_KBUILD], s);
        factor_append_rows(m, keep->t0);
        factorize(m, keep->t0);
    } else {
        launch_kbuild(m->prec, m->cov, n, np, m->t_x, m->t_y, m->t_z, m->t_s2, m->Kmat, m->d_tmax, m->d_tij, s);
        launch_reduce_tilemax(ntiles, m->d_tmax, m->d_tij, m->d_info + 2, s);
        (void)hipEventRecord(m->ev[EV_KBUILD], s);
        // ---- factorisation ----
        factorize(m);
    }
    (void)hipEventRecord(m->ev[EV_FACTOR], s);
    // ---- alpha = K^-1 y with fp64-residual refinement ----
    // ir_steps >= 0: exactly that many steps.  Default: adaptive -- at least one step, then until the fp64 residual
    // max|y - K alpha| is below 1e-9 max|y| (at most 4 steps).  Measured at N = 16384 with an fp32 factor, alpha
    // error vs fp64 after 1 / 2 / 3 steps: Matern-5/2 2e-9 / 7e-13 / 3e-14 (stops after 1), thin-plate R=4
    // 2e-4 / 8e-6 / 2e-7 (runs 3); each step costs one substitution pair + one matrix-free residual (2.3 ms).
    const bool ir_adaptive = m->opt.ir_steps < 0;
    const int ir_max = ir_adaptive ? 4 : m->opt.ir_steps;
    double ymax = 0.0;
    for (int i = 0; i < n; ++i)
        ymax = std::max(ymax, std::fabs(m->hlabel[i]));
    const double ir_tol = 1e-9 * std::max(ymax, 1e-300);
    int ir = 0;
    for (int it = 0;; ++it) {
        // right-hand side: y (first pass) or the fp64 residual
        launch_cast_vec(m->prec, n, np, it == 0 ? m->d_lab : m->d_r, m->t_b, s);
        solve_ldl(m, m->t_b, m->t_yv, m->t_xs);
        launch_axpy_cast(m->prec, n, np, m->d_alpha, m->t_xs, m->t_alpha, s);
        // r = y - K alpha in fp64, matrix-free from the fp64 points
        launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, n, m->d_x, m->d_y, m->d_z,
                       m->d_f, nullptr, m->ws_pred, s);
        HIPCHK(hipMemsetAsync(m->d_rmax, 0, sizeof(double), s));
        launch_residual(n, m->d_lab, m->d_f, m->d_s2, m->d_alpha, m->d_r, m->d_rmax, s);
        ir = it;
        if (it >= ir_max)
            break;
        if (ir_adaptive && it >= 1) {
            double r_now = 0.0;
            HIPCHK(hipMemcpyAsync(&r_now, m->d_rmax, sizeof(double), hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            if (!(r_now > ir_tol))
                break;
        }
    }
    m->stats.ir_steps_done = ir;
    (void)hipEventRecord(m->ev[EV_SOLVE], s);
    // ---- normals at the training points (create<true>, gp_regressor.hpp:166-181) ----
    if (m->opt.with_normals) {
        if (!m->d_normals)
            HIPCHK(hipMalloc((void **)&m->d_normals, sizeof(double) * 3 * (size_t)n));
        launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, n, m->d_x, m->d_y, m->d_z, m->d_f,
                       m->d_normals, m->ws_pred, s);
        launch_normalize_rows3(n, m->d_normals, s);
        m->has_normals = true;
    }
    (void)hipEventRecord(m->ev[EV_NORMALS], s);
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipGetLastError());
    int info[4];
    double rmax = 0;
    HIPCHK(hipMemcpy(info, m->d_info, sizeof(info), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&rmax, m->d_rmax, sizeof(double), hipMemcpyDeviceToHost));
    float ms;
    m->stats = gpx_stats{};
    if (hipEventElapsedTime(&ms, m->ev[EV_T0], m->ev[EV_KBUILD]) == hipSuccess)
        m->stats.t_kbuild_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_KBUILD], m->ev[EV_FACTOR]) == hipSuccess)
        m->stats.t_factor_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_FACTOR], m->ev[EV_SOLVE]) == hipSuccess)
        m->stats.t_solve_ms = ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_SOLVE], m->ev[EV_NORMALS]) == hipSuccess)
        m->stats.t_normals_ms = ms;
    double tg = 0;
    for (size_t i = 0; i < m->gemm_ev_used_factor; ++i)
        if (hipEventElapsedTime(&ms, m->gemm_ev[2 * i], m->gemm_ev[2 * i + 1]) == hipSuccess)
            tg += ms;
    m->stats.t_factor_gemm_ms = tg;
    m->stats.factor_gemm_launches = (int64_t)m->gemm_ev_used_factor;
    m->stats.n = n;
    m->stats.n_padded = np;
    m->stats.n_negative_pivots = info[1] + (keep ? keep->n_neg : 0);
    m->stats.ir_steps_done = ir;
    m->stats.alpha_residual = rmax;
    if (info[0] != 0)
        return fail(GPX_E_SINGULAR, "LDL^T: zero or non-finite pivot at internal row " + std::to_string(info[0] - 1));
    // Model::R (gp_regressor.hpp:135): the device found the arg-max pair, the distance is fp64
    if (!(keep && keep->t0 > 0)) {
        const int a = info[2], b = info[3];
        if (a >= 0 && a < n && b >= 0 && b < n) {
            const int ia = m->perm[a], ib = m->perm[b];
            const double dx = m->hx[ia] - m->hx[ib], dy = m->hy[ia] - m->hy[ib], dz = m->hz[ia] - m->hz[ib];
            m->R = std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    m->ready = true;
    if (m->opt.prepare_variance || m->opt.precision == GPX_PREC_MIXED) {
        int rc = build_inverse(m);
        if (rc)
            return rc;
    }
    if (m->opt.precision == GPX_PREC_MIXED)
        return demote_to_f32(m);
    return GPX_OK;
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

static void set_query_batch(gpx_model *m)
{
    if (m->opt.query_batch > 0) {
        m->qbatch = m->opt.query_batch;
        return;
    }
    // ~512 MiB of Kqp per batch: 8192 queries at N = 16384, more for small models so that one variance
    // launch still fills the chip (N = 724: 131072 queries -> 6 x 1024 tiles)
    size_t qb = ((size_t)512 << 20) / ((size_t)m->npad * 4);
    qb = std::min<size_t>(std::max<size_t>(qb, 8192), 131072);
    m->qbatch = (int)(qb / 256 * 256);
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
    m->prec = (o.precision == GPX_PREC_F32 || o.precision == GPX_PREC_F32_SPLIT) ? GPX_PREC_F32
                                                                                  : GPX_PREC_F64;  // MIXED trains in fp64
    m->esz = m->prec == GPX_PREC_F64 ? 8 : 4;
    m->kern = *kernel;
    m->cov = make_cov(*kernel);
    m->opt = o;
    m->n = (int)n;
    m->npad = (int)gpx_padded_n(n);
    m->nblk = m->npad / TILE;
    set_query_batch(m);
    if (const char *e64 = std::getenv("GPX_INV64"))
        m->inv64 = std::atoi(e64) != 0;
    if (hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking) != hipSuccess) {
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
    const char *app_env = std::getenv("GPX_UPDATE_APPEND");  // 0: always rebuild (tests compare the two)
    const bool append_on = !app_env || std::atoi(app_env) != 0;
    if (append_on && m->ready && m->Kmat && m->linv && !m->x_packed && m->opt.precision != GPX_PREC_MIXED &&
        n_old >= TILE) {
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
            }
        }
    }
    free_dev(m);
    m->ready = m->has_inverse = m->has_normals = false;
    m->prec = (m->opt.precision == GPX_PREC_F32 || m->opt.precision == GPX_PREC_F32_SPLIT) ? GPX_PREC_F32
                                                                                              : GPX_PREC_F64;
    m->esz = m->prec == GPX_PREC_F64 ? 8 : 4;
    m->hD.clear();
    m->x_packed = false;
    m->n = (int)m->hx.size();
    m->npad = (int)gpx_padded_n(m->n);
    m->nblk = m->npad / TILE;
    set_query_batch(m);
    rc = build_model(m, keep.t0 > 0 ? &keep : nullptr);  // :457-459 refactors from scratch; same results
    keep.release();
    m->R = keepR;
    return rc;
}

// ---- evaluate ------------------------------------------------------------------------------------
static int evaluate_locked(gpx_model *m, size_t nq, const double *qx, const double *qy, const double *qz,
                           double *f, double *v, double *grad, double *tx, double *ty, hipStream_t s)
{
    const int np = m->npad;
    const size_t e = m->esz;
    const bool want_basis = tx || ty;
    double *g = grad;
    if (want_basis && !g) {
        int rc = ensure((void **)&m->ws_grad, &m->ws_grad_doubles, sizeof(double) * 3 * nq);
        if (rc)
            return rc;
        g = m->ws_grad;
    }
    size_t need = predict_ws_doubles((long)nq, np, g != nullptr) * sizeof(double);
    if (need) {
        int rc = ensure((void **)&m->ws_pred, &m->ws_pred_doubles, need);
        if (rc)
            return rc;
    }
    if (v) {
        int rc = build_inverse(m);
        if (rc)
            return rc;
        const size_t qb = (size_t)std::min<size_t>((size_t)m->qbatch, ((nq + TILE - 1) / TILE) * TILE);
        if ((rc = ensure(&m->ws_kqp, &m->ws_kqp_bytes, e * qb * np)))
            return rc;
        if ((rc = ensure(&m->ws_partial, &m->ws_partial_bytes, e * qb * m->nblk)))
            return rc;
    }
    // The workspaces (prediction partials, K tile, variance partials) are shared by all evaluations of this model,
    // which may be enqueued on different streams (gpx_model_evaluate_device): order them behind the previous user.
    if (m->ws_in_flight)
        (void)hipStreamWaitEvent(s, m->ev[EV_WS], 0);
    (void)hipEventRecord(m->ev[EV_M0], s);
    // mean and gradient always in fp64 from the fp64 points and alpha (cheap next to the variance, and
    // the long alternating sum of a thin-plate GP at N = 16k is not within 1e-5 in fp32)
    launch_predict(GPX_PREC_F64, m->cov, np, m->d_x, m->d_y, m->d_z, m->d_alpha, (long)nq, qx, qy, qz, f, g,
                   m->ws_pred, s);
    if (want_basis)
        launch_tangent_basis((long)nq, g, tx, ty, s);
    (void)hipEventRecord(m->ev[EV_M1], s);
    m->gemm_ev_used_var = 0;
    if (v) {
        const size_t qb = (size_t)std::min<size_t>((size_t)m->qbatch, ((nq + TILE - 1) / TILE) * TILE);
        const int np_rows = std::min(np, (m->n + TILE - 1) / TILE * TILE);  // 128-row blocks that hold training points
        size_t gi = 0;
        for (size_t q0 = 0; q0 < nq; q0 += qb) {
            const size_t nv = std::min(qb, nq - q0);
            const size_t ntile = ((nv + TILE - 1) / TILE) * TILE;
            if (m->x_packed) {  // F32_SPLIT: fp16 hi/lo operands, three MFMA products per k-step
                launch_kqp_split(m->cov, m->sk, m->n, np, m->t_x, m->t_y, m->t_z, (long)nv, (long)ntile, qx + q0,
                                 qy + q0, qz + q0, m->ws_kqp, s);
                hipEvent_t *ev2 = (s == m->stream) ? gemm_events(m, m->gemm_ev_used_factor + gi) : nullptr;
                if (ev2)
                    (void)hipEventRecord(ev2[0], s);
                launch_vsplit_gemm(m->X, m->ws_kqp, np, (int)ntile, (const float *)m->t_dinv, (float *)m->ws_partial,
                                   (long)qb, 2, s, np_rows);
                if (ev2) {
                    (void)hipEventRecord(ev2[1], s);
                    ++gi;
                }
                launch_var_finish(m->prec, m->cov.k0, np_rows / TILE, (long)qb, m->ws_partial, (long)nv, v + q0, s);
                continue;
            }
            launch_kqp(m->prec, m->cov, m->n, np, m->t_x, m->t_y, m->t_z, (long)nv, (long)ntile, qx + q0, qy + q0,
                       qz + q0, m->ws_kqp, s, np_rows);
            GemmArgs a;  // partial[mt][q] = sum_rows (X * Kqp^T)^2 / D
            a.A = m->X, a.lda = np;
            a.B = m->ws_kqp, a.ldb = np;
            a.M = np_rows, a.N = (int)ntile, a.K = np;  // rows of X in the identity padding see only zeros of Kqp
            a.a_lower = 1;
            a.epi = EPI_COLSQ;
            // 256 x 256 tiles (fp32) halve the L2-miss traffic at equal speed, but only when there are enough of
            // them to fill 256 CUs; small models use 128 x 128 tiles
            a.cfg = (np_rows % 256 == 0 && (size_t)(np_rows / 256) * (ntile / 256) >= 1024) ? 2 : 0;
            a.rowweight = m->t_dinv;
            a.partial = m->ws_partial, a.ldp = (long)qb;
            hipEvent_t *ev = (s == m->stream) ? gemm_events(m, m->gemm_ev_used_factor + gi) : nullptr;
            if (ev)
                (void)hipEventRecord(ev[0], s);
            launch_gemm(m->prec, a, s);
            if (ev) {
                (void)hipEventRecord(ev[1], s);
                ++gi;
            }
            const int bm = gemm_rows_per_partial(m->prec, a);
            launch_var_finish(m->prec, m->cov.k0, np_rows / bm, (long)qb, m->ws_partial, (long)nv, v + q0, s);
        }
        m->gemm_ev_used_var = gi;
    }
    (void)hipEventRecord(m->ev[EV_V1], s);
    (void)hipEventRecord(m->ev[EV_WS], s);
    m->ws_in_flight = true;
    m->stats_eval_pending = true;
    m->eval_had_var = v != nullptr;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
    return GPX_OK;
}

static int check_query(const gpx_model *m, size_t nq, const void *qx, const void *qy, const void *qz, const void *f)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!m->ready)
        return fail(GPX_E_STATE, "model is not ready (shell not committed or create failed)");
    if (nq == 0)
        return fail(GPX_E_EMPTY, "All input data is empty!");
    if (!qx || !qy || !qz)
        return fail(GPX_E_NULL, "Empty data pointer");
    if (!f)
        return fail(GPX_E_NULL, "Empty output pointer");
    return GPX_OK;
}

extern "C" int gpx_model_evaluate_device(const gpx_model *cm, size_t nq, const void *d_qx, const void *d_qy,
                                         const void *d_qz, void *d_f, void *d_v, void *d_grad, void *d_tx,
                                         void *d_ty, void *stream)
{
    int rc = check_query(cm, nq, d_qx, d_qy, d_qz, d_f);
    if (rc)
        return rc;
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = stream ? (hipStream_t)stream : m->stream;
    return evaluate_locked(m, nq, (const double *)d_qx, (const double *)d_qy, (const double *)d_qz, (double *)d_f,
                           (double *)d_v, (double *)d_grad, (double *)d_tx, (double *)d_ty, s);
}

// One device batch for a list of host requests: queries are concatenated into pinned staging, evaluated
// once (the union of the requested outputs), and the results scattered back.
constexpr size_t SMALL_EVAL_MAX_NQ = 64;  // a handful of queries on a small model: one launch (gpx_predict.hip)

static int run_requests(gpx_model *m, const std::vector<gpx_pending *> &reqs)
{
    size_t total = 0;
    bool wv = false, wg = false, wtx = false, wty = false;
    for (const gpx_pending *r : reqs) {
        total += r->nq;
        wv |= r->v != nullptr;
        wg |= r->grad != nullptr;
        wtx |= r->tx != nullptr;
        wty |= r->ty != nullptr;
    }
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    // layout (host pinned and device alike): qx qy qz | f | v | grad | tx | ty
    const size_t doubles = total * (3 + 1 + 1 + 3 + 3 + 3);
    int rc;
    if ((rc = ensure((void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    if (m->pin_doubles < doubles) {
        if (m->pin)
            HIPCHK(hipHostFree(m->pin));
        m->pin = nullptr;
        m->pin_doubles = 0;
        HIPCHK(hipHostMalloc((void **)&m->pin, sizeof(double) * doubles, hipHostMallocDefault));
        m->pin_doubles = doubles;
    }
    double *h = m->pin, *d = m->ws_host_io;
    size_t off = 0;
    for (const gpx_pending *r : reqs) {
        std::memcpy(h + off, r->qx, sizeof(double) * r->nq);
        std::memcpy(h + total + off, r->qy, sizeof(double) * r->nq);
        std::memcpy(h + 2 * total + off, r->qz, sizeof(double) * r->nq);
        off += r->nq;
    }
    double *dqx = d, *dqy = d + total, *dqz = d + 2 * total, *df = d + 3 * total, *dv = d + 4 * total,
           *dg = d + 5 * total, *dtx = d + 8 * total, *dty = d + 11 * total;
    hipStream_t s = m->stream;
    // a handful of queries on a small model: one launch that reads and writes the pinned buffer directly
    static const bool small_on = [] {
        const char *e = std::getenv("GPX_SMALL_EVAL");
        return !e || std::atoi(e) != 0;
    }();
    if (small_on && total <= SMALL_EVAL_MAX_NQ && m->npad <= SMALL_EVAL_NP_MAX &&
        !(wv && m->opt.precision == GPX_PREC_F32_SPLIT)) {
        if (wv && (rc = build_inverse(m)))
            return rc;
        if (!m->ws_small) {
            const size_t sb = small_eval_scratch_bytes((int)SMALL_EVAL_MAX_NQ, SMALL_EVAL_NP_MAX);
            HIPCHK(hipMalloc(&m->ws_small, sb));
            HIPCHK(hipMemsetAsync(m->ws_small, 0, sb, s));
        }
        launch_small_eval(m->prec, m->cov, m->n, m->npad, m->d_x, m->d_y, m->d_z, m->d_alpha, m->X, m->t_dinv,
                          (int)total, (int)SMALL_EVAL_MAX_NQ, h, h + 3 * total, wv ? h + 4 * total : nullptr,
                          wg ? h + 5 * total : nullptr, wtx ? h + 8 * total : nullptr,
                          wty ? h + 11 * total : nullptr, m->ws_small, s);
        hipError_t le = hipGetLastError();
        if (le != hipSuccess)
            return fail(GPX_E_HIP, std::string("kernel launch: ") + hipGetErrorString(le));
        HIPCHK(hipStreamSynchronize(s));
        off = 0;
        for (const gpx_pending *r : reqs) {
            std::memcpy(r->f, h + 3 * total + off, sizeof(double) * r->nq);
            if (r->v)
                std::memcpy(r->v, h + 4 * total + off, sizeof(double) * r->nq);
            if (r->grad)
                std::memcpy(r->grad, h + 5 * total + 3 * off, sizeof(double) * 3 * r->nq);
            if (r->tx)
                std::memcpy(r->tx, h + 8 * total + 3 * off, sizeof(double) * 3 * r->nq);
            if (r->ty)
                std::memcpy(r->ty, h + 11 * total + 3 * off, sizeof(double) * 3 * r->nq);
            off += r->nq;
        }
        return GPX_OK;
    }
    HIPCHK(hipMemcpyAsync(d, h, sizeof(double) * 3 * total, hipMemcpyHostToDevice, s));
    rc = evaluate_locked(m, total, dqx, dqy, dqz, df, wv ? dv : nullptr, wg ? dg : nullptr, wtx ? dtx : nullptr,
                         wty ? dty : nullptr, s);
    if (rc)
        return rc;
    HIPCHK(hipMemcpyAsync(h + 3 * total, df, sizeof(double) * total * (wv ? 2 : 1), hipMemcpyDeviceToHost, s));
    if (wg)
        HIPCHK(hipMemcpyAsync(h + 5 * total, dg, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    if (wtx)
        HIPCHK(hipMemcpyAsync(h + 8 * total, dtx, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    if (wty)
        HIPCHK(hipMemcpyAsync(h + 11 * total, dty, sizeof(double) * 3 * total, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    off = 0;
    for (const gpx_pending *r : reqs) {
        std::memcpy(r->f, h + 3 * total + off, sizeof(double) * r->nq);
        if (r->v)
            std::memcpy(r->v, h + 4 * total + off, sizeof(double) * r->nq);
        if (r->grad)
            std::memcpy(r->grad, h + 5 * total + 3 * off, sizeof(double) * 3 * r->nq);
        if (r->tx)
            std::memcpy(r->tx, h + 8 * total + 3 * off, sizeof(double) * 3 * r->nq);
        if (r->ty)
            std::memcpy(r->ty, h + 11 * total + 3 * off, sizeof(double) * 3 * r->nq);
        off += r->nq;
    }
    return GPX_OK;
}

// Large host batches: slices of 2^18 queries through a pinned double buffer on the model's stream.  The stream runs
// copy-in / kernels / copy-out of slice i while the host fills the other buffer with slice i+1 and, once slice i-1 has
// signalled, hands its results to the caller -- so the pageable <-> pinned copies (the larger part of the PCIe-side
// cost) hide behind the device work, and the staging stays bounded (a 256^3 grid would be 1.9 GB in one piece).
static int run_large(gpx_model *m, const gpx_pending &r)
{
    constexpr size_t SLICE = (size_t)1 << 18;
    const size_t per_q = 3 + 1 + 1 + 3 + 3 + 3;  // qx qy qz | f | v | grad | tx | ty
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    const size_t S = std::min(SLICE, r.nq);
    int rc;
    if ((rc = ensure((void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * S * per_q)))
        return rc;
    if (m->pin2_doubles < S * per_q) {
        for (int b = 0; b < 2; ++b) {
            if (m->pin2[b])
                HIPCHK(hipHostFree(m->pin2[b]));
            m->pin2[b] = nullptr;
        }
        m->pin2_doubles = 0;
        for (int b = 0; b < 2; ++b)
            HIPCHK(hipHostMalloc((void **)&m->pin2[b], sizeof(double) * S * per_q, hipHostMallocDefault));
        m->pin2_doubles = S * per_q;
    }
    for (int b = 0; b < 2; ++b)
        if (!m->pin2_done[b])
            HIPCHK(hipEventCreateWithFlags(&m->pin2_done[b], hipEventDisableTiming));
    hipStream_t s = m->stream;
    double *d = m->ws_host_io;
    const size_t nslices = (r.nq + S - 1) / S;
    // results of slice i (already in pinned buffer i & 1 once its event has fired) -> the caller's arrays
    auto deliver = [&](size_t i) -> int {
        const int b = (int)(i & 1);
        const size_t q0 = i * S, nn = std::min(S, r.nq - q0);
        HIPCHK(hipEventSynchronize(m->pin2_done[b]));
        const double *h = m->pin2[b];
        std::memcpy(r.f + q0, h + 3 * S, sizeof(double) * nn);
        if (r.v)
            std::memcpy(r.v + q0, h + 4 * S, sizeof(double) * nn);
        if (r.grad)
            std::memcpy(r.grad + 3 * q0, h + 5 * S, sizeof(double) * 3 * nn);
        if (r.tx)
            std::memcpy(r.tx + 3 * q0, h + 8 * S, sizeof(double) * 3 * nn);
        if (r.ty)
            std::memcpy(r.ty + 3 * q0, h + 11 * S, sizeof(double) * 3 * nn);
        return GPX_OK;
    };
    for (size_t i = 0; i < nslices; ++i) {
        const int b = (int)(i & 1);
        const size_t q0 = i * S, nn = std::min(S, r.nq - q0);
        if (i >= 2 && (rc = deliver(i - 2)))  // frees pinned buffer b; the device is busy with slice i-1 meanwhile
            break;
        double *h = m->pin2[b];
        std::memcpy(h, r.qx + q0, sizeof(double) * nn);
        std::memcpy(h + S, r.qy + q0, sizeof(double) * nn);
        std::memcpy(h + 2 * S, r.qz + q0, sizeof(double) * nn);
        HIPCHK(hipMemcpyAsync(d, h, sizeof(double) * 3 * S, hipMemcpyHostToDevice, s));
        if ((rc = evaluate_locked(m, nn, d, d + S, d + 2 * S, d + 3 * S, r.v ? d + 4 * S : nullptr,
                                  r.grad ? d + 5 * S : nullptr, r.tx ? d + 8 * S : nullptr,
                                  r.ty ? d + 11 * S : nullptr, s)))
            break;
        HIPCHK(hipMemcpyAsync(h + 3 * S, d + 3 * S, sizeof(double) * S * (r.v ? 2 : 1), hipMemcpyDeviceToHost, s));
        if (r.grad)
            HIPCHK(hipMemcpyAsync(h + 5 * S, d + 5 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        if (r.tx)
            HIPCHK(hipMemcpyAsync(h + 8 * S, d + 8 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        if (r.ty)
            HIPCHK(hipMemcpyAsync(h + 11 * S, d + 11 * S, sizeof(double) * 3 * S, hipMemcpyDeviceToHost, s));
        HIPCHK(hipEventRecord(m->pin2_done[b], s));
    }
    if (rc) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    for (size_t i = nslices >= 2 ? nslices - 2 : 0; i < nslices; ++i)
        if ((rc = deliver(i)))
            return rc;
    return GPX_OK;
}

constexpr size_t COMBINE_MAX_NQ = 4096;  // larger calls fill the device on their own

extern "C" int gpx_model_evaluate(const gpx_model *cm, size_t nq, const double *qx, const double *qy,
                                  const double *qz, double *f, double *v, double *grad, double *tx, double *ty)
{
    int rc = check_query(cm, nq, qx, qy, qz, f);
    if (rc)
        return rc;
    gpx_model *m = const_cast<gpx_model *>(cm);
    gpx_pending req{nq, qx, qy, qz, f, v, grad, tx, ty};
    if (nq > COMBINE_MAX_NQ)
        return run_large(m, req);
    // flat combining: the calling thread either becomes the leader of a batch or waits for one
    std::unique_lock<std::mutex> lk(m->qmtx);
    m->pending.push_back(&req);
    while (!req.done) {
        if (!m->leader_active) {
            m->leader_active = true;
            std::vector<gpx_pending *> batch;
            batch.swap(m->pending);
            lk.unlock();
            const int brc = run_requests(m, batch);
            const std::string berr = brc ? g_err : std::string();
            lk.lock();
            for (gpx_pending *p : batch) {
                p->rc = brc;
                p->err = berr;
                p->done = true;
            }
            m->leader_active = false;
            m->qcv.notify_all();
        } else {
            m->qcv.wait(lk);
        }
    }
    lk.unlock();
    if (req.rc)
        g_err = req.err;
    return req.rc;
}

extern "C" int gpx_model_sample_surface(const gpx_model *cm, size_t nq, const double *qx, const double *qy,
                                        const double *qz, double f_tol, size_t capacity, int64_t *idx, double *f,
                                        double *v, size_t *n_out)
{
    if (!n_out || !idx)
        return fail(GPX_E_NULL, "Empty output pointer");
    *n_out = 0;
    int rc = check_query(cm, nq, qx, qy, qz, f);
    if (rc)
        return rc;
    if (!(f_tol >= 0.0))
        return fail(GPX_E_BAD_ARG, "f_tol must be non-negative");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    const size_t cap = std::min(capacity, nq);
    const size_t nb = (nq + 255) / 256;
    // device staging: qx qy qz f_all | compacted sx sy sz fs vs (cap each) | idx (cap int64) | block counters | total
    const size_t doubles = nq * 4 + cap * 5 + cap + nb / 2 + 4;
    if ((rc = ensure((void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    double *d = m->ws_host_io;
    double *dqx = d, *dqy = d + nq, *dqz = d + 2 * nq, *dfa = d + 3 * nq;
    double *sx = d + 4 * nq, *sy = sx + cap, *sz = sy + cap, *fs = sz + cap, *vs = fs + cap;
    long long *didx = (long long *)(vs + cap);
    unsigned *bc = (unsigned *)(didx + cap);
    unsigned long long *dtotal = (unsigned long long *)(d + doubles - 2);
    HIPCHK(hipMemcpyAsync(dqx, qx, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dqy, qy, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dqz, qz, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    if ((rc = evaluate_locked(m, nq, dqx, dqy, dqz, dfa, nullptr, nullptr, nullptr, nullptr, s)))
        return rc;
    launch_surface_select((long)nq, dfa, f_tol, bc, dtotal, cap, dqx, dqy, dqz, didx, fs, sx, sy, sz, s);
    unsigned long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, dtotal, sizeof(total), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_out = (size_t)total;
    const size_t ns = std::min((size_t)total, cap);
    if (ns > 0) {
        if (v) {  // variance of the survivors only (their mean is recomputed by the same call; it is cheap)
            if ((rc = evaluate_locked(m, ns, sx, sy, sz, fs, vs, nullptr, nullptr, nullptr, s)))
                return rc;
            HIPCHK(hipMemcpyAsync(v, vs, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
        }
        HIPCHK(hipMemcpyAsync(f, fs, sizeof(double) * ns, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(idx, didx, sizeof(int64_t) * ns, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    if (total > cap)
        return fail(GPX_E_SIZE_MISMATCH, "more surface points than capacity");
    return GPX_OK;
}

// ---- AtlasBase::project, batched and device-resident (reference include/atlas/atlas.hpp:201-276) -------------
extern "C" int gpx_model_project(const gpx_model *cm, size_t nq, const double *x, const double *y, const double *z,
                                 const double *normal, const gpx_project_options *opt, double *out_xyz, double *out_f,
                                 int32_t *out_iter, int32_t *out_status)
{
    if (!out_xyz || !normal)
        return fail(GPX_E_NULL, "Empty data pointer");
    int rc = check_query(cm, nq, x, y, z, out_xyz);
    if (rc)
        return rc;
    gpx_project_options o{1e-2, 1e-7, 0.001, 500, {0, 0, 0}};
    if (opt)
        o = *opt;
    if (!(o.f_tol >= 0.0) || !(o.improve_tol >= 0.0) || o.max_iter < 0 || !std::isfinite(o.step_mul))
        return fail(GPX_E_BAD_ARG, "project options: tolerances and max_iter must be non-negative, step_mul finite");
    gpx_model *m = const_cast<gpx_model *>(cm);
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    hipStream_t s = m->stream;
    // device state: cx cy cz f_cur f_new (nq each) | g grad_new (3 nq each) | iter status (nq ints each) | active
    const size_t doubles = nq * 5 + nq * 6 + nq + 2;
    if ((rc = ensure((void **)&m->ws_host_io, &m->ws_host_io_doubles, sizeof(double) * doubles)))
        return rc;
    double *d = m->ws_host_io;
    double *cx = d, *cy = d + nq, *cz = d + 2 * nq, *fcur = d + 3 * nq, *fnew = d + 4 * nq;
    double *g = d + 5 * nq, *gnew = d + 8 * nq;
    int *iter = (int *)(d + 11 * nq), *status = iter + nq;
    unsigned *active = (unsigned *)(d + 12 * nq);
    HIPCHK(hipMemcpyAsync(cx, x, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(cy, y, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(cz, z, sizeof(double) * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(g, normal, sizeof(double) * 3 * nq, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemsetAsync(iter, 0, sizeof(int) * 2 * nq, s));
    bool fused = true;
    if (const char *e = std::getenv("GPX_PROJECT_FUSED"))
        fused = std::atoi(e) != 0;
    if (fused)  // the whole loop in one launch when the model fits the LDS (N <= 4096)
        fused = launch_project_fused(m->cov, m->npad, m->d_x, m->d_y, m->d_z, m->d_alpha, (long)nq, o.f_tol,
                                     o.improve_tol, o.step_mul, o.max_iter, cx, cy, cz, g, fcur, iter, status, s);
    if (!fused) {
        // the mean at the start points (:225 of the first iteration; also the answer when max_iter == 0)
        if ((rc = evaluate_locked(m, nq, cx, cy, cz, fcur, nullptr, nullptr, nullptr, nullptr, s)))
            return rc;
        for (int it = 0; it < o.max_iter; ++it) {
            HIPCHK(hipMemsetAsync(active, 0, sizeof(unsigned), s));
            launch_project_pre((long)nq, o.f_tol, o.step_mul, cx, cy, cz, g, fcur, status, s);
            if ((rc = evaluate_locked(m, nq, cx, cy, cz, fnew, nullptr, gnew, nullptr, nullptr, s)))
                return rc;
            launch_project_post((long)nq, o.improve_tol, o.max_iter, fnew, gnew, g, fcur, iter, status, active, s);
            if ((it & 7) == 7 || it + 1 == o.max_iter) {  // look at the device only every 8 iterations
                unsigned left = 0;
                HIPCHK(hipMemcpyAsync(&left, active, sizeof(left), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (left == 0)
                    break;
            }
        }
    }
    std::vector<double> hc(3 * nq);
    std::vector<int> hs(2 * nq);
    HIPCHK(hipMemcpyAsync(hc.data(), cx, sizeof(double) * 3 * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(hs.data(), iter, sizeof(int) * 2 * nq, hipMemcpyDeviceToHost, s));
    if (out_f)
        HIPCHK(hipMemcpyAsync(out_f, fcur, sizeof(double) * nq, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    for (size_t i = 0; i < nq; ++i) {
        out_xyz[3 * i] = hc[i];
        out_xyz[3 * i + 1] = hc[nq + i];
        out_xyz[3 * i + 2] = hc[2 * nq + i];
        int st = hs[nq + i];
        if (st == 0)
            st = 3;  // max_iter == 0: the loop of the reference is never entered
        if (out_iter)
            out_iter[i] = hs[i];
        if (out_status)
            out_status[i] = st;
    }
    return GPX_OK;
}

extern "C" int gpx_model_prepare_variance(gpx_model *m)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    if (!m->ready)
        return fail(GPX_E_STATE, "model is not ready");
    std::lock_guard<std::mutex> lk(m->mtx);
    HIPCHK(hipSetDevice(m->device));
    return build_inverse(m);
}

extern "C" int gpx_model_sync(const gpx_model *m)
{
    if (!m)
        return fail(GPX_E_NULL, "Empty Model pointer");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    return GPX_OK;
}

static void resolve_eval_stats(gpx_model *m)
{
    if (!m->stats_eval_pending)
        return;
    if (hipStreamSynchronize(m->stream) != hipSuccess)
        return;
    float ms;
    if (hipEventElapsedTime(&ms, m->ev[EV_M0], m->ev[EV_M1]) == hipSuccess)
        m->stats.t_mean_ms = ms;
    m->stats.t_var_ms = 0;
    if (m->eval_had_var && hipEventElapsedTime(&ms, m->ev[EV_M1], m->ev[EV_V1]) == hipSuccess)
        m->stats.t_var_ms = ms;
    double tg = 0;
    for (size_t i = 0; i < m->gemm_ev_used_var; ++i) {
        const size_t k = m->gemm_ev_used_factor + i;
        if (hipEventElapsedTime(&ms, m->gemm_ev[2 * k], m->gemm_ev[2 * k + 1]) == hipSuccess)
            tg += ms;
    }
    m->stats.t_var_gemm_ms = tg;
    m->stats.var_gemm_launches = (int64_t)m->gemm_ev_used_var;
    m->stats_eval_pending = false;
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
        for (size_t i = 0; i < n; ++i) {
            st[i] = m->hx[i];
            st[np + i] = m->hy[i];
            st[2 * (size_t)np + i] = m->hz[i];
            st[3 * (size_t)np + i] = m->hs2[i];
        }
        double *dd = nullptr;
        void *tt = nullptr, *Kt = nullptr;
        float *tm = nullptr;
        int *tj = nullptr;
        const int nt = np / TILE, ntiles = nt * (nt + 1) / 2;
        HIPCHK(hipMalloc((void **)&dd, sizeof(double) * 4 * np));
        HIPCHK(hipMalloc(&tt, e * 4 * np));
        HIPCHK(hipMalloc(&Kt, e * (size_t)np * np));
        HIPCHK(hipMalloc((void **)&tm, sizeof(float) * ntiles));
        HIPCHK(hipMalloc((void **)&tj, sizeof(int) * 2 * ntiles));
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
        (void)hipFree(dd);
        (void)hipFree(tt);
        (void)hipFree(Kt);
        (void)hipFree(tm);
        (void)hipFree(tj);
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
    if (o.precision == GPX_PREC_MIXED) {  // the committed state of a MIXED model is the fp32 layout
        m->prec = GPX_PREC_F32;
        m->esz = 4;
    }
    rc = alloc_model(m);
    if (rc == GPX_OK) {
        hipError_t e = hipMalloc(&m->X, m->esz * (size_t)m->npad * m->npad);
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
    m->ready = true;
    if (with_variance) {
        if (!m->X)
            return fail(GPX_E_STATE, "inverse factor buffer missing");
        m->has_inverse = true;
        if (m->opt.precision == GPX_PREC_F32_SPLIT) {  // the received blobs are already packed / scaled
            int e2 = 0;
            (void)std::frexp(m->cov.k0 > 0 ? m->cov.k0 : 1.0, &e2);
            m->sk = (float)std::ldexp(1.0, -e2);
            m->x_packed = true;
        }
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
    static thread_local float *tm = nullptr;
    static thread_local int *tj = nullptr;
    static thread_local int cap = 0;
    if (cap < ntiles) {
        if (tm)
            (void)hipFree(tm);
        if (tj)
            (void)hipFree(tj);
        HIPCHK(hipMalloc((void **)&tm, sizeof(float) * ntiles));
        HIPCHK(hipMalloc((void **)&tj, sizeof(int) * (2 * ntiles + 2)));
        cap = ntiles;
    }
    CovHost c = make_cov(*kernel);
    launch_kbuild(precision, c, (int)n, (int)n_padded, d_x, d_y, d_z, d_s2, d_K, tm, tj, (hipStream_t)stream);
    (void)d_rmax;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess)
        return fail(GPX_E_HIP, hipGetErrorString(le));
    return GPX_OK;
}
