// gpx_internal.hpp -- declarations shared by the HIP translation units of libgpx.so.
// Device code is written for gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/gpx.h"
#include "gpx_host.hpp"

namespace gpx {

constexpr int TILE = 128;   // order of a diagonal block / GEMM tile edge
constexpr int PANEL = 256;  // padding unit of N, and the narrow outer panel of the factorisation (2 diagonal blocks)
constexpr int WIDE_PANEL = 512;  // optional wider outer panel (GPX_PANEL=512, 4 diagonal blocks) and the workspace width
constexpr int WAVE = 64;
// Terms of the low-rank fit taken out of the kernel operand of the variance contraction.  The fit is a polynomial of
// degree two in s = |q - p|^2, which is rank 14 in (q, p): basis functions of the training point, relative to the
// model's centre c (p' = p - c):  1 | p'_x p'_y p'_z | p'_x^2 p'_y^2 p'_z^2 p'_x p'_y p'_x p'_z p'_y p'_z |
// |p'|^2 p'_x  |p'|^2 p'_y  |p'|^2 p'_z | |p'|^4
constexpr int VAR_NCORR = 14;
constexpr int VAR_NFIT = 3;                      // a_q, b_q, c_q of the fit a + b s + c s^2
constexpr int VAR_NCOEF = VAR_NCORR + VAR_NFIT;  // rows of the per-batch coefficient array (doubles)
constexpr int BLOB_META = 8;  // doubles at the end of state part 0: centre x y z, 1 / (sx sk) of the split operands, weight offset of the fit, reserved

// Kernel attributes such as hipFuncAttributeMaxDynamicSharedMemorySize are PER DEVICE (and per kernel instantiation):
// one flag per device ordinal, run under std::call_once so that concurrent builders on several host threads (or a
// process that drives more than one GPU through gpx_options.device / gpx_model_replicate) each set them exactly once.
// (gpx_host.hpp: the host-side machinery that is also built and run under sanitizers on the CPU)
using gpxh::MAX_DEVICES;
using gpxh::PerDeviceOnce;

// Host-side description of a covariance function, lowered to Cov<T> for the device.
struct CovHost {
    int id;
    double a;   // amplitude: sigma^2 (Gaussian, Matern) or 2*sigma (Laplace)
    double s;   // decay:     1/l^2 (Gaussian), 1/l (Laplace), sqrt3/l, sqrt5/l
    double R;   // thin plate
    double R3;
    double k0;  // k(0)
};
CovHost make_cov(const gpx_kernel &k);

template <typename T>
struct Cov {
    T a, s, R, R3;
};
template <typename T>
inline Cov<T> lower_cov(const CovHost &h)
{
    return Cov<T>{(T)h.a, (T)h.s, (T)h.R, (T)h.R3};
}

// linear index t of a lower-triangular tile enumeration -> (ti, tj), ti >= tj, t = ti (ti + 1) / 2 + tj
__device__ __forceinline__ void tri_decode(int t, int &ti, int &tj)
{
    int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > t)
        --i;
    while ((i + 1) * (i + 2) / 2 <= t)
        ++i;
    ti = i;
    tj = t - i * (i + 1) / 2;
}

// ---- pairwise (kernel-matrix) stages : gpx_pairwise.hip -----------------------------------
// K (lower block-triangle, identity on padding), per-tile maxima of the squared distance.
// returns the number of (tile_max_d2, tile_max_ij) entries it writes (what launch_reduce_tilemax then scans)
int launch_kbuild(int prec, const CovHost &cov, int n, int npad, const void *x, const void *y, const void *z,
                  const void *s2, void *K, float *tile_max_d2, int *tile_max_ij, hipStream_t st,
                  int first_tile_row = 0);  // > 0: only the tile rows from that row block on (rank-n update)
// picks the global maximum of the per-tile maxima -> out_ij[2]
void launch_reduce_tilemax(int ntiles, const float *tile_max_d2, const int *tile_max_ij, int *out_ij,
                           hipStream_t st);
// ---- low-rank fit taken out of the kernel operand of the variance contraction (gpx_eval.hip) ----
// Per query q the B operand holds k'(q, p) = k(d) - (a_q + b_q s + c_q s^2), s = d^2 = |q - p|^2: what a polynomial of
// degree two in s cannot represent.  ANY (a_q, b_q, c_q) gives the same variance in exact arithmetic -- the fit is
// rank 14 in (q, p) and its product with the inverse factor is added back in the GEMM epilogue, in fp64, from 14
// per-model vectors X b_c accumulated in fp64; a good fit shrinks the operand and with it every fp32 error of the
// N-term contraction (rounding of X, accumulation in the matrix core), which all scale with |k'|.  Round 3 measured
// the pieces at N = 16384, thin-plate R = 4 (profiles/r03_tp_fit_probe.txt; variance error / max|v_ref|): degree-one
// fit with the operand formed in fp32 and an fp32 add-back (round 2) 3.9e-5; degree two, operand formed in fp64 and
// rounded once, fp64 add-back 7e-6.  The fit is a weighted least-squares parabola of k against s over a strided
// sample of the training points, so it follows the distances the query actually sees.
// samples of the fit: every stride-th training point, at most VAR_FIT_SAMPLES (register-resident, 8 per lane), default 64.
// Measured (scripts/fit_samples_check.py, variance error / max|v_ref| at N = 16384): 512 / 256 / 128 / 64 / 32 samples give
// thin plate 5.0 / 3.3 / 3.9 / 3.7 / 3.6e-6 and Matern-5/2 1.2 / 1.1 / 1.4 / 1.2 / 1.5e-6 -- the fit is a 3-parameter
// parabola of a smooth function, more samples buy nothing -- while the kernel's time is proportional to them: 29 % of the
// variance time of a C5-shaped model (N = 724, 128^3 queries) at 512 samples, 4 % at 64.
constexpr int VAR_FIT_SAMPLES = 128;
constexpr int VAR_FIT_SAMPLES_DEFAULT = 64;  // GPX_VAR_FIT_SAMPLES (16 .. 128)
// weights 1 / (s + delta) of the least squares, delta = R_max^2 / VAR_FIT_WDELTA_DIV with R_max = Model::R, the largest
// pairwise training distance (0.05 for the node's unit-ball clouds with their exterior sphere of radius 2)
constexpr double VAR_FIT_WDELTA_DIV = 320.0;
// coef (doubles, [VAR_NCOEF][ldcc]) for the queries [0, nq_tile) of a batch (zero from nq_valid on): rows 0..13 the
// query-side coefficients of the 14 basis functions, rows 14..16 a_q, b_q, c_q.
// px, py, pz: the model's fp64 points; cen (device): the model's meta block (centre in [0..2], weight offset in [4]).  op64: everything in fp64 (for the
// operand kernel that forms k - fit in fp64); else centred points and query are rounded to fp32 and so are a, b, c --
// the values the fp32 operand kernel uses, so that the identity fit = sum_c coef_c b_c holds exactly for what it subtracts.
// compact: only rows a_q, b_q, c_q are written ([3][ldcc]); the consumer derives the 14 coefficients with var_fit_coefs()
void launch_var_fit(bool op64, const CovHost &cov, int n, const double *px, const double *py, const double *pz,
                    const double *cen, long nq_valid, long nq_tile, const double *qx, const double *qy,
                    const double *qz, double *coef, long ldcc, hipStream_t st, bool compact = false);
// query-side coefficients of the 14 basis functions (list above) for the fit a + b s + c s^2, s = |q' - p'|^2, from
// (a, b, c) and the centred query q' -- ONE definition, so that the fit kernel and the kernels that derive the coefficients
// themselves produce the same bits
__device__ __forceinline__ void var_fit_coefs(double fa, double fb, double fc, double ax, double ay, double az, double (&cf)[VAR_NCORR])
{
    const double q2 = ax * ax + ay * ay + az * az;
    const double lin = -2.0 * fb - 4.0 * fc * q2, dg = fb + 2.0 * fc * q2;
    cf[0] = fa + q2 * (fb + fc * q2);
    cf[1] = lin * ax;
    cf[2] = lin * ay;
    cf[3] = lin * az;
    cf[4] = dg + 4.0 * fc * ax * ax;
    cf[5] = dg + 4.0 * fc * ay * ay;
    cf[6] = dg + 4.0 * fc * az * az;
    cf[7] = 8.0 * fc * ax * ay;
    cf[8] = 8.0 * fc * ax * az;
    cf[9] = 8.0 * fc * ay * az;
    cf[10] = -4.0 * fc * ax;
    cf[11] = -4.0 * fc * ay;
    cf[12] = -4.0 * fc * az;
    cf[13] = fc;
}
// Kqp[q][j] = k(|q - p_j|) [- (a_q + b_q s + c_q s^2)], q in [0,nq_tile) (rows >= nq_valid and columns >= n are zero).
// compute64: distances, kernel and fit in fp64 from fp64 points px.. (differences are translation invariant: no
// centring), rounded once to the output type; else fp32 arithmetic on the centred fp32 points, queries centred by
// cen before rounding.  out_prec: GPX_PREC_F32 / _F64 = type of Kqp.  fab: rows a_q, b_q, c_q (doubles, stride ldcc) or
// null for the plain kernel values.  accurate_math: the compiler library's sqrt / exp (fp64 models: pinned to the
// reference classes) instead of the fast fp64 forms of gpx_cov.hpp.
void launch_kqp(bool compute64, int out_prec, bool accurate_math, const CovHost &cov, int n, int npad, const void *px,
                const void *py, const void *pz, const double *cen, long nq_valid, long nq_tile, const double *qx,
                const double *qy, const double *qz, void *Kqp, hipStream_t st,
                int ncols = 0,  // > 0: only the first ncols columns are written (columns in the padding are never read)
                const double *fab = nullptr, long ldcc = 0);

// out (doubles, [VAR_NCORR][np]) = X b_c over the training points l < n, fp64 accumulation; X: np x np lower-triangular,
// fp64 if x_is_f64 else fp32.  px..: the fp64 points, p' = p - cen (rounded to fp32 unless op64, as in launch_var_fit).
void launch_var_rowcorr(bool x_is_f64, bool op64, int n, int np, const void *X, long ldx, const double *px,
                        const double *py, const double *pz, const double *cen, double *out, hipStream_t st);

// ---- prediction : gpx_predict.hip -----------------------------------------------------------
// f[q] = sum_j k(|q-p_j|) alpha_j ; grad[q] = sum_j alpha_j k'(.)(q-p_j)  (double outputs).
// NaN into every given output of the queries that have a NaN or infinite coordinate (last launch of an evaluate; gpx_predict.hip)
void launch_poison_nonfinite(long nq, const double *qx, const double *qy, const double *qz, double *f, double *v, double *grad,
                             double *tx, double *ty, hipStream_t st);
// ws: device doubles, at least predict_ws_doubles(nq, n, grad) long.
// plan_nq > 0: the split of the point range over workgroups (and with it the order of the sum) is the one a call of plan_nq
// queries would get, whatever nq is (iso-surface sampling: the same arithmetic for the whole grid, a candidate set, a slab)
size_t predict_ws_doubles(long nq, int n_chunk_src, bool grad, long plan_nq = 0);
void launch_predict(int prec, const CovHost &cov, int n_pad_pts, const void *px, const void *py, const void *pz,
                    const void *alpha, long nq, const double *qx, const double *qy, const double *qz, double *f,
                    double *grad /*nq x 3 row-major or null*/, double *ws, hipStream_t st,
                    int nvalid = 0,  // > 0: points from nvalid on are padding with alpha = 0 and are skipped
                    long plan_nq = 0);
// v[q] = k0 - sum_m partial[m][q]
void launch_var_finish(int prec, double k0, int mtiles, long ldp, const void *partial, long nq, double *v,
                       hipStream_t st);
// survivors of |f| <= tol in query order: idx / f / coordinates compacted (capacity entries at most), *total = count
void launch_surface_select(long nq, const double *f, double tol, unsigned *block_cnt, unsigned long long *total,
                           size_t capacity, const double *qx, const double *qy, const double *qz, long long *idx,
                           double *fs, double *sx, double *sy, double *sz, hipStream_t st,
                           const long long *idx_map = nullptr);  // idx_map: idx[pos] = idx_map[q] instead of q
// fp32 screen in front of that selection (gpx_predict.hip): g[q] = a proved lower bound of |f(q)|; ws: surface_screen_ws_doubles
bool surface_screen_takes(const CovHost &cov);
size_t surface_screen_ws_doubles(int npts);
void launch_surface_screen(const CovHost &cov, int n, int npts, const double *px, const double *py, const double *pz,
                           const double *alpha, const double *cen, long nq, const double *qx, const double *qy,
                           const double *qz, double *g, double *ws, hipStream_t st);
void launch_tangent_basis(long nq, const double *grad, double *tx, double *ty, hipStream_t st);
// AtlasBase::project, one iteration = pre (tolerance test + step) -> mean/gradient at the new points -> post
constexpr int SMALL_EVAL_NP_MAX = 1024;  // one launch for a handful of queries: models up to this padded size
size_t small_eval_scratch_bytes(int nq_max, int npts_max);
void launch_small_eval(int prec, const CovHost &cov, int n, int npts, const double *px, const double *py,
                       const double *pz, const double *alpha, const void *X, const void *dinv, int nq, int nq_max,
                       const double *q, double *f, double *v, double *grad, double *tx, double *ty, void *scratch,
                       hipStream_t st);
bool launch_project_fused(const CovHost &h, int npts, const double *px, const double *py, const double *pz,
                          const double *alpha, long nq, double f_tol, double improve_tol, double step_mul,
                          int max_iter, double *cx, double *cy, double *cz, const double *g, double *f, int *iter,
                          int *status, hipStream_t st);
void launch_project_pre(long nq, double f_tol, double step_mul, double *cx, double *cy, double *cz, const double *g,
                        const double *f_cur, int *status, hipStream_t st);
void launch_project_post(long nq, double improve_tol, int max_iter, const double *f_new, const double *grad_new,
                         double *g, double *f_cur, int *iter, int *status, unsigned *active, hipStream_t st);
// r = y - f - s2*alpha (all double, n entries); also max|r| -> *rmax (atomic, pre-zeroed)
void launch_residual(int n, const double *y, const double *f, const double *s2, const double *alpha, double *r,
                     double *rmax, hipStream_t st);
void launch_axpy_cast(int prec, int n, int npad, double *alpha_d, const void *delta /*T*/, void *alpha_t,
                      hipStream_t st);  // alpha_d += delta ; alpha_t = (T)alpha_d (zero padded)
void launch_cast_vec(int prec, int n, int npad, const double *src, void *dst, hipStream_t st,
                     double offset = 0.0);  // dst[i] = (T)(src[i] - offset), zero from n on
void launch_normalize_rows3(long n, double *g, hipStream_t st);
void launch_cast_d2f(size_t n, const double *src, float *dst, hipStream_t st);
void launch_cast_f2d(size_t n, const float *src, double *dst, hipStream_t st);
// np x np lower block-triangular matrices: tiles above the block diagonal are skipped, or written as zeros (never read)
void launch_cast_lower_f2d(int np, const float *src, double *dst, bool zero_upper, hipStream_t st, int row0 = 0,
                           int row1 = -1);  // rows [row0, row1) only (multiples of 128; -1: to the end)
void launch_cast_lower_d2f(int np, const double *src, float *dst, bool zero_upper, hipStream_t st);

// ---- MFMA GEMM core : gpx_gemm.hip ----------------------------------------------------------
enum GemmEpi { EPI_STORE = 0, EPI_TRSM = 1, EPI_COLSQ = 2 };
struct GemmArgs {
    const void *A = nullptr, *B = nullptr;
    void *C = nullptr;
    long lda = 0, ldb = 0, ldc = 0;
    int M = 0, N = 0, K = 0;        // multiples of TILE (K: of the k-tile)
    long sA = 0, sB = 0, sC = 0;    // batch strides, elements
    int batch = 1;
    int M_last = -1;                // M of the last batch entry (<= M), -1: same
    int k_eq_m = 0;                 // K of an entry equals its M (ragged last entry)
    double alpha = 1.0;
    int beta = 0;                   // 0 or 1
    int nn = 0;                     // 0: C = A * B^T (B is [n][k]); 1: C = A * B (B is [k][n])
    int lower_only = 0;             // square C: compute only tiles with m-tile >= n-tile
    int a_lower = 0;                // A lower-triangular: k < m0 + TILE
    int m_valid = 0;                // EPI_COLSQ, cfg 6: rows of A (= columns of B) that hold data, the rest being the identity / zero
                                    // padding (0 = all): the one-wave fp32 tile leaves the padding's share of the work out
    int b_lower = 0;                // B lower-triangular: nn: k >= n0 ; nt: k < n0 + TILE
    int epi = EPI_STORE;
    int cfg = 0;                    // preferred tile: 0 = 128x128, 2 = 256x256 (falls back if it does not divide); EPI_COLSQ fp32
                                    // also 3 = 128x128 with 64-byte k rows, 6 = one-wave tile (gpx_vargemm.hip)
    void *W = nullptr;              // EPI_TRSM: un-scaled product
    long ldw = 0;
    const void *colscale = nullptr; // EPI_TRSM: C = acc * colscale[n]
    const void *rowweight = nullptr; // EPI_COLSQ: partial[mt][n] = sum_rows acc^2 * rowweight[m]
    void *partial = nullptr;
    long ldp = 0;
    // EPI_COLSQ, fp32 product with the low-rank fit added back (colcoef != null): the epilogue runs in fp64 --
    // w = (double)acc[m][n] + sum_c colcoef[c][n] * rowcorr[c][m], partial64[mt][n] = sum_rows w^2 * rowweight64[m]
    const double *rowcorr = nullptr;     // [VAR_NCORR][ldrc]
    const double *colcoef = nullptr;     // [VAR_NCORR][ldcc]
    const double *rowweight64 = nullptr; // 1/D in fp64
    long ldrc = 0, ldcc = 0;             // (partial then holds doubles)
    int no_pair = 0;  // EPI_COLSQ one-wave tiles: 1 = never the paired launch (which walks half of the row tiles in descending k
                      // WHEN the number of column tiles fills the SIMDs in whole rounds): the order of every sum is then the same
                      // whatever the batch a query sits in (iso-surface sampling)
};
void launch_gemm(int prec, const GemmArgs &g, hipStream_t st);
// cfg 6 (fp32 EPI_COLSQ with a lower-triangular A only): one wave per workgroup, 128 x 128 tile, operands from global memory
// straight into the MFMA fragments -- gpx_vargemm.hip.  launch_gemm routes cfg 6 there when var_w1_fits(), else to cfg 3.
bool var_w1_fits(const GemmArgs &g);
void launch_var_w1(const GemmArgs &g, hipStream_t st);
bool var_w1_f64_fits(const GemmArgs &g);   // the fp64 form: 128 x 64 tile per wave, plain epilogue
void launch_var_w1_f64(const GemmArgs &g, hipStream_t st);
// Rows of the variance operand K' are np + KQP_LDPAD elements apart: with the power-of-two stride np the 128 rows of a
// tile's K' slab and of its X slab fall on the same L2 sets and evict each other (N = 16384: 19.9 GB of L2 misses per
// launch against 7.6 GB with the padded stride, paired launch; profiles/r03_w1_traffic.txt)
constexpr int KQP_LDPAD = 32;
// fp64 C = alpha A B (B in [k][n] form, EPI_STORE, beta = 0) with K >= W1_NN_MIN_K goes to the one-wave kernel as well
// (the three largest levels of the inverse-factor assembly carry 98 % of its flops)
constexpr int W1_NN_MIN_K = 1024;
bool w1_f64_nn_fits(const GemmArgs &g);
void launch_w1_f64_nn(const GemmArgs &g, hipStream_t st);
int gemm_rows_per_partial(int prec, const GemmArgs &g);  // BM of the tile launch_gemm will pick for g
int gemm_tile_m(int cfg);
int gemm_tile_n(int cfg);

// ---- variance contraction of small models : gpx_varcols.hip ----------------------------------------
// One wave per workgroup holds every row fragment of w = X K'^T for 16 CF queries (all rows resident in the AGPRs, the
// triangle of X skipped per 16-row fragment, ONE fp64 epilogue per accumulator fragment) and writes v = k(0) - sum w^2 / D
// directly: no partial sums, no var_finish.  For fp32 contractions with the fit (var_fit) of up to VARCOLS_MAX_N points.
constexpr int VARCOLS_MAX_N = 1024;
constexpr int SPLIT_MIN_N = 1024;  // F32_SPLIT models of up to this many points keep X in fp32: their variance kernel (gpx_varcols16.hip) splits and packs it per call
struct VarColsArgs {
    const float *X = nullptr;  // inverse factor [np][ldx]
    long ldx = 0;
    int n = 0, np = 0;
    const float *Kq = nullptr;  // operand K' = k - fit of the batch, [nq_tile][ldk] (only read when the wave cannot form it itself)
    long ldk = 0;
    const double *rowcorr = nullptr, *colcoef = nullptr, *dinv64 = nullptr;
    long ldrc = 0, ldcc = 0;
    double k0 = 0;
    long nq_valid = 0, nq_tile = 0;
    double *v = nullptr;
    // for the operand formed inside the wave (fp32 arithmetic): covariance function, centred fp32 points, queries, centre
    CovHost cov{};
    bool op64 = false;  // the model's operand is formed in fp64 (thin plate): read it from Kq
    const float *px = nullptr, *py = nullptr, *pz = nullptr;
    const double *qx = nullptr, *qy = nullptr, *qz = nullptr;
    double cen[3] = {0, 0, 0};
    bool compact_coef = false;  // colcoef holds only rows a_q, b_q, c_q (launch_var_fit(compact)); operand formed in the wave only
};
bool var_cols_fits(int n, int np, long ldx, long ldk);
bool var_cols_gen(const VarColsArgs &a);  // true: launch_var_cols forms the operand itself -- no launch_kqp needed
void launch_var_cols(const VarColsArgs &a, hipStream_t st);

// ---- small models in the split-fp16 mode : gpx_varcols16.hip ------------------------------------------
// the same contraction on v_mfma_f32_16x16x32_f16 with hi / lo halves (fp32 accuracy), operand formed in the wave, fit added
// back in fp64; a: as for launch_var_cols with compact coefficients; ws: var_cols16_ws_bytes(n), filled by ..._pack once per call
bool var_cols16_takes(const VarColsArgs &a);
size_t var_cols16_ws_bytes(int n);
void launch_var_cols16_pack(const VarColsArgs &a, float sk, void *ws, hipStream_t st);
void launch_var_cols16(const VarColsArgs &a, float sk, const void *ws, hipStream_t st);

// ---- variance of small fp64 models : gpx_varcols64.hip ----------------------------------------------
// v = k(0) - sum_m (X k_q)_m^2 / D_m for nq queries in one launch: operand formed in the wave, every row fragment resident,
// the triangle of X skipped per 16-row fragment, v written directly (X: np x ldx fp64 lower triangular; px..: the fp64 points)
constexpr int VARCOLS64_MAX_N = 1024;    // points the kernel holds in LDS
constexpr int VARCOLS64_DEFAULT_N = 992;  // models routed to it (the general path is ahead again from ~1000 points)
bool var_cols64_fits(int n, int np, long ldx);
void launch_var_cols64(const CovHost &cov, int n, int np, const double *X, long ldx, const double *px, const double *py,
                       const double *pz, const double *dinv, long nq, const double *qx, const double *qy, const double *qz,
                       double *v, double *xp_ws /* var_cols64_ws_bytes(n) */, hipStream_t st, const double *alpha = nullptr,
                       double *f = nullptr /* with alpha: the mean too, f[q] = sum_p alpha_p k(q, p) */);
size_t var_cols64_ws_bytes(int n);

// ---- split-fp16 variance contraction : gpx_vsplit.hip ------------------------------------------
// in place: X (fp32, np x np) -> packed hi/lo halves with a device-chosen power-of-two scale sx;
// dinv -> w = dinv / (sx sk)^2 (the weights of the plain epilogue); *inv_scale = 1 / (sx sk) (the fp64 epilogue
// brings the accumulators back to true units with it before the fit is added)
void launch_split_prepare(float *X, int np, float *dinv_to_w, float sk, unsigned *amax_bits, hipStream_t st,
                          double *inv_scale = nullptr);
// fab: rows a_q, b_q, c_q (doubles) of the fit or null; compute64: the operand is formed in fp64 from the fp64 points
// px64.., else in fp32 from the centred fp32 points px.. (queries centred by cen)
void launch_kqp_split(bool compute64, const CovHost &cov, float sk, int n, int npad, const void *px, const void *py, const void *pz,
                      const double *px64, const double *py64, const double *pz64, const double *cen, long nq_valid,
                      long nq_tile, const double *qx, const double *qy, const double *qz, void *P, hipStream_t st,
                      const double *fab = nullptr, long ldcc = 0, long ldk = 0);
// colcoef != null: fp64 epilogue (rowcorr, colcoef, dinv64, inv_scale as above; partial holds doubles); else the plain
// fp32 epilogue with the scaled weights w
void launch_vsplit_gemm(const void *Xp, const void *Kp, int np, int nq_tile, const float *w, void *partial, long ldp,
                        hipStream_t st, int m_rows = 0, const double *rowcorr = nullptr, long ldrc = 0,
                        const double *colcoef = nullptr, long ldcc = 0, const double *dinv64 = nullptr,
                        const double *inv_scale = nullptr, long ldk = 0);

// ---- factorisation helpers : gpx_factor.hip -------------------------------------------------
// LDL^T of one TILE x TILE diagonal block in place (strict lower = L, diagonal = D), its unit-lower
// inverse to linv (TILE x TILE, row-major, zeros above the diagonal), d / 1/d vectors, info.
void launch_diag_ldl(int prec, void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk,
                     hipStream_t st, bool narrow = false);  // narrow: 4-wave variant that fits beside a GEMM workgroup
// diagonal blocks blk0 .. nblk-1 lie in the identity padding: linv = I, d = dinv = 1
void launch_identity_blocks(int prec, int blk0, int nblk, void *linv, void *d, void *dinv, hipStream_t st);
void launch_place_diag(int prec, int nblk, const void *linv_blocks, void *X, long ldx, hipStream_t st);
// forward/backward block substitution steps on a vector (T), see gpx_factor.hip
// fwd: step kb of L y = b (b is consumed, y receives block kb); bwd: step kb of L^T x = y.
void launch_fwd_step(int prec, int kb, int nblk, const void *L, long ld, const void *linv_blocks, void *b, void *y,
                     hipStream_t st);
void launch_bwd_step(int prec, int kb, const void *L, long ld, const void *linv_blocks, void *y, void *x,
                     hipStream_t st);
// the same substitution, one launch per direction (workgroup per block row, self-validating entries between them):
// x = (L D L^T)^-1 b, y: scratch of the same length; info[5] = 1 when a wait gave up after wait_ticks of the 100 MHz clock
// (<= 0: the default limit) -- the result is then invalid and the caller redoes the solve with the step kernels
void launch_tri_solve(int prec, int nblk, const void *L, long ld, const void *linv_blocks, const void *dinv,
                      const void *b, void *y, void *x, int *info, hipStream_t st, long long wait_ticks);
void factor_init(int prec);  // per-device one-time kernel attributes (LDS size of diag_ldl)
void launch_scale_vec(int prec, int npad, void *b, const void *dinv, hipStream_t st);

}  // namespace gpx
