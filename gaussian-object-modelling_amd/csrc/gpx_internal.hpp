// gpx_internal.hpp -- declarations shared by the HIP translation units of libgpx.so.
// Device code is written for gfx950 (CDNA4, wave64) only.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <mutex>

#include "../../include/gpx.h"

namespace gpx {

constexpr int TILE = 128;   // order of a diagonal block / GEMM tile edge
constexpr int PANEL = 256;  // padding unit of N, and the narrow outer panel of the factorisation (2 diagonal blocks)
constexpr int WIDE_PANEL = 512;  // optional wider outer panel (GPX_PANEL=512, 4 diagonal blocks) and the workspace width
constexpr int WAVE = 64;
// terms of the low-rank fit taken out of the kernel operand of the variance contraction: {1, p_x, p_y, p_z, |p|^2}
constexpr int VAR_NCORR = 5;

// Kernel attributes such as hipFuncAttributeMaxDynamicSharedMemorySize are PER DEVICE (and per kernel instantiation):
// one flag per device ordinal, run under std::call_once so that concurrent builders on several host threads (or a
// process that drives more than one GPU through gpx_options.device / gpx_model_replicate) each set them exactly once.
constexpr int MAX_DEVICES = 64;
struct PerDeviceOnce {
    std::once_flag flag[MAX_DEVICES];
    template <typename F>
    void run(F &&f)
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) {
            f();  // unknown ordinal: set the attribute every time (idempotent)
            return;
        }
        std::call_once(flag[dev], f);
    }
};

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
void launch_kbuild(int prec, const CovHost &cov, int n, int npad, const void *x, const void *y, const void *z,
                   const void *s2, void *K, float *tile_max_d2, int *tile_max_ij, hipStream_t st,
                   int first_tile_row = 0);  // > 0: only the tile rows from that row block on (rank-n update)
// picks the global maximum of the per-tile maxima -> out_ij[2]
void launch_reduce_tilemax(int ntiles, const float *tile_max_d2, const int *tile_max_ij, int *out_ij,
                           hipStream_t st);
// ---- low-rank fit taken out of the kernel operand of the variance contraction (gpx_eval.hip) ----
// Per query q the B operand holds k(d) - (a_q + b_q d^2): what a polynomial of degree one in
// d^2 = |q|^2 - 2 q.p + |p|^2 cannot represent.  ANY (a_q, b_q) gives the same variance in exact arithmetic (the fit is
// rank 5 in (q, p) and its product with the inverse factor is added back in the GEMM epilogue from five per-model
// vectors); a good one shrinks the operand, and with it the fp32 rounding of the N-term contractions, by an order of
// magnitude (thin-plate R = 4, N = 16384: variance error 2.0e-5 k(0) -> 1.1e-6).  (a_q, b_q) is the
// least-squares line of k against d^2 over a strided sample of ~512 training points (all of them for small models),
// so it follows the actual distances of the query, outliers such as the node's exterior sphere included.
constexpr int VAR_FIT_SAMPLES = 512;
constexpr int VAR_NCOEF = VAR_NCORR + 2;  // rows of the per-batch coefficient array: the 5 epilogue coefficients, a_q, b_q
// coef (prec scalars, [VAR_NCOEF][ldcc]) for the queries [0, nq_tile) of a batch (zero from nq_valid on); px, py, pz:
// the n training points as prec scalars
void launch_var_fit(int prec, const CovHost &cov, int n, const void *px, const void *py, const void *pz, long nq_valid,
                    long nq_tile, const double *qx, const double *qy, const double *qz, void *coef, long ldcc,
                    hipStream_t st);
// Kqp[q][j] = k(|q - p_j|) - (a_q + b_q |q - p_j|^2), q in [0,nq_tile) (rows >= nq_valid and columns >= n are zero).
// fab: rows a_q, b_q of launch_var_fit's array (prec scalars, stride ldcc), or null for the plain kernel values.
void launch_kqp(int prec, const CovHost &cov, int n, int npad, const void *px, const void *py, const void *pz,
                long nq_valid, long nq_tile, const double *qx, const double *qy, const double *qz, void *Kqp,
                hipStream_t st,
                int ncols = 0,  // > 0: only the first ncols columns are written (columns in the padding are never read)
                const void *fab = nullptr, long ldcc = 0);

// out (prec scalars, [VAR_NCORR][np]) = X b_c for b = {1, p_x, p_y, p_z, |p|^2} (fp64 accumulation); X: np x np
// lower-triangular, fp64 if x_is_f64 (or prec is F64) else fp32; the points are `prec` scalars
void launch_var_rowcorr(bool x_is_f64, int prec, int n, int np, const void *X, long ldx, const void *px, const void *py,
                        const void *pz, void *out, hipStream_t st);

// ---- prediction : gpx_predict.hip -----------------------------------------------------------
// f[q] = sum_j k(|q-p_j|) alpha_j ; grad[q] = sum_j alpha_j k'(.)(q-p_j)  (double outputs).
// ws: device doubles, at least predict_ws_doubles(nq, n, grad) long.
size_t predict_ws_doubles(long nq, int n_chunk_src, bool grad);
void launch_predict(int prec, const CovHost &cov, int n_pad_pts, const void *px, const void *py, const void *pz,
                    const void *alpha, long nq, const double *qx, const double *qy, const double *qz, double *f,
                    double *grad /*nq x 3 row-major or null*/, double *ws, hipStream_t st);
// v[q] = k0 - sum_m partial[m][q]
void launch_var_finish(int prec, double k0, int mtiles, long ldp, const void *partial, long nq, double *v,
                       hipStream_t st);
// survivors of |f| <= tol in query order: idx / f / coordinates compacted (capacity entries at most), *total = count
void launch_surface_select(long nq, const double *f, double tol, unsigned *block_cnt, unsigned long long *total,
                           size_t capacity, const double *qx, const double *qy, const double *qz, long long *idx,
                           double *fs, double *sx, double *sy, double *sz, hipStream_t st);
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
void launch_cast_vec(int prec, int n, int npad, const double *src, void *dst, hipStream_t st);
void launch_normalize_rows3(long n, double *g, hipStream_t st);
void launch_cast_d2f(size_t n, const double *src, float *dst, hipStream_t st);
void launch_cast_f2d(size_t n, const float *src, double *dst, hipStream_t st);
// np x np lower block-triangular matrices: tiles above the block diagonal are skipped, or written as zeros (never read)
void launch_cast_lower_f2d(int np, const float *src, double *dst, bool zero_upper, hipStream_t st);
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
    int b_lower = 0;                // B lower-triangular: nn: k >= n0 ; nt: k < n0 + TILE
    int epi = EPI_STORE;
    int cfg = 0;                    // preferred tile: 0 = 128x128, 1 = 256x128, 2 = 256x256 (falls back if it does not divide)
    void *W = nullptr;              // EPI_TRSM: un-scaled product
    long ldw = 0;
    const void *colscale = nullptr; // EPI_TRSM: C = acc * colscale[n]
    const void *rowweight = nullptr; // EPI_COLSQ: partial[mt][n] = sum_rows acc^2 * rowweight[m]
    void *partial = nullptr;
    long ldp = 0;
    const void *rowcorr = nullptr;   // EPI_COLSQ: acc[m][n] += sum_c colcoef[c][n] * rowcorr[c][m] before squaring
    const void *colcoef = nullptr;   //   (c < VAR_NCORR; leading dimensions ldrc / ldcc; null: no correction)
    long ldrc = 0, ldcc = 0;
};
void launch_gemm(int prec, const GemmArgs &g, hipStream_t st);
int gemm_rows_per_partial(int prec, const GemmArgs &g);  // BM of the tile launch_gemm will pick for g
int gemm_tile_m(int cfg);
int gemm_tile_n(int cfg);

// ---- split-fp16 variance contraction : gpx_vsplit.hip ------------------------------------------
// in place: X (fp32, np x np) -> packed hi/lo halves with a device-chosen power-of-two scale sx;
// dinv -> w = dinv / (sx sk)^2
//   rowcorr ([VAR_NCORR][np] or null): the row-correction vectors of the fit, scaled by sx sk in place
void launch_split_prepare(float *X, int np, float *dinv_to_w, float sk, unsigned *amax_bits, hipStream_t st,
                          float *rowcorr = nullptr);
void launch_kqp_split(const CovHost &cov, float sk, int n, int npad, const void *px, const void *py, const void *pz,
                      long nq_valid, long nq_tile, const double *qx, const double *qy, const double *qz, void *P,
                      hipStream_t st, const float *fab = nullptr, long ldcc = 0);
void launch_vsplit_gemm(const void *Xp, const void *Kp, int np, int nq_tile, const float *w, float *partial, long ldp,
                        int prefetch, hipStream_t st, int m_rows = 0, const float *rowcorr = nullptr, long ldrc = 0,
                        const float *colcoef = nullptr, long ldcc = 0);

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
// x = (L D L^T)^-1 b, y: scratch of the same length; info[5] = 1 when a wait gave up after spin_limit polls
// (<= 0: the default limit) -- the result is then invalid and the caller redoes the solve with the step kernels
void launch_tri_solve(int prec, int nblk, const void *L, long ld, const void *linv_blocks, const void *dinv,
                      const void *b, void *y, void *x, int *info, hipStream_t st, int spin_limit = 0);
void factor_init(int prec);  // per-device one-time kernel attributes (LDS size of diag_ldl)
void launch_scale_vec(int prec, int npad, void *b, const void *dinv, hipStream_t st);

}  // namespace gpx
