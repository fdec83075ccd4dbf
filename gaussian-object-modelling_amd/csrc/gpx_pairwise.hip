// gpx_pairwise.hip -- kernel-matrix construction on gfx950.
//
//   kbuild : K[i][j] = k(|p_i - p_j|) + sigma2_i * delta_ij   (lower block-triangle, 128x128 tiles)
//            replaces buildEuclideanDistanceMatrix + the kernel loop of GPRegressor::create
//            (reference gp_regressor.hpp:132-159, :548-557) and Kpp.maxCoeff() (:135).
//   kqp    : Kqp[q][j] = k(|q - p_j|) [- the per-query fit] for one batch of queries (gp_regressor.hpp:300-303),
//            the B operand of the variance GEMM; var_fit / var_rowcorr: the fit and its per-model row vectors.
//
// Both are HBM-write-bound: every lane produces 4 consecutive columns and issues one 16-byte
// (fp32) / two 16-byte (fp64) stores, a wave writes 2 x 512 contiguous bytes per instruction.
// Distances are direct differences (dx^2+dy^2+dz^2): exact 0 on the diagonal, never negative
// (documented deviation from the reference's norm expansion).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "gpx_cov.hpp"

namespace gpx {

template <typename T>
struct Vec4 {
    T v[4];
};

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f64x2_t = __attribute__((ext_vector_type(2))) double;

// (Measured and closed in round 3, profiles/r03_kqp_store_patterns.txt: non-temporal stores and 1-KiB row segments per wave
// store -- 128 x 256 blocks -- change the traffic pattern, not the time, of kbuild and kqp; the variants were removed.)
template <typename T>
__device__ __forceinline__ void store4(T *dst, const T (&o)[4])
{
    if constexpr (sizeof(T) == 4) {
        const f32x4_t v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4_t *>(dst) = v;
    } else {
        const f64x2_t v0 = {o[0], o[1]}, v1 = {o[2], o[3]};
        *reinterpret_cast<f64x2_t *>(dst) = v0;
        *reinterpret_cast<f64x2_t *>(dst + 2) = v1;
    }
}

template <typename T, int KID>
__global__ __launch_bounds__(256) void kbuild_kernel(Cov<T> cov, int n, int npad, const T *__restrict__ x,
                                                     const T *__restrict__ y, const T *__restrict__ z,
                                                     const T *__restrict__ s2, T *__restrict__ K,
                                                     float *__restrict__ tmax, int *__restrict__ tij, int tile0)
{
    __shared__ T rx[TILE], ry[TILE], rz[TILE], rs[TILE];
    __shared__ float wbest[4];
    __shared__ int wbi[4], wbj[4];
    int ti, tj;
    const int tile = (int)blockIdx.x + tile0;  // tile0 > 0: only the tile rows from some row block on (update)
    tri_decode(tile, ti, tj);
    const int tid = threadIdx.x;
    if (tid < TILE) {
        int gi = ti * TILE + tid;
        rx[tid] = x[gi];
        ry[tid] = y[gi];
        rz[tid] = z[gi];
        rs[tid] = s2[gi];
    }
    const int tx = tid & 31, ty = tid >> 5;
    const int gj0 = tj * TILE + tx * 4;
    T cx[4], cy[4], cz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cx[c] = x[gj0 + c];
        cy[c] = y[gj0 + c];
        cz[c] = z[gj0 + c];
    }
    __syncthreads();
    float best = -1.0f;
    int bi = 0, bj = 0;
    // Interior tiles -- strictly below the block diagonal and clear of the padding, all but ~2 / nt of them -- have no
    // diagonal entry, no identity padding and no index to carry through the loop: the maximum of d2 alone is tracked (one
    // v_max per entry, a handful of compares / selects per row of four for its column).  The general body below costs
    // 4 compare / select instructions per entry and a divergent branch per store for conditions that are uniformly false
    // here; with them the kernel is VALU-bound at 5.2 TB/s (104.7 us at N = 16384, profiles/r03_bench_kernel_stats.txt).
    const bool interior = tj < ti && (ti + 1) * TILE <= n;
    if (interior) {
        int br = 0, bc = 0;  // the row pass in which the lane's running maximum was last raised, and its column there
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int li = ty + 8 * r;
            const T ax = rx[li], ay = ry[li], az = rz[li];
            T out[4];
            float d2f[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const T dx = ax - cx[c], dy = ay - cy[c], dz = az - cz[c];
                const T d2 = dx * dx + dy * dy + dz * dz;
                out[c] = cov_k<T, KID>(cov, d2);
                d2f[c] = (float)d2;
            }
            const float m01 = fmaxf(d2f[0], d2f[1]), m23 = fmaxf(d2f[2], d2f[3]);
            const float m = fmaxf(m01, m23);
            // the first of the four entries that reaches the row's maximum (what the general body's strict comparison keeps),
            // from the values compared -- nothing is recomputed, so no second evaluation has to round the same way
            const int cm = m01 >= m23 ? (d2f[0] >= d2f[1] ? 0 : 1) : (d2f[2] >= d2f[3] ? 2 : 3);
            const bool up = m > best;
            br = up ? r : br;
            bc = up ? cm : bc;
            best = fmaxf(best, m);
            store4<T>(K + (size_t)(ti * TILE + li) * npad + gj0, out);
        }
        bi = ti * TILE + ty + 8 * br, bj = gj0 + bc;
    } else {
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int li = ty + 8 * r;
        const int gi = ti * TILE + li;
        const T ax = rx[li], ay = ry[li], az = rz[li];
        T out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int gj = gj0 + c;
            T dx = ax - cx[c], dy = ay - cy[c], dz = az - cz[c];
            T d2 = dx * dx + dy * dy + dz * dz;
            T kv = cov_k<T, KID>(cov, d2);
            if (gi == gj)
                kv += rs[li];
            if (gi < n && gj < n) {
                if ((float)d2 > best) {
                    best = (float)d2;
                    bi = gi;
                    bj = gj;
                }
            } else {
                kv = (gi == gj) ? T(1) : T(0);  // identity on the padding
            }
            out[c] = kv;
        }
        store4<T>(K + (size_t)gi * npad + gj0, out);
    }
    }
    // block maximum of the squared distance (for Model::R)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off);
        int oi = __shfl_xor(bi, off), oj = __shfl_xor(bj, off);
        if (ob > best) {
            best = ob;
            bi = oi;
            bj = oj;
        }
    }
    if ((tid & 63) == 0) {
        wbest[tid >> 6] = best;
        wbi[tid >> 6] = bi;
        wbj[tid >> 6] = bj;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w)
            if (wbest[w] > best) {
                best = wbest[w];
                bi = wbi[w];
                bj = wbj[w];
            }
        tmax[tile] = best;
        tij[2 * tile] = bi;
        tij[2 * tile + 1] = bj;
    }
}

__global__ __launch_bounds__(256) void reduce_tilemax_kernel(int ntiles, const float *__restrict__ tmax,
                                                             const int *__restrict__ tij, int *__restrict__ out)
{
    __shared__ float sb[256];
    __shared__ int si[256];
    float best = -2.0f;
    int bt = 0;
    for (int t = threadIdx.x; t < ntiles; t += 256)
        if (tmax[t] > best) {
            best = tmax[t];
            bt = t;
        }
    sb[threadIdx.x] = best;
    si[threadIdx.x] = bt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s && sb[threadIdx.x + s] > sb[threadIdx.x]) {
            sb[threadIdx.x] = sb[threadIdx.x + s];
            si[threadIdx.x] = si[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = tij[2 * si[0]];
        out[1] = tij[2 * si[0] + 1];
    }
}

// ---- the per-query fit of the variance contraction, once per query batch ------------------------------------------
// (a, b, c) = weighted least-squares parabola of k against s = d^2 over every `stride`-th training point (<= 128 samples),
// weights 1 / (s + wdelta) (wdelta <= 0: uniform): the rows of the
// inverse factor weigh a query's NEAREST training points most, so that is where the residual should be smallest
// (measured at N = 16384, variance error / max|v_ref|, thin-plate R = 4: uniform 7.1e-6, wdelta = 0.05 3.8e-6;
// Matern-5/2 1.8e-6 -> 8.5e-7; profiles/r03_fit_variants_*.txt).  wdelta is a per-model number (R_max^2 / 320).  The normal
// equations are solved by orthogonalising 1, s, s^2 against the weights; a degenerate direction (all sampled distances
// equal, fewer than three samples) simply gets no coefficient -- any (a, b, c) yields the same variance.
// coef rows 0..13: query-side coefficients of the 14 basis functions of gpx_internal.hpp for
//   a + b s + c s^2,   s = |q'|^2 - 2 q'.p' + |p'|^2   (q', p' relative to the model's centre)
// rows 14..16: a, b, c (read by the operand kernels).
// px.. are the model's fp64 points.  R32 = false: p' = p - cen, q' = q - cen in fp64.  R32 = true (fp32 operand kernel):
// p' = (float)(p - cen) -- the stored centred fp32 points --, q' = (float)(q - cen), and a, b, c are rounded to fp32:
// exactly the numbers that kernel works with, so that fit = sum_c coef_c b_c holds for what it subtracts.
// ONE lane per query: the samples are the same strided training points for every query, so their coordinates are scalar
// loads and the eight weighted moments sum_w {1, s, s^2, s^3, s^4, k, k s, k s^2} accumulate in registers; the
// orthogonal-polynomial solution is formed from the moments.  (Until round 4 sixteen lanes shared a query and orthogonalised
// in three passes over their samples: 72 shuffles per query and coefficient stores from one lane in sixteen -- 3.8 instead
// of 1.5 ms over the eight objects of C5.  What cancellation does to the expanded moments, a few digits of fp64, only moves
// the fit, which any (a, b, c) may be; the variances agree to every printed digit, profiles/r04_fit_lane.txt.)
// COMPACT: only rows a, b, c are written (coef[0 .. 2][ldcc]) -- the small-model kernel that forms its operand in the wave derives
// the 14 query-side coefficients from them with var_fit_coefs() below (gpx_internal.hpp): 24 instead of 136 bytes per query.
template <bool R32, int KID, bool COMPACT>
__global__ __launch_bounds__(256) void var_fit_kernel(Cov<double> cov, int n, int stride, int ns,
                                                           const double *__restrict__ px, const double *__restrict__ py,
                                                           const double *__restrict__ pz, const double *__restrict__ cen,
                                                           long nq_valid, long nq_tile, const double *__restrict__ qx,
                                                           const double *__restrict__ qy, const double *__restrict__ qz,
                                                           double *__restrict__ coef, long ldcc)
{
    constexpr bool P64 = !R32;
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_tile)
        return;
    double fa = 0, fb = 0, fc = 0, ax = 0, ay = 0, az = 0;
    const double c0 = cen[0], c1 = cen[1], c2c = cen[2], wdelta = cen[4];
    if (q < nq_valid) {
        ax = qx[q] - c0, ay = qy[q] - c1, az = qz[q] - c2c;
        if constexpr (!P64)
            ax = (double)(float)ax, ay = (double)(float)ay, az = (double)(float)az;
        double sw = 0, S1 = 0, S2 = 0, S3 = 0, S4 = 0, K0 = 0, K1 = 0, K2 = 0;
        for (int i = 0; i < ns; ++i) {
            const int l = i * stride;  // (the 16-lane form visits (sub + 16 i) stride: the same set of points)
            if (l >= n)
                break;
            double bx = px[l] - c0, by = py[l] - c1, bz = pz[l] - c2c;
            if constexpr (!P64)
                bx = (double)(float)bx, by = (double)(float)by, bz = (double)(float)bz;
            const double dx = ax - bx, dy = ay - by, dz = az - bz;
            const double u = dx * dx + dy * dy + dz * dz;
            const double k = cov_k<double, KID, MathFast>(cov, u + 1e-300);
            const double w = wdelta > 0.0 ? 1.0 / (u + wdelta) : 1.0;
            const double wu = w * u, wu2 = wu * u, wk = w * k;
            sw += w;
            S1 += wu;
            S2 += wu2;
            S3 = fma(wu2, u, S3);
            S4 = fma(wu2 * u, u, S4);
            K0 += wk;
            K1 = fma(wk, u, K1);
            K2 = fma(wk * u, u, K2);
        }
        // orthogonal polynomials under the weights: p0 = 1, p1 = s - m1, p2 = s^2 - al p1 - be, from the moments
        const double m1 = S1 / sw, a0 = K0 / sw;
        const double s11 = S2 - m1 * S1, s1k = K1 - m1 * K0, s21 = S3 - m1 * S2, s20 = S2;
        const double scale2 = S1 * S1 / sw;
        const bool ok1 = s11 > 1e-10 * scale2 && s11 > 0.0;
        const double b1 = ok1 ? s1k / s11 : 0.0, al = ok1 ? s21 / s11 : 0.0, be = s20 / sw;
        const double k0c = al * m1 - be;  // p2 = s^2 - al s + k0c
        const double s22 = S4 + al * al * S2 + k0c * k0c * sw - 2.0 * al * S3 + 2.0 * k0c * S2 - 2.0 * al * k0c * S1;
        const double s2k = K2 - al * K1 + k0c * K0;
        // (s22 is a difference of O(S4) terms: below 1e-9 of them it is rounding, not a direction)
        const bool ok2 = ok1 && s22 > 1e-10 * s20 * be && s22 > 1e-9 * S4 && s22 > 0.0;
        const double c2 = ok2 ? s2k / s22 : 0.0;
        fa = a0 - b1 * m1 + c2 * (al * m1 - be);
        fb = b1 - c2 * al;
        fc = c2;
        if constexpr (!P64)
            fa = (double)(float)fa, fb = (double)(float)fb, fc = (double)(float)fc;
    }
    if constexpr (COMPACT) {
        coef[q] = fa;
        coef[ldcc + q] = fb;
        coef[2 * ldcc + q] = fc;
    } else {
        double cf[VAR_NCORR];
        var_fit_coefs(fa, fb, fc, ax, ay, az, cf);
#pragma unroll
        for (int c = 0; c < VAR_NCORR; ++c)
            coef[(size_t)c * ldcc + q] = cf[c];
        coef[14 * ldcc + q] = fa;
        coef[15 * ldcc + q] = fb;
        coef[16 * ldcc + q] = fc;
    }
}

void launch_var_fit(bool op64, const CovHost &h, int n, const double *px, const double *py, const double *pz,
                    const double *cen, long nq_valid, long nq_tile, const double *qx, const double *qy,
                    const double *qz, double *coef, long ldcc, hipStream_t st, bool compact)
{
    constexpr int nsamp = VAR_FIT_SAMPLES_DEFAULT;  // strided training points per query (sweep of 16 .. 128: profiles/r03_fit_variants_*.txt)
    // Small models (the small-model variance kernel's range): 32 samples.  The fit kernel's time is proportional to the
    // samples and independent of N -- 0.19 ms for 2^21 queries at 32 samples, 8 % of the variance stage at N = 277 -- while
    // the accuracy is not (see gpx_internal.hpp: 3.6 / 3.7e-6 and 1.5 / 1.2e-6 at 32 / 64 samples).
    const int ns = n <= VARCOLS_MAX_N ? std::min(nsamp, 32) : nsamp;
    const int stride = (n + ns - 1) / ns;
    Cov<double> c = lower_cov<double>(h);
    const int ns16 = (ns + 15) / 16 * 16;  // (the sample counts are multiples of 16: what the 16-lane form of rounds 2-3 visited)
    const dim3 g1((unsigned)((nq_tile + 255) / 256));
    if (op64) {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_fit_kernel<false, KID, false>), g1, dim3(256), 0, st, c, n, stride, ns16, px, py, pz,
                                                  cen, nq_valid, nq_tile, qx, qy, qz, coef, ldcc));
    } else if (compact) {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_fit_kernel<true, KID, true>), g1, dim3(256), 0, st, c, n, stride, ns16, px, py, pz,
                                                  cen, nq_valid, nq_tile, qx, qy, qz, coef, ldcc));
    } else {
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_fit_kernel<true, KID, false>), g1, dim3(256), 0, st, c, n, stride, ns16, px, py, pz,
                                                  cen, nq_valid, nq_tile, qx, qy, qz, coef, ldcc));
    }
}

// ---- the kernel operand of one variance batch ---------------------------------------------------------------------
// TC: arithmetic of distances, kernel and fit (double: from the fp64 points as they are -- differences do not see the
// position of the cloud; float: from the centred fp32 points, the queries centred before rounding); TO: what is stored.
// With TC = double, TO = float the residual k - fit is formed in fp64 and rounded ONCE: its error is 6e-8 of the
// (small) residual, where forming k and the fit separately in fp32 costs 6e-8 of k(0) each (measured: 1.2e-7 k(0) of
// variance error from the operand alone at N = 16384 thin-plate, against 2e-9 -- profiles/r03_tp_fit_probe.txt).
template <typename TC, typename TO, int KID, typename M>
__global__ __launch_bounds__(256) void kqp_kernel(Cov<TC> cov, int n, int npad, const TC *__restrict__ px,
                                                  const TC *__restrict__ py, const TC *__restrict__ pz,
                                                  const double *__restrict__ cen, long nq_valid,
                                                  const double *__restrict__ qx, const double *__restrict__ qy,
                                                  const double *__restrict__ qz, TO *__restrict__ Kqp,
                                                  const double *__restrict__ fab, long ldcc)
{
    __shared__ TC rx[TILE], ry[TILE], rz[TILE], rfa[TILE], rfb[TILE], rfc[TILE];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.y * TILE;
    if (tid < TILE) {
        long q = q0 + tid;
        bool ok = q < nq_valid;
        // fp32 arithmetic works in coordinates relative to the model's centre (the fp32 points are stored that way)
        const double c0 = sizeof(TC) == 4 ? cen[0] : 0.0, c1 = sizeof(TC) == 4 ? cen[1] : 0.0,
                     c2 = sizeof(TC) == 4 ? cen[2] : 0.0;
        rx[tid] = ok ? (TC)(qx[q] - c0) : TC(0);
        ry[tid] = ok ? (TC)(qy[q] - c1) : TC(0);
        rz[tid] = ok ? (TC)(qz[q] - c2) : TC(0);
        rfa[tid] = fab ? (TC)fab[q] : TC(0);
        rfb[tid] = fab ? (TC)fab[ldcc + q] : TC(0);
        rfc[tid] = fab ? (TC)fab[2 * ldcc + q] : TC(0);
    }
    constexpr int LANES_X = 32, ROWS_PER_PASS = 256 / LANES_X, PASSES = TILE / ROWS_PER_PASS;
    const int tx = tid & (LANES_X - 1), ty = tid / LANES_X;
    const int gj0 = blockIdx.x * (4 * LANES_X) + tx * 4;
    TC cx[4], cy[4], cz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cx[c] = px[gj0 + c];
        cy[c] = py[gj0 + c];
        cz[c] = pz[gj0 + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < PASSES; ++r) {
        const int li = ty + ROWS_PER_PASS * r;
        const long q = q0 + li;
        const TC ax = rx[li], ay = ry[li], az = rz[li], fa = rfa[li], fb = rfb[li], fc = rfc[li];
        TO out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            TC dx = ax - cx[c], dy = ay - cy[c], dz = az - cz[c];
            TC d2 = dx * dx + dy * dy + dz * dz;
            TC kv;
            if constexpr (sizeof(TC) == 8 && std::is_same<M, MathFast>::value)
                kv = cov_k<TC, KID, M>(cov, d2 + 1e-300);  // the fast fp64 sqrt wants a positive argument
            else
                kv = cov_k<TC, KID, M>(cov, d2);
            kv -= fa + d2 * (fb + fc * d2);
            out[c] = (q < nq_valid && gj0 + c < n) ? (TO)kv : TO(0);
        }
        store4<TO>(Kqp + (size_t)q * npad + gj0, out);
    }
}

// Row-correction vectors of the variance contraction: out[c][j] = sum_l X[j][l] b_c(p'_l) for the 14 basis functions of
// gpx_internal.hpp over the training points l < n, accumulated in fp64 (TX: type of X; px..: the model's fp64 points,
// p' = p - cen, rounded to fp32 when R32 -- the coordinates the fp32 operand kernel uses).  X is lower-triangular: row j
// is read up to its diagonal.  One workgroup per row; ~N^2/2 reads once per model.
template <typename TX, bool R32>
__global__ __launch_bounds__(256) void var_rowcorr_kernel(int n, int np, const TX *__restrict__ X, long ldx,
                                                          const double *__restrict__ px, const double *__restrict__ py,
                                                          const double *__restrict__ pz, const double *__restrict__ cen,
                                                          double *__restrict__ out)
{
    __shared__ double red[4][VAR_NCORR];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int lend = min(n, j + 1);
    const double c0 = cen[0], c1 = cen[1], c2 = cen[2];
    double s[VAR_NCORR];
#pragma unroll
    for (int c = 0; c < VAR_NCORR; ++c)
        s[c] = 0.0;
    const TX *row = X + (size_t)j * ldx;
    for (int l = tid; l < lend; l += 256) {
        const double xv = (double)row[l];
        double x = px[l] - c0, y = py[l] - c1, z = pz[l] - c2;
        if constexpr (R32)
            x = (double)(float)x, y = (double)(float)y, z = (double)(float)z;
        const double r2 = x * x + y * y + z * z;
        const double xx = xv * x, xy = xv * y, xz = xv * z, xr = xv * r2;
        s[0] += xv;
        s[1] += xx;
        s[2] += xy;
        s[3] += xz;
        s[4] = fma(xx, x, s[4]);
        s[5] = fma(xy, y, s[5]);
        s[6] = fma(xz, z, s[6]);
        s[7] = fma(xx, y, s[7]);
        s[8] = fma(xx, z, s[8]);
        s[9] = fma(xy, z, s[9]);
        s[10] = fma(xr, x, s[10]);
        s[11] = fma(xr, y, s[11]);
        s[12] = fma(xr, z, s[12]);
        s[13] = fma(xr, r2, s[13]);
    }
#pragma unroll
    for (int c = 0; c < VAR_NCORR; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s[c] += __shfl_xor(s[c], off);
        if ((tid & 63) == 0)
            red[tid >> 6][c] = s[c];
    }
    __syncthreads();
    if (tid < VAR_NCORR)
        out[(size_t)tid * np + j] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
}

void launch_var_rowcorr(bool x_is_f64, bool op64, int n, int np, const void *X, long ldx, const double *px,
                        const double *py, const double *pz, const double *cen, double *out, hipStream_t st)
{
#define GPX_ROWCORR(TX, R32)                                                                                         \
    hipLaunchKernelGGL((var_rowcorr_kernel<TX, R32>), dim3(np), dim3(256), 0, st, n, np, (const TX *)X, ldx, px, py, pz, \
                       cen, out)
    if (x_is_f64) {
        if (op64)
            GPX_ROWCORR(double, false);
        else
            GPX_ROWCORR(double, true);
    } else {
        if (op64)
            GPX_ROWCORR(float, false);
        else
            GPX_ROWCORR(float, true);
    }
#undef GPX_ROWCORR
}

template <typename T>
static int kbuild_t(const CovHost &h, int n, int npad, const void *x, const void *y, const void *z,
                    const void *s2, void *K, float *tmax, int *tij, int first_tile_row, hipStream_t st)
{
    const int nt = npad / TILE;
    const int ntiles = nt * (nt + 1) / 2;
    const int tile0 = first_tile_row * (first_tile_row + 1) / 2;  // row-major enumeration of the lower tiles
    if (tile0 >= ntiles)
        return 0;
    Cov<T> c = lower_cov<T>(h);
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kbuild_kernel<T, KID>), dim3(ntiles - tile0), dim3(256), 0, st, c, n,
                                              npad, (const T *)x, (const T *)y, (const T *)z, (const T *)s2, (T *)K,
                                              tmax, tij, tile0));
    return ntiles;
}

int launch_kbuild(int prec, const CovHost &cov, int n, int npad, const void *x, const void *y, const void *z,
                  const void *s2, void *K, float *tmax, int *tij, hipStream_t st, int first_tile_row)
{
    if (prec == GPX_PREC_F64)
        return kbuild_t<double>(cov, n, npad, x, y, z, s2, K, tmax, tij, first_tile_row, st);
    return kbuild_t<float>(cov, n, npad, x, y, z, s2, K, tmax, tij, first_tile_row, st);
}

void launch_reduce_tilemax(int ntiles, const float *tmax, const int *tij, int *out_ij, hipStream_t st)
{
    hipLaunchKernelGGL(reduce_tilemax_kernel, dim3(1), dim3(256), 0, st, ntiles, tmax, tij, out_ij);
}

template <typename TC, typename TO, typename M>
static void kqp_t(const CovHost &h, int n, int npad, const void *px, const void *py, const void *pz, const double *cen,
                  long nq_valid, long nq_tile, const double *qx, const double *qy, const double *qz, void *Kqp,
                  hipStream_t st, int ncols, const double *fab, long ldcc)
{
    Cov<TC> c = lower_cov<TC>(h);
    const int cols = ncols > 0 ? ncols : npad;
    const dim3 grid((cols + 127) / 128, (unsigned)(nq_tile / TILE));
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kqp_kernel<TC, TO, KID, M>), grid, dim3(256), 0, st, c, n, npad, (const TC *)px,
                                              (const TC *)py, (const TC *)pz, cen, nq_valid, qx, qy, qz, (TO *)Kqp, fab, ldcc));
}

void launch_kqp(bool compute64, int out_prec, bool accurate_math, const CovHost &cov, int n, int npad, const void *px,
                const void *py, const void *pz, const double *cen, long nq_valid, long nq_tile, const double *qx,
                const double *qy, const double *qz, void *Kqp, hipStream_t st, int ncols, const double *fab, long ldcc)
{
#define GPX_KQP(TC, TO, M) \
    kqp_t<TC, TO, M>(cov, n, npad, px, py, pz, cen, nq_valid, nq_tile, qx, qy, qz, Kqp, st, ncols, fab, ldcc)
    if (out_prec == GPX_PREC_F64) {  // fp64 models: fp64 throughout, the library's sqrt / exp
        GPX_KQP(double, double, MathAcc);
    } else if (compute64) {
        if (accurate_math)
            GPX_KQP(double, float, MathAcc);
        else
            GPX_KQP(double, float, MathFast);
    } else {
        GPX_KQP(float, float, MathAcc);
    }
#undef GPX_KQP
}

}  // namespace gpx
