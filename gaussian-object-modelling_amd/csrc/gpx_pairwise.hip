// gpx_pairwise.hip -- kernel-matrix construction on gfx950.
//
//   kbuild : K[i][j] = k(|p_i - p_j|) + sigma2_i * delta_ij   (lower block-triangle, 128x128 tiles)
//            replaces buildEuclideanDistanceMatrix + the kernel loop of GPRegressor::create
//            (reference gp_regressor.hpp:132-159, :548-557) and Kpp.maxCoeff() (:135).
//   kqp    : Kqp[q][j] = k(|q - p_j|) for one batch of queries (gp_regressor.hpp:300-303), the
//            B operand of the variance GEMM.
//
// Both are HBM-write-bound: every lane produces 4 consecutive columns and issues one 16-byte
// (fp32) / two 16-byte (fp64) stores, a wave writes 2 x 512 contiguous bytes per instruction.
// Distances are direct differences (dx^2+dy^2+dz^2): exact 0 on the diagonal, never negative
// (documented deviation from the reference's norm expansion).
#include "gpx_cov.hpp"

namespace gpx {

template <typename T>
struct Vec4 {
    T v[4];
};

template <typename T>
__device__ __forceinline__ void store4(T *dst, const T (&o)[4])
{
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<float4 *>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
        *reinterpret_cast<double2 *>(dst) = make_double2(o[0], o[1]);
        *reinterpret_cast<double2 *>(dst + 2) = make_double2(o[2], o[3]);
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

// Per-query fit of the variance contraction, once per query batch: (a_q, b_q) = least-squares line of k against
// u = d^2 over every `stride`-th training point (sums in fp64: the normal equations cancel).  coef rows 0..4 = the
// fit's query-side coefficients for b = {1, p_x, p_y, p_z, |p|^2}: {a_q + b_q |q|^2, -2 b_q q_xyz, b_q} (read by the GEMM
// epilogue), rows 5, 6 = a_q, b_q (read by the kqp kernels).  Queries are rounded to T first: they are the coordinates
// the kqp kernels use.
template <typename T, int KID>
__global__ __launch_bounds__(256) void var_fit_kernel(Cov<T> cov, int n, int stride, const T *__restrict__ px,
                                                      const T *__restrict__ py, const T *__restrict__ pz, long nq_valid,
                                                      long nq_tile, const double *__restrict__ qx,
                                                      const double *__restrict__ qy, const double *__restrict__ qz,
                                                      T *__restrict__ coef, long ldcc)
{
    // 16 lanes share a query: lane `sub` takes the samples sub, sub + 16, ... of the strided list
    const long q = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    if (q >= nq_tile)
        return;
    T fa = T(0), fb = T(0), ax = T(0), ay = T(0), az = T(0);
    if (q < nq_valid) {
        ax = (T)qx[q], ay = (T)qy[q], az = (T)qz[q];
        double s1 = 0, su = 0, suu = 0, sk = 0, suk = 0;
        for (int l = sub * stride; l < n; l += 16 * stride) {
            const T dx = ax - px[l], dy = ay - py[l], dz = az - pz[l];
            const T d2 = dx * dx + dy * dy + dz * dz;
            const double u = (double)d2, kv = (double)cov_k<T, KID>(cov, d2);
            s1 += 1.0;
            su += u;
            suu = fma(u, u, suu);
            sk += kv;
            suk = fma(u, kv, suk);
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) {
            s1 += __shfl_xor(s1, off);
            su += __shfl_xor(su, off);
            suu += __shfl_xor(suu, off);
            sk += __shfl_xor(sk, off);
            suk += __shfl_xor(suk, off);
        }
        const double det = s1 * suu - su * su;
        double b = 0.0;
        if (det > 1e-12 * s1 * suu)
            b = (s1 * suk - su * sk) / det;
        fb = (T)b;
        fa = (T)((sk - (double)fb * su) / s1);
    }
    if (sub != 0)
        return;
    coef[q] = fa + fb * (ax * ax + ay * ay + az * az);
    coef[ldcc + q] = T(-2) * fb * ax;
    coef[2 * ldcc + q] = T(-2) * fb * ay;
    coef[3 * ldcc + q] = T(-2) * fb * az;
    coef[4 * ldcc + q] = fb;
    coef[5 * ldcc + q] = fa;
    coef[6 * ldcc + q] = fb;
}

void launch_var_fit(int prec, const CovHost &h, int n, const void *px, const void *py, const void *pz, long nq_valid,
                    long nq_tile, const double *qx, const double *qy, const double *qz, void *coef, long ldcc,
                    hipStream_t st)
{
    const dim3 grid((unsigned)((nq_tile + 15) / 16));
    const int stride = (n + VAR_FIT_SAMPLES - 1) / VAR_FIT_SAMPLES;
    if (prec == GPX_PREC_F64) {
        Cov<double> c = lower_cov<double>(h);
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_fit_kernel<double, KID>), grid, dim3(256), 0, st, c, n, stride,
                                                  (const double *)px, (const double *)py, (const double *)pz, nq_valid,
                                                  nq_tile, qx, qy, qz, (double *)coef, ldcc));
    } else {
        Cov<float> c = lower_cov<float>(h);
        GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((var_fit_kernel<float, KID>), grid, dim3(256), 0, st, c, n, stride,
                                                  (const float *)px, (const float *)py, (const float *)pz, nq_valid,
                                                  nq_tile, qx, qy, qz, (float *)coef, ldcc));
    }
}

// fab != null: Kqp holds k(d) - (a_q + b_q d^2) with the per-query fit (a_q, b_q) = fab[q], fab[ldcc + q] of
// var_fit_kernel; fab == null: the plain kernel values.
template <typename T, int KID>
__global__ __launch_bounds__(256) void kqp_kernel(Cov<T> cov, int n, int npad, const T *__restrict__ px,
                                                  const T *__restrict__ py, const T *__restrict__ pz,
                                                  long nq_valid, const double *__restrict__ qx,
                                                  const double *__restrict__ qy, const double *__restrict__ qz,
                                                  T *__restrict__ Kqp, const T *__restrict__ fab, long ldcc)
{
    __shared__ T rx[TILE], ry[TILE], rz[TILE], rfa[TILE], rfb[TILE];
    const int tid = threadIdx.x;
    const long q0 = (long)blockIdx.y * TILE;
    if (tid < TILE) {
        long q = q0 + tid;
        bool ok = q < nq_valid;
        rx[tid] = ok ? (T)qx[q] : T(0);
        ry[tid] = ok ? (T)qy[q] : T(0);
        rz[tid] = ok ? (T)qz[q] : T(0);
        rfa[tid] = fab ? fab[q] : T(0);
        rfb[tid] = fab ? fab[ldcc + q] : T(0);
    }
    const int tx = tid & 31, ty = tid >> 5;
    const int gj0 = blockIdx.x * TILE + tx * 4;
    T cx[4], cy[4], cz[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        cx[c] = px[gj0 + c];
        cy[c] = py[gj0 + c];
        cz[c] = pz[gj0 + c];
    }
    __syncthreads();
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        const int li = ty + 8 * r;
        const long q = q0 + li;
        const T ax = rx[li], ay = ry[li], az = rz[li], fa = rfa[li], fb = rfb[li];
        T out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            T dx = ax - cx[c], dy = ay - cy[c], dz = az - cz[c];
            T d2 = dx * dx + dy * dy + dz * dz;
            T kv = cov_k<T, KID>(cov, d2) - (fa + fb * d2);
            out[c] = (q < nq_valid && gj0 + c < n) ? kv : T(0);
        }
        store4<T>(Kqp + (size_t)q * npad + gj0, out);
    }
}

// Row-correction vectors of the variance contraction: out[c][j] = sum_l X[j][l] b_c[l], b = {1, p_x, p_y, p_z, |p|^2}
// over the training points l < n, accumulated in fp64 whatever the types (TX: X, T: points and output).  X is
// lower-triangular: row j is read up to its diagonal.  One workgroup per row; ~N^2/2 reads once per model.
template <typename TX, typename T>
__global__ __launch_bounds__(256) void var_rowcorr_kernel(int n, int np, const TX *__restrict__ X, long ldx,
                                                          const T *__restrict__ px, const T *__restrict__ py,
                                                          const T *__restrict__ pz, T *__restrict__ out)
{
    __shared__ double red[4][VAR_NCORR];
    const int j = blockIdx.x, tid = threadIdx.x;
    const int lend = min(n, j + 1);
    double s[VAR_NCORR] = {0, 0, 0, 0, 0};
    const TX *row = X + (size_t)j * ldx;
    for (int l = tid; l < lend; l += 256) {
        const double xv = (double)row[l];
        const double x = (double)px[l], y = (double)py[l], z = (double)pz[l];
        s[0] += xv;
        s[1] = fma(xv, x, s[1]);
        s[2] = fma(xv, y, s[2]);
        s[3] = fma(xv, z, s[3]);
        s[4] = fma(xv, x * x + y * y + z * z, s[4]);
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
        out[(size_t)tid * np + j] = (T)(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

void launch_var_rowcorr(bool x_is_f64, int prec, int n, int np, const void *X, long ldx, const void *px, const void *py,
                        const void *pz, void *out, hipStream_t st)
{
    if (prec == GPX_PREC_F64)  // fp64 model: X is fp64 too
        hipLaunchKernelGGL((var_rowcorr_kernel<double, double>), dim3(np), dim3(256), 0, st, n, np, (const double *)X, ldx,
                           (const double *)px, (const double *)py, (const double *)pz, (double *)out);
    else if (x_is_f64)
        hipLaunchKernelGGL((var_rowcorr_kernel<double, float>), dim3(np), dim3(256), 0, st, n, np, (const double *)X, ldx,
                           (const float *)px, (const float *)py, (const float *)pz, (float *)out);
    else
        hipLaunchKernelGGL((var_rowcorr_kernel<float, float>), dim3(np), dim3(256), 0, st, n, np, (const float *)X, ldx,
                           (const float *)px, (const float *)py, (const float *)pz, (float *)out);
}

template <typename T>
static void kbuild_t(const CovHost &h, int n, int npad, const void *x, const void *y, const void *z,
                     const void *s2, void *K, float *tmax, int *tij, int first_tile_row, hipStream_t st)
{
    const int nt = npad / TILE;
    const int ntiles = nt * (nt + 1) / 2;
    const int tile0 = first_tile_row * (first_tile_row + 1) / 2;  // row-major enumeration of the lower tiles
    if (tile0 >= ntiles)
        return;
    Cov<T> c = lower_cov<T>(h);
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kbuild_kernel<T, KID>), dim3(ntiles - tile0), dim3(256), 0, st, c, n,
                                              npad, (const T *)x, (const T *)y, (const T *)z, (const T *)s2, (T *)K,
                                              tmax, tij, tile0));
}

void launch_kbuild(int prec, const CovHost &cov, int n, int npad, const void *x, const void *y, const void *z,
                   const void *s2, void *K, float *tmax, int *tij, hipStream_t st, int first_tile_row)
{
    if (prec == GPX_PREC_F64)
        kbuild_t<double>(cov, n, npad, x, y, z, s2, K, tmax, tij, first_tile_row, st);
    else
        kbuild_t<float>(cov, n, npad, x, y, z, s2, K, tmax, tij, first_tile_row, st);
}

void launch_reduce_tilemax(int ntiles, const float *tmax, const int *tij, int *out_ij, hipStream_t st)
{
    hipLaunchKernelGGL(reduce_tilemax_kernel, dim3(1), dim3(256), 0, st, ntiles, tmax, tij, out_ij);
}

template <typename T>
static void kqp_t(const CovHost &h, int n, int npad, const void *px, const void *py, const void *pz,
                  long nq_valid, long nq_tile, const double *qx, const double *qy, const double *qz, void *Kqp,
                  hipStream_t st, int ncols, const void *fab, long ldcc)
{
    Cov<T> c = lower_cov<T>(h);
    dim3 grid((ncols > 0 ? ncols : npad) / TILE, (unsigned)(nq_tile / TILE));
    GPX_DISPATCH_KID(h.id, hipLaunchKernelGGL((kqp_kernel<T, KID>), grid, dim3(256), 0, st, c, n, npad,
                                              (const T *)px, (const T *)py, (const T *)pz, nq_valid, qx, qy, qz,
                                              (T *)Kqp, (const T *)fab, ldcc));
}

void launch_kqp(int prec, const CovHost &cov, int n, int npad, const void *px, const void *py, const void *pz,
                long nq_valid, long nq_tile, const double *qx, const double *qy, const double *qz, void *Kqp,
                hipStream_t st, int ncols, const void *fab, long ldcc)
{
    if (prec == GPX_PREC_F64)
        kqp_t<double>(cov, n, npad, px, py, pz, nq_valid, nq_tile, qx, qy, qz, Kqp, st, ncols, fab, ldcc);
    else
        kqp_t<float>(cov, n, npad, px, py, pz, nq_valid, nq_tile, qx, qy, qz, Kqp, st, ncols, fab, ldcc);
}

}  // namespace gpx
