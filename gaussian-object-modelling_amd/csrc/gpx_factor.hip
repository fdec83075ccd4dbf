// gpx_factor.hip -- the latency-bound pieces of the blocked LDL^T factorisation and of the
// triangular solves (gfx950).  Replaces Eigen::LDLT<MatrixXd>::compute / ::solve as used by the
// reference (gp_regressor.hpp:161-163); the O(N^3) work is in gpx_gemm.hip.
//
//   diag_ldl   : one 128 x 128 diagonal block, LDS resident: unblocked right-looking LDL^T
//                (no pivoting inside the block: the Eigen rule picks pivots from the ORIGINAL
//                diagonal, so the permutation is applied to the points before kbuild), then the
//                unit-lower inverse of L in place.  The inverse blocks turn every panel solve
//                and every block substitution into matrix products.
//   fwd / bwd  : one launch per block step of L y = b / L^T x = y using the inverse blocks.
#include "gpx_internal.hpp"

namespace gpx {

constexpr int DLD = TILE + 1;  // LDS leading dimension (conflict-free row and column walks)
constexpr int DT = 1024;  // threads of the diagonal-block kernel

// One 128 x 128 diagonal block, LDS resident: unblocked right-looking LDL^T (no pivoting inside the
// block), then the unit-lower inverse of L in place.  1024 threads: the trailing update of step j is
// spread over a 32 x 32 thread grid (<= 16 dependent LDS read-modify-writes per thread and step), the
// inverse uses 8 lanes per row.  A fully register-resident variant (rows in registers, both loops
// unrolled 128x) was tried: hipcc needs 4.5 min for it and spills 2.8 KB per lane.
template <typename T>
__global__ __launch_bounds__(DT) void diag_ldl_kernel(T *__restrict__ A, long lda, T *__restrict__ linv,
                                                      T *__restrict__ d, T *__restrict__ dinv,
                                                      int *__restrict__ info, int blk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *S = reinterpret_cast<T *>(smem_raw);  // [TILE][DLD]
    __shared__ T tmp[TILE];
    __shared__ T Ds[TILE];
    const int tid = threadIdx.x;

    for (int idx = tid; idx < TILE * TILE; idx += DT) {
        const int i = idx >> 7, j = idx & 127;
        S[i * DLD + j] = A[(size_t)i * lda + j];
    }
    // ---- right-looking LDL^T; column j keeps u_ij = l_ij * d_j until the final scaling ----
    const int ti = tid >> 5, tk = tid & 31;
    int nneg = 0;
    for (int j = 0; j < TILE; ++j) {
        __syncthreads();
        const T dj = S[j * DLD + j];
        if (tid == 0) {
            if (!(fabs((double)dj) > 0.0) || !(fabs((double)dj) < 1e300))
                atomicCAS(&info[0], 0, blk * TILE + j + 1);
            if (dj < T(0))
                ++nneg;
        }
        const T inv = T(1) / dj;
        for (int i = j + 1 + ti; i < TILE; i += 32) {
            const T lij = S[i * DLD + j] * inv;
            for (int k = j + 1 + tk; k <= i; k += 32)
                S[i * DLD + k] -= lij * S[k * DLD + j];
        }
    }
    __syncthreads();
    if (tid == 0 && nneg)
        atomicAdd(&info[1], nneg);
    if (tid < TILE) {
        const T dj = S[tid * DLD + tid];
        Ds[tid] = T(1) / dj;
        d[blk * TILE + tid] = dj;
        dinv[blk * TILE + tid] = T(1) / dj;
    }
    __syncthreads();
    // scale to the unit-lower L, write L (strict lower) and D (diagonal) back
    for (int idx = tid; idx < TILE * TILE; idx += DT) {
        const int i = idx >> 7, j = idx & 127;
        if (j < i) {
            const T l = S[i * DLD + j] * Ds[j];
            S[i * DLD + j] = l;
            A[(size_t)i * lda + j] = l;
        } else if (j == i) {
            A[(size_t)i * lda + j] = S[i * DLD + j];
        }
    }
    // ---- in-place inverse of the unit-lower L:  X L = I, columns from right to left ----
    //   X[i][j] = -( L[i][j] + sum_{j<k<i} X[i][k] L[k][j] ),  8 lanes per row
    const int row_off = tid >> 3, h = tid & 7;
    for (int j = TILE - 2; j >= 0; --j) {
        __syncthreads();
        if (tid < TILE && tid > j)
            tmp[tid] = S[tid * DLD + j];
        __syncthreads();
        const int i = j + 1 + row_off;
        T s = T(0);
        if (i < TILE)
            for (int k = j + 1 + h; k < i; k += 8)
                s += S[i * DLD + k] * tmp[k];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (i < TILE && h == 0)
            S[i * DLD + j] = -(tmp[i] + s);
    }
    __syncthreads();
    for (int idx = tid; idx < TILE * TILE; idx += DT) {
        const int i = idx >> 7, j = idx & 127;
        linv[(size_t)blk * TILE * TILE + idx] = j < i ? S[i * DLD + j] : (j == i ? T(1) : T(0));
    }
}

template <typename T>
static void diag_t(void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk, hipStream_t st)
{
    const size_t shmem = (size_t)TILE * DLD * sizeof(T);
    hipLaunchKernelGGL(diag_ldl_kernel<T>, dim3(1), dim3(DT), shmem, st, (T *)Ablk, lda, (T *)linv, (T *)d,
                       (T *)dinv, info, blk);
}

void factor_init(int prec)
{
    if (prec == GPX_PREC_F64)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldl_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TILE * DLD * sizeof(double)));
    else
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldl_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)(TILE * DLD * sizeof(float)));
}

void launch_diag_ldl(int prec, void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk,
                     hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        diag_t<double>(Ablk, lda, linv, d, dinv, info, blk, st);
    else
        diag_t<float>(Ablk, lda, linv, d, dinv, info, blk, st);
}

template <typename T>
__global__ __launch_bounds__(256) void place_diag_kernel(const T *__restrict__ linv, T *__restrict__ X, long ldx)
{
    const int blk = blockIdx.x;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256) {
        const int i = idx >> 7, j = idx & 127;
        X[(size_t)(blk * TILE + i) * ldx + blk * TILE + j] = linv[(size_t)blk * TILE * TILE + idx];
    }
}

void launch_place_diag(int prec, int nblk, const void *linv, void *X, long ldx, hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(place_diag_kernel<double>, dim3(nblk), dim3(256), 0, st, (const double *)linv,
                           (double *)X, ldx);
    else
        hipLaunchKernelGGL(place_diag_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float *)linv, (float *)X,
                           ldx);
}

// y[r] = sum_c M[r][c] v[c] for a 128 x 128 row-major block; one wave per 32 rows.
template <typename T>
__device__ __forceinline__ void block_matvec(const T *__restrict__ M, long ldm, const T *v_lds, T *out_lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T v0 = v_lds[lane], v1 = v_lds[lane + 64];
    T m0[32], m1[32];  // issue all 64 loads of the wave's 32 rows before any reduction (latency-bound otherwise)
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const T *row = M + (size_t)(wave * 32 + rr) * ldm;
        m0[rr] = row[lane];
        m1[rr] = row[lane + 64];
    }
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        T s = m0[rr] * v0 + m1[rr] * v1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            s += __shfl_xor(s, off);
        if (lane == 0)
            out_lds[wave * 32 + rr] = s;
    }
}

// y[c] = sum_r M[r][c] v[r]; threads 0..127 own a column, two halves of r combined through LDS.
template <typename T>
__device__ __forceinline__ void block_matvec_t(const T *__restrict__ M, long ldm, const T *v_lds, T *out_lds,
                                               T *scratch_lds)
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
    T mv[64];
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        mv[rr] = M[(size_t)(hh * 64 + rr) * ldm + c];
    T s = T(0);
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        s += mv[rr] * v_lds[hh * 64 + rr];
    if (hh == 1)
        scratch_lds[c] = s;
    __syncthreads();
    if (hh == 0)
        out_lds[c] = s + scratch_lds[c];
}

// Step kb of L y = b (unit lower L, 128-blocks): y_kb = Linv_kb b_kb ; b_rb -= L[rb,kb] y_kb (rb > kb).
// b is the working right-hand side (updated), y the output.  grid = nblk - kb.
template <typename T>
__global__ __launch_bounds__(256) void fwd_step_kernel(int kb, const T *__restrict__ L, long ld,
                                                       const T *__restrict__ linv, T *__restrict__ b,
                                                       T *__restrict__ y)
{
    __shared__ T bk[TILE], yk[TILE], upd[TILE];
    const int tid = threadIdx.x;
    if (tid < TILE)
        bk[tid] = b[kb * TILE + tid];
    __syncthreads();
    block_matvec<T>(linv + (size_t)kb * TILE * TILE, TILE, bk, yk);
    __syncthreads();
    const int rb = kb + blockIdx.x;
    if (blockIdx.x == 0) {
        if (tid < TILE)
            y[kb * TILE + tid] = yk[tid];
        return;
    }
    block_matvec<T>(L + (size_t)rb * TILE * ld + (size_t)kb * TILE, ld, yk, upd);
    __syncthreads();
    if (tid < TILE)
        b[rb * TILE + tid] -= upd[tid];
}

// Step kb of L^T x = y: x_kb = Linv_kb^T y_kb ; y_cb -= L[kb,cb]^T x_kb (cb < kb).  grid = kb + 1.
template <typename T>
__global__ __launch_bounds__(256) void bwd_step_kernel(int kb, const T *__restrict__ L, long ld,
                                                       const T *__restrict__ linv, T *__restrict__ y,
                                                       T *__restrict__ x)
{
    __shared__ T yk[TILE], xk[TILE], upd[TILE], scratch[TILE];
    const int tid = threadIdx.x;
    if (tid < TILE)
        yk[tid] = y[kb * TILE + tid];
    __syncthreads();
    block_matvec_t<T>(linv + (size_t)kb * TILE * TILE, TILE, yk, xk, scratch);
    __syncthreads();
    const int cb = blockIdx.x;
    if (cb == kb) {
        if (tid < TILE)
            x[kb * TILE + tid] = xk[tid];
        return;
    }
    block_matvec_t<T>(L + (size_t)kb * TILE * ld + (size_t)cb * TILE, ld, xk, upd, scratch);
    __syncthreads();
    if (tid < TILE)
        y[cb * TILE + tid] -= upd[tid];
}

void launch_fwd_step(int prec, int kb, int nblk, const void *L, long ld, const void *linv, void *b, void *y,
                     hipStream_t st)
{
    dim3 grid(nblk - kb);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(fwd_step_kernel<double>, grid, dim3(256), 0, st, kb, (const double *)L, ld,
                           (const double *)linv, (double *)b, (double *)y);
    else
        hipLaunchKernelGGL(fwd_step_kernel<float>, grid, dim3(256), 0, st, kb, (const float *)L, ld,
                           (const float *)linv, (float *)b, (float *)y);
}

void launch_bwd_step(int prec, int kb, const void *L, long ld, const void *linv, void *y, void *x, hipStream_t st)
{
    dim3 grid(kb + 1);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(bwd_step_kernel<double>, grid, dim3(256), 0, st, kb, (const double *)L, ld,
                           (const double *)linv, (double *)y, (double *)x);
    else
        hipLaunchKernelGGL(bwd_step_kernel<float>, grid, dim3(256), 0, st, kb, (const float *)L, ld,
                           (const float *)linv, (float *)y, (float *)x);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_vec_kernel(int n, T *__restrict__ b, const T *__restrict__ s)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        b[i] *= s[i];
}

void launch_scale_vec(int prec, int npad, void *b, const void *dinv, hipStream_t st)
{
    dim3 grid((npad + 255) / 256);
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(scale_vec_kernel<double>, grid, dim3(256), 0, st, npad, (double *)b,
                           (const double *)dinv);
    else
        hipLaunchKernelGGL(scale_vec_kernel<float>, grid, dim3(256), 0, st, npad, (float *)b, (const float *)dinv);
}

}  // namespace gpx
