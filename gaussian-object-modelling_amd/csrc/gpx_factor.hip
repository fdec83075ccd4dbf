// gpx_factor.hip -- the latency-bound pieces of the blocked LDL^T factorisation and of the
// triangular solves (gfx950).  Replaces Eigen::LDLT<MatrixXd>::compute / ::solve as used by the
// reference (gp_regressor.hpp:161-163); the O(N^3) work is in gpx_gemm.hip.
//
//   diag_ldl   : one 128 x 128 diagonal block by one workgroup of 8 (4) waves: right-looking LDL^T blocked by 32, no
//                pivoting inside the block (the Eigen rule picks pivots from the ORIGINAL diagonal, so the permutation
//                is applied to the points before kbuild), and the unit-lower inverse of its L.  The inverse blocks turn
//                every panel solve and every block substitution into matrix products.  Round 2 (diag_ldlm_kernel):
//                the 32 x 32 sub-block is factorised AND inverted by rank-1 MFMA updates of accumulators that hold it,
//                the rows below by one product with that inverse, the trailing blocks stay in registers, the 128 x 128
//                inverse is assembled left-looking in the shadow of the panel steps.  identity_blocks: the same result
//                for blocks that lie in the padding.
//   tri_solve  : L y = b and L^T x = D^-1 y in one launch each (a workgroup per block row, results handed on
//                through self-validating entries); fwd / bwd step kernels: the same, one launch per block step.
#include "gpx_internal.hpp"
#include "gpx_blk.hpp"
#include "gpx_diag128.hpp"

// phase timing of diag_ldlm_kernel for scripts/diag_bench.hip (which defines GPX_STAMP); nothing in the library build
#ifndef GPX_STAMP
#define GPX_STAMP(i)
#endif

namespace gpx {
template <typename T, int DT>
static void diagm_t(void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk, hipStream_t st)
{
    hipLaunchKernelGGL((diag_ldlm_kernel<T, DT>), dim3(1), dim3(DT), diagm_shmem_bytes(sizeof(T)), st, (T *)Ablk, lda, (T *)linv,
                       (T *)d, (T *)dinv, info, blk);
}
void factor_init(int prec)
{
    static PerDeviceOnce once64, once32;  // per device, see gpx_internal.hpp
    if (prec == GPX_PREC_F64) {
        once64.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<double, DIAG_THREADS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(double)));
        });
    } else {
        once32.run([] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<float, DIAG_THREADS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(float)));
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldlm_kernel<float, DIAG_THREADS_NARROW>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)diagm_shmem_bytes(sizeof(float)));
        });
    }
}

// narrow: the 4-wave instantiation (fp32 only), for launches that run beside a GEMM on another stream
void launch_diag_ldl(int prec, void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk,
                     hipStream_t st, bool narrow)
{
    if (prec == GPX_PREC_F64)
        diagm_t<double, DIAG_THREADS>(Ablk, lda, linv, d, dinv, info, blk, st);
    else if (narrow)
        diagm_t<float, DIAG_THREADS_NARROW>(Ablk, lda, linv, d, dinv, info, blk, st);
    else
        diagm_t<float, DIAG_THREADS>(Ablk, lda, linv, d, dinv, info, blk, st);
}

// Diagonal blocks that lie entirely in the padding (the kernel matrix is the identity there): L = I, D = 1, inverse = I
template <typename T>
__global__ __launch_bounds__(256) void identity_blocks_kernel(int blk0, T *__restrict__ linv, T *__restrict__ d,
                                                              T *__restrict__ dinv)
{
    const int blk = blk0 + blockIdx.x;
    for (int idx = threadIdx.x; idx < TILE * TILE; idx += 256)
        linv[(size_t)blk * TILE * TILE + idx] = (idx >> 7) == (idx & 127) ? T(1) : T(0);
    if (threadIdx.x < TILE) {
        d[blk * TILE + threadIdx.x] = T(1);
        dinv[blk * TILE + threadIdx.x] = T(1);
    }
}

void launch_identity_blocks(int prec, int blk0, int nblk, void *linv, void *d, void *dinv, hipStream_t st)
{
    if (blk0 >= nblk)
        return;
    if (prec == GPX_PREC_F64)
        hipLaunchKernelGGL(identity_blocks_kernel<double>, dim3(nblk - blk0), dim3(256), 0, st, blk0, (double *)linv,
                           (double *)d, (double *)dinv);
    else
        hipLaunchKernelGGL(identity_blocks_kernel<float>, dim3(nblk - blk0), dim3(256), 0, st, blk0, (float *)linv,
                           (float *)d, (float *)dinv);
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
    // 32 row sums over 64 lanes by recursive halving: a lane gives away the half of the rows its partner keeps, so
    // the wave needs 16 + 8 + 4 + 2 + 1 + 1 = 32 shuffles instead of 32 x 6 for one butterfly per row; afterwards
    // lane l holds row l >> 1
    T s[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
        s[rr] = m0[rr] * v0 + m1[rr] * v1;
#define GPX_HALVE(N, OFF)                                       \
    _Pragma("unroll") for (int i = 0; i < (N); ++i)             \
    {                                                           \
        const bool up = (lane & (OFF)) != 0;                    \
        const T send = up ? s[i] : s[i + (N)];                  \
        const T keep = up ? s[i + (N)] : s[i];                  \
        s[i] = keep + __shfl_xor(send, (OFF));                  \
    }
    GPX_HALVE(16, 32)
    GPX_HALVE(8, 16)
    GPX_HALVE(4, 8)
    GPX_HALVE(2, 4)
    GPX_HALVE(1, 2)
#undef GPX_HALVE
    s[0] += __shfl_xor(s[0], 1);
    if ((lane & 1) == 0)
        out_lds[wave * 32 + (lane >> 1)] = s[0];
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

// ---- the whole substitution in one launch per direction ------------------------------------------------------
// Workgroup i owns block row i of L y = b: it subtracts L(i,k) y_k for k = 0 .. i-1 as the y_k appear and then
// publishes y_i = Linv_i b_i.  The output vector is filled with a sentinel (all-ones NaN, which arithmetic never
// produces) before the launch and every consumer polls the entries it needs with cache-bypassing loads until they
// are no sentinel any more: the data validates itself, no flag, no fence.  Workgroups are dispatched in blockIdx
// order and wait only for lower ones, so the scheme cannot deadlock even when not all of them are resident.  The
// 128 x 128 block of the NEXT step (the last one: Linv_i) is loaded into a second register set before the poll of
// the current one, so a block costs a poll round trip and a register matvec (~2.5 us) -- with the loads behind the
// poll it was 8 us and the 127 blocks of the last row took as long as 256 step launches.  In-order dispatch is what
// the hardware does, not a guarantee of the programming model (other streams, host threads and processes share the
// GPU): a poll that exceeds its time budget gives up with 0 and raises info[5], every workgroup still reaches its
// end, and the host then REDOES the solve with the launch-per-step kernels below (build_model, gpx_build.hip;
// gpx_stats.solve_fallbacks) -- neither a hung GPU nor a failed call.
// The backward direction runs the same scheme bottom-up on L^T, with D^-1 folded into its start.

template <typename T>
__device__ __forceinline__ void block_load(const T *__restrict__ M, long ldm, T (&m0)[32], T (&m1)[32])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int rr = 0; rr < 32; ++rr) {
        const T *row = M + (size_t)(wave * 32 + rr) * ldm;
        m0[rr] = row[lane];
        m1[rr] = row[lane + 64];
    }
}
// out = M v with M in the registers filled by block_load
template <typename T>
__device__ __forceinline__ void block_mv_regs(const T (&m0)[32], const T (&m1)[32], const T *v_lds, T *out_lds)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const T v0 = v_lds[lane], v1 = v_lds[lane + 64];
    T s[32];
#pragma unroll
    for (int rr = 0; rr < 32; ++rr)
        s[rr] = m0[rr] * v0 + m1[rr] * v1;
#define GPX_HALVE(N, OFF)                                       \
    _Pragma("unroll") for (int i = 0; i < (N); ++i)             \
    {                                                           \
        const bool up = (lane & (OFF)) != 0;                    \
        const T send = up ? s[i] : s[i + (N)];                  \
        const T keep = up ? s[i + (N)] : s[i];                  \
        s[i] = keep + __shfl_xor(send, (OFF));                  \
    }
    GPX_HALVE(16, 32)
    GPX_HALVE(8, 16)
    GPX_HALVE(4, 8)
    GPX_HALVE(2, 4)
    GPX_HALVE(1, 2)
#undef GPX_HALVE
    s[0] += __shfl_xor(s[0], 1);
    if ((lane & 1) == 0)
        out_lds[wave * 32 + (lane >> 1)] = s[0];
}
// transposed: thread (c = tid & 127, hh = tid >> 7) holds rows hh*64 .. +63 of column c
template <typename T>
__device__ __forceinline__ void block_load_t(const T *__restrict__ M, long ldm, T (&mv)[64])
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
#pragma unroll
    for (int rr = 0; rr < 64; ++rr)
        mv[rr] = M[(size_t)(hh * 64 + rr) * ldm + c];
}
template <typename T>
__device__ __forceinline__ void block_mv_regs_t(const T (&mv)[64], const T *v_lds, T *out_lds, T *scratch_lds)
{
    const int c = threadIdx.x & 127, hh = threadIdx.x >> 7;
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

// entry *p of the shared vector once it is no sentinel any more (0 and info[5] = 1 once the time budget of the wait -- ticks of
// the constant 100 MHz clock, looked at every 32 polls -- is spent, or at once when the factor itself is void: info[6])
__device__ __forceinline__ float poll_entry(const float *p, int *info, long long wait_ticks)
{
    const unsigned *u = reinterpret_cast<const unsigned *>(p);
    const unsigned long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
        const unsigned b = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != 0xffffffffu)
            return __uint_as_float(b);
        if (wait_ticks <= 0 || ((spins & 31) == 31 && ((long long)(wall_clock64() - t0) > wait_ticks ||
                                                       __hip_atomic_load(&info[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            break;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(&info[5], 1);
    return 0.0f;
}
__device__ __forceinline__ double poll_entry(const double *p, int *info, long long wait_ticks)
{
    const unsigned long long *u = reinterpret_cast<const unsigned long long *>(p);
    const unsigned long long t0 = wall_clock64();
    for (int spins = 0;; ++spins) {
        const unsigned long long b = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b != 0xffffffffffffffffull)
            return __longlong_as_double((long long)b);
        if (wait_ticks <= 0 || ((spins & 31) == 31 && ((long long)(wall_clock64() - t0) > wait_ticks ||
                                                       __hip_atomic_load(&info[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            break;
        __builtin_amdgcn_s_sleep(1);
    }
    atomicExch(&info[5], 1);
    return 0.0;
}
__device__ __forceinline__ void publish_entry(float *p, float v)
{
    __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_entry(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// TRANS = false: L y = b, rows top-down.  TRANS = true: L^T x = D^-1 y, rows bottom-up (scale = 1/D).
// `out` must hold the sentinel in every entry at launch.
template <typename T, bool TRANS>
__global__ __launch_bounds__(256) void tri_solve_kernel(int nblk, const T *__restrict__ L, long ld,
                                                        const T *__restrict__ linv, const T *__restrict__ scale,
                                                        const T *__restrict__ rhs, T *out, int *info, long long wait_ticks)
{
    __shared__ T vk[TILE], upd[TILE], scratch[TILE];
    constexpr int RS = TRANS ? 64 : 32;  // registers of one half block set
    const int tid = threadIdx.x;
    const int i = TRANS ? nblk - 1 - (int)blockIdx.x : (int)blockIdx.x;
    const int nprev = TRANS ? nblk - 1 - i : i;  // blocks this row depends on, in the order they get ready
    T acc = T(0);                                // entry tid of the working right-hand side (tid < TILE)
    if (tid < TILE)
        acc = TRANS ? rhs[i * TILE + tid] * scale[i * TILE + tid] : rhs[i * TILE + tid];
    // block s of the sequence: L(i, s) (TRANS: L(nblk-1-s, i)) for s < nprev, then the inverse diagonal block
    auto src = [&](int s) -> const T * {
        if (s >= nprev)
            return linv + (size_t)i * TILE * TILE;
        return TRANS ? L + (size_t)(nblk - 1 - s) * TILE * ld + (size_t)i * TILE : L + (size_t)i * TILE * ld + (size_t)s * TILE;
    };
    auto ldm = [&](int s) -> long { return s >= nprev ? (long)TILE : ld; };
    T a0[RS], a1[TRANS ? 1 : 32], b0[RS], b1[TRANS ? 1 : 32];
    auto load = [&](int s, T(&r0)[RS], T(&r1)[TRANS ? 1 : 32]) {
        if constexpr (TRANS)
            block_load_t<T>(src(s), ldm(s), r0);
        else
            block_load<T>(src(s), ldm(s), r0, r1);
    };
    auto mv = [&](const T(&r0)[RS], const T(&r1)[TRANS ? 1 : 32]) {
        if constexpr (TRANS)
            block_mv_regs_t<T>(r0, vk, upd, scratch);
        else
            block_mv_regs<T>(r0, r1, vk, upd);
    };
    // one step: the block in (r0, r1) times the published vector block of step s
    auto step = [&](int s, const T(&r0)[RS], const T(&r1)[TRANS ? 1 : 32]) {
        const int k = TRANS ? nblk - 1 - s : s;
        if (tid < TILE)
            vk[tid] = poll_entry(out + k * TILE + tid, info, wait_ticks);
        __syncthreads();
        mv(r0, r1);
        __syncthreads();
        if (tid < TILE)
            acc -= upd[tid];
    };
    load(0, a0, a1);
    bool last_in_b = false;
    for (int s = 0; s < nprev; s += 2) {
        load(s + 1, b0, b1);
        step(s, a0, a1);
        if (s + 1 < nprev) {
            load(s + 2, a0, a1);
            step(s + 1, b0, b1);
        } else {
            last_in_b = true;
        }
    }
    if (tid < TILE)
        vk[tid] = acc;
    __syncthreads();
    if (last_in_b)
        mv(b0, b1);
    else
        mv(a0, a1);
    __syncthreads();
    if (tid < TILE)
        publish_entry(out + i * TILE + tid, upd[tid]);
}

template <typename T>
static void tri_solve_t(int nblk, const void *L, long ld, const void *linv, const void *dinv, const void *b, void *y,
                        void *x, int *info, hipStream_t st, long long wait_ticks)
{
    const size_t words = (size_t)nblk * TILE * (sizeof(T) / 4);
    if ((char *)x == (char *)y + words * 4) {  // adjacent (the model's vectors are): one fill
        (void)hipMemsetD32Async((hipDeviceptr_t)y, (int)0xffffffff, 2 * words, st);
    } else {
        (void)hipMemsetD32Async((hipDeviceptr_t)y, (int)0xffffffff, words, st);
        (void)hipMemsetD32Async((hipDeviceptr_t)x, (int)0xffffffff, words, st);
    }
    hipLaunchKernelGGL((tri_solve_kernel<T, false>), dim3(nblk), dim3(256), 0, st, nblk, (const T *)L, ld, (const T *)linv,
                       (const T *)nullptr, (const T *)b, (T *)y, info, wait_ticks);
    hipLaunchKernelGGL((tri_solve_kernel<T, true>), dim3(nblk), dim3(256), 0, st, nblk, (const T *)L, ld, (const T *)linv,
                       (const T *)dinv, (const T *)y, (T *)x, info, wait_ticks);
}

// x = (L D L^T)^-1 b (y: scratch of the same length), two launches
void launch_tri_solve(int prec, int nblk, const void *L, long ld, const void *linv, const void *dinv, const void *b,
                      void *y, void *x, int *info, hipStream_t st, long long wait_ticks)
{
    if (prec == GPX_PREC_F64)
        tri_solve_t<double>(nblk, L, ld, linv, dinv, b, y, x, info, st, wait_ticks);
    else
        tri_solve_t<float>(nblk, L, ld, linv, dinv, b, y, x, info, st, wait_ticks);
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
