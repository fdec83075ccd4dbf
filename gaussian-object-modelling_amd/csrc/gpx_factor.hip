// gpx_factor.hip -- the latency-bound pieces of the blocked LDL^T factorisation and of the
// triangular solves (gfx950).  Replaces Eigen::LDLT<MatrixXd>::compute / ::solve as used by the
// reference (gp_regressor.hpp:161-163); the O(N^3) work is in gpx_gemm.hip.
//
//   diag_ldl   : one 128 x 128 diagonal block by one workgroup: right-looking LDL^T blocked by 32 (the 32 x 32
//                sub-block is factorised and inverted by one wave in registers), no pivoting inside the block
//                (the Eigen rule picks pivots from the ORIGINAL diagonal, so the permutation is applied to the
//                points before kbuild), then the unit-lower inverse of L assembled from the four 32 x 32
//                inverses.  The inverse blocks turn every panel solve and every block substitution into
//                matrix products.  identity_blocks: the same result for blocks that lie in the padding.
//   fwd / bwd  : one launch per block step of L y = b / L^T x = y using the inverse blocks.
#include "gpx_internal.hpp"

// phase timing of diag_ldl_kernel for scripts/diag_bench.hip (which defines GPX_STAMP); nothing in the library build
#ifndef GPX_STAMP
#define GPX_STAMP(i)
#endif

namespace gpx {

// threads of the diagonal-block kernel: step A keeps two 32-entry rows in registers (fp32 ~180, fp64 ~300 VGPRs)
template <typename T>
struct DiagThreads {
    static constexpr int value = sizeof(T) == 8 ? 256 : 512;
};
constexpr int NB = 32;    // sub-block order inside the 128 x 128 diagonal block
constexpr int PLD = NB + 1;

// value of `v` in lane `src` (wave-uniform, here a compile-time constant after unrolling): v_readlane_b32 puts
// the result in a SCALAR register, so the 1000 broadcasts of step A cost no vector registers
// (with __shfl = ds_bpermute the scheduler kept hundreds in flight and spilled 1.3 KB per lane)
__device__ __forceinline__ float bcast_lane(float v, int src)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
__device__ __forceinline__ double bcast_lane(double v, int src)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), src);
    const int hi = __builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// One 128 x 128 diagonal block: LDL^T without pivoting + the inverse of its unit-lower L, blocked by 32.
//
//   per 32-column panel  A) the 32 x 32 diagonal sub-block is factorised by wave 0 with the rows in registers
//                           (static indices; pivot rows / columns travel by v_readlane, no barrier); in the same
//                           interval wave 1 inverts the PREVIOUS sub-block's L the same way, so the inverses --
//                           needed only by the assembly at the end -- are off the critical path,
//                        B) the rows below: W L11^T = A21 by forward substitution, one thread per row with the
//                           row in registers (L11 is read as LDS broadcasts); L21 = W D^-1,
//                        C) the trailing update A22 -= W L21^T on the (L2-resident) global block, panel
//                           operands in LDS.
//   then the 128 x 128 inverse is assembled from the four 32 x 32 inverses, block column by block column:
//        X[i][j] = -Xd[i] * sum_{j<=k<i} L[i][k] X[k][j].
// ~8 barriers per panel instead of one per column; replaces an unblocked LDS kernel (348 us, then 177 us with
// 1024 threads) that sat on the critical path of the factorisation 128 times at N = 16384.  A fully
// register-resident 128-wide variant was tried too: hipcc needs 4.5 min for it and spills 2.8 KB per lane.
// Until the sub-block inverse moved to the second wave and B became a substitution (it was W = A21 X11^T, which
// needed X11 first) the kernel took 105 us (fp32) / 190 us (fp64).

// step C_ of the row substitution of phase B: prefetch column C_ + 1 of L11, eliminate with column C_
template <typename T, int C_>
struct SubstStep {
    static __device__ __forceinline__ void run(T (&a)[NB], T (&lc)[NB], T (&ln)[NB], const T *lt)
    {
        if constexpr (C_ < NB - 1) {
            constexpr int Q0 = ((C_ + 2) / 4) * 4;  // first aligned quad of column C_ + 1 that is still needed
#pragma unroll
            for (int c2 = Q0; c2 < NB; ++c2)
                ln[c2] = lt[(C_ + 1) * NB + c2];
            const T w = a[C_];
#pragma unroll
            for (int c2 = C_ + 1; c2 < NB; ++c2)
                a[c2] -= w * lc[c2];
            constexpr int NREAD = (NB - Q0) / 4 * (sizeof(T) == 8 ? 2 : 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);       // DS reads of the next column first
            __builtin_amdgcn_sched_group_barrier(0x002, NB - 1 - C_, 0);  // then this column's FMAs
            SubstStep<T, C_ + 1>::run(a, ln, lc, lt);
        }
    }
};

// unit-lower inverse of a 32 x 32 L held as r[c] = L[l][c] in lane l:  X[l][j] = -( L[l][j] + sum_{j<k<l} X[l][k] L[k][j] )
template <typename T>
__device__ __forceinline__ void sub_inverse(const T (&r)[NB], T (&x)[NB], int l)
{
#pragma unroll
    for (int c = 0; c < NB; ++c)
        x[c] = T(0);
#pragma unroll
    for (int j = NB - 2; j >= 0; --j) {
        T s0 = T(0), s1 = T(0);
#pragma unroll
        for (int k = j + 1; k < NB; ++k) {
            const T lkj = bcast_lane(r[j], k);  // L[k][j]
            // x[k] is still 0 for k >= l (set below only when l > k), so no select is needed
            if (k & 1)
                s1 += x[k] * lkj;
            else
                s0 += x[k] * lkj;
        }
        x[j] = (l > j) ? -(r[j] + s0 + s1) : T(0);
    }
}

template <typename T>
__global__ __launch_bounds__(DiagThreads<T>::value) void diag_ldl_kernel(T *__restrict__ A, long lda, T *__restrict__ linv,
                                                      T *__restrict__ d, T *__restrict__ dinv,
                                                      int *__restrict__ info, int blk)
{
    constexpr int DT = DiagThreads<T>::value;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *Pa = reinterpret_cast<T *>(smem_raw);       // [TILE][PLD]  panel: A entries, then W = L D
    T *Lp = Pa + TILE * PLD;                       // [TILE][PLD]  panel: L
    T *Xd = Lp + TILE * PLD;                       // [4][NB][PLD] inverses of the diagonal sub-blocks
    T *Di = Xd + 4 * NB * PLD;                     // [TILE]       1 / D
    T *Lt = Di + TILE;                             // [2][NB][NB]  L11 of the current / previous panel, TRANSPOSED
    // the inverse assembly re-uses the panel region: Xs = 3 off-diagonal blocks, Tb = 3 product blocks
    T *Xs = Pa;                                    // [3][NB][PLD]  X(1,0), X(2,0), X(2,1)
    T *Tb = Pa + 3 * NB * PLD;                     // [3][NB][PLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    GPX_STAMP(0);
    for (int jb = 0; jb <= 4; ++jb) {
        const int c0 = NB * jb;
        const int nrows = TILE - c0;
        for (int idx = tid; idx < nrows * NB; idx += DT) {
            const int r_ = idx >> 5, c_ = idx & 31;
            Pa[(c0 + r_) * PLD + c_] = A[(size_t)(c0 + r_) * lda + c0 + c_];
        }
        __syncthreads();
        GPX_STAMP(1 + 4 * jb);
        // ---- A: wave 0 factorises sub-block jb, wave 1 inverts L of sub-block jb - 1; lane l (and its twin
        //         l + 32) owns row l ----
        if (wave == 0 && jb < 4) {
            const int l = lane & 31;
            T r[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c)
                r[c] = Pa[(c0 + l) * PLD + c];
            T dmine = T(1);
            int nneg = 0;
            bool bad = false;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const T dj = bcast_lane(r[j], j);
                if (!(fabs((double)dj) > 0.0) || !(fabs((double)dj) < 1e300))
                    bad = true;
                if (dj < T(0))
                    ++nneg;
                if (l == j)
                    dmine = dj;
                const T lij = r[j] * (T(1) / dj);
#pragma unroll
                for (int k = j + 1; k < NB; ++k) {
                    const T akj = bcast_lane(r[j], k);  // a_kj (still un-scaled in lane k)
                    r[k] -= lij * akj;
                }
                if (l > j)
                    r[j] = lij;
            }
            if (lane < NB) {
                T *lt = Lt + (jb & 1) * NB * NB;
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    lt[c * NB + l] = c < l ? r[c] : T(0);  // L11[l][c], column c contiguous
                    if (c < l)
                        A[(size_t)(c0 + l) * lda + c0 + c] = r[c];
                }
                A[(size_t)(c0 + l) * lda + c0 + l] = dmine;
                Di[c0 + l] = T(1) / dmine;
                d[blk * TILE + c0 + l] = dmine;
                dinv[blk * TILE + c0 + l] = T(1) / dmine;
                if (l == 0) {
                    if (bad)
                        atomicCAS(&info[0], 0, blk * TILE + c0 + 1);
                    if (nneg)
                        atomicAdd(&info[1], nneg);
                }
            }
        } else if (wave == 1 && jb > 0) {
            const int l = lane & 31;
            const T *lt = Lt + ((jb - 1) & 1) * NB * NB;
            T r[NB], x[NB];
#pragma unroll
            for (int c = 0; c < NB; ++c)
                r[c] = lt[c * NB + l];
            sub_inverse(r, x, l);
            if (lane < NB) {
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    Xd[((jb - 1) * NB + l) * PLD + c] = c < l ? x[c] : (c == l ? T(1) : T(0));
            }
        }
        __syncthreads();
        GPX_STAMP(2 + 4 * jb);
        const int nb_rows = nrows - NB;  // rows below the diagonal sub-block
        if (nb_rows > 0) {
            // ---- B: row i of W solves W L11^T = A21:  w_c = a_ic - sum_{k<c} w_k L11[c][k] ----
            if (tid < nb_rows) {
                const int i_ = c0 + NB + tid;
                const T *lt = Lt + (jb & 1) * NB * NB;
                T a[NB];
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    a[c] = Pa[i_ * PLD + c];
                // column c of L11 (contiguous in Lt) is fetched whole, one column ahead of its use: left to itself
                // the compiler issued every LDS read just before its FMA and waited for it (7 us for this loop)
                T lc[NB], ln[NB];
#pragma unroll
                for (int c2 = 0; c2 < NB; ++c2)
                    lc[c2] = lt[c2];
                SubstStep<T, 0>::run(a, lc, ln, lt);
#pragma unroll
                for (int c = 0; c < NB; ++c) {
                    Pa[i_ * PLD + c] = a[c];
                    Lp[i_ * PLD + c] = a[c] * Di[c0 + c];
                }
            }
            __syncthreads();
            GPX_STAMP(3 + 4 * jb);
            for (int idx = tid; idx < nb_rows * NB; idx += DT) {
                const int i_ = c0 + NB + (idx >> 5), c_ = idx & 31;
                A[(size_t)i_ * lda + c0 + c_] = Lp[i_ * PLD + c_];
            }
            // ---- C: trailing update on the global block: A[i][k] -= sum_c W[i][c] L[k][c], k <= i ----
            // lower-triangle entries only, CPT per thread; all global loads are issued before the first use (as a
            // read-modify-write inside the loop every entry paid its own L2 round trip: ~18 in a row for jb = 0)
            const int r0 = c0 + NB;
            constexpr int CPT = ((TILE - NB) * (TILE - NB + 1) / 2 + DT - 1) / DT;
            const int ntri = nb_rows * (nb_rows + 1) / 2;
            T creg[CPT];
            int cpos[CPT];
#pragma unroll
            for (int e = 0; e < CPT; ++e) {
                const int t = tid + DT * e;
                int ii = 0, kk = 0;
                if (t < ntri)
                    tri_decode(t, ii, kk);
                cpos[e] = (ii << 8) | kk;
                creg[e] = t < ntri ? A[(size_t)(r0 + ii) * lda + r0 + kk] : T(0);
            }
#pragma unroll
            for (int e = 0; e < CPT; ++e) {
                const int t = tid + DT * e;
                if (t < ntri) {
                    const int ii = cpos[e] >> 8, kk = cpos[e] & 255;
                    const T *wrow = Pa + (r0 + ii) * PLD;
                    const T *lrow = Lp + (r0 + kk) * PLD;
                    T s = T(0);
#pragma unroll
                    for (int c = 0; c < NB; ++c)
                        s += wrow[c] * lrow[c];
                    A[(size_t)(r0 + ii) * lda + r0 + kk] = creg[e] - s;
                }
            }
        }
        __threadfence_block();
        __syncthreads();
        GPX_STAMP(4 + 4 * jb);
    }

    // ---- inverse of the 128 x 128 unit-lower L from the four 32 x 32 inverses ----
    for (int s_ = 1; s_ < 4; ++s_) {
        // phase 1: T(i,j) = sum_{k=j}^{i-1} L[i][k] X[k][j]  for the 4 - s_ blocks (i = j + s_, j)
        const int nblocks = 4 - s_;
        for (int idx = tid; idx < nblocks * NB * NB; idx += DT) {
            const int b = idx >> 10, r_ = (idx >> 5) & 31, c_ = idx & 31;
            const int jj = b, ii = b + s_;
            T acc = T(0);
            for (int k = jj; k < ii; ++k) {
                const T *lrow = A + (size_t)(NB * ii + r_) * lda + NB * k;  // L(ii,k) row r_, from global
                const T *xb = (k == jj) ? Xd + (jj * NB) * PLD
                                        : Xs + ((k == 1 ? 0 : (jj == 0 ? 1 : 2)) * NB) * PLD;  // X(1,0) | X(2,0) | X(2,1)
                for (int t = 0; t < NB; ++t)
                    acc += lrow[t] * xb[t * PLD + c_];
            }
            Tb[(b * NB + r_) * PLD + c_] = acc;
        }
        __syncthreads();
        // phase 2: X(i,j) = -Xd[i] * T(i,j)
        for (int idx = tid; idx < nblocks * NB * NB; idx += DT) {
            const int b = idx >> 10, r_ = (idx >> 5) & 31, c_ = idx & 31;
            const int jj = b, ii = b + s_;
            const T *xdrow = Xd + (ii * NB + r_) * PLD;
            T acc = T(0);
            for (int t = 0; t <= r_; ++t)
                acc += xdrow[t] * Tb[(b * NB + t) * PLD + c_];
            const T xv = -acc;
            linv[(size_t)blk * TILE * TILE + (size_t)(NB * ii + r_) * TILE + NB * jj + c_] = xv;
            if (ii < 3)  // X(1,0), X(2,0), X(2,1) are operands of later steps
                Xs[((ii == 1 ? 0 : (jj == 0 ? 1 : 2)) * NB + r_) * PLD + c_] = xv;
        }
        __syncthreads();
        GPX_STAMP(20 + s_);
    }
    // diagonal blocks and the zero upper part
    for (int idx = tid; idx < TILE * TILE; idx += DT) {
        const int r_ = idx >> 7, c_ = idx & 127;
        const int bi = r_ >> 5, bj = c_ >> 5;
        if (bj > bi)
            linv[(size_t)blk * TILE * TILE + idx] = T(0);
        else if (bj == bi)
            linv[(size_t)blk * TILE * TILE + idx] = Xd[(bi * NB + (r_ & 31)) * PLD + (c_ & 31)];
    }
    GPX_STAMP(24);
}

static size_t diag_shmem_bytes(size_t esz)
{
    return esz * (size_t)(2 * TILE * PLD + 4 * NB * PLD + TILE + 2 * NB * NB);  // Pa + Lp + Xd + Di + Lt
}

template <typename T>
static void diag_t(void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk, hipStream_t st)
{
    const size_t shmem = diag_shmem_bytes(sizeof(T));
    hipLaunchKernelGGL(diag_ldl_kernel<T>, dim3(1), dim3(DiagThreads<T>::value), shmem, st, (T *)Ablk, lda, (T *)linv, (T *)d,
                       (T *)dinv, info, blk);
}

void factor_init(int prec)
{
    if (prec == GPX_PREC_F64)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldl_kernel<double>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_shmem_bytes(sizeof(double)));
    else
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&diag_ldl_kernel<float>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_shmem_bytes(sizeof(float)));
}

void launch_diag_ldl(int prec, void *Ablk, long lda, void *linv, void *d, void *dinv, int *info, int blk,
                     hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        diag_t<double>(Ablk, lda, linv, d, dinv, info, blk, st);
    else
        diag_t<float>(Ablk, lda, linv, d, dinv, info, blk, st);
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
