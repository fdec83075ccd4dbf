// gpx_gemm.hip -- LDS-tiled MFMA GEMM core for gfx950 (fp32: v_mfma_f32_16x16x4_f32,
// fp64: v_mfma_f64_16x16x4_f64), shared by every dense contraction of the GP path:
//
//   * LDL^T trailing update  C -= W * L^T            (replaces Eigen::LDLT::compute, gp_regressor.hpp:161-162)
//   * panel solve            W = A21 * Linv11^T, L21 = W * D^-1   (same)
//   * inverse factor         T = L21 * X11,  X21 = -X22 * T
//   * predictive variance    partial[mt][q] = sum_rows (X * Kqp^T)^2 / D    (gp_regressor.hpp:316-319)
//
// Tile 128 x 128 x (128 bytes of k), 256 threads = 4 waves in 2 x 2, each wave 64 x 64 = 4 x 4
// MFMA fragments of 16 x 16.  Operand fragments: lane l holds A[row l&15][k = 4*(l>>4)+s] for
// the 4 k-steps s of a 16-deep chunk, read as ONE 16-byte (fp32) LDS vector; the same k
// permutation is applied to B, so the products pair up.  Global -> register -> LDS staging is
// double-buffered with one barrier per k-tile.  All dimensions are multiples of the tile by
// construction (matrices are padded to 256), so there is no edge handling in the hot loop.
#include <cstdlib>
#include "gpx_internal.hpp"

namespace gpx {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

using f32x16 = __attribute__((ext_vector_type(16))) float;

// MFMA shape policy.  FR = rows/cols of a fragment, NACC = accumulator registers per lane,
// LG = 64 / FR lane groups; a lane of group g supplies k = 4 g + s (s = 0..3) of a 4*LG-deep chunk.
template <typename T, bool M32>
struct MfmaT;
template <>
struct MfmaT<float, false> {  // v_mfma_f32_16x16x4_f32
    using acc_t = f32x4;
    static constexpr int FR = 16, NACC = 4, LG = 4;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + r
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <>
struct MfmaT<float, true> {  // v_mfma_f32_32x32x2_f32
    using acc_t = f32x16;
    static constexpr int FR = 32, NACC = 16, LG = 2;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    static __device__ __forceinline__ int crow(int lane, int r) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
};
template <>
struct MfmaT<double, false> {  // v_mfma_f64_16x16x4_f64
    using acc_t = f64x4;
    static constexpr int FR = 16, NACC = 4, LG = 4;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * r
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// 16-byte chunks of padding per staged [row][k] LDS row.  One chunk (144-byte rows) leaves 2-way bank conflicts in the
// fragment reads: a ds_read_b128 is served in four groups of 16 lanes -- {0-3, 12-15, 20-27}, ... -- and with a row
// stride of 9 chunks the rows 4..11 of the second k sub-block fall on the banks of rows 0-3 / 12-15 of the first
// (SQ_LDS_BANK_CONFLICT = 37 % of the LDS cycles of the variance GEMM).  Two chunks (160-byte rows) are
// conflict-free for every group; the 128 x 128 tile then needs exactly 80 KiB, so two workgroups still fit 160 KiB.
#ifndef GEMM_KPAD
#define GEMM_KPAD 2
#endif
// 1: give the fp32 instantiations named staging registers and a structured scheduling hint (VMEM reads, then per
// 16-deep chunk the LDS reads and the MFMAs) as well.  Measured: variance 256^2 tile 137.8 (unhinted 138.2), 128^2
// tile 131 (136), trailing-update tile 91 (104-107) TFLOP/s -- the fp32 code is better left to the scheduler.
#ifndef GEMM_F32_HINT
#define GEMM_F32_HINT 0
#endif
// Waves per SIMD the 4-wave (128 x 128) tiles are compiled for.  Without a bound hipcc budgets 512 registers per lane
// (252 VGPRs + 80 AGPRs for the read-modify-write tile), which leaves ONE workgroup per CU although two fit its LDS;
// 2 caps the tile at 256 registers (234 used, nothing spilled) so that two workgroups share a CU and one hides the
// other's prologue / epilogue -- what the short k-loops of the LDL^T trailing update need.
#ifndef GEMM_WAVES_PER_EU
#define GEMM_WAVES_PER_EU 2
#endif
// Padding (elements) of a [k][n] LDS row of the fp64 NN tiles (inverse-factor assembly).  A lane group fg reads k-rows
// 2 fg and 2 fg + 1: with 2 elements of padding rows two apart start 8 banks apart and the 16-lane groups of one half
// wave collide; see DESIGN.md for the measurement.
#ifndef GEMM_NPAD64
#define GEMM_NPAD64 2
#endif

template <typename T>
struct GemmDev {
    const T *A, *B;
    T *C;
    long lda, ldb, ldc;
    int M, N, K;
    long sA, sB, sC;
    int batch, M_last, k_eq_m;
    T alpha;
    int beta, lower_only, a_lower, b_lower;
    T *W;
    long ldw;
    const T *colscale;
    const T *rowweight;
    T *partial;
    long ldp;
    // EPI_COLSQ of an fp32 product with the low-rank fit added back (colcoef != null): the epilogue runs in fp64,
    // w = acc[m][n] + sum_c colcoef[c][n] * rowcorr[c][m] (c < VAR_NCORR), partial64[mt][n] = sum_rows w^2 rowweight64[m]
    const double *rowcorr, *colcoef, *rowweight64;
    double *partial64;
    long ldrc, ldcc;
};

// FM x FN MFMA fragments (16 x 16) per wave, WGM x WGN waves per workgroup:
// block tile BM x BN = (WGM * FM * FR) x (WGN * FN * FR), 64 * WGM * WGN threads.
// The k-loop starts on a 32-byte boundary: where it starts relative to the instruction-fetch window matters (round 3,
// profiles/r03_var_gemm_variants.txt: the 128-byte-k variance tile runs 16.0 ms per launch with the loop at 0, 16, 24 bytes
// past a 32-byte boundary and 16.45 at 8 bytes past it; what surrounds the kernel in the code object decided that before).
#ifndef GEMM_LOOP_P2ALIGN
#define GEMM_LOOP_P2ALIGN 5
#endif
#ifndef GEMM_KERNEL_ALIGN
#define GEMM_KERNEL_ALIGN 256
#endif
template <typename T, bool NN, int EPI, int FM, int FN, int WGM, int WGN, int KBYTES, bool M32>
__global__ __attribute__((aligned(GEMM_KERNEL_ALIGN))) __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 4) ? (KBYTES == 64 ? 3 : GEMM_WAVES_PER_EU) : 1) void gemm_kernel(GemmDev<T> g)
{
    using MF = MfmaT<T, M32>;
    using acc_t = typename MF::acc_t;
    constexpr int FR = MF::FR, NACC = MF::NACC, KCH = 4 * MF::LG;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int BM = WGM * FM * FR, BN = WGN * FN * FR;
    constexpr int BK = KBYTES / sizeof(T);      // k-tile: KBYTES bytes of k per staged row
    constexpr int CPRW = KBYTES / 16;           // 16-byte chunks per staged [row][k] row
    constexpr int EPC = 16 / sizeof(T);         // elements per 16-byte chunk
    constexpr int BKP = BK + GEMM_KPAD * EPC;   // padded k extent of a [row][k] LDS tile
    constexpr int BNP = BN + (sizeof(T) == 8 ? GEMM_NPAD64 : EPC);  // padded n extent of a [k][n] LDS tile (NN)
    constexpr int A_TILE = BM * BKP;
    constexpr int B_TILE = NN ? BK * BNP : BN * BKP;
    constexpr int A_CH = BM * CPRW / NT;        // 16-byte chunks per thread, A tile
    constexpr int B_CH = BN * CPRW / NT;        // same for B (both layouts hold BN * BK elements)
    static_assert((BM * CPRW) % NT == 0 && (BN * CPRW) % NT == 0 && BK % KCH == 0, "tile / thread mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem_raw[];
    T *smem = reinterpret_cast<T *>(gemm_smem_raw);
    T *As = smem;
    T *Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;

    // ---- which tile ----
    int mt, nt;
    if (g.lower_only) {
        tri_decode((int)blockIdx.x, mt, nt);
    } else if (NN && g.b_lower) {
        // the k-range of a tile depends on its COLUMN here (k >= n0): neighbours in launch order share the column,
        // hence the k-range, and stay in step on the shared B panel -- the mirror image of the a_lower case below.
        // With the row index fastest the same product ran 1.4-1.8x slower than its a_lower sibling (trtri, fp64).
        mt = blockIdx.x;
        nt = blockIdx.y;  // heavy column-tiles first
    } else {
        nt = blockIdx.x;
        mt = g.a_lower ? (int)(gridDim.y - 1 - blockIdx.y) : (int)blockIdx.y;  // heavy row-tiles first
    }
    const int z = blockIdx.z;
    int Mz = g.M, Kz = g.K;
    if (g.M_last >= 0 && z == g.batch - 1) {
        Mz = g.M_last;
        if (g.k_eq_m)
            Kz = g.M_last;
    }
    const int m0 = mt * BM, n0 = nt * BN;
    if (m0 >= Mz)
        return;
    const T *A = g.A + (size_t)z * g.sA;
    const T *B = g.B + (size_t)z * g.sB;

    int klo = 0, khi = Kz;
    if (g.a_lower)
        khi = min(khi, m0 + BM);
    if (g.b_lower) {
        if (NN)
            klo = n0;
        else
            khi = min(khi, n0 + BN);
    }
    const int kt0 = klo / BK, kt1 = khi / BK;

    acc_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < NACC; ++r)
                acc[i][j][r] = T(0);

    // ---- staging: 16-byte chunks, global -> registers -> LDS.  Written as straight-line macros on
    // named register arrays with NO conditionals around them: with lambdas + `if (more)` hipcc demoted
    // the staging registers to scratch and waited for every global load before the MFMAs.
    const int a_row = tid / CPRW, a_kc = tid % CPRW;  // chunk c = tid + NT * i  ->  row = a_row + (NT/CPRW) * i
    const T *a_src = A + (size_t)(m0 + a_row) * g.lda + a_kc * EPC;
    const T *b_src;
    int b_lds_off;
    if constexpr (NN) {
        constexpr int CPR = BN / EPC;  // chunks per k-row
        b_src = B + (size_t)(tid / CPR) * g.ldb + n0 + (tid % CPR) * EPC;
        b_lds_off = (tid / CPR) * BNP + (tid % CPR) * EPC;
    } else {
        b_src = B + (size_t)(n0 + a_row) * g.ldb + a_kc * EPC;
        b_lds_off = a_row * BKP + a_kc * EPC;
    }
    const int a_lds_off = a_row * BKP + a_kc * EPC;
    constexpr int ROWS_PER_PASS = NT / CPRW;                     // rows per pass of the workgroup ([row][k])
    constexpr int KROWS_PER_PASS = NN ? NT / (BN / EPC) : 1;     // k-rows per pass ([k][n])

    // Staging registers.  fp32: small arrays (fully unrolled accesses; the compiler schedules that form best).  fp64:
    // NAMED scalars, because its main loop carries scheduling hints and with those hipcc keeps arrays in scratch
    // memory (at most 2 + 2 chunks per thread there); the fp32 tiles lose ~5 % with named scalars.
    constexpr bool NAMED = sizeof(T) == 8 || GEMM_F32_HINT;
    static_assert(!NAMED || (A_CH <= 4 && B_CH <= 4), "named staging is written out for 4 chunks per operand");
    uint4 ra[NAMED ? 1 : A_CH], rb[NAMED ? 1 : B_CH];
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GPX_A_PTR(I) (a_src + (size_t)(ROWS_PER_PASS * (I)) * g.lda + k0_)
#define GPX_B_PTR(I) (NN ? b_src + (k0_ + KROWS_PER_PASS * (I)) * g.ldb : b_src + (size_t)(ROWS_PER_PASS * (I)) * g.ldb + k0_)
#define GPX_GLOAD(KT)                                                                                       \
    {                                                                                                       \
        const size_t k0_ = (size_t)(KT) * BK;                                                               \
        if constexpr (NAMED) {                                                                              \
            ra0 = *reinterpret_cast<const uint4 *>(GPX_A_PTR(0));                                           \
            rb0 = *reinterpret_cast<const uint4 *>(GPX_B_PTR(0));                                           \
            if constexpr (A_CH > 1) ra1 = *reinterpret_cast<const uint4 *>(GPX_A_PTR(1));                   \
            if constexpr (B_CH > 1) rb1 = *reinterpret_cast<const uint4 *>(GPX_B_PTR(1));                   \
            if constexpr (A_CH > 2) ra2 = *reinterpret_cast<const uint4 *>(GPX_A_PTR(2));                   \
            if constexpr (B_CH > 2) rb2 = *reinterpret_cast<const uint4 *>(GPX_B_PTR(2));                   \
            if constexpr (A_CH > 3) ra3 = *reinterpret_cast<const uint4 *>(GPX_A_PTR(3));                   \
            if constexpr (B_CH > 3) rb3 = *reinterpret_cast<const uint4 *>(GPX_B_PTR(3));                   \
        } else {                                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < A_CH; ++i_)                                             \
                ra[i_] = *reinterpret_cast<const uint4 *>(GPX_A_PTR(i_));                                   \
            _Pragma("unroll") for (int i_ = 0; i_ < B_CH; ++i_)                                             \
                rb[i_] = *reinterpret_cast<const uint4 *>(GPX_B_PTR(i_));                                   \
        }                                                                                                   \
    }
#define GPX_A_LDS(I, BUF) (As + (BUF) * A_TILE + a_lds_off + ROWS_PER_PASS * (I) * BKP)
#define GPX_B_LDS(I, BUF) (Bs + (BUF) * B_TILE + b_lds_off + (NN ? KROWS_PER_PASS * (I) * BNP : ROWS_PER_PASS * (I) * BKP))
#define GPX_SSTORE(BUF)                                                                                     \
    {                                                                                                       \
        if constexpr (NAMED) {                                                                              \
            *reinterpret_cast<uint4 *>(GPX_A_LDS(0, BUF)) = ra0;                                            \
            *reinterpret_cast<uint4 *>(GPX_B_LDS(0, BUF)) = rb0;                                            \
            if constexpr (A_CH > 1) *reinterpret_cast<uint4 *>(GPX_A_LDS(1, BUF)) = ra1;                    \
            if constexpr (B_CH > 1) *reinterpret_cast<uint4 *>(GPX_B_LDS(1, BUF)) = rb1;                    \
            if constexpr (A_CH > 2) *reinterpret_cast<uint4 *>(GPX_A_LDS(2, BUF)) = ra2;                    \
            if constexpr (B_CH > 2) *reinterpret_cast<uint4 *>(GPX_B_LDS(2, BUF)) = rb2;                    \
            if constexpr (A_CH > 3) *reinterpret_cast<uint4 *>(GPX_A_LDS(3, BUF)) = ra3;                    \
            if constexpr (B_CH > 3) *reinterpret_cast<uint4 *>(GPX_B_LDS(3, BUF)) = rb3;                    \
        } else {                                                                                            \
            _Pragma("unroll") for (int i_ = 0; i_ < A_CH; ++i_)                                             \
                *reinterpret_cast<uint4 *>(GPX_A_LDS(i_, BUF)) = ra[i_];                                    \
            _Pragma("unroll") for (int i_ = 0; i_ < B_CH; ++i_)                                             \
                *reinterpret_cast<uint4 *>(GPX_B_LDS(i_, BUF)) = rb[i_];                                    \
        }                                                                                                   \
    }

    const int fr = lane % FR, fg = lane / FR;
    // k owned by lane group fg inside a 16-deep chunk.  fp32: the 4 consecutive k of ONE 16-byte LDS chunk (4 fg + s).
    // fp64: TWO 16-byte chunks, fg and fg + 4, i.e. k = 2 fg + (s & 1) + 8 (s >> 1) -- with the natural 4 fg + s a
    // lane group would step 32 bytes per fg and the 16-lane service groups of ds_read_b128 would collide on the
    // banks whatever the row padding; stepping ONE chunk per fg is conflict-free with 160-byte rows, as in fp32.
    constexpr int FGS = sizeof(T) == 8 ? 2 : 4;  // elements per lane-group step
    const int a_frag_off = (wm * FM * FR + fr) * BKP + FGS * fg;
    const int b_frag_off = NN ? (sizeof(T) == 8 ? 2 * fg : 4 * fg) * BNP + wn * FN * FR + fr
                              : (wn * FN * FR + fr) * BKP + FGS * fg;

    // ---- EPI_STORE: fetch the C tile NOW, in fragment layout, so that its latency hides under the k-loop
    // (the trailing update of the factorisation has only 8 k-tiles per tile; a read-modify-write epilogue
    // that starts its loads after the last MFMA leaves the matrix pipe idle for ~20 % of the tile).
    constexpr bool C_PREFETCH = (EPI == EPI_STORE) && (FM * FN * NACC <= 64);
    T cpre[C_PREFETCH ? FM : 1][C_PREFETCH ? FN : 1][NACC];
    if constexpr (C_PREFETCH) {
        const T *Cin = g.C + (size_t)z * g.sC;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r)
                    cpre[i][j][r] = Cin[(size_t)(m0 + (wm * FM + i) * FR + MF::crow(lane, r)) * g.ldc + n0 +
                                        (wn * FN + j) * FR + fr];
    }

    // (fp32 only: the fp64 contraction needs no correction, and its hinted main loop is sensitive to every extra
    // register -- with the correction code compiled in, the fp64 variance product dropped from 72.5 to 64 TFLOP/s)
    constexpr bool CORR_OK = (EPI == EPI_COLSQ) && sizeof(T) == 4;
    const bool corr = CORR_OK && g.colcoef != nullptr;

    // one k-tile of MFMAs on LDS buffer BUF
#define GPX_COMPUTE(BUF) \
            { \
                const T *as = As + (BUF) * A_TILE + a_frag_off; \
                const T *bs = Bs + (BUF) * B_TILE + b_frag_off; \
_Pragma("unroll") \
                for (int kc = 0; kc < BK / KCH; ++kc) { \
                    T a[FM][4], b[FN][4]; \
_Pragma("unroll") \
                    for (int f = 0; f < FM; ++f) { \
                        const T *p = as + f * FR * BKP + kc * KCH; \
                        if constexpr (sizeof(T) == 4) { \
                            float4 v = *reinterpret_cast<const float4 *>(p); \
                            a[f][0] = v.x, a[f][1] = v.y, a[f][2] = v.z, a[f][3] = v.w; \
                        } else { \
                            double2 v0 = *reinterpret_cast<const double2 *>(p); \
                            double2 v1 = *reinterpret_cast<const double2 *>(p + 8); \
                            a[f][0] = v0.x, a[f][1] = v0.y, a[f][2] = v1.x, a[f][3] = v1.y; \
                        } \
                    } \
_Pragma("unroll") \
                    for (int f = 0; f < FN; ++f) { \
                        if constexpr (NN) { \
_Pragma("unroll") \
                            for (int s = 0; s < 4; ++s) \
                                b[f][s] = bs[(kc * KCH + (sizeof(T) == 8 ? (s & 1) + 8 * (s >> 1) : s)) * BNP + f * FR]; \
                        } else { \
                            const T *p = bs + f * FR * BKP + kc * KCH; \
                            if constexpr (sizeof(T) == 4) { \
                                float4 v = *reinterpret_cast<const float4 *>(p); \
                                b[f][0] = v.x, b[f][1] = v.y, b[f][2] = v.z, b[f][3] = v.w; \
                            } else { \
                                double2 v0 = *reinterpret_cast<const double2 *>(p); \
                                double2 v1 = *reinterpret_cast<const double2 *>(p + 8); \
                                b[f][0] = v0.x, b[f][1] = v0.y, b[f][2] = v1.x, b[f][3] = v1.y; \
                            } \
                        } \
                    } \
_Pragma("unroll") \
                    for (int s = 0; s < 4; ++s) \
_Pragma("unroll") \
                        for (int i = 0; i < FM; ++i) \
_Pragma("unroll") \
                            for (int j = 0; j < FN; ++j) \
                                acc[i][j] = MF::run(a[i][s], b[j][s], acc[i][j]); \
                } \
            }

    // ---- main loop: one barrier per k-tile; the next tile's global loads are in flight over the MFMAs.
    // The last iteration re-loads its own tile (clamped index) so that nothing in the loop is conditional.
    if (kt0 < kt1) {
        GPX_GLOAD(kt0);
        GPX_SSTORE(0);
        __syncthreads();
        int buf = 0;
#if GEMM_LOOP_P2ALIGN
        asm volatile(".p2align %0" ::"n"(GEMM_LOOP_P2ALIGN));
#endif
#ifdef GEMM_LOOP_NOPS  /* placement experiment: N 4-byte instructions after the alignment point */
        asm volatile(".rept %0\n s_nop 0\n .endr" ::"n"(GEMM_LOOP_NOPS));
#endif
        for (int kt = kt0; kt < kt1; ++kt) {
            const int ktn = min(kt + 1, kt1 - 1);
            GPX_GLOAD(ktn);
            // fp64: keep the prefetch HERE.  Left alone, the machine scheduler sinks these loads below ~90 % of the
            // k-tile's MFMAs (shorter live ranges) and the wave then waits for them right before the LDS stores:
            // pinning them lifts the fp64 variance product from 66 to 73 TFLOP/s (93 % of the fp64 peak).  The same
            // hint costs the fp32 instantiations 6-27 % (their own schedule interleaves the LDS reads better), so
            // they are left to the scheduler.
            if constexpr (sizeof(T) == 8) {
                __builtin_amdgcn_sched_group_barrier(0x020, A_CH + B_CH, 0);  // the VMEM reads first ...
                __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);            // ... then (at least) the first MFMAs
            } else if (GEMM_F32_HINT) {
                __builtin_amdgcn_sched_group_barrier(0x020, A_CH + B_CH, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, FM * FN * 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, FM + FN, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, FM * FN * 4, 0);
            }
            GPX_COMPUTE(buf);
            GPX_SSTORE(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
#undef GPX_GLOAD
#undef GPX_COMPUTE
#undef GPX_SSTORE
#undef GPX_A_PTR
#undef GPX_B_PTR
#undef GPX_A_LDS
#undef GPX_B_LDS

    // ---- epilogues ----
    if constexpr (EPI == EPI_STORE) {
        T *C = g.C + (size_t)z * g.sC;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int row = m0 + (wm * FM + i) * FR + MF::crow(lane, r);
                    const int col = n0 + (wn * FN + j) * FR + fr;
                    T *p = C + (size_t)row * g.ldc + col;
                    T v = g.alpha * acc[i][j][r];
                    if constexpr (C_PREFETCH) {
                        v += g.beta ? cpre[i][j][r] : T(0);
                    } else {
                        if (g.beta)
                            v += *p;
                    }
                    *p = v;
                }
    } else if constexpr (EPI == EPI_TRSM) {
        T *C = g.C + (size_t)z * g.sC;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
            const int col = n0 + (wn * FN + j) * FR + fr;
            const T cs = g.colscale[col];
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < NACC; ++r) {
                    const int row = m0 + (wm * FM + i) * FR + MF::crow(lane, r);
                    const T v = acc[i][j][r];
                    g.W[(size_t)row * g.ldw + col] = v;
                    C[(size_t)row * g.ldc + col] = v * cs;
                }
        }
    } else {  // EPI_COLSQ
        // plain form: partial[mt][n] = sum over the tile's rows of acc^2 * rowweight[row], in T
        // (the fp64 instantiation keeps exactly this form: its main loop's schedule -- 72.5 TFLOP/s -- changed with every
        // other arrangement of the epilogue, down to 57 TFLOP/s)
#define GPX_COLSQ_PLAIN()                                                                          \
    {                                                                                              \
        T *red = smem; /* [WGM][BN], re-using the staging buffers (all waves are past the k-loop barrier) */ \
        T w[FM][NACC];                                                                             \
        _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int r = 0; r < NACC; ++r)      \
            w[i][r] = g.rowweight[m0 + (wm * FM + i) * FR + MF::crow(lane, r)];                    \
        _Pragma("unroll") for (int j = 0; j < FN; ++j)                                             \
        {                                                                                          \
            T s = T(0);                                                                            \
            _Pragma("unroll") for (int i = 0; i < FM; ++i) _Pragma("unroll") for (int r = 0; r < NACC; ++r)  \
                s += acc[i][j][r] * acc[i][j][r] * w[i][r];                                        \
            if constexpr (FR == 16)                                                                \
                s += __shfl_xor(s, 16);                                                            \
            s += __shfl_xor(s, 32);                                                                \
            if (fg == 0)                                                                           \
                red[wm * BN + (wn * FN + j) * FR + fr] = s;                                        \
        }                                                                                          \
        __syncthreads();                                                                           \
        for (int c = tid; c < BN; c += NT) {                                                       \
            T s = T(0);                                                                            \
            _Pragma("unroll") for (int w2 = 0; w2 < WGM; ++w2) s += red[w2 * BN + c];              \
            g.partial[(size_t)mt * g.ldp + n0 + c] = s;                                            \
        }                                                                                          \
    }
        if constexpr (!CORR_OK) {
            GPX_COLSQ_PLAIN();
        } else {
            if (!corr) {
                GPX_COLSQ_PLAIN();
            } else {
                // Low-rank correction of the contraction (gpx_eval.hip, "centred kernel operand"): the B operand holds
                // k - fit with fit[n][k] = sum_c colcoef[c][n] b_c[k]; the product of A with the fit comes back here from
                // rowcorr[c][m] = sum_k A[m][k] b_c[k] (fp64, once per model).  The 14 terms cancel among themselves and
                // against the accumulator, so everything from here on is fp64: the sum, the square, 1/D, the column sums
                // (fp32 row vectors and coefficients cost 9.5e-7 k(0) at N = 16384 thin-plate against 1.3e-7 in fp64,
                // profiles/r03_tp_fit_probe.txt).  ~900 fp64 FMAs and LDS reads per lane: < 1 % of a tile's MFMA time.
                // This tile's slices of the row vectors, the coefficient vectors and 1/D go through the (now idle) LDS.
                double *ds = reinterpret_cast<double *>(gemm_smem_raw);
                double *rowc = ds;                           // [VAR_NCORR][BM]
                double *colc = rowc + VAR_NCORR * BM;        // [VAR_NCORR][BN]
                double *roww = colc + VAR_NCORR * BN;        // [BM]
                double *red64 = roww + BM;                   // [WGM][BN]
                static_assert(sizeof(double) * (VAR_NCORR * (BM + BN) + BM + WGM * BN) <=
                                  sizeof(T) * (2 * (size_t)A_TILE + 2 * (size_t)B_TILE),
                              "the fp64 epilogue must fit the staging buffers");
                // (all of a thread's loads are issued before its first LDS store)
                constexpr int NEL = VAR_NCORR * (BM + BN) + BM, NLD = (NEL + NT - 1) / NT;
                double pv[NLD];
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + NT * i;
                    double v = 0.0;
                    if (e < VAR_NCORR * BM) {
                        v = g.rowcorr[(size_t)(e / BM) * g.ldrc + m0 + e % BM];
                    } else if (e < VAR_NCORR * (BM + BN)) {
                        const int e2 = e - VAR_NCORR * BM;
                        v = g.colcoef[(size_t)(e2 / BN) * g.ldcc + n0 + e2 % BN];
                    } else if (e < NEL) {
                        v = g.rowweight64[m0 + e - VAR_NCORR * (BM + BN)];
                    }
                    pv[i] = v;
                }
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    const int e = tid + NT * i;
                    if (e < NEL)
                        ds[e] = pv[i];
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int lcol = (wn * FN + j) * FR + fr;
                    double ca[VAR_NCORR];
#pragma unroll
                    for (int c = 0; c < VAR_NCORR; ++c)
                        ca[c] = colc[c * BN + lcol];
                    double sj = 0.0;
#pragma unroll
                    for (int i = 0; i < FM; ++i)
#pragma unroll
                        for (int r = 0; r < NACC; ++r) {
                            const int lrow = (wm * FM + i) * FR + MF::crow(lane, r);
                            double w = (double)acc[i][j][r];
#pragma unroll
                            for (int c = 0; c < VAR_NCORR; ++c)
                                w = fma(ca[c], rowc[c * BM + lrow], w);
                            sj = fma(w * w, roww[lrow], sj);
                        }
                    if constexpr (FR == 16)
                        sj += __shfl_xor(sj, 16);
                    sj += __shfl_xor(sj, 32);
                    if (fg == 0)
                        red64[wm * BN + lcol] = sj;
                }
                __syncthreads();
                for (int c = tid; c < BN; c += NT) {
                    double s2 = 0.0;
#pragma unroll
                    for (int w2 = 0; w2 < WGM; ++w2)
                        s2 += red64[w2 * BN + c];
                    g.partial64[(size_t)mt * g.ldp + n0 + c] = s2;
                }
            }
        }
#undef GPX_COLSQ_PLAIN
    }
}

template <typename T, bool NN, int EPI, int FM, int FN, int WGM, int WGN, int KBYTES = 128, bool M32 = false>
static void gemm_launch_cfg(const GemmDev<T> &g, const GemmArgs &a, hipStream_t st)
{
    constexpr int FR = M32 ? 32 : 16;
    constexpr int BM = WGM * FM * FR, BN = WGN * FN * FR;
    constexpr int EPC = 16 / sizeof(T), BK = KBYTES / sizeof(T);
    constexpr int BKP = BK + GEMM_KPAD * EPC, BNP = BN + (sizeof(T) == 8 ? GEMM_NPAD64 : EPC);
    constexpr size_t shmem = sizeof(T) * (2 * (size_t)BM * BKP + 2 * (size_t)(NN ? BK * BNP : BN * BKP));
    static PerDeviceOnce attr_once;  // the LDS-size attribute is per device (one static per instantiation)
    auto kern = gemm_kernel<T, NN, EPI, FM, FN, WGM, WGN, KBYTES, M32>;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)shmem);
    });
    const int mt = a.M / BM, nt = a.N / BN;
    if (mt <= 0 || nt <= 0 || a.batch <= 0)
        return;
    dim3 grid = a.lower_only ? dim3(mt * (mt + 1) / 2, 1, a.batch)
                             : ((NN && a.b_lower) ? dim3(mt, nt, a.batch) : dim3(nt, mt, a.batch));
    hipLaunchKernelGGL(kern, grid, dim3(64 * WGM * WGN), shmem, st, g);
}

// tile choice: cfg 0 = 128 x 128 (4 waves of 64 x 64), cfg 2 = 256 x 256 (8 waves of 128 x 64, fp32 only);
// cfg 1 (256 x 128, 8 waves of 64 x 64) is not instantiated any more
int gemm_tile_m(int cfg) { return cfg == 0 ? 128 : 256; }
int gemm_tile_n(int cfg) { return cfg == 2 ? 256 : 128; }

static int pick_cfg(const GemmArgs &a, size_t esz)
{
    int cfg = (a.cfg > 2 || a.cfg < 0) ? 0 : a.cfg;
    // a tile must divide the problem (and the ragged last batch entry); square tiles for lower_only
    auto fits = [&](int c) {
        const int bm = gemm_tile_m(c), bn = gemm_tile_n(c);
        if (a.M % bm || a.N % bn)
            return false;
        if (c == 1)
            return false;  // 256 x 128 with 8 waves of 64 x 64 measured slower than both others (98 vs 134 TF)
        if (esz == 8 && c == 2)
            return false;  // 128 x 64 fp64 accumulators per wave do not fit 256 VGPRs at 2 waves/SIMD
        if (a.M_last >= 0 && a.M_last % bm)
            return false;
        if (a.lower_only && bm != bn)
            return false;
        if (a.b_lower && !a.nn && bn != TILE)
            return false;
        return true;
    };
    while (cfg > 0 && !fits(cfg))
        --cfg;
    return cfg;
}

int gemm_rows_per_partial(int prec, const GemmArgs &g)
{
    return gemm_tile_m(pick_cfg(g, prec == GPX_PREC_F64 ? 8 : 4));
}

template <typename T>
static void gemm_t(const GemmArgs &a, hipStream_t st)
{
    GemmDev<T> g;
    g.A = (const T *)a.A;
    g.B = (const T *)a.B;
    g.C = (T *)a.C;
    g.lda = a.lda, g.ldb = a.ldb, g.ldc = a.ldc;
    g.M = a.M, g.N = a.N, g.K = a.K;
    g.sA = a.sA, g.sB = a.sB, g.sC = a.sC;
    g.batch = a.batch, g.M_last = a.M_last, g.k_eq_m = a.k_eq_m;
    g.alpha = (T)a.alpha;
    g.beta = a.beta, g.lower_only = a.lower_only, g.a_lower = a.a_lower, g.b_lower = a.b_lower;
    g.W = (T *)a.W, g.ldw = a.ldw;
    g.colscale = (const T *)a.colscale;
    g.rowweight = (const T *)a.rowweight;
    g.partial = (T *)a.partial, g.ldp = a.ldp;
    g.rowcorr = a.rowcorr, g.colcoef = a.colcoef, g.rowweight64 = a.rowweight64;
    g.partial64 = (double *)a.partial;
    g.ldrc = a.ldrc, g.ldcc = a.ldcc;
    const int cfg = pick_cfg(a, sizeof(T));
#define GPX_GEMM_CFG(NN_, EPI_)                                                \
    do {                                                                       \
        if constexpr (sizeof(T) == 8) {                                        \
            /* fp64: 128 x 128 tile as 8 waves of 64 x 32 (2-3 waves/SIMD): 66 TF vs 43.5 TF for 4 waves of 64 x 64 */ \
            gemm_launch_cfg<T, NN_, EPI_, 4, 2, 2, 4>(g, a, st);               \
        } else if (cfg == 2) {                                                 \
            gemm_launch_cfg<T, NN_, EPI_, 8, 4, 2, 4>(g, a, st);               \
        } else {                                                               \
            gemm_launch_cfg<T, NN_, EPI_, 4, 4, 2, 2>(g, a, st);               \
        }                                                                      \
    } while (0)
    if (a.epi == EPI_STORE) {
        if (a.nn)
            GPX_GEMM_CFG(true, EPI_STORE);
        else
            GPX_GEMM_CFG(false, EPI_STORE);
    } else if (a.epi == EPI_TRSM) {
        GPX_GEMM_CFG(false, EPI_TRSM);
    } else if constexpr (sizeof(T) == 8) {  // EPI_COLSQ, fp64: the LDS-staged 128 x 128 tile (what the one-wave fp64 tile falls back to)
        gemm_launch_cfg<T, false, EPI_COLSQ, 4, 2, 2, 4>(g, a, st);
    } else {
        // EPI_COLSQ, fp32: the documented fallback of the one-wave tiles (GPX_VAR_TILE=3) -- 128 x 128 with 64-byte k rows,
        // 40 KiB of LDS per workgroup, compiled for 3 workgroups per CU.  The 128-byte-k and 256 x 256 variants measured
        // within 1.5 % of it on the variance shape (profiles/r03_var_gemm_variants.txt) and were removed in round 4.
        gemm_launch_cfg<T, false, EPI_COLSQ, 4, 4, 2, 2, 64>(g, a, st);
    }
#undef GPX_GEMM_CFG
}

void launch_gemm(int prec, const GemmArgs &g, hipStream_t st)
{
    if (g.cfg == 6) {
        if (prec != GPX_PREC_F64 && var_w1_fits(g)) {
            launch_var_w1(g, st);
            return;
        }
        if (prec == GPX_PREC_F64 && var_w1_f64_fits(g)) {
            launch_var_w1_f64(g, st);
            return;
        }
        GemmArgs h = g;
        h.cfg = 3;
        launch_gemm(prec, h, st);
        return;
    }
    if (prec == GPX_PREC_F64) {
        if (w1_f64_nn_fits(g)) {
            launch_w1_f64_nn(g, st);
            return;
        }
        gemm_t<double>(g, st);
    } else {
        gemm_t<float>(g, st);
    }
}

}  // namespace gpx
