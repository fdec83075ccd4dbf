// gpx_blk.hpp -- 32 x 32 block primitives on the matrix cores shared by the factorisation kernels (gpx_factor.hip:
// the 128 x 128 diagonal block of the blocked LDL^T; gpx_small.hip: the one-launch factorisation of small models):
// lane broadcasts, the reciprocal, block products of LDS operands (BlkAcc) and the LDL^T + inverse of one 32 x 32
// sub-block by rank-1 MFMA updates (subblock_ldl).  gfx950 only.
#pragma once
#include "gpx_internal.hpp"

namespace gpx {

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

// 1 / x to within an ulp or two by Newton steps on the hardware estimate: the IEEE division sequence (12 fp32 /
// ~25 fp64 instructions) sat on the dependency chain of every pivot column.  Zero, infinite or NaN pivots give a
// non-finite result, which the caller has already flagged.
__device__ __forceinline__ float fast_rcp(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}
__device__ __forceinline__ double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = fma(fma(-x, y, 1.0), y, y);
    return fma(fma(-x, y, 1.0), y, y);
}
__device__ __forceinline__ float pivot_huge(float) { return 3.0e38f; }
__device__ __forceinline__ double pivot_huge(double) { return 1e300; }

// 32 x 32 x 32 block product of two LDS operands (leading dimension PLD) by one wave on the 16x16x4 MFMA:
// four 16 x 16 accumulator tiles.  A operand: lane supplies A[i = lane & 15][k = lane >> 4]; B: B[k][j = lane & 15].
template <typename T>
struct BlkMma;
template <>
struct BlkMma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <>
struct BlkMma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <typename T>
struct BlkAcc {
    typename BlkMma<T>::acc_t t[2][2];
    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                t[i][j] = typename BlkMma<T>::acc_t{T(0), T(0), T(0), T(0)};
    }
    // += Ab (32 x 32) * Bb (32 x 32)
    __device__ __forceinline__ void mac(const T *Ab, const T *Bb, int lane)
    {
        const int i = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const int k = 4 * kk + kq;
            const T a0 = Ab[i * PLD + k], a1 = Ab[(16 + i) * PLD + k];
            const T b0 = Bb[k * PLD + i], b1 = Bb[k * PLD + 16 + i];
            t[0][0] = BlkMma<T>::mma(a0, b0, t[0][0]);
            t[0][1] = BlkMma<T>::mma(a0, b1, t[0][1]);
            t[1][0] = BlkMma<T>::mma(a1, b0, t[1][0]);
            t[1][1] = BlkMma<T>::mma(a1, b1, t[1][1]);
        }
    }
    // -= / += Ab (32 x 32) * Bb^T  (Bb row-major [j][k])
    __device__ __forceinline__ void msub_nt(const T *Ab, const T *Bb, int lane) { mac_nt<true>(Ab, Bb, lane); }
    template <bool NEG>
    __device__ __forceinline__ void mac_nt(const T *Ab, const T *Bb, int lane)
    {
        const int i = lane & 15, kq = lane >> 4;
#pragma unroll
        for (int kk = 0; kk < NB / 4; ++kk) {
            const int k = 4 * kk + kq;
            const T a0 = NEG ? -Ab[i * PLD + k] : Ab[i * PLD + k], a1 = NEG ? -Ab[(16 + i) * PLD + k] : Ab[(16 + i) * PLD + k];
            const T b0 = Bb[i * PLD + k], b1 = Bb[(16 + i) * PLD + k];
            t[0][0] = BlkMma<T>::mma(a0, b0, t[0][0]);
            t[0][1] = BlkMma<T>::mma(a0, b1, t[0][1]);
            t[1][0] = BlkMma<T>::mma(a1, b0, t[1][0]);
            t[1][1] = BlkMma<T>::mma(a1, b1, t[1][1]);
        }
    }
    __device__ __forceinline__ void load(const T *g, long ldg, int lane)
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    t[i][j][r] = g[(size_t)(16 * i + BlkMma<T>::crow(lane, r)) * ldg + 16 * j + (lane & 15)];
    }
    // sign * block -> LDS block (ld PLD, may be null) and / or global (ld ldg, may be null)
    __device__ __forceinline__ void store(T sign, T *lds, T *g, long ldg, int lane) const
    {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + BlkMma<T>::crow(lane, r), col = 16 * j + (lane & 15);
                    const T v = sign * t[i][j][r];
                    if (lds)
                        lds[row * PLD + col] = v;
                    if (g)
                        g[(size_t)row * ldg + col] = v;
                }
    }
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// workgroup barrier that waits for this wave's LDS traffic only (no vmcnt wait: see above)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Sub-block step of diag_ldlm_kernel, fp32: the 32 x 32 block Dn (lower triangle valid) is factorised in the accumulator of
// v_mfma_f32_32x32x2_f32 by one wave; L11 (strictly lower) -> Lx, L11^-1 -> Xdb, D -> dvec (lane = row).
__device__ __forceinline__ void subblock_ldl(const float *Dn, float *Lx, float *Xdb, int lane, float &dvec)
{
    typedef float T;
    const int half = lane >> 5, col = lane & 31;
    f32x16 M, X;
    {
        // symmetric from the lower triangle: element (row, col) and its mirror image, two base addresses and immediate offsets
        const T *b1 = Dn + 4 * half * PLD + col, *b2 = Dn + col * PLD + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = 8 * (r >> 2) + (r & 3), row = o + 4 * half;
            const T v1 = b1[o * PLD], v2 = b2[o];
            M[r] = row > col ? v1 : v2;
            X[r] = row == col ? 1.0f : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int rj = (j >> 3) * 4 + (j & 3), hj = (j >> 2) & 1;  // row j: register rj of wave half hj
        const T rowj = M[rj];
        const T dj = bcast_lane(rowj, j + 32 * hj);
        const T lj = rowj * fast_rcp(dj);
        const bool act = half == hj && col > j;
        const T a = act ? -lj : 0.0f;
        if (lane == j)
            dvec = dj;
        M = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rowj, M, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a, X[rj], X, 0, 0, 0);
        if (act)
            Lx[col * PLD + j] = lj;  // L11[col][j]
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        Xdb[(8 * (r >> 2) + 4 * half + (r & 3)) * PLD + col] = X[r];  // exact zeros above, ones on the diagonal
}

// fp64: no 32 x 32 form; the block is three 16 x 16 tiles of v_mfma_f64_16x16x4_f64 -- T00, T01 (rows 0-15 of the right
// half: the mirror image of the lower-left tile, which is never needed because the steps read ROWS) and T11 -- and its
// inverse three more (X00, X10, X11).  Element (row 4 r + g, col c) of a tile is register r of lane 16 g + c, so row j is
// register j / 4 of lane group j % 4: again the B operand as it stands (k = lane group) and, scaled, the A operand.
// The one-wave fp64 routine follows the two-wave one below (it shares the rank-4 panel helpers).
typedef double f64x4 __attribute__((ext_vector_type(4)));
// ---- the same on TWO waves: wave 0 factorises, wave 1 inverts one step behind -----------------------------------------------
// The updates of X = L11^-1 take the same A operand as those of the block (the scaled column l_j) but feed nothing back into the
// chain of pivots, and they are 2 of the 3-5 MFMAs of a step.  Wave 0 keeps the block: per step it publishes column j of L in
// Lx and then the step count in *prog (LDS operations of a wave execute in order); wave 1 -- on another SIMD -- waits for the
// count, reads the column back as its A operand and applies it to its rows of X.  prog counts from `base` (the caller zeroes it
// behind a barrier and passes 32 * (number of sub-blocks done)).  Same arithmetic as subblock_ldl (the pivot of the next step is
// formed ahead of the update, by the same fused multiply-add).  Measured (rank-1 steps, the fp32 form below): 7.3 -> 6.0 us per
// sub-block -- the step is a chain of latencies (MFMA result -> lane broadcast -> scaled column -> MFMA), not of MFMA issue slots:
// wave 0 alone, with neither the inverse nor the reciprocal on its chain, still needs 6.2 us.  The fp64 form takes RANK-4 steps
// (further down): 4.3 us.
//
// WAVE 1's MFMAs ARE INLINE ASM WITH THEIR OWN WAIT STATES, and that is the point (round 6).  gfx950 does not
// interlock a read of an MFMA result still in the pipe; the ISA asks for 11 (fp64 16x16x4; 18 for LDS / memory reads) or 18
// (fp32 32x32x2, 16 passes) software wait states, which hipcc pads behind a builtin MFMA -- walking the CFG backwards with ONE
// visited set.  ldl_column has two ways to the next MFMA: the poll loop (long) and the poll skipped because the count read
// earlier already covers the column (short).  The recogniser reached the MFMA through the loop first, marked its block visited,
// never walked the short edge, and sized the s_nop for the long path: on the short one the fp32 build read the B operand of the
// next step (v_accvgpr_read of X[r]) 6-7 wait states behind a 16-pass MFMA -- only when wave 0 happened to be two columns ahead,
// hence inverse blocks that differed from run to run (round 5: "cause not found").  The fp64 build had the same padding error
// and survived on arithmetic: its shortest feasible path holds 12 instructions and a DGEMM needs 11.  With the wait states
// written behind the MFMA itself no path can be shorter; gaussian-object-modelling_amd/codeobj.py (guard_mfma_result_hazards,
// run by build()) walks every path of the built code from every MFMA to the first touch of its result and fails the build on
// a short one.
// (the pointers are cast to the LDS address space by hand: through a generic volatile pointer hipcc emits flat loads and
// stores with system scope, hundreds of cycles each -- the first build of the pair was no faster than one wave for that)
#define GPX_LDS(T_) __attribute__((address_space(3))) T_
__device__ __forceinline__ void ldl_publish(volatile int *prog, int v)
{
    asm volatile("" ::: "memory");
    *(volatile GPX_LDS(int) *)prog = v;
    asm volatile("" ::: "memory");
}
// wave 1: the value at `src` of a column that is complete once *prog >= need.  `seen` is the last count read: a column known to be
// published is read without a poll; otherwise count and value are requested together, count first -- LDS operations execute in
// order, so a value read behind a count that was high enough is the published one.
template <typename T>
__device__ __forceinline__ T ldl_column(volatile int *prog, int need, int &seen, const T *src, bool take)
{
    const volatile GPX_LDS(T) *vs = (const volatile GPX_LDS(T) *)src;
    if (seen >= need)
        return take ? *vs : T(0);
    T val;
    do {
        seen = __builtin_amdgcn_readfirstlane(*(volatile GPX_LDS(int) *)prog);
        val = take ? *vs : T(0);
    } while (seen < need);
    return val;
}

// wave 1's rank-1 updates: the MFMA and the wait states the ISA wants before ANY later instruction may touch its result
// (18 covers a VALU read and an LDS / memory read of either form) in one asm statement -- see the comment above; the two in front
// cover a VALU write of an operand (hipcc pads nothing around an asm statement)
typedef double f64x4_blk __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ldl_x_update(f64x4_blk &x, double a, double b)
{
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 1" : "+a"(x) : "v"(a), "v"(b));
}
__device__ __forceinline__ void ldl_x_update2(f64x4_blk &x0, f64x4_blk &x1, double a0, double a1, double b)
{
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %2, %4, %0\n\tv_mfma_f64_16x16x4_f64 %1, %3, %4, %1\n\ts_nop 15\n\ts_nop 1"
                 : "+a"(x0), "+a"(x1)
                 : "v"(a0), "v"(a1), "v"(b));
}
__device__ __forceinline__ void ldl_x_update(f32x16 &x, float a, float b)
{
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 1" : "+a"(x) : "v"(a), "v"(b));
}

// ---- round 6: the fp64 pair in RANK-4 steps --------------------------------------------------------------------------------------
// A rank-1 step uses one of the four k slots of v_mfma_f64_16x16x4_f64 and pays, per COLUMN, the chain MFMA result -> pivot ->
// reciprocal -> scaled column -> MFMA (6.0 us per 32 x 32 sub-block: 450 cycles a column, the matrix pipe busy with 3 x 64 of them).
// In the accumulator layout the four rows 4 p + g (g = lane group) of a tile are register p of the four lane groups -- the B operand
// of a FULL-k MFMA as it stands.  A panel of four columns is therefore eliminated at once:
//   1. the panel's 4 x 4 diagonal block G (ten lane broadcasts) is factorised by every lane alike, G = L4 D4 L4^T, and W = L4^-1 formed;
//   2. ONE MFMA per tile applies W to the panel rows: U = W T_panel are the rows as the four rank-1 steps would have left them
//      (A operand: W in the lanes of rows 0-3, zero elsewhere; result row g in register 0 of lane group g -- the panel's own layout);
//   3. l = U / d are the four columns of L (published for wave 1 with ONE count), and ONE MFMA per tile subtracts all four from the
//      trailing rows (A = -l of the lane's own (row, k), zero for the rows of the panel and above; B = U).
// First half of the sub-block: 2 + 3 MFMAs per panel instead of 12, second half 1 + 1 instead of 4, and the chain per FOUR columns is
// broadcasts -> 4 x 4 factorisation (four reciprocals in sequence) -> two MFMA latencies.  Wave 1 applies the same panels to the inverse:
// the panel rows of X are REPLACED by W X_panel (they are final), the rows below take -l X'_panel; it rebuilds W from the six
// multipliers of the panel's 4 x 4 block that wave 0 published with the columns.  Same factorisation as the rank-1 form to rounding
// (the sums inside a panel are taken in another order); measured: profiles/r06_ldl_rank4.txt.
struct LdlPanel4 {
    double d[4], r[4], w10, w20, w21, w30, w31, w32;
};
// (g_ab = element (a, b) of the symmetric 4 x 4 block, a >= b)
__device__ __forceinline__ LdlPanel4 ldl_panel4(double g00, double g10, double g11, double g20, double g21, double g22, double g30,
                                                double g31, double g32, double g33)
{
    LdlPanel4 P;
    P.d[0] = g00, P.r[0] = fast_rcp(g00);
    const double l10 = g10 * P.r[0], l20 = g20 * P.r[0], l30 = g30 * P.r[0];
    P.d[1] = fma(-l10, g10, g11), P.r[1] = fast_rcp(P.d[1]);
    const double u21 = fma(-l20, g10, g21), u31 = fma(-l30, g10, g31);
    const double l21 = u21 * P.r[1], l31 = u31 * P.r[1];
    P.d[2] = fma(-l21, u21, fma(-l20, g20, g22)), P.r[2] = fast_rcp(P.d[2]);
    const double u32 = fma(-l31, u21, fma(-l30, g20, g32));
    const double l32 = u32 * P.r[2];
    P.d[3] = fma(-l32, u32, fma(-l31, u31, fma(-l30, g30, g33))), P.r[3] = fast_rcp(P.d[3]);
    P.w10 = -l10, P.w21 = -l21, P.w32 = -l32;
    P.w20 = fma(-l21, P.w10, -l20);
    P.w31 = fma(-l32, P.w21, -l31);
    P.w30 = fma(-l32, P.w20, fma(-l31, P.w10, -l30));
    return P;
}
// W = L4^-1 from the six multipliers of a unit lower 4 x 4 block
__device__ __forceinline__ void ldl_w_from_l(double l10, double l20, double l21, double l30, double l31, double l32, LdlPanel4 &P)
{
    P.w10 = -l10, P.w21 = -l21, P.w32 = -l32;
    P.w20 = fma(-l21, P.w10, -l20);
    P.w31 = fma(-l32, P.w21, -l31);
    P.w30 = fma(-l32, P.w20, fma(-l31, P.w10, -l30));
}
// the lane's element of the A operand that applies W to a panel: A[i][k] = W[i][k] for i < 4 (lane (k, i) = (g, c)), zero below
__device__ __forceinline__ double ldl_w_operand(const LdlPanel4 &P, int g, int c)
{
    double a = (c < 4 && c == g) ? 1.0 : 0.0;
    a = (c == 1 && g == 0) ? P.w10 : a;
    a = (c == 2 && g == 0) ? P.w20 : a;
    a = (c == 2 && g == 1) ? P.w21 : a;
    a = (c == 3 && g == 0) ? P.w30 : a;
    a = (c == 3 && g == 1) ? P.w31 : a;
    a = (c == 3 && g == 2) ? P.w32 : a;
    return a;
}
// wave 1: W X_panel into a VGPR tile, with the wait states of an asm MFMA (see ldl_x_update)
__device__ __forceinline__ double ldl_x_transform(double aw, double xp)
{
    f64x4_blk t;
    asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 1" : "=v"(t) : "v"(aw), "v"(xp));
    return t[0];
}

__device__ __forceinline__ void subblock_ldl_pair(const double *Dn, double *Lx, double *Xdb, int lane, int wave, double &dvec,
                                                  volatile int *prog, int base)
{
    typedef double T;
    const int g = lane >> 4, c = lane & 15;
    GPX_LDS(T) *LxL = (GPX_LDS(T) *)Lx;
    if (wave == 0) {
        f64x4 T00, T01, T11;
        {
            const T *b1 = Dn + g * PLD + c, *b2 = Dn + c * PLD + g;  // (row 4 r + g, col c) and its mirror image
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * r + g;
                const T lo = b1[4 * r * PLD], up = b2[4 * r];
                const T lo11 = b1[(16 + 4 * r) * PLD + 16], up11 = b2[16 * PLD + 16 + 4 * r];
                T00[r] = row > c ? lo : up;
                T01[r] = b2[16 * PLD + 4 * r];
                T11[r] = row > c ? lo11 : up11;
            }
        }
        const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // columns 4 p .. 4 p + 3: rows 4 p + g of T00 | T01 are register p
            const T tp0 = T00[p], tp1 = T01[p];
            const int c0 = 4 * p;
            const LdlPanel4 P = ldl_panel4(bcast_lane(tp0, c0), bcast_lane(tp0, 16 + c0), bcast_lane(tp0, 16 + c0 + 1),
                                           bcast_lane(tp0, 32 + c0), bcast_lane(tp0, 32 + c0 + 1), bcast_lane(tp0, 32 + c0 + 2),
                                           bcast_lane(tp0, 48 + c0), bcast_lane(tp0, 48 + c0 + 1), bcast_lane(tp0, 48 + c0 + 2),
                                           bcast_lane(tp0, 48 + c0 + 3));
            const T aw = ldl_w_operand(P, g, c);
            const T u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp0, zero4, 0, 0, 0)[0];
            const T u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp1, zero4, 0, 0, 0)[0];
            const T rk = g == 0 ? P.r[0] : (g == 1 ? P.r[1] : (g == 2 ? P.r[2] : P.r[3]));
            const T l0 = u0 * rk, l1 = u1 * rk;  // l[c][4 p + g], l[16 + c][4 p + g]
            if (lane >= c0 && lane < c0 + 4)
                dvec = (lane & 3) == 0 ? P.d[0] : ((lane & 3) == 1 ? P.d[1] : ((lane & 3) == 2 ? P.d[2] : P.d[3]));
            if (c > c0 + g)
                LxL[c * PLD + c0 + g] = l0;
            LxL[(16 + c) * PLD + c0 + g] = l1;
            ldl_publish(prog, base + c0 + 4);
            const T a00 = c > c0 + 3 ? -l0 : 0.0, a01 = -l1;
            T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, u0, T00, 0, 0, 0);
            T01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, u1, T01, 0, 0, 0);
            T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, u1, T11, 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {  // columns 16 + 4 p .. + 3: rows 16 + 4 p + g of T11
            const T tp = T11[p];
            const int c0 = 4 * p;
            const LdlPanel4 P = ldl_panel4(bcast_lane(tp, c0), bcast_lane(tp, 16 + c0), bcast_lane(tp, 16 + c0 + 1),
                                           bcast_lane(tp, 32 + c0), bcast_lane(tp, 32 + c0 + 1), bcast_lane(tp, 32 + c0 + 2),
                                           bcast_lane(tp, 48 + c0), bcast_lane(tp, 48 + c0 + 1), bcast_lane(tp, 48 + c0 + 2),
                                           bcast_lane(tp, 48 + c0 + 3));
            const T aw = ldl_w_operand(P, g, c);
            const T u = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp, zero4, 0, 0, 0)[0];
            const T rk = g == 0 ? P.r[0] : (g == 1 ? P.r[1] : (g == 2 ? P.r[2] : P.r[3]));
            const T l = u * rk;  // l[16 + c][16 + 4 p + g]
            if (lane >= 16 + c0 && lane < 16 + c0 + 4)
                dvec = (lane & 3) == 0 ? P.d[0] : ((lane & 3) == 1 ? P.d[1] : ((lane & 3) == 2 ? P.d[2] : P.d[3]));
            if (c > c0 + g)
                LxL[(16 + c) * PLD + 16 + c0 + g] = l;
            ldl_publish(prog, base + 16 + c0 + 4);
            if (p < 3)
                T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(c > c0 + 3 ? -l : 0.0, u, T11, 0, 0, 0);
        }
    } else {
        f64x4_blk X00, X10, X11;
        int seen = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X00[r] = 4 * r + g == c ? 1.0 : 0.0;
            X10[r] = 0.0;
            X11[r] = 4 * r + g == c ? 1.0 : 0.0;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c0 = 4 * p, need = base + c0 + 4;
            // the lane's multipliers of the four columns (rows c / 16 + c, column 4 p + g) and the panel's own 4 x 4 block
            const T a00 = -ldl_column(prog, need, seen, Lx + c * PLD + c0 + g, c > c0 + 3);
            const T a01 = -ldl_column(prog, need, seen, Lx + (16 + c) * PLD + c0 + g, true);
            LdlPanel4 P;
            ldl_w_from_l(ldl_column(prog, need, seen, Lx + (c0 + 1) * PLD + c0, true), ldl_column(prog, need, seen, Lx + (c0 + 2) * PLD + c0, true),
                         ldl_column(prog, need, seen, Lx + (c0 + 2) * PLD + c0 + 1, true), ldl_column(prog, need, seen, Lx + (c0 + 3) * PLD + c0, true),
                         ldl_column(prog, need, seen, Lx + (c0 + 3) * PLD + c0 + 1, true),
                         ldl_column(prog, need, seen, Lx + (c0 + 3) * PLD + c0 + 2, true), P);
            const T aw = ldl_w_operand(P, g, c);
            const T xp = ldl_x_transform(aw, X00[p]);  // rows 4 p + g of X: final
            X00[p] = xp;
            ldl_x_update2(X00, X10, a00, a01, xp);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c0 = 4 * p, need = base + 16 + c0 + 4;
            const T a11 = -ldl_column(prog, need, seen, Lx + (16 + c) * PLD + 16 + c0 + g, c > c0 + 3);
            LdlPanel4 P;
            const T *Lb = Lx + (16 + c0) * PLD + 16 + c0;  // the panel's 4 x 4 block of L11's lower right tile
            ldl_w_from_l(ldl_column(prog, need, seen, Lb + PLD, true), ldl_column(prog, need, seen, Lb + 2 * PLD, true),
                         ldl_column(prog, need, seen, Lb + 2 * PLD + 1, true), ldl_column(prog, need, seen, Lb + 3 * PLD, true),
                         ldl_column(prog, need, seen, Lb + 3 * PLD + 1, true), ldl_column(prog, need, seen, Lb + 3 * PLD + 2, true), P);
            const T aw = ldl_w_operand(P, g, c);
            const T x10p = ldl_x_transform(aw, X10[p]), x11p = ldl_x_transform(aw, X11[p]);
            X10[p] = x10p, X11[p] = x11p;
            if (p < 3) {
                ldl_x_update(X10, a11, x10p);
                ldl_x_update(X11, a11, x11p);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + g;
            Xdb[row * PLD + c] = X00[r];
            Xdb[row * PLD + 16 + c] = 0.0;
            Xdb[(16 + row) * PLD + c] = X10[r];
            Xdb[(16 + row) * PLD + 16 + c] = X11[r];
        }
    }
}

// The one-wave form (the 128 x 128 diagonal-block routine of the launch chain and of wide_factor_kernel, gpx_diag128.hpp) in the same
// rank-4 panels: block and inverse in one wave, 3 + 5 MFMAs per panel in the first half (rank-1 steps: 20), 3 + 3 in the second (12).
__device__ __forceinline__ void subblock_ldl(const double *Dn, double *Lx, double *Xdb, int lane, double &dvec)
{
    typedef double T;
    const int g = lane >> 4, c = lane & 15;
    f64x4 T00, T01, T11, X00, X10, X11;
    {
        const T *b1 = Dn + g * PLD + c, *b2 = Dn + c * PLD + g;  // (row 4 r + g, col c) and its mirror image
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * r + g;
            const T lo = b1[4 * r * PLD], up = b2[4 * r];
            const T lo11 = b1[(16 + 4 * r) * PLD + 16], up11 = b2[16 * PLD + 16 + 4 * r];
            T00[r] = row > c ? lo : up;
            T01[r] = b2[16 * PLD + 4 * r];           // (row, 16 + c) = mirror of (16 + c, row)
            T11[r] = row > c ? lo11 : up11;
            X00[r] = row == c ? 1.0 : 0.0;
            X10[r] = 0.0;
            X11[r] = row == c ? 1.0 : 0.0;
        }
    }
    const f64x4 zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const T tp0 = T00[p], tp1 = T01[p];
        const int c0 = 4 * p;
        const LdlPanel4 P = ldl_panel4(bcast_lane(tp0, c0), bcast_lane(tp0, 16 + c0), bcast_lane(tp0, 16 + c0 + 1),
                                       bcast_lane(tp0, 32 + c0), bcast_lane(tp0, 32 + c0 + 1), bcast_lane(tp0, 32 + c0 + 2),
                                       bcast_lane(tp0, 48 + c0), bcast_lane(tp0, 48 + c0 + 1), bcast_lane(tp0, 48 + c0 + 2),
                                       bcast_lane(tp0, 48 + c0 + 3));
        const T aw = ldl_w_operand(P, g, c);
        const T u0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp0, zero4, 0, 0, 0)[0];
        const T u1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp1, zero4, 0, 0, 0)[0];
        const T xp = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, X00[p], zero4, 0, 0, 0)[0];  // rows 4 p + g of the inverse: final
        const T rk = g == 0 ? P.r[0] : (g == 1 ? P.r[1] : (g == 2 ? P.r[2] : P.r[3]));
        const T l0 = u0 * rk, l1 = u1 * rk;
        if (lane >= c0 && lane < c0 + 4)
            dvec = (lane & 3) == 0 ? P.d[0] : ((lane & 3) == 1 ? P.d[1] : ((lane & 3) == 2 ? P.d[2] : P.d[3]));
        if (c > c0 + g)
            Lx[c * PLD + c0 + g] = l0;
        Lx[(16 + c) * PLD + c0 + g] = l1;
        const T a00 = c > c0 + 3 ? -l0 : 0.0, a01 = -l1;
        T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, u0, T00, 0, 0, 0);
        T01 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, u1, T01, 0, 0, 0);
        T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, u1, T11, 0, 0, 0);
        X00[p] = xp;
        X00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a00, xp, X00, 0, 0, 0);
        X10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a01, xp, X10, 0, 0, 0);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const T tp = T11[p];
        const int c0 = 4 * p;
        const LdlPanel4 P = ldl_panel4(bcast_lane(tp, c0), bcast_lane(tp, 16 + c0), bcast_lane(tp, 16 + c0 + 1),
                                       bcast_lane(tp, 32 + c0), bcast_lane(tp, 32 + c0 + 1), bcast_lane(tp, 32 + c0 + 2),
                                       bcast_lane(tp, 48 + c0), bcast_lane(tp, 48 + c0 + 1), bcast_lane(tp, 48 + c0 + 2),
                                       bcast_lane(tp, 48 + c0 + 3));
        const T aw = ldl_w_operand(P, g, c);
        const T u = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, tp, zero4, 0, 0, 0)[0];
        const T x10p = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, X10[p], zero4, 0, 0, 0)[0];
        const T x11p = __builtin_amdgcn_mfma_f64_16x16x4f64(aw, X11[p], zero4, 0, 0, 0)[0];
        const T rk = g == 0 ? P.r[0] : (g == 1 ? P.r[1] : (g == 2 ? P.r[2] : P.r[3]));
        const T l = u * rk;
        if (lane >= 16 + c0 && lane < 16 + c0 + 4)
            dvec = (lane & 3) == 0 ? P.d[0] : ((lane & 3) == 1 ? P.d[1] : ((lane & 3) == 2 ? P.d[2] : P.d[3]));
        if (c > c0 + g)
            Lx[(16 + c) * PLD + 16 + c0 + g] = l;
        X10[p] = x10p, X11[p] = x11p;
        if (p < 3) {
            const T a11 = c > c0 + 3 ? -l : 0.0;
            T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, u, T11, 0, 0, 0);
            X10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, x10p, X10, 0, 0, 0);
            X11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a11, x11p, X11, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * r + g;
        Xdb[row * PLD + c] = X00[r];
        Xdb[row * PLD + 16 + c] = 0.0;
        Xdb[(16 + row) * PLD + c] = X10[r];
        Xdb[(16 + row) * PLD + 16 + c] = X11[r];
    }
}

// fp32 twin: the block and its inverse are one 32 x 32 accumulator of v_mfma_f32_32x32x2_f32 each (layout of the one-wave routine
// above: register r of lane (half, col) is element (8 (r >> 2) + 4 half + (r & 3), col)).
__device__ __forceinline__ void subblock_ldl_pair(const float *Dn, float *Lx, float *Xdb, int lane, int wave, float &dvec,
                                                  volatile int *prog, int base)
{
    typedef float T;
    const int half = lane >> 5, col = lane & 31;
    GPX_LDS(T) *LxL = (GPX_LDS(T) *)Lx;
    if (wave == 0) {
        f32x16 M;
        {
            const T *b1 = Dn + 4 * half * PLD + col, *b2 = Dn + col * PLD + 4 * half;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = 8 * (r >> 2) + (r & 3), row = o + 4 * half;
                const T v1 = b1[o * PLD], v2 = b2[o];
                M[r] = row > col ? v1 : v2;
            }
        }
        T dj = bcast_lane(M[0], 0);
        T rinv = fast_rcp(dj);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rj = (j >> 3) * 4 + (j & 3), hj = (j >> 2) & 1;
            const T rowj = M[rj];
            const T lj = rowj * rinv;
            const bool act = half == hj && col > j;
            const T a = act ? -lj : 0.0f;
            if (lane == j)
                dvec = dj;
            if (act)
                LxL[col * PLD + j] = lj;
            ldl_publish(prog, base + j + 1);
            // next pivot: element (j + 1, j + 1) minus l_{j+1,j} times element (j, j + 1)
            const int jn = j + 1, rn = (jn >> 3) * 4 + (jn & 3), hn = (jn >> 2) & 1;
            const T m = j < NB - 1 ? bcast_lane(rowj, 32 * hj + jn) : 0.0f;
            const T old = j < NB - 1 ? bcast_lane(M[rn & 15], 32 * hn + (jn & 31)) : 1.0f;
            M = __builtin_amdgcn_mfma_f32_32x32x2f32(a, rowj, M, 0, 0, 0);
            dj = fmaf(-(m * rinv), m, old);
            rinv = fast_rcp(dj);
        }
    } else {
        f32x16 X;
        int seen = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            X[r] = 8 * (r >> 2) + 4 * half + (r & 3) == col ? 1.0f : 0.0f;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int rj = (j >> 3) * 4 + (j & 3), hj = (j >> 2) & 1;
            const bool act = half == hj && col > j;
            const T a = -ldl_column(prog, base + j + 1, seen, Lx + col * PLD + j, act);
            const T xr = X[rj];
            ldl_x_update(X, a, xr);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            Xdb[(8 * (r >> 2) + 4 * half + (r & 3)) * PLD + col] = X[r];
    }
}

}  // namespace gpx
