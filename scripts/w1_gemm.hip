// w1_gemm.hip -- prototype: the fp32 variance contraction with ONE wave per workgroup (one wave per SIMD, the whole
// 512-register file), a 128 x 128 tile per wave, operands straight from global memory into MFMA fragment registers
// (no LDS, no barriers), register double buffering one 16-deep k chunk ahead.  Plain (fp32) COLSQ epilogue; compared
// with the library's LDS-staged tiles on the same data.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaussian-object-modelling_amd/csrc scripts/w1_gemm.hip \
//          gaussian-object-modelling_amd/csrc/gpx_gemm.hip -o scripts/w1_gemm.bin
//   run  : scripts/w1_gemm.bin [N = 16384] [NQ = 8192]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#include "gpx_internal.hpp"
using namespace gpx;
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(err__), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f4v __attribute__((ext_vector_type(4)));

template <typename T>
__global__ void fill_kernel(T *d, size_t n, unsigned seed, double scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)(i * 2654435761u) ^ seed;
        s = s * 1664525u + 1013904223u;
        s ^= s >> 15;
        s = s * 1664525u + 1013904223u;
        d[i] = (T)(scale * ((double)(s >> 8) / (1 << 24) - 0.5));
    }
}
template <typename T>
static void fill(T *d, size_t n, unsigned seed, double scale)
{
    hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, d, n, seed, scale);
    CK(hipDeviceSynchronize());
}

#ifndef W1_KCH
#define W1_KCH 16
#endif

// partial[mt][n] = sum_{m in tile mt} (sum_k X[m][k] Kq[n][k])^2 dinv[m],   k < (mt + 1) * 128  (X lower triangular)
template <int VARIANT>
__global__ __attribute__((aligned(256))) __launch_bounds__(64, 1) void w1_kernel(const float *__restrict__ X, long ldx, const float *__restrict__ Kq, long ldk,
                                                                              const float *__restrict__ dinv, float *__restrict__ partial, long ldp, unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, mt = (int)(gridDim.y - 1 - blockIdx.y);
    const int m0 = mt * 128, n0 = nt * 128;
    const int r16 = lane & 15, g = lane >> 4;
    const int nch = (m0 + 128) / 16;  // 16-deep k chunks, a multiple of 8
    // uniform row-block bases (SGPR pairs) + one per-lane byte offset each for A and B
    const char *abase = (const char *)(X + (size_t)m0 * ldx);
    const char *bbase = (const char *)(Kq + (size_t)n0 * ldk);
    const unsigned aoff = (unsigned)(r16 * ldx * 4 + g * 16);
    const unsigned boff = (unsigned)(r16 * ldk * 4 + g * 16);
    const size_t astep = (size_t)16 * ldx * 4, bstep = (size_t)16 * ldk * 4;

    f4v acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            acc[i][j] = f4v{0.f, 0.f, 0.f, 0.f};

    float4 a0[8], b0[8], a1[8], b1[8];
#define W1_LOAD(A_, B_, C_)                                                                      \
    {                                                                                            \
        const unsigned kb_ = (unsigned)(C_) * 64u;                                               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                         \
            A_[i_] = *reinterpret_cast<const float4 *>(abase + i_ * astep + (aoff + kb_));       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                         \
            B_[i_] = *reinterpret_cast<const float4 *>(bbase + i_ * bstep + (boff + kb_));       \
    }
#define W1_MFMA4(A_, B_, S_)                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_)                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                         \
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(A_[i_].S_), "v"(B_[j_].S_));
#define W1_COMPUTE(A_, B_) \
    W1_MFMA4(A_, B_, x) W1_MFMA4(A_, B_, y) W1_MFMA4(A_, B_, z) W1_MFMA4(A_, B_, w)

    // one piece (a 16-byte load per lane) of the next chunk's operands: pieces 0-7 are A fragments, 8-15 B fragments
#define W1_PIECE(A_, B_, KB_, P_)                                                                              \
    {                                                                                                          \
        if ((P_) < 8)                                                                                          \
            A_[(P_) & 7] = *reinterpret_cast<const float4 *>(abase + ((P_) & 7) * astep + (aoff + (KB_)));     \
        else                                                                                                   \
            B_[(P_) & 7] = *reinterpret_cast<const float4 *>(bbase + ((P_) & 7) * bstep + (boff + (KB_)));     \
    }
#define W1_ROW(A_, B_, S_, I_)                                                                   \
    _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_)                                             \
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[I_][j_]) : "v"(A_[I_].S_), "v"(B_[j_].S_));
#define W1_HROW(A_, B_, S_, I_, H_)                                                              \
    _Pragma("unroll") for (int j_ = 4 * (H_); j_ < 4 * (H_) + 4; ++j_)                           \
        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[I_][j_]) : "v"(A_[I_].S_), "v"(B_[j_].S_));
    // the same with one piece per 4 MFMAs (all 16 pieces in the first quarter)
#define W1_COMPUTE_LD4(A_, B_, AN_, BN_, KB_)                                                    \
    {                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_PIECE(AN_, BN_, KB_, 2 * i_) W1_HROW(A_, B_, x, i_, 0) W1_PIECE(AN_, BN_, KB_, 2 * i_ + 1) W1_HROW(A_, B_, x, i_, 1) } \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, y, i_) }               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, z, i_) }               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, w, i_) }               \
    }
    // compute on (A_, B_) with the 16 pieces of (AN_, BN_) issued one per 8 MFMAs over the first half
#define W1_COMPUTE_LD(A_, B_, AN_, BN_, KB_)                                                     \
    {                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_PIECE(AN_, BN_, KB_, i_) W1_ROW(A_, B_, x, i_) }       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_PIECE(AN_, BN_, KB_, 8 + i_) W1_ROW(A_, B_, y, i_) }   \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, z, i_) }               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, w, i_) }               \
    }
    // VARIANT 7: buffer loads -- one descriptor per operand (uniform), the row-block step in the scalar offset, one
    // shared per-lane 32-bit offset: no per-load address arithmetic at all
    const auto arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(abase), 0, (int)(128 * ldx * 4), 0x00020000);
    const auto brsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(bbase), 0, (int)(128 * ldk * 4), 0x00020000);
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
#define W1_BPIECE(A_, B_, KB_, P_)                                                                                         \
    {                                                                                                                      \
        if ((P_) < 8)                                                                                                      \
            A_[(P_) & 7] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(arsrc, (int)(aoff + (KB_)), (int)(((P_) & 7) * astep), 0)); \
        else                                                                                                               \
            B_[(P_) & 7] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(brsrc, (int)(boff + (KB_)), (int)(((P_) & 7) * bstep), 0)); \
    }
#define W1_COMPUTE_BLD(A_, B_, AN_, BN_, KB_)                                                    \
    {                                                                                            \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_BPIECE(AN_, BN_, KB_, i_) W1_ROW(A_, B_, x, i_) }       \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_BPIECE(AN_, BN_, KB_, 8 + i_) W1_ROW(A_, B_, y, i_) }   \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, z, i_) }               \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) { W1_ROW(A_, B_, w, i_) }               \
    }
    W1_LOAD(a0, b0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (VARIANT == 6) {  // diagnostic: no loads in the loop at all (results are wrong; what the MFMA issue alone costs)
        W1_LOAD(a1, b1, 1);
        asm volatile(".p2align 6");
        for (int c = 0; c < nch; c += 2) {
            W1_COMPUTE(a0, b0);
            W1_COMPUTE(a1, b1);
        }
    } else if (VARIANT == 7) {
        asm volatile(".p2align 6");
        for (int c = 0; c < nch; c += 2) {
            const unsigned kb1 = (unsigned)(c + 1) * 64u, kb2 = (unsigned)min(c + 2, nch - 1) * 64u;
            W1_COMPUTE_BLD(a0, b0, a1, b1, kb1);
            W1_COMPUTE_BLD(a1, b1, a0, b0, kb2);
        }
    } else if (VARIANT == 0) {
        asm volatile(".p2align 6");
        for (int c = 0; c < nch; c += 2) {
            W1_LOAD(a1, b1, c + 1);
            W1_COMPUTE(a0, b0);
            W1_LOAD(a0, b0, min(c + 2, nch - 1));
            W1_COMPUTE(a1, b1);
        }
    } else if (VARIANT == 1 || VARIANT == 4) {
        // VARIANT 4: the diagnostic build -- shader-clock and 100-MHz stamps around the loop (stamps[] is read by nothing else)
        asm volatile(".p2align 6");
        for (int c = 0; c < nch; c += 2) {
            const unsigned kb1 = (unsigned)(c + 1) * 64u, kb2 = (unsigned)min(c + 2, nch - 1) * 64u;
            W1_COMPUTE_LD(a0, b0, a1, b1, kb1);
            W1_COMPUTE_LD(a1, b1, a0, b0, kb2);
        }
    } else if (VARIANT == 2) {
        asm volatile(".p2align 6");
        for (int c = 0; c < nch; c += 2) {
            const unsigned kb1 = (unsigned)(c + 1) * 64u, kb2 = (unsigned)min(c + 2, nch - 1) * 64u;
            W1_COMPUTE_LD4(a0, b0, a1, b1, kb1);
            W1_COMPUTE_LD4(a1, b1, a0, b0, kb2);
        }
    } else {
        // three register buffers, loads two chunks ahead (nch is a multiple of 8; the loop handles 3 chunks per trip
        // and the remainder is finished on clamped indices -- extra loads only, never extra MFMAs)
        float4 a2[8], b2[8];
        W1_LOAD(a1, b1, 1);
        asm volatile(".p2align 6");
        int c = 0;
        for (; c + 3 <= nch; c += 3) {
            const unsigned kb2 = (unsigned)min(c + 2, nch - 1) * 64u, kb3 = (unsigned)min(c + 3, nch - 1) * 64u, kb4 = (unsigned)min(c + 4, nch - 1) * 64u;
            W1_COMPUTE_LD(a0, b0, a2, b2, kb2);
            W1_COMPUTE_LD(a1, b1, a0, b0, kb3);
            W1_COMPUTE_LD(a2, b2, a1, b1, kb4);
        }
        // nch mod 3 chunks left (nch = 8 m: remainder 2, 1 or 0), operands already in (a0, b0), (a1, b1)
        if (c < nch) { W1_COMPUTE(a0, b0); ++c; }
        if (c < nch) { W1_COMPUTE(a1, b1); ++c; }
    }

    {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = stamps + 4 * ((size_t)mt * gridDim.x + nt);
            o[0] = t1 - t0, o[1] = r1 - r0, o[2] = (unsigned long long)nch, o[3] = r0;
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");  // the last MFMAs' results before the accumulators are read
    // plain COLSQ epilogue: acc[i][j][r] is row 16 i + 4 g + r, column 16 j + r16
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 w = *reinterpret_cast<const float4 *>(dinv + m0 + 16 * i + 4 * g);
            s = fmaf(acc[i][j][0] * acc[i][j][0], w.x, s);
            s = fmaf(acc[i][j][1] * acc[i][j][1], w.y, s);
            s = fmaf(acc[i][j][2] * acc[i][j][2], w.z, s);
            s = fmaf(acc[i][j][3] * acc[i][j][3], w.w, s);
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0)
            partial[(size_t)mt * ldp + n0 + 16 * j + r16] = s;
    }
}

typedef float f16v __attribute__((ext_vector_type(16)));
// the same with v_mfma_f32_32x32x2_f32: 4 x 4 fragments of 32 x 32, lane (r32 = lane & 31, g2 = lane >> 5) holds k = 8 g2 .. 8 g2 + 7
// of a 16-deep chunk (two 16-byte loads per fragment row), MFMA step s uses k = 8 g2 + s
__global__ __attribute__((aligned(256))) __launch_bounds__(64, 1) void w1_kernel32(const float *__restrict__ X, long ldx, const float *__restrict__ Kq, long ldk,
                                                                                const float *__restrict__ dinv, float *__restrict__ partial, long ldp, unsigned long long *__restrict__ stamps)
{
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, mt = (int)(gridDim.y - 1 - blockIdx.y);
    const int m0 = mt * 128, n0 = nt * 128;
    const int r32 = lane & 31, g2 = lane >> 5;
    const int nch = (m0 + 128) / 16;
    const char *abase = (const char *)(X + (size_t)m0 * ldx);
    const char *bbase = (const char *)(Kq + (size_t)n0 * ldk);
    const unsigned aoff = (unsigned)(r32 * ldx * 4 + g2 * 32);
    const unsigned boff = (unsigned)(r32 * ldk * 4 + g2 * 32);
    const size_t astep = (size_t)32 * ldx * 4, bstep = (size_t)32 * ldk * 4;
    f16v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                acc[i][j][r] = 0.f;
    // pieces 0-7: A (fragment p >> 1, half p & 1), 8-15: B
    float4 a0[8], b0[8], a1[8], b1[8];
#define W32_PIECE(A_, B_, KB_, P_)                                                                                       \
    {                                                                                                                    \
        if ((P_) < 8)                                                                                                    \
            A_[(P_) & 7] = *reinterpret_cast<const float4 *>(abase + (((P_) & 7) >> 1) * astep + (aoff + (KB_) + 16 * ((P_) & 1))); \
        else                                                                                                             \
            B_[(P_) & 7] = *reinterpret_cast<const float4 *>(bbase + (((P_) & 7) >> 1) * bstep + (boff + (KB_) + 16 * ((P_) & 1))); \
    }
#define W32_STEP(A_, B_, H_, S_)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                             \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                         \
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i_][j_]) : "v"(A_[2 * i_ + (H_)].S_), "v"(B_[2 * j_ + (H_)].S_));
#define W32_COMPUTE_LD(A_, B_, AN_, BN_, KB_)                                                    \
    {                                                                                            \
        W32_PIECE(AN_, BN_, KB_, 0) W32_PIECE(AN_, BN_, KB_, 1) W32_PIECE(AN_, BN_, KB_, 2) W32_PIECE(AN_, BN_, KB_, 3)      \
        W32_STEP(A_, B_, 0, x)                                                                   \
        W32_PIECE(AN_, BN_, KB_, 4) W32_PIECE(AN_, BN_, KB_, 5) W32_PIECE(AN_, BN_, KB_, 6) W32_PIECE(AN_, BN_, KB_, 7)      \
        W32_STEP(A_, B_, 0, y)                                                                   \
        W32_PIECE(AN_, BN_, KB_, 8) W32_PIECE(AN_, BN_, KB_, 9) W32_PIECE(AN_, BN_, KB_, 10) W32_PIECE(AN_, BN_, KB_, 11)    \
        W32_STEP(A_, B_, 0, z)                                                                   \
        W32_PIECE(AN_, BN_, KB_, 12) W32_PIECE(AN_, BN_, KB_, 13) W32_PIECE(AN_, BN_, KB_, 14) W32_PIECE(AN_, BN_, KB_, 15)  \
        W32_STEP(A_, B_, 0, w)                                                                   \
        W32_STEP(A_, B_, 1, x) W32_STEP(A_, B_, 1, y) W32_STEP(A_, B_, 1, z) W32_STEP(A_, B_, 1, w) \
    }
#pragma unroll
    for (int p = 0; p < 16; ++p)
        W32_PIECE(a0, b0, 0u, p)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    asm volatile(".p2align 6");
    for (int c = 0; c < nch; c += 2) {
        const unsigned kb1 = (unsigned)(c + 1) * 64u, kb2 = (unsigned)min(c + 2, nch - 1) * 64u;
        W32_COMPUTE_LD(a0, b0, a1, b1, kb1);
        W32_COMPUTE_LD(a1, b1, a0, b0, kb2);
    }
    {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = stamps + 4 * ((size_t)mt * gridDim.x + nt);
            o[0] = t1 - t0, o[1] = r1 - r0, o[2] = (unsigned long long)nch, o[3] = r0;
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    // acc[i][j][r]: row 32 i + 8 (r / 4) + 4 g2 + (r % 4), column 32 j + r32
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w = *reinterpret_cast<const float4 *>(dinv + m0 + 32 * i + 8 * q + 4 * g2);
                s = fmaf(acc[i][j][4 * q + 0] * acc[i][j][4 * q + 0], w.x, s);
                s = fmaf(acc[i][j][4 * q + 1] * acc[i][j][4 * q + 1], w.y, s);
                s = fmaf(acc[i][j][4 * q + 2] * acc[i][j][4 * q + 2], w.z, s);
                s = fmaf(acc[i][j][4 * q + 3] * acc[i][j][4 * q + 3], w.w, s);
            }
        s += __shfl_xor(s, 32);
        if (g2 == 0)
            partial[(size_t)mt * ldp + n0 + 32 * j + r32] = s;
    }
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 16384;
    const int NQ = argc > 2 ? atoi(argv[2]) : 8192;
    float *X, *Kqp, *dinv, *p_ref, *p_new;
    CK(hipMalloc(&X, 4 * (size_t)N * N));
    CK(hipMalloc(&Kqp, 4 * (size_t)NQ * N));
    CK(hipMalloc(&dinv, 4 * (size_t)N));
    CK(hipMalloc(&p_ref, 4 * (size_t)NQ * (N / 128)));
    CK(hipMalloc(&p_new, 4 * (size_t)NQ * (N / 128)));
    fill(X, (size_t)N * N, 1, 1e-2);
    fill(Kqp, (size_t)NQ * N, 2, 1.0);
    fill(dinv, N, 3, 1.0);
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double flop = (double)N * N * NQ;
    for (int cfg : {3, 0}) {
        GemmArgs a;
        a.A = X, a.lda = N; a.B = Kqp, a.ldb = N; a.M = N, a.N = NQ, a.K = N; a.a_lower = 1; a.epi = EPI_COLSQ;
        a.rowweight = dinv; a.partial = p_ref, a.ldp = NQ; a.cfg = cfg;
        launch_gemm(0, a, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < 10; ++r) launch_gemm(0, a, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
        printf("library cfg%d (plain epilogue): %.3f ms  %.1f TFLOP/s\n", cfg, ms, flop / ms / 1e9);
    }
    dim3 grid(NQ / 128, N / 128);
    std::vector<float> hr((size_t)NQ * (N / 128)), hn(hr.size());
    CK(hipMemcpy(hr.data(), p_ref, 4 * hr.size(), hipMemcpyDeviceToHost));
    unsigned long long *stamps;
    CK(hipMalloc(&stamps, 32 * (size_t)grid.x * grid.y));
    for (int round = 0; round < 2; ++round)
    for (int variant : {1, 7, 6}) {
        auto kern = variant == 1 ? w1_kernel<1> : variant == 7 ? w1_kernel<7> : variant == 6 ? w1_kernel<6> : w1_kernel32;
        CK(hipMemsetAsync(p_new, 0, 4 * hr.size(), st));
        hipLaunchKernelGGL(kern, grid, dim3(64), 0, st, X, (long)N, Kqp, (long)N, dinv, p_new, (long)NQ, stamps);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        const int reps = 10;
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(kern, grid, dim3(64), 0, st, X, (long)N, Kqp, (long)N, dinv, p_new, (long)NQ, stamps);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        CK(hipMemcpy(hn.data(), p_new, 4 * hn.size(), hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        for (size_t i = 0; i < hr.size(); ++i) {
            md = fmax(md, fabs((double)hr[i] - hn[i]));
            mx = fmax(mx, fabs((double)hr[i]));
        }
        {
            std::vector<unsigned long long> hs(4 * (size_t)grid.x * grid.y);
            CK(hipMemcpy(hs.data(), stamps, 8 * hs.size(), hipMemcpyDeviceToHost));
            std::vector<double> clk, cyc;
            for (size_t w = 0; w < hs.size() / 4; ++w)
                if (hs[4 * w + 2] >= 256) {  // long loops only
                    clk.push_back((double)hs[4 * w] / (double)hs[4 * w + 1] * 100.0);
                    cyc.push_back((double)hs[4 * w] / (double)hs[4 * w + 2]);
                }
            std::sort(clk.begin(), clk.end());
            std::sort(cyc.begin(), cyc.end());
            printf("  in-kernel clock (MHz) min / median / max: %.0f / %.0f / %.0f ; shader cycles per 16-deep chunk (256 MFMAs = 8192 pipe cycles): min / median / max %.0f / %.0f / %.0f\n",
                   clk.front(), clk[clk.size() / 2], clk.back(), cyc.front(), cyc[cyc.size() / 2], cyc.back());
        }
        printf("one-wave 128x128 tile, no LDS, variant %d : %.3f ms  %.1f TFLOP/s   max |partial - library| / max = %.3e\n", variant, ms, flop / ms / 1e9, md / mx);
    }
    return 0;
}
