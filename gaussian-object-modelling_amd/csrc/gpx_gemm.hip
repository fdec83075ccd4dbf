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
#include "gpx_internal.hpp"

namespace gpx {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T>
struct MfmaT;
template <>
struct MfmaT<float> {
    using acc_t = f32x4;
    static constexpr int BK = 32;  // k-tile: 128 bytes per row
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + r
    static __device__ __forceinline__ int crow(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <>
struct MfmaT<double> {
    using acc_t = f64x4;
    static constexpr int BK = 16;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * r
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};

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
};

__device__ __forceinline__ void tri_decode_g(int t, int &ti, int &tj)
{
    int i = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (i * (i + 1) / 2 > t)
        --i;
    while ((i + 1) * (i + 2) / 2 <= t)
        ++i;
    ti = i;
    tj = t - i * (i + 1) / 2;
}

template <typename T, bool NN, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmDev<T> g)
{
    using MF = MfmaT<T>;
    using acc_t = typename MF::acc_t;
    constexpr int BK = MF::BK;
    constexpr int EPC = 16 / sizeof(T);         // elements per 16-byte chunk
    constexpr int BKP = BK + EPC;               // padded k extent of a [row][k] LDS tile
    constexpr int BNP = TILE + EPC;             // padded n extent of a [k][n] LDS tile (NN)
    constexpr int A_TILE = TILE * BKP;
    constexpr int B_TILE = NN ? BK * BNP : TILE * BKP;
    __shared__ __attribute__((aligned(16))) T smem[2 * A_TILE + 2 * B_TILE];
    T *As = smem;
    T *Bs = smem + 2 * A_TILE;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- which tile ----
    int mt, nt;
    if (g.lower_only) {
        tri_decode_g((int)blockIdx.x, mt, nt);
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
    const int m0 = mt * TILE, n0 = nt * TILE;
    if (m0 >= Mz)
        return;
    const T *A = g.A + (size_t)z * g.sA;
    const T *B = g.B + (size_t)z * g.sB;

    int klo = 0, khi = Kz;
    if (g.a_lower)
        khi = min(khi, m0 + TILE);
    if (g.b_lower) {
        if (NN)
            klo = n0;
        else
            khi = min(khi, n0 + TILE);
    }
    const int kt0 = klo / BK, kt1 = khi / BK;

    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[i][j][r] = T(0);

    // ---- staging maps (4 x 16-byte chunks per thread per operand) ----
    uint4 ra[4], rb[4];
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            {
                const int row = c >> 3, kc = c & 7;
                ra[i] = *reinterpret_cast<const uint4 *>(A + (size_t)(m0 + row) * g.lda + k0 + kc * EPC);
            }
            if constexpr (NN) {
                constexpr int CPR = TILE / EPC;  // chunks per k-row
                const int kr = c / CPR, cc = c % CPR;
                rb[i] = *reinterpret_cast<const uint4 *>(B + (size_t)(k0 + kr) * g.ldb + n0 + cc * EPC);
            } else {
                const int row = c >> 3, kc = c & 7;
                rb[i] = *reinterpret_cast<const uint4 *>(B + (size_t)(n0 + row) * g.ldb + k0 + kc * EPC);
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = tid + 256 * i;
            {
                const int row = c >> 3, kc = c & 7;
                *reinterpret_cast<uint4 *>(As + buf * A_TILE + row * BKP + kc * EPC) = ra[i];
            }
            if constexpr (NN) {
                constexpr int CPR = TILE / EPC;
                const int kr = c / CPR, cc = c % CPR;
                *reinterpret_cast<uint4 *>(Bs + buf * B_TILE + kr * BNP + cc * EPC) = rb[i];
            } else {
                const int row = c >> 3, kc = c & 7;
                *reinterpret_cast<uint4 *>(Bs + buf * B_TILE + row * BKP + kc * EPC) = rb[i];
            }
        }
    };

    const int fr = lane & 15, fg = lane >> 4;
    auto compute = [&](int buf) {
        const T *as = As + buf * A_TILE + (wm * 64 + fr) * BKP + 4 * fg;
        const T *bs = NN ? Bs + buf * B_TILE + (4 * fg) * BNP + wn * 64 + fr
                         : Bs + buf * B_TILE + (wn * 64 + fr) * BKP + 4 * fg;
#pragma unroll
        for (int kc = 0; kc < BK / 16; ++kc) {
            T a[4][4], b[4][4];
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const T *p = as + f * 16 * BKP + kc * 16;
                if constexpr (sizeof(T) == 4) {
                    float4 v = *reinterpret_cast<const float4 *>(p);
                    a[f][0] = v.x, a[f][1] = v.y, a[f][2] = v.z, a[f][3] = v.w;
                } else {
                    double2 v0 = *reinterpret_cast<const double2 *>(p);
                    double2 v1 = *reinterpret_cast<const double2 *>(p + 2);
                    a[f][0] = v0.x, a[f][1] = v0.y, a[f][2] = v1.x, a[f][3] = v1.y;
                }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if constexpr (NN) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        b[f][s] = bs[(kc * 16 + s) * BNP + f * 16];
                } else {
                    const T *p = bs + f * 16 * BKP + kc * 16;
                    if constexpr (sizeof(T) == 4) {
                        float4 v = *reinterpret_cast<const float4 *>(p);
                        b[f][0] = v.x, b[f][1] = v.y, b[f][2] = v.z, b[f][3] = v.w;
                    } else {
                        double2 v0 = *reinterpret_cast<const double2 *>(p);
                        double2 v1 = *reinterpret_cast<const double2 *>(p + 2);
                        b[f][0] = v0.x, b[f][1] = v0.y, b[f][2] = v1.x, b[f][3] = v1.y;
                    }
                }
            }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = MF::run(a[i][s], b[j][s], acc[i][j]);
        }
    };

    // ---- main loop: one barrier per k-tile, next tile's global loads in flight over the MFMAs
    if (kt0 < kt1) {
        gload(kt0);
        sstore(0);
        __syncthreads();
        int buf = 0;
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more)
                gload(kt + 1);
            compute(buf);
            if (more)
                sstore(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }

    // ---- epilogues ----
    if constexpr (EPI == EPI_STORE) {
        T *C = g.C + (size_t)z * g.sC;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 64 + i * 16 + MF::crow(lane, r);
                    const int col = n0 + wn * 64 + j * 16 + fr;
                    T *p = C + (size_t)row * g.ldc + col;
                    T v = g.alpha * acc[i][j][r];
                    if (g.beta)
                        v += *p;
                    *p = v;
                }
    } else if constexpr (EPI == EPI_TRSM) {
        T *C = g.C + (size_t)z * g.sC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + fr;
            const T cs = g.colscale[col];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 64 + i * 16 + MF::crow(lane, r);
                    const T v = acc[i][j][r];
                    g.W[(size_t)row * g.ldw + col] = v;
                    C[(size_t)row * g.ldc + col] = v * cs;
                }
        }
    } else {  // EPI_COLSQ
        __shared__ T red[2][TILE];
        T w[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                w[i][r] = g.rowweight[m0 + wm * 64 + i * 16 + MF::crow(lane, r)];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            T s = T(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s += acc[i][j][r] * acc[i][j][r] * w[i][r];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (fg == 0)
                red[wm][wn * 64 + j * 16 + fr] = s;
        }
        __syncthreads();
        if (tid < TILE)
            g.partial[(size_t)mt * g.ldp + n0 + tid] = red[0][tid] + red[1][tid];
    }
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
    const int mt = a.M / TILE, nt = a.N / TILE;
    if (mt <= 0 || nt <= 0 || a.batch <= 0)
        return;
    dim3 grid = a.lower_only ? dim3(mt * (mt + 1) / 2, 1, a.batch) : dim3(nt, mt, a.batch);
#define GPX_GEMM_LAUNCH(NN_, EPI_) hipLaunchKernelGGL((gemm_kernel<T, NN_, EPI_>), grid, dim3(256), 0, st, g)
    if (a.epi == EPI_STORE) {
        if (a.nn)
            GPX_GEMM_LAUNCH(true, EPI_STORE);
        else
            GPX_GEMM_LAUNCH(false, EPI_STORE);
    } else if (a.epi == EPI_TRSM) {
        GPX_GEMM_LAUNCH(false, EPI_TRSM);
    } else {
        GPX_GEMM_LAUNCH(false, EPI_COLSQ);
    }
#undef GPX_GEMM_LAUNCH
}

void launch_gemm(int prec, const GemmArgs &g, hipStream_t st)
{
    if (prec == GPX_PREC_F64)
        gemm_t<double>(g, st);
    else
        gemm_t<float>(g, st);
}

}  // namespace gpx
