// mfma_lat_probe.hip -- development probe: issue / dependent latency of the f32-input MFMAs and of the elimination step
// of diag_ldlm_kernel<float> (one wave, one workgroup), in shader clocks (s_memtime) and ns (s_memrealtime, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_lat_probe.hip -o scripts/mfma_lat_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rl(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
__device__ __forceinline__ float frcp(float x)
{
    float y = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, y, 1.0f), y, y);
}

constexpr int IT = 64;

template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, long long *t)
{
    const int lane = threadIdx.x;
    f32x16 M, X;
    f32x4 m4[4];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        M[r] = 1.0f + 0.001f * (float)((lane * 7 + r * 3) % 13);
        X[r] = 0.5f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        m4[q] = f32x4{1.0f, 1.1f, 1.2f, 1.3f};
    float a = 1e-3f * (float)(lane & 3), b = 1e-3f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < IT; ++it) {
        if (MODE == 0) {  // dependent 32x32x2 chain
#pragma unroll
            for (int u = 0; u < 8; ++u)
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, M, 0, 0, 0);
        } else if (MODE == 1) {  // two independent accumulators
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, M, 0, 0, 0);
                X = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, X, 0, 0, 0);
            }
        } else if (MODE == 2 || MODE == 3) {  // the elimination step (with / without the X update)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int rj = (j >> 3) * 4 + (j & 3), hj = (j >> 2) & 1;
                const float rowj = M[rj];
                const float dj = rl(rowj, j + 32 * hj);
                const float lj = rowj * frcp(dj);
                const bool act = (lane >> 5) == hj && (lane & 31) > j;
                const float aa = act ? -lj : 0.0f;
                M = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, rowj, M, 0, 0, 0);
                if (MODE == 2)
                    X = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, X[rj], X, 0, 0, 0);
            }
        } else if (MODE == 4) {  // dependent 16x16x4 chain
#pragma unroll
            for (int u = 0; u < 8; ++u)
                m4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m4[0], 0, 0, 0);
        } else if (MODE == 5) {  // four independent 16x16x4
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    m4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m4[q], 0, 0, 0);
        } else if (MODE == 6) {  // the VALU part of the step alone, chained through a register
            float v = M[0];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dj = rl(v, j);
                const float lj = v * frcp(dj);
                v = (lane & 31) > j ? -lj : 1.0f;
            }
            M[0] = v;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        s += M[r] + X[r];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        s += m4[q][0] + m4[q][3];
    out[lane] = s;
    if (lane == 0)
        t[0] = c1 - c0, t[1] = w1 - w0;
}

template <int MODE>
static void run(const char *what, float *out, long long *t)
{
    long long h[2];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, out, t);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
    const double n = 8.0 * IT;
    printf("%-58s %7.1f s_memtime ticks  %7.1f ns  per unit\n", what, (double)h[0] / n, (double)h[1] * 10.0 / n);
}

int main()
{
    float *out;
    long long *t;
    (void)hipMalloc(&out, 256), (void)hipMalloc(&t, 16);
    run<0>("32x32x2 f32, dependent accumulator", out, t);
    run<1>("32x32x2 f32, two accumulators alternating (per MFMA)", out, t);
    run<2>("elimination step: pivot row -> rcp -> 2 MFMA", out, t);
    run<3>("elimination step without the inverse's MFMA", out, t);
    run<4>("16x16x4 f32, dependent accumulator", out, t);
    run<5>("16x16x4 f32, four accumulators (per MFMA)", out, t);
    run<6>("VALU part of the step (readlane, rcp + Newton, mul, select)", out, t);
    return 0;
}
