// mfma_filler_probe2.hip -- development probe (round 5): the question of mfma_filler_probe.hip asked of three kinds of MFMA:
// does INDEPENDENT vector-ALU work of the same wave hide behind the matrix instruction?
//   f32 16x16x4 (f32 inputs: the small-model kernel's), f64 16x16x4 (the fp64 kernels'), f32 16x16x32 f16 (a split-fp16 form's).
// Loop body = 8 x [1 MFMA on its own accumulator + K independent fillers]; shader clocks per MFMA for K = 0 .. 12, with 1 wave,
// 4 waves (one per SIMD) and 8 waves (two per SIMD) in the workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_filler_probe2.hip -o scripts/mfma_filler_probe2.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef double d4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
constexpr int IT = 128, NMF = 8, KMAX = 12;
enum { M_F32 = 0, M_F64, M_F16 };
enum { F_FMA32 = 0, F_FMA64, F_EXP32 };

template <int MF, int KIND, int K>
__global__ __launch_bounds__(512) void probe(float *out, long long *t)
{
    const int lane = threadIdx.x & 63;
    float a = 1e-3f * lane, b = 1e-3f;
    double ad = 1e-3 * lane, bd = 1e-3;
    h8v ah, bh;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        ah[i] = (_Float16)(1e-3f * lane), bh[i] = (_Float16)1e-3f;
    float xs[KMAX];
    double ds[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        xs[k] = 0.5f + 1e-3f * k, ds[k] = 0.5 + 1e-3 * k;
    f4v acc[NMF];
    d4v accd[NMF];
#pragma unroll
    for (int c = 0; c < NMF; ++c)
        acc[c] = f4v{0, 0, 0, 0}, accd[c] = d4v{0, 0, 0, 0};
    __syncthreads();
    const long long c0 = clock64();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int c = 0; c < NMF; ++c) {
            if (MF == M_F32)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b));
            else if (MF == M_F64)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(accd[c]) : "v"(ad), "v"(bd));
            else
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(ah), "v"(bh));
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == F_FMA32)
                    asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(xs[k]));
                else if (KIND == F_FMA64)
                    asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(ds[k]));
                else
                    asm volatile("v_exp_f32 %0, %0" : "+v"(xs[k]));
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        s += xs[k] + (float)ds[k];
#pragma unroll
    for (int c = 0; c < NMF; ++c)
        s += acc[c].x + (float)accd[c].x;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0)
        t[0] = c1 - c0;
}

template <int MF, int KIND, int K>
double run1(float *out, long long *t, int threads)
{
    hipLaunchKernelGGL((probe<MF, KIND, K>), dim3(1), dim3(threads), 0, 0, out, t);
    hipLaunchKernelGGL((probe<MF, KIND, K>), dim3(1), dim3(threads), 0, 0, out, t);
    long long h = 0;
    (void)hipMemcpy(&h, t, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h / (IT * NMF);
}

template <int MF, int KIND>
void run(const char *name, float *out, long long *t)
{
    const int th[3] = {64, 256, 512};
    const char *wn[3] = {"1 wave ", "4 waves", "8 waves"};
    for (int w = 0; w < 3; ++w) {
        printf("%-34s %s:", w == 0 ? name : "", wn[w]);
        printf(" %5.1f", run1<MF, KIND, 0>(out, t, th[w])), printf(" %5.1f", run1<MF, KIND, 1>(out, t, th[w]));
        printf(" %5.1f", run1<MF, KIND, 2>(out, t, th[w])), printf(" %5.1f", run1<MF, KIND, 4>(out, t, th[w]));
        printf(" %5.1f", run1<MF, KIND, 6>(out, t, th[w])), printf(" %5.1f", run1<MF, KIND, 8>(out, t, th[w]));
        printf(" %5.1f\n", run1<MF, KIND, 12>(out, t, th[w]));
    }
}

int main()
{
    float *out;
    long long *t;
    (void)hipMalloc(&out, 512 * sizeof(float));
    (void)hipMalloc(&t, sizeof(long long));
    printf("shader clocks per MFMA (wave 0's clock) with K independent fillers behind each MFMA, K = 0 1 2 4 6 8 12\n");
    run<M_F32, F_FMA32>("f32 16x16x4      + v_fma_f32", out, t);
    run<M_F32, F_FMA64>("f32 16x16x4      + v_fma_f64", out, t);
    run<M_F64, F_FMA32>("f64 16x16x4      + v_fma_f32", out, t);
    run<M_F64, F_FMA64>("f64 16x16x4      + v_fma_f64", out, t);
    run<M_F16, F_FMA32>("f32 16x16x32 f16 + v_fma_f32", out, t);
    run<M_F16, F_FMA64>("f32 16x16x32 f16 + v_fma_f64", out, t);
    run<M_F16, F_EXP32>("f32 16x16x32 f16 + v_exp_f32", out, t);
    return 0;
}
