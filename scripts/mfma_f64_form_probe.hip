#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4v __attribute__((ext_vector_type(4)));
constexpr int IT = 256, NMF = 8;
template <int MODE>
__global__ __launch_bounds__(64) void probe(float *out, long long *t)
{
    const int lane = threadIdx.x & 63;
    double ad = 1e-3 * lane, bd = 1e-3;
    double av[NMF];
    for (int c = 0; c < NMF; ++c) av[c] = 1e-3 * (lane + c);
    d4v acc[NMF];
#pragma unroll
    for (int c = 0; c < NMF; ++c) acc[c] = d4v{0, 0, 0, 0};
    const long long c0 = clock64();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int c = 0; c < NMF; ++c) {
            if (MODE == 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(ad), "v"(bd));
            else if (MODE == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(ad), "v"(bd));
            else if (MODE == 2) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd, acc[c], 0, 0, 0);
            else if (MODE == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(av[c]), "v"(bd));
            else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc[c & 1]) : "v"(ad), "v"(bd));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NMF; ++c) s += (float)acc[c].x;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = c1 - c0;
}
template <int MODE> double run(float *out, long long *t)
{
    hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(64), 0, 0, out, t);
    hipLaunchKernelGGL((probe<MODE>), dim3(1), dim3(64), 0, 0, out, t);
    long long h = 0; (void)hipMemcpy(&h, t, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h / (IT * NMF);
}
int main()
{
    float *out; long long *t;
    (void)hipMalloc(&out, 256 * sizeof(float)); (void)hipMalloc(&t, sizeof(long long));
    printf("v_mfma_f64_16x16x4_f64, one wave, 8 accumulators in turn, shader clocks per MFMA:\n");
    printf("asm, AGPR accumulators, same A/B registers : %.1f\n", run<0>(out, t));
    printf("asm, VGPR accumulators                     : %.1f\n", run<1>(out, t));
    printf("builtin (compiler's choice)                : %.1f\n", run<2>(out, t));
    printf("asm, AGPR accumulators, A differs per MFMA : %.1f\n", run<3>(out, t));
    printf("asm, AGPR, two accumulators alternating    : %.1f\n", run<4>(out, t));
    return 0;
}
