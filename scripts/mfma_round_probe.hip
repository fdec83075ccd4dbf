// mfma_round_probe.hip -- how does v_mfma_f32_32x32x16_f16 (and _32x32x2_f32) round acc + product?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void probe16(float a0, float b0, int nprod, float c0, float *out)
{
    half8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    // k index of lane group 0 (lanes 0..31) element i is i; put nprod equal products there
    if (threadIdx.x < 32)
        for (int i = 0; i < nprod && i < 8; ++i) {
            a[i] = (_Float16)a0;
            b[i] = (_Float16)b0;
        }
    float16v acc;
    for (int i = 0; i < 16; ++i)
        acc[i] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0)
        out[0] = acc[0];
}
__global__ void probe32(float a0, float b0, float c0, float *out)
{
    float a = threadIdx.x < 32 ? a0 : 0.0f, b = threadIdx.x < 32 ? b0 : 0.0f;
    float16v acc;
    for (int i = 0; i < 16; ++i)
        acc[i] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0)
        out[0] = acc[0];
}
int main()
{
    float *d, h;
    hipMalloc(&d, 4);
    const float ulp = ldexpf(1.0f, -23);  // ulp of 1.0
    const float fr[] = {0.25f, 0.5f, 0.75f, 0.96875f, 1.25f, 1.5f, 1.75f};
    for (float sgn : {1.0f, -1.0f})
        for (float c0 : {1.0f, -1.0f})
            for (float f : fr) {
                // product = sgn * f * ulp : a = 2^-12, b = sgn * f * 2^-11  (both exact in fp16)
                const float a0 = ldexpf(1.0f, -12), b0 = sgn * f * ldexpf(1.0f, -11);
                hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, a0, b0, 1, c0, d);
                hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                const float r16 = (h - c0) / ulp;
                hipLaunchKernelGGL(probe32, dim3(1), dim3(64), 0, 0, a0, b0, c0, d);
                hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                const float r32 = (h - c0) / ulp;
                const float rne = (float)(((double)(float)((double)c0 + (double)sgn * f * ulp)) - c0) / ulp;
                printf("c0=%+.0f  product=%+.5f ulp :  f16 mfma -> %+.2f ulp   f32 mfma -> %+.2f ulp   (RNE %+.2f)\n", c0, sgn * f, r16, r32, rne);
            }
    // several small products that only matter together: 8 products of 0.2 ulp each = 1.6 ulp
    for (float c0 : {1.0f, -1.0f}) {
        hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, ldexpf(1.0f, -12), 0.2001953125f * ldexpf(1.0f, -11), 8, c0, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("c0=%+.0f  8 products of +0.2 ulp : f16 mfma -> %+.2f ulp (exact sum 1.60)\n", c0, (h - c0) / ulp);
    }
    return 0;
}
