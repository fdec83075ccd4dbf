// mfma_tree_probe.hip -- precision of the internal 16-product sum of v_mfma_f32_32x32x16_f16 (and the 2-product sum of
// the fp32 one): two large products that cancel exactly plus one small product; exact answer = the small product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void probe16(float big, float sa, float sb, float c0, int small_in_other_group, float *out)
{
    half8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool g0 = threadIdx.x < 32;  // k = 0..7 live in lanes 0..31, k = 8..15 in lanes 32..63
    if (g0) {
        a[0] = (_Float16)big, b[0] = (_Float16)1.0f;
        a[1] = (_Float16)big, b[1] = (_Float16)-1.0f;
    }
    if (small_in_other_group ? !g0 : g0)
        a[2] = (_Float16)sa, b[2] = (_Float16)sb;
    float16v acc;
    for (int i = 0; i < 16; ++i)
        acc[i] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0)
        out[0] = acc[0];
}
int main()
{
    float *d, h;
    hipMalloc(&d, 4);
    for (int other = 0; other < 2; ++other)
        for (float big : {1.0f, 0.125f})
            for (int e : {12, 16, 20, 22, 24, 26, 28}) {
                const int ea = e / 2, eb = e - ea;
                const float sa = ldexpf(1.0f, -ea), sb = 1.5f * ldexpf(1.0f, -eb);
                hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, big, sa, sb, 0.0f, other, d);
                hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                printf("big=+-%g, small product 1.5*2^-%d (%s k-group), c=0 : mfma = %.6g (exact %.6g)  ratio %.4f\n", big, e,
                       other ? "other" : "same", h, 1.5 * ldexp(1.0, -e), h / (1.5 * ldexp(1.0, -e)));
            }
    // and against a non-zero accumulator that cancels with a product
    for (int e : {16, 20, 24, 28}) {
        const int ea = e / 2, eb = e - ea;
        // products: +1 (a=1,b=1) and small; c0 = -1 -> exact = small
        hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, 0.0f, ldexpf(1.0f, -ea), 1.5f * ldexpf(1.0f, -eb), 0.0f, 0, d);
        hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("no big products, small 1.5*2^-%d : mfma = %.6g (exact %.6g)\n", e, h, 1.5 * ldexp(1.0, -e));
    }
    return 0;
}
