// mfma_denorm_probe.hip -- does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs, and how are tiny products
// added to a large accumulator?   hipcc --offload-arch=gfx950 -O2 scripts/mfma_denorm_probe.hip -o scripts/mfma_denorm_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void probe(float aval, float bval, float c0, float *out)
{
    half8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (_Float16)aval;
        b[i] = (_Float16)bval;
    }
    float16v acc;
    for (int i = 0; i < 16; ++i)
        acc[i] = c0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = (float)a[0];
    }
}

int main()
{
    float *d, h[2];
    hipMalloc(&d, 8);
    const float cases[][3] = {{1e-6f, 1.0f, 0.0f},   {3e-5f, 1.0f, 0.0f},  {1e-4f, 1.0f, 0.0f}, {1e-6f, 1e-6f, 0.0f},
                              {1e-3f, 1e-3f, 1.0f},  {1e-3f, 1e-3f, 64.0f}, {0.5f, 0.5f, 1.0f}, {3e-5f, 0.5f, 1.0f}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, c[0], c[1], c[2], d);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        const double ah = (double)(_Float16)c[0], bh = (double)(_Float16)c[1];
        printf("a=%g (fp16 %.9g) b=%g c0=%g : mfma = %.9g   exact = %.9g\n", c[0], h[1], c[1], c[2], h[0],
               c[2] + 16.0 * ah * bh);
    }
    return 0;
}
