// mfma_f64_chain_probe.hip -- development probe: v_mfma_f64_16x16x4_f64 (and v_mfma_f32_16x16x4_f32) with NCH independent
// accumulation chains issued round-robin, one wave: shader clocks per MFMA.  Tells how many independent chains the fp64
// add-back of gpx_varcols_kernel.hpp needs to run at the issue rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_f64_chain_probe.hip -o scripts/mfma_f64_chain_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4v __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int IT = 256;

template <int NCH, bool F64>
__global__ __launch_bounds__(64) void probe(double *out, long long *t)
{
    const int lane = threadIdx.x;
    double a = 1e-3 * lane, b = 1e-3;
    float af = 1e-3f * lane, bf = 1e-3f;
    d4v d[8];
    f4v f[8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
        d[c] = d4v{0, 0, 0, 0}, f[c] = f4v{0, 0, 0, 0};
    const long long c0 = clock64();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int u = 0; u < 8 / NCH; ++u)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (F64)
                    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(d[c]) : "v"(a), "v"(b));
                else
                    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(f[c]) : "v"(af), "v"(bf));
            }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c)
        s += d[c][0] + f[c][0];
    out[lane] = s;
    if (lane == 0)
        t[0] = c1 - c0;
}

template <int NCH, bool F64>
void run(double *out, long long *t)
{
    hipLaunchKernelGGL((probe<NCH, F64>), dim3(1), dim3(64), 0, 0, out, t);
    hipLaunchKernelGGL((probe<NCH, F64>), dim3(1), dim3(64), 0, 0, out, t);
    long long h = 0;
    hipMemcpy(&h, t, sizeof(h), hipMemcpyDeviceToHost);
    printf("%s 16x16x4, %d independent chain(s): %.1f shader clocks per MFMA\n", F64 ? "f64" : "f32", NCH, (double)h / (IT * 8));
}

int main()
{
    double *out;
    long long *t;
    hipMalloc(&out, 64 * sizeof(double));
    hipMalloc(&t, sizeof(long long));
    run<1, true>(out, t), run<2, true>(out, t), run<4, true>(out, t), run<8, true>(out, t);
    run<1, false>(out, t), run<2, false>(out, t), run<4, false>(out, t), run<8, false>(out, t);
    return 0;
}
