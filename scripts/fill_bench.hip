// fill_bench.hip -- the sustained HBM WRITE rate of MI355X with the footprint and store shapes of kbuild / kqp: a kernel
// that does nothing but store a constant, 512 MiB per launch (8192 x 16384 floats), 20 launches back to back.
// Answers VERDICT r2 #7: is ~5.3 TB/s (66 % of the 8 TB/s spec) what the operand kernels CAN reach, or do they leave
// bandwidth on the table?  Variants: 16-byte stores with a wave covering two 512-byte row pieces (the kernels' shape) or
// one 1-KiB row segment; plain or non-temporal; and a grid-stride fill with perfectly linear addresses.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/fill_bench.hip -o scripts/fill_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e__), __LINE__); exit(1);} } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

// tile kernel: block = 256 threads covers 128 rows x (WIDE ? 256 : 128) columns, as kqp_kernel does
template <bool WIDE, bool NT>
__global__ __launch_bounds__(256) void tile_fill(float *out, int ld, float val)
{
    constexpr int LX = WIDE ? 64 : 32, RPP = 256 / LX, PASSES = 128 / RPP;
    const int tx = threadIdx.x & (LX - 1), ty = threadIdx.x / LX;
    const size_t col = (size_t)blockIdx.x * (4 * LX) + tx * 4;
    const size_t row0 = (size_t)blockIdx.y * 128;
    const f32x4 v = {val, val + 1, val + 2, val + 3};
#pragma unroll 4
    for (int r = 0; r < PASSES; ++r) {
        f32x4 *p = reinterpret_cast<f32x4 *>(out + (row0 + ty + RPP * r) * ld + col);
        if constexpr (NT)
            __builtin_nontemporal_store(v, p);
        else
            *p = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void linear_fill(f32x4 *out, size_t n4, float val)
{
    const f32x4 v = {val, val + 1, val + 2, val + 3};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        if constexpr (NT)
            __builtin_nontemporal_store(v, out + i);
        else
            out[i] = v;
    }
}

int main()
{
    const int rows = 8192, ld = 16384;
    const size_t bytes = (size_t)rows * ld * 4;
    float *buf;
    CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i)
            launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 20;
        printf("%-58s %7.1f us per 512 MiB  %.2f TB/s  (%.0f %% of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0 * 100);
    };
    timed("tile 128 x 128, 2 x 512 B per wave store", [&] { hipLaunchKernelGGL((tile_fill<false, false>), dim3(ld / 128, rows / 128), dim3(256), 0, 0, buf, ld, 1.0f); });
    timed("tile 128 x 128, 2 x 512 B per wave store, non-temporal", [&] { hipLaunchKernelGGL((tile_fill<false, true>), dim3(ld / 128, rows / 128), dim3(256), 0, 0, buf, ld, 1.0f); });
    timed("tile 128 x 256, 1 KiB per wave store", [&] { hipLaunchKernelGGL((tile_fill<true, false>), dim3(ld / 256, rows / 128), dim3(256), 0, 0, buf, ld, 1.0f); });
    timed("tile 128 x 256, 1 KiB per wave store, non-temporal", [&] { hipLaunchKernelGGL((tile_fill<true, true>), dim3(ld / 256, rows / 128), dim3(256), 0, 0, buf, ld, 1.0f); });
    timed("linear grid-stride fill, 2048 blocks", [&] { hipLaunchKernelGGL((linear_fill<false>), dim3(2048), dim3(256), 0, 0, (f32x4 *)buf, bytes / 16, 1.0f); });
    timed("linear grid-stride fill, 2048 blocks, non-temporal", [&] { hipLaunchKernelGGL((linear_fill<true>), dim3(2048), dim3(256), 0, 0, (f32x4 *)buf, bytes / 16, 1.0f); });
    timed("hipMemsetAsync", [&] { (void)hipMemsetAsync(buf, 0, bytes, 0); });
    return 0;
}
