// stream_hop_bench.hip -- cost of a cross-stream dependency (event record on one stream, wait on the other) between
// two ordinary HIP streams, and of the same chain on one stream.
//   hipcc --offload-arch=gfx950 -O3 scripts/stream_hop_bench.hip -o scripts/stream_hop_bench.bin
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(int *p) { if (threadIdx.x == 0) atomicAdd(p, 1); }
int main(int argc, char **)
{
    int *d;
    hipMalloc(&d, 4), hipMemset(d, 0, 4);
    hipStream_t a, b;
    if (argc > 1) {  // CU-masked variant: a on all but the first 8 CUs of each 32, b on those 8
        const unsigned ma[8] = {0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u, 0xffffff00u};
        const unsigned mb[8] = {0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu};
        printf("CU-masked streams: %d %d\n", (int)hipExtStreamCreateWithCUMask(&a, 8, ma), (int)hipExtStreamCreateWithCUMask(&b, 8, mb));
    } else {
        hipStreamCreateWithFlags(&a, hipStreamNonBlocking), hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
    }
    const int H = 200;
    hipEvent_t ev[2 * H];
    for (auto &e : ev)
        hipEventCreateWithFlags(&e, hipEventDisableTiming);
    for (int rep = 0; rep < 3; ++rep) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < H; ++i) {
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
            hipEventRecord(ev[2 * i], a);
            hipStreamWaitEvent(b, ev[2 * i], 0);
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, b, d);
            hipEventRecord(ev[2 * i + 1], b);
            hipStreamWaitEvent(a, ev[2 * i + 1], 0);
        }
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < 2 * H; ++i)
            hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, a, d);
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("rep %d: %d kernels ping-pong over two streams: %.1f us per kernel; on one stream: %.1f us per kernel\n", rep, 2 * H,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (2 * H),
               std::chrono::duration<double, std::micro>(t2 - t1).count() / (2 * H));
    }
    return 0;
}
