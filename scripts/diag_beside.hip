// diag_beside.hip -- development harness: what the 128 x 128 diagonal-block kernel costs when it shares the GPU with a
// trailing update of the LDL^T on another stream (the look-ahead schedule of gpx_build.hip), phase by phase.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaussian-object-modelling_amd/csrc -I include scripts/diag_beside.hip \
//          gaussian-object-modelling_amd/csrc/gpx_gemm.hip -o scripts/diag_beside.bin
//   run  : scripts/diag_beside.bin [M of the trailing update = 12288]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
__device__ long long gpx_dbg_stamps[32];
#define GPX_STAMP(i)                                            \
    do {                                                        \
        if (threadIdx.x == 0)                                   \
            gpx_dbg_stamps[(i)] = (long long)wall_clock64();    \
    } while (0)
#include "../gaussian-object-modelling_amd/csrc/gpx_factor.hip"
using namespace gpx;
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(err__), __FILE__, __LINE__); exit(1);} } while (0)

static void spin_us(double us)
{
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {
    }
}

int main(int argc, char **argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 12288;
    const int n = 128, LD = 16384;
    factor_init(0);
    std::vector<float> A((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            A[(size_t)i * n + j] = (float)(std::exp(-0.05 * std::fabs((double)(i - j))) + (i == j ? 0.5 : 0.0));
    float *dA, *dL, *dd, *ddi, *C, *W, *B;
    int *info;
    CK(hipMalloc(&dA, 4 * n * n)); CK(hipMalloc(&dL, 4 * n * n)); CK(hipMalloc(&dd, 4 * n)); CK(hipMalloc(&ddi, 4 * n));
    CK(hipMalloc(&info, 64)); CK(hipMemset(info, 0, 64));
    CK(hipMalloc(&C, 4 * (size_t)M * LD)); CK(hipMalloc(&W, 4 * (size_t)M * 512)); CK(hipMalloc(&B, 4 * (size_t)M * LD));
    CK(hipMemset(C, 0, 4 * (size_t)M * LD)); CK(hipMemset(W, 0, 4 * (size_t)M * 512)); CK(hipMemset(B, 0, 4 * (size_t)M * LD));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1, g0, g1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&g0)); CK(hipEventCreate(&g1));
    GemmArgs s;  // the `rest` update: C -= W L^T, lower tiles only, K = 256
    s.A = W, s.lda = 512, s.B = B, s.ldb = LD, s.C = C, s.ldc = LD, s.M = M, s.N = M, s.K = 256, s.alpha = -1, s.beta = 1, s.lower_only = 1;
    launch_gemm(0, s, sa);
    CK(hipDeviceSynchronize());
    auto report = [&](const char *what, float ev_us, float gemm_us) {
        long long st[32];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gpx_dbg_stamps), sizeof(st)));
        auto us = [&](int a, int b) { return (st[b] - st[a]) / 100.0; };
        double pa = 0, pb = 0, pc = 0;
        for (int jb = 0; jb < 4; ++jb) {
            pa += us(1 + 4 * jb, 2 + 4 * jb);
            if (jb < 3)
                pb += us(2 + 4 * jb, 3 + 4 * jb), pc += us(3 + 4 * jb, 4 + 4 * jb);
            else
                pc += us(2 + 4 * jb, 4 + 4 * jb);
        }
        printf("%-58s events %6.1f us | in-kernel %6.1f = A %5.1f  B %5.1f  C %5.1f  last inv %5.1f  assembly %5.1f | GEMM %6.1f us\n", what, ev_us,
               us(0, 24), pa, pb, pc, us(16, 20), us(20, 24), gemm_us);
    };
    for (int narrow = 0; narrow < 2; ++narrow) {
        const char *nm = narrow ? "4-wave" : "8-wave";
        char buf[128];
        float ms, gms;
        for (int rep = 0; rep < 2; ++rep) {  // alone
            CK(hipMemcpy(dA, A.data(), 4 * n * n, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, sb));
            launch_diag_ldl(0, dA, n, dL, dd, ddi, info, 0, sb, narrow);
            CK(hipEventRecord(e1, sb));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        snprintf(buf, sizeof buf, "%s alone", nm);
        report(buf, ms * 1e3f, 0.f);
        for (double delay : {0.0, 20.0, 60.0}) {  // GEMM first, the diagonal block `delay` us later
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipMemcpy(dA, A.data(), 4 * n * n, hipMemcpyHostToDevice));
                CK(hipEventRecord(g0, sa));
                launch_gemm(0, s, sa);
                CK(hipEventRecord(g1, sa));
                spin_us(delay);
                CK(hipEventRecord(e0, sb));
                launch_diag_ldl(0, dA, n, dL, dd, ddi, info, 0, sb, narrow);
                CK(hipEventRecord(e1, sb));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&gms, g0, g1));
            }
            snprintf(buf, sizeof buf, "%s launched %2.0f us (host) after the GEMM (M = %d)", nm, delay, M);
            report(buf, ms * 1e3f, gms * 1e3f);
        }
        for (int rep = 0; rep < 2; ++rep) {  // the diagonal block first
            CK(hipMemcpy(dA, A.data(), 4 * n * n, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, sb));
            launch_diag_ldl(0, dA, n, dL, dd, ddi, info, 0, sb, narrow);
            CK(hipEventRecord(e1, sb));
            CK(hipEventRecord(g0, sa));
            launch_gemm(0, s, sa);
            CK(hipEventRecord(g1, sa));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipEventElapsedTime(&gms, g0, g1));
        }
        snprintf(buf, sizeof buf, "%s launched just BEFORE the GEMM", nm);
        report(buf, ms * 1e3f, gms * 1e3f);
    }
    return 0;
}
