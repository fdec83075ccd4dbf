// diag_bench.hip -- development harness: phase timing (s_memtime) of the 128 x 128 diagonal-block kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/diag_bench.hip -I gaussian-object-modelling_amd/csrc -I include -o scripts/diag_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
__device__ long long gpx_dbg_stamps[32];
#define GPX_STAMP(i)                                            \
    do {                                                        \
        if (threadIdx.x == 0)                                   \
            gpx_dbg_stamps[(i)] = (long long)wall_clock64();    \
    } while (0)
#include "../gaussian-object-modelling_amd/csrc/gpx_factor.hip"
using namespace gpx;

// reads the block once, so that the timed launch finds it in L2 (as it does inside the factorisation, where the
// preceding update has just written it); GPX_BENCH_COLD=1 skips this
template <typename T>
__global__ void touch_kernel(const T *a, int n, T *sink)
{
    T s = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        s += a[i];
    if (s == T(12345.678))
        *sink = s;
}

template <typename T>
static void run(const char *name, int prec, bool narrow = false)
{
    const int n = 128;
    factor_init(prec);
    std::vector<T> A((size_t)n * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            A[(size_t)i * n + j] = (T)(std::exp(-0.05 * std::fabs((double)(i - j))) + (i == j ? 0.5 : 0.0));
    T *dA, *dL, *dd, *ddi;
    int *info;
    hipMalloc(&dA, sizeof(T) * n * n), hipMalloc(&dL, sizeof(T) * n * n), hipMalloc(&dd, sizeof(T) * n), hipMalloc(&ddi, sizeof(T) * n);
    hipMalloc(&info, 64), hipMemset(info, 0, 64);
    long long st[32];
    for (int rep = 0; rep < 3; ++rep) {
        {   // only the lower triangle of a diagonal tile is valid in the factorisation: poison the rest
            std::vector<T> P(A);
            for (int i = 0; i < n; ++i)
                for (int j = i + 1; j < n; ++j)
                    P[(size_t)i * n + j] = (T)NAN;
            hipMemcpy(dA, P.data(), sizeof(T) * n * n, hipMemcpyHostToDevice);
        }
        if (!getenv("GPX_BENCH_COLD"))
            hipLaunchKernelGGL(touch_kernel<T>, dim3(1), dim3(256), 0, 0, dA, n * n, dd);
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        launch_diag_ldl(prec, dA, n, dL, dd, ddi, info, 0, 0, narrow);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpyFromSymbol(st, HIP_SYMBOL(gpx_dbg_stamps), sizeof(st));
        if (rep == 2) {
            printf("%s: event %.1f us; phases in us (100 MHz wall clock):\n", name, ms * 1e3);
            auto us = [&](int a, int b) { return (st[b] - st[a]) / 100.0; };
            for (int jb = 0; jb < 4; ++jb)
                printf("  panel %d: load %.1f  A(LDL wave 0 | inv wave 1) %.1f  B(W, L21) %.1f  C(trailing) %.1f\n", jb,
                       us(jb ? 4 * jb : 0, 1 + 4 * jb), us(1 + 4 * jb, 2 + 4 * jb), jb < 3 ? us(2 + 4 * jb, 3 + 4 * jb) : 0.0,
                       jb < 3 ? us(3 + 4 * jb, 4 + 4 * jb) : us(2 + 4 * jb, 4 + 4 * jb));
            printf("  last sub-block inverse %.1f  inverse assembly: %.1f %.1f %.1f  final copy %.1f   total %.1f\n", us(16, 20),
                   us(20, 21), us(21, 22), us(22, 23), us(23, 24), us(0, 24));
            // correctness: L D L^T = A and linv L = I, in double on the host
            std::vector<T> F((size_t)n * n), X((size_t)n * n);
            hipMemcpy(F.data(), dA, sizeof(T) * n * n, hipMemcpyDeviceToHost);
            hipMemcpy(X.data(), dL, sizeof(T) * n * n, hipMemcpyDeviceToHost);
            double err1 = 0, err2 = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j <= i; ++j) {
                    double s = 0, t = 0;
                    for (int k = 0; k <= j; ++k) {
                        const double lik = k == i ? 1.0 : (double)F[(size_t)i * n + k];
                        const double ljk = k == j ? 1.0 : (double)F[(size_t)j * n + k];
                        s += lik * (double)F[(size_t)k * n + k] * ljk;
                    }
                    for (int k = j; k <= i; ++k)
                        t += (double)X[(size_t)i * n + k] * (k == j ? 1.0 : (double)F[(size_t)k * n + j]);
                    err1 = std::fmax(err1, std::fabs(s - (double)A[(size_t)i * n + j]));
                    err2 = std::fmax(err2, std::fabs(t - (i == j ? 1.0 : 0.0)));
                }
            printf("  max |L D L^T - A| = %.3e   max |linv L - I| = %.3e\n", err1, err2);
            // back-to-back launches on fresh copies of the block: average duration incl. dispatch
            const int NL = 256;
            T *many;
            hipMalloc(&many, sizeof(T) * n * n * NL);
            for (int c = 0; c < NL; ++c)
                hipMemcpy(many + (size_t)c * n * n, A.data(), sizeof(T) * n * n, hipMemcpyHostToDevice);
            hipEventRecord(e0, 0);
            for (int c = 0; c < NL; ++c)
                launch_diag_ldl(prec, many + (size_t)c * n * n, n, dL, dd, ddi, info, 0, 0, narrow);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
            printf("  %d back-to-back launches: %.1f us each\n", NL, ms * 1e3 / NL);
            hipFree(many);
        }
    }
}
int main()
{
    run<float>("fp32", 0);
    run<float>("fp32, 4-wave variant", 0, true);
    run<double>("fp64", 1);
    return 0;
}
