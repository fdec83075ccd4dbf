// gemm_bench.hip -- development harness: times the GEMM core on the two hot shapes (variance contraction with / without
// the low-rank correction in its epilogue; trailing update of the LDL^T for K = 128 .. 2048).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gaussian-object-modelling_amd/csrc scripts/gemm_bench.hip \
//          gaussian-object-modelling_amd/csrc/gpx_gemm.hip gaussian-object-modelling_amd/csrc/gpx_vargemm.hip \
//          gaussian-object-modelling_amd/csrc/gpx_host.cpp -o scripts/gemm_bench.bin       [-DGEMM_WAVES_PER_EU=1|2]
//   (compiled TOGETHER with the GEMM source, not linked to libgpx.so: struct GemmArgs is internal to the library and a
//   harness built against another revision of it passes garbage pointers -- the round-1 faults)
//   run  : scripts/gemm_bench.bin N NQ prec(0 = f32, 1 = f64) [with_correction = 1] [inverse-assembly shapes = 0] [variance only = 0]
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gpx_internal.hpp"
using namespace gpx;
#define CK(x) do { hipError_t err__ = (x); if (err__ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(err__), __FILE__, __LINE__); exit(1);} } while (0)

template <typename T>
__global__ void fill_kernel(T *d, size_t n, unsigned seed, double scale)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned s = (unsigned)(i * 2654435761u) ^ seed;
        s = s * 1664525u + 1013904223u;
        s ^= s >> 15;
        s = s * 1664525u + 1013904223u;
        d[i] = (T)(scale * ((double)(s >> 8) / (1 << 24) - 0.5));
    }
}
template <typename T>
static void fill(T *d, size_t n, unsigned seed, double scale)
{
    hipLaunchKernelGGL(fill_kernel<T>, dim3(4096), dim3(256), 0, 0, d, n, seed, scale);
    CK(hipDeviceSynchronize());
}

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 16384;  // matrix order
    const int NQ = argc > 2 ? atoi(argv[2]) : 8192;  // queries per launch
    const int prec = argc > 3 ? atoi(argv[3]) : 0;
    const size_t e = prec ? 8 : 4;
    // GEMM_BENCH_LDPAD=p: leading dimension N + p of X and Kqp in the variance shape (is the power-of-two row stride a cache problem?)
    const int ldpad = getenv("GEMM_BENCH_LDPAD") ? atoi(getenv("GEMM_BENCH_LDPAD")) : 0;
    const long LDV = N + ldpad;
    const long LDA = getenv("GEMM_BENCH_NOPAD_A") ? N : LDV, LDB = getenv("GEMM_BENCH_NOPAD_B") ? N : LDV;
    void *X, *Kqp, *dinv, *partial, *C, *W, *rowcorr, *colcoef;
    const bool with_corr = argc > 4 ? atoi(argv[4]) != 0 : true;  // the low-rank correction in the COLSQ epilogue
    CK(hipMalloc(&X, e * (size_t)N * (N + ldpad)));
    CK(hipMalloc(&Kqp, e * (size_t)NQ * (N + ldpad)));
    CK(hipMalloc(&dinv, e * N));
    CK(hipMalloc(&partial, 8 * (size_t)NQ * (N / 128)));  // doubles with the fp64 epilogue (low-rank correction)
    CK(hipMalloc(&C, e * (size_t)N * (N + ldpad)));
    CK(hipMalloc(&W, e * (size_t)N * 2048));
    // the low-rank correction is fp64 data whatever the product's type (round 3: fp64 epilogue)
    void *dinv64;
    CK(hipMalloc(&rowcorr, 8 * (size_t)N * VAR_NCORR));
    CK(hipMalloc(&colcoef, 8 * (size_t)NQ * VAR_NCORR));
    CK(hipMalloc(&dinv64, 8 * (size_t)N));
    fill((double *)rowcorr, (size_t)N * VAR_NCORR, 5, 1e-2);
    fill((double *)colcoef, (size_t)NQ * VAR_NCORR, 6, 1.0);
    fill((double *)dinv64, N, 7, 1.0);
    if (prec) { fill((double *)X, (size_t)N * (N + ldpad), 1, 1e-2); fill((double *)Kqp, (size_t)NQ * (N + ldpad), 2, 1.0); fill((double *)dinv, N, 3, 1.0); fill((double*)W, (size_t)N*2048, 4, 1e-2); }
    else { fill((float *)X, (size_t)N * (N + ldpad), 1, 1e-2); fill((float *)Kqp, (size_t)NQ * (N + ldpad), 2, 1.0); fill((float *)dinv, N, 3, 1.0); fill((float*)W, (size_t)N*2048, 4, 1e-2); }
    CK(hipMemset(C, 0, e * (size_t)N * (N + ldpad)));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#ifdef GEMM_BENCH_M32
    for (int cfg : {0, 2, 3, 4, 5}) {
#else
    for (int cfg : {0, 2, 3, 6}) {
        if (cfg == 2 && prec)
            continue;
#endif
        GemmArgs a;
        a.A = X, a.lda = LDA; a.B = Kqp, a.ldb = LDB; a.M = N, a.N = NQ, a.K = N; a.a_lower = 1; a.epi = EPI_COLSQ;
        a.rowweight = dinv; a.partial = partial, a.ldp = NQ; a.cfg = cfg;
        if (with_corr && !prec) { a.rowcorr = (const double *)rowcorr, a.ldrc = N; a.colcoef = (const double *)colcoef, a.ldcc = NQ; a.rowweight64 = (const double *)dinv64; }
        launch_gemm(prec, a, st);
        CK(hipStreamSynchronize(st));
        const int reps = 3;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch_gemm(prec, a, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        double flop = (double)N * N * NQ;
        // checksum of the partial sums of the first row tile (epilogue variants of one product must agree)
        double chk = 0;
        {
            const bool p64 = (with_corr && !prec) || prec;
            std::vector<char> h((size_t)NQ * 8);
            CK(hipMemcpy(h.data(), partial, (size_t)NQ * (p64 ? 8 : 4), hipMemcpyDeviceToHost));
            for (int q = 0; q < NQ; ++q)
                chk += p64 ? ((double *)h.data())[q] : (double)((float *)h.data())[q];
        }
        printf("VAR  cfg%d prec%d N=%d NQ=%d : %.3f ms  %.1f TFLOP/s (algorithmic N^2 per query)  checksum %.10e\n", cfg, prec, N, NQ, ms, flop / ms / 1e9, chk);
    }
    if (argc > 6 && atoi(argv[6]) != 0)
        return 0;  // variance shapes only
    for (int KK : {128, 256, 512, 1024, 2048})
    for (int cfg = 0; cfg < 1; cfg += 2) {
        const int M = N - 256;
        GemmArgs s;
        s.A = W, s.lda = 2048; s.B = X, s.ldb = LDV; s.C = C, s.ldc = LDV; s.M = M, s.N = M, s.K = KK; s.alpha = -1, s.beta = 1; s.lower_only = 1; s.cfg = cfg;
        launch_gemm(prec, s, st);
        CK(hipStreamSynchronize(st));
        const int reps = 5;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch_gemm(prec, s, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        double flop = (double)M * M * KK;  // lower triangle only: M^2/2 * K * 2
        const double tiles = (M / 128) * (M / 128 + 1) / 2.0;
        printf("SYRK cfg%d prec%d M=%d K=%d : %.3f ms  %.1f TFLOP/s   %.2f us per tile-slot (512 slots)\n", cfg, prec, M, KK, ms, flop / ms / 1e9, ms * 1e3 / (tiles / 512.0));
    }
    // the two products of the top level of the inverse-factor assembly (h = N / 2), as launched (NN) and as NT products
    // of the same size, for the "what would a transposed copy buy" question
    if (argc > 5 && atoi(argv[5]) != 0) {
        const int h = N / 2;
        struct { const char *what; int nn, a_lower, b_lower; } cases[] = {
            {"T = L21 X11   (NN, B lower)        ", 1, 0, 1}, {"X21 = -X22 T  (NN, A lower)        ", 1, 1, 0},
            {"same size     (NT, A lower)        ", 0, 1, 0}, {"same size     (NT, B lower [n][k]) ", 0, 0, 1},
            {"same size     (NT, full)           ", 0, 0, 0}, {"same size     (NN, full)           ", 1, 0, 0}};
        for (auto &c : cases) {
            GemmArgs g;
            g.A = X, g.lda = LDV; g.B = Kqp, g.ldb = LDV; g.C = C, g.ldc = LDV; g.M = h, g.N = h, g.K = h;
            g.nn = c.nn, g.a_lower = c.a_lower, g.b_lower = c.b_lower;
            launch_gemm(prec, g, st);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < 3; ++r) launch_gemm(prec, g, st);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
            const double flop = 2.0 * h * (double)h * h * ((c.a_lower || c.b_lower) ? 0.5 : 1.0);
            printf("INV  prec%d h=%d %s: %.3f ms  %.1f TFLOP/s\n", prec, h, c.what, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
