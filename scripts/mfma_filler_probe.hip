// mfma_filler_probe.hip -- development probe: how much other work one wave can issue behind each v_mfma_f32_16x16x4_f32
// (32 cycles in the matrix pipe) before the pipe runs dry.  Loop body = NMF x [1 MFMA + K fillers of one kind], independent
// accumulators; prints shader clocks per MFMA for K = 0 .. 10.  One wave (and, second column, 4 waves on the 4 SIMDs of a CU).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/mfma_filler_probe.hip -o scripts/mfma_filler_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));
constexpr int IT = 128, NMF = 8;

enum { F_FMA32 = 0, F_PKFMA32, F_EXP32, F_FMA64, F_CVT64, F_ACCREAD, F_DSREAD, F_SALU, F_FMA32_IND, F_PKFMA32_IND, F_FMA64_IND };
constexpr int KMAX = 12;

// the *_IND kinds: filler k of a group works on ITS OWN register, so the K fillers behind one MFMA are independent of each
// other (a register is touched again one MFMA later, >= 32 cycles: no dependency stall) -- the column the round-4 table lacked
template <int KIND, int K>
__device__ __forceinline__ void fillers(float &x, f2v &p, double &dd, f4v &acc0, unsigned lds_addr, int &sc, float (&xs)[KMAX],
                                        f2v (&ps)[KMAX], double (&ds)[KMAX])
{
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (KIND == F_FMA32_IND)
            asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(xs[k]));
        else if (KIND == F_PKFMA32_IND)
            asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(ps[k]));
        else if (KIND == F_FMA64_IND)
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(ds[k]));
        else if (KIND == F_FMA32)
            asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
        else if (KIND == F_PKFMA32)
            asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
        else if (KIND == F_EXP32)
            asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else if (KIND == F_FMA64)
            asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(dd));
        else if (KIND == F_CVT64)
            asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd) : "v"(x));
        else if (KIND == F_ACCREAD)
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(acc0.x));
        else if (KIND == F_DSREAD)
            asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"(lds_addr));
        else
            asm volatile("s_add_i32 %0, %0, 1" : "+s"(sc));
    }
    if (KIND == F_DSREAD && K > 0)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int KIND, int K>
__global__ __launch_bounds__(256) void probe(float *out, long long *t)
{
    __shared__ float lds[64];
    const int lane = threadIdx.x & 63;
    lds[lane] = 1.0f;
    float a = 1e-3f * lane, b = 1e-3f, x = 0.5f;
    f2v p = {0.5f, 0.25f};
    double dd = 0.5;
    int sc = 0;
    float xs[KMAX];
    f2v ps[KMAX];
    double ds[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        xs[k] = 0.5f + 1e-3f * k, ps[k] = f2v{0.5f, 0.25f + 1e-3f * k}, ds[k] = 0.5 + 1e-3 * k;
    const unsigned lds_addr = (unsigned)(size_t)(lds + lane);
    f4v acc[NMF];
#pragma unroll
    for (int c = 0; c < NMF; ++c)
        acc[c] = f4v{0, 0, 0, 0};
    const long long c0 = clock64();
    for (int it = 0; it < IT; ++it) {
#pragma unroll
        for (int c = 0; c < NMF; ++c) {
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[c]) : "v"(a), "v"(b));
            fillers<KIND, K>(x, p, dd, acc[(c + 4) % NMF], lds_addr, sc, xs, ps, ds);
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long c1 = clock64();
    float s = x + p.x + (float)dd + (float)sc;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        s += xs[k] + ps[k].x + (float)ds[k];
#pragma unroll
    for (int c = 0; c < NMF; ++c)
        s += acc[c].x;
    out[threadIdx.x] = s;
    if (threadIdx.x == 0)
        t[0] = c1 - c0;
}

template <int KIND, int K>
double run1(float *out, long long *t, int threads)
{
    hipLaunchKernelGGL((probe<KIND, K>), dim3(1), dim3(threads), 0, 0, out, t);
    hipLaunchKernelGGL((probe<KIND, K>), dim3(1), dim3(threads), 0, 0, out, t);
    long long h = 0;
    (void)hipMemcpy(&h, t, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h / (IT * NMF);
}

template <int KIND>
void run(const char *name, float *out, long long *t)
{
    printf("%-22s 1 wave :", name);
    printf(" %5.1f", run1<KIND, 0>(out, t, 64)), printf(" %5.1f", run1<KIND, 1>(out, t, 64)), printf(" %5.1f", run1<KIND, 2>(out, t, 64));
    printf(" %5.1f", run1<KIND, 3>(out, t, 64)), printf(" %5.1f", run1<KIND, 4>(out, t, 64)), printf(" %5.1f", run1<KIND, 5>(out, t, 64));
    printf(" %5.1f", run1<KIND, 6>(out, t, 64)), printf(" %5.1f", run1<KIND, 7>(out, t, 64)), printf(" %5.1f", run1<KIND, 8>(out, t, 64));
    printf(" %5.1f", run1<KIND, 10>(out, t, 64)), printf(" %5.1f\n", run1<KIND, 12>(out, t, 64));
    printf("%-22s 4 waves:", "");
    printf(" %5.1f", run1<KIND, 0>(out, t, 256)), printf(" %5.1f", run1<KIND, 1>(out, t, 256)), printf(" %5.1f", run1<KIND, 2>(out, t, 256));
    printf(" %5.1f", run1<KIND, 3>(out, t, 256)), printf(" %5.1f", run1<KIND, 4>(out, t, 256)), printf(" %5.1f", run1<KIND, 5>(out, t, 256));
    printf(" %5.1f", run1<KIND, 6>(out, t, 256)), printf(" %5.1f", run1<KIND, 7>(out, t, 256)), printf(" %5.1f", run1<KIND, 8>(out, t, 256));
    printf(" %5.1f", run1<KIND, 10>(out, t, 256)), printf(" %5.1f\n", run1<KIND, 12>(out, t, 256));
}

int main()
{
    float *out;
    long long *t;
    (void)hipMalloc(&out, 256 * sizeof(float));
    (void)hipMalloc(&t, sizeof(long long));
    printf("shader clocks per MFMA with K fillers behind each MFMA, K = 0 1 2 3 4 5 6 7 8 10 12\n");
    run<F_FMA32>("v_fma_f32", out, t);
    run<F_PKFMA32>("v_pk_fma_f32", out, t);
    run<F_EXP32>("v_exp_f32", out, t);
    run<F_FMA64>("v_fma_f64", out, t);
    run<F_CVT64>("v_cvt_f64_f32", out, t);
    run<F_ACCREAD>("v_accvgpr_read_b32", out, t);
    run<F_DSREAD>("ds_read_b32 + wait", out, t);
    run<F_SALU>("s_add_i32", out, t);
    printf("independent fillers (filler k of every group on its own register)\n");
    run<F_FMA32_IND>("v_fma_f32 indep.", out, t);
    run<F_PKFMA32_IND>("v_pk_fma_f32 indep.", out, t);
    run<F_FMA64_IND>("v_fma_f64 indep.", out, t);
    return 0;
}
