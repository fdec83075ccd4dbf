// mfma_acc_probe.hip -- how accurately do the fp16 and fp32 MFMA instructions accumulate a long dot product?
// One wave computes C(32x32) = A(32xK) * B(Kx32) with (a) v_mfma_f32_32x32x16_f16 and (b) v_mfma_f32_32x32x2_f32 on
// the SAME fp16-representable inputs, so that every product is exact in fp32 and only the accumulation differs.
// Error is reported against an fp64 host sum, relative to sum |a||b|.
//   hipcc --offload-arch=gfx950 -O2 scripts/mfma_acc_probe.hip -o scripts/mfma_acc_probe.bin
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

__global__ void probe_f16(const _Float16 *A, const _Float16 *B, int K, float *C)
{
    const int l = threadIdx.x, r = l & 31, kg = l >> 5;
    float16v acc = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = A[(size_t)r * K + k0 + kg * 8 + i];
            b[i] = B[(size_t)r * K + k0 + kg * 8 + i];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j)
        C[((j / 4) * 8 + kg * 4 + (j % 4)) * 32 + r] = acc[j];
}

__global__ void probe_f32(const _Float16 *A, const _Float16 *B, int K, float *C)
{
    const int l = threadIdx.x, r = l & 31, kg = l >> 5;
    float16v acc = {0};
    for (int k0 = 0; k0 < K; k0 += 2) {
        float a = (float)A[(size_t)r * K + k0 + kg];
        float b = (float)B[(size_t)r * K + k0 + kg];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    for (int j = 0; j < 16; ++j)
        C[((j / 4) * 8 + kg * 4 + (j % 4)) * 32 + r] = acc[j];
}

// fp16 MFMA with the accumulator flushed into a separate fp32 sum every `flush` k-steps
__global__ void probe_f16_flush(const _Float16 *A, const _Float16 *B, int K, int flush, float *C)
{
    const int l = threadIdx.x, r = l & 31, kg = l >> 5;
    float16v acc = {0}, tot = {0};
    int cnt = 0;
    for (int k0 = 0; k0 < K; k0 += 16) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) {
            a[i] = A[(size_t)r * K + k0 + kg * 8 + i];
            b[i] = B[(size_t)r * K + k0 + kg * 8 + i];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (++cnt == flush) {
            tot += acc;
            acc = float16v{0};
            cnt = 0;
        }
    }
    tot += acc;
    for (int j = 0; j < 16; ++j)
        C[((j / 4) * 8 + kg * 4 + (j % 4)) * 32 + r] = tot[j];
}

int main()
{
    const int K = 16384;
    for (int mode = 0; mode < 3; ++mode) {  // 0: positive values, 1: mixed signs, 2: decaying magnitudes, mixed signs
        std::vector<_Float16> hA(32 * (size_t)K), hB(32 * (size_t)K);
        srand(1234 + mode);
        auto rnd = [&](size_t k) {
            double u = (rand() + 0.5) / ((double)RAND_MAX + 1.0);
            double v = mode == 0 ? 0.5 + 0.5 * u : 2.0 * u - 1.0;
            if (mode == 2)
                v *= std::exp(-6.0 * (double)(K - 1 - k) / K);
            return (_Float16)v;
        };
        for (int r = 0; r < 32; ++r)
            for (size_t k = 0; k < (size_t)K; ++k) {
                hA[r * (size_t)K + k] = rnd(k);
                hB[r * (size_t)K + k] = rnd(k);
            }
        _Float16 *dA, *dB;
        float *dC;
        hipMalloc(&dA, hA.size() * 2), hipMalloc(&dB, hB.size() * 2), hipMalloc(&dC, 4096);
        hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        std::vector<double> ref(1024), mag(1024);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double s = 0, m = 0;
                for (int k = 0; k < K; ++k) {
                    double p = (double)hA[i * (size_t)K + k] * (double)hB[j * (size_t)K + k];
                    s += p, m += std::fabs(p);
                }
                ref[i * 32 + j] = s, mag[i * 32 + j] = m;
            }
        auto report = [&](const char *name) {
            std::vector<float> hC(1024);
            hipMemcpy(hC.data(), dC, 4096, hipMemcpyDeviceToHost);
            double worst = 0, rms = 0;
            for (int i = 0; i < 1024; ++i) {
                double e = std::fabs(hC[i] - ref[i]) / mag[i];
                worst = std::fmax(worst, e), rms += e * e;
            }
            printf("mode %d  %-22s max err / sum|ab| = %.3e   rms = %.3e   (2^-24 = 5.96e-08)\n", mode, name, worst,
                   std::sqrt(rms / 1024));
        };
        hipLaunchKernelGGL(probe_f16, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
        report("mfma 32x32x16 f16");
        hipLaunchKernelGGL(probe_f32, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
        report("mfma 32x32x2 f32");
        for (int fl : {2, 8, 32}) {
            hipLaunchKernelGGL(probe_f16_flush, dim3(1), dim3(64), 0, 0, dA, dB, K, fl, dC);
            char nm[64];
            snprintf(nm, sizeof nm, "f16 flush every %d", fl);
            report(nm);
        }
        hipFree(dA), hipFree(dB), hipFree(dC);
    }
    return 0;
}
