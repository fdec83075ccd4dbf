// gpx_cov.hpp -- device covariance functions on the UN-squared distance (reference semantics).
//   kernels/gaussian.hpp:15-27, laplace.hpp:37-49, thin_plate.hpp:12-20 of the reference;
//   Matern closed forms from matlab_src/test_gp_regression_3Dsurf.m:117-123.
#pragma once
#include "gpx_internal.hpp"

namespace gpx {

template <typename T>
__device__ __forceinline__ T dev_sqrt(T x);
template <>
__device__ __forceinline__ float dev_sqrt<float>(float x)
{
    // v_sqrt_f32 (1 ulp): the fp32 kernels are HBM-write-bound only if the math stays this cheap;
    // the IEEE sqrtf/expf expansions made kbuild VALU-bound (41 % of HBM peak)
    return __builtin_amdgcn_sqrtf(x);
}
template <>
__device__ __forceinline__ double dev_sqrt<double>(double x)
{
    return sqrt(x);
}
template <typename T>
__device__ __forceinline__ T dev_exp(T x);
template <>
__device__ __forceinline__ float dev_exp<float>(float x)
{
    // v_exp_f32 on x * log2(e); arguments here are -s*d in [-40, 0]: relative error <= |x| * 6e-8
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
template <>
__device__ __forceinline__ double dev_exp<double>(double x)
{
    return exp(x);
}

// ---- fp64 sqrt / exp for the N-term prediction sums (gpx_predict.hip) -------------------------------------------------
// The mean of 2^20 queries against 16384 points is 1.7e10 kernel evaluations in fp64 and VALU-issue bound; the
// library's sqrt / exp spend most of their ~45 instructions on ranges this path never sees.  These take
// x in [0, 1e300) / x <= 0 and stay within ~2 ulp:
//   sqrt: v_rsq_f64 seed (>= 26 bits) + one coupled Newton step on (g, h) = (x y, y / 2):  error 1.5 eps_seed^2
//   exp : 2^n e^r, n = rint(x log2 e), r = x - n ln 2, Taylor degree 12 on |r| <= 0.347 (1.7e-16), exponent-field add
struct MathAcc {  // the compiler library's functions (kernel matrix, Kqp: pinned to the reference classes at 4e-15)
    template <typename T>
    static __device__ __forceinline__ T sqrt_(T x) { return dev_sqrt<T>(x); }
    template <typename T>
    static __device__ __forceinline__ T exp_(T x) { return dev_exp<T>(x); }
};
struct MathFast {
    static __device__ __forceinline__ float sqrt_(float x) { return dev_sqrt<float>(x); }
    static __device__ __forceinline__ float exp_(float x) { return dev_exp<float>(x); }
    static __device__ __forceinline__ double sqrt_(double x)  // x > 0 (callers add 1e-300 to the squared distance)
    {
        const double y = __builtin_amdgcn_rsq(x);
        const double g = x * y, h = 0.5 * y;
        const double r = fma(-h, g, 0.5);
        return fma(g, r, g);
    }
    static __device__ __forceinline__ double exp_(double x)
    {
        // n = rint(x log2 e) by the 1.5 * 2^52 trick: the integer sits in the low word of the biased sum, so neither
        // v_rndne_f64 nor v_cvt_i32_f64 (both slower than an FMA) is needed; x is clamped where e^x is ~1e-304
        constexpr double MAGIC = 6755399441055744.0;
        x = fmax(x, -700.0);
        const double nb = fma(x, 1.4426950408889634, MAGIC);
        const int ni = __double2loint(nb);
        const double n = nb - MAGIC;
        const double r = fma(n, -0.6931471805599453, x);  // |n| <= 1010: n (ln 2 - fl(ln 2)) <= 3e-14 at the far end, 4e-16 for |x| < 12
        double p = 2.08767569878680989792e-09;  // 1/12!
        p = fma(p, r, 2.50521083854417187751e-08);
        p = fma(p, r, 2.75573192239858906526e-07);
        p = fma(p, r, 2.75573192239858906526e-06);
        p = fma(p, r, 2.48015873015873015873e-05);
        p = fma(p, r, 1.98412698412698412698e-04);
        p = fma(p, r, 1.38888888888888888889e-03);
        p = fma(p, r, 8.33333333333333333333e-03);
        p = fma(p, r, 4.16666666666666666667e-02);
        p = fma(p, r, 1.66666666666666666667e-01);
        p = fma(p, r, 0.5);
        p = fma(p, r, 1.0);
        p = fma(p, r, 1.0);
        // p in [0.70, 1.42]: scale by 2^n with one integer add on the exponent field (no underflow: n >= -1010)
        return __hiloint2double(__double2hiint(p) + (ni << 20), __double2loint(p));
    }
};

// The exponential kernels a e^(-s d) [x polynomial in s d] for the fp64 mean / gradient kernel (gpx_predict.hip), VALU-issue
// bound at ~30 instructions per (query, point) pair, 13 of them the exponential:
//   e^x from a 512-entry table of 2^(j/512) in LDS (4 KB): x = -s d = (512 m + j) ln2/512 + r, |r| <= 0.00068, so that a
//   degree-4 polynomial is exact to 1e-18 (was: Taylor degree 12);
//   everything that depends on s alone is prepared once per thread: the polynomial runs in u = d + n ln2/(512 s) with the
//   coefficients (-s)^k / k!, the Matern factors in d, so t = s d is never formed;
// The kernel that uses it calls ExpTab::init() with all its threads and passes a barrier before the first value.
struct ExpTab {
    static constexpr int NTAB = 512;
    static __device__ __forceinline__ double *tab()
    {
        __shared__ double t[NTAB];
        return t;
    }
    static __device__ __forceinline__ void init(int tid, int nthreads)
    {
        for (int j = tid; j < NTAB; j += nthreads)
            tab()[j] = exp2((double)j * (1.0 / NTAB));
    }
};
template <int KID>
struct ExpMean {
    double k1, lq, a1, a2, a3, a4, dmax, s, s2_3;
    __device__ __forceinline__ void prep(const Cov<double> &c)
    {
        s = c.s;
        k1 = -s * (ExpTab::NTAB * 1.4426950408889634);          // nb = MAGIC + rint(x 512 / ln 2)
        lq = 0.6931471805599453 / ExpTab::NTAB / s;             // u = d + n ln2 / (512 s),  r = -s u
        a1 = -s, a2 = 0.5 * s * s, a3 = -s * s * s * (1.0 / 6.0), a4 = s * s * s * s * (1.0 / 24.0);
        dmax = 700.0 / s;                                       // e^x is ~1e-304 there
        s2_3 = s * s * (1.0 / 3.0);
    }
    // e^(-s d), d >= 0
    __device__ __forceinline__ double e(double d) const
    {
        constexpr double MAGIC = 6755399441055744.0;            // 1.5 * 2^52: n sits in the low word of the sum
        d = fmin(d, dmax);
        const double nb = fma(d, k1, MAGIC);
        const int n = __double2loint(nb);
        const double u = fma(nb - MAGIC, lq, d);
        double p = fma(a4, u, a3);
        p = fma(p, u, a2);
        p = fma(p, u, a1);
        p = fma(p, u, 1.0);
        p *= ExpTab::tab()[n & (ExpTab::NTAB - 1)];
        // p in [0.999, 2.002): scale by 2^m, m = n >> 9 >= -1010, with one integer add on the exponent field
        return __hiloint2double(__double2hiint(p) + ((n >> 9) << 20), __double2loint(p));
    }
    // k(d) without the amplitude (the caller has folded it into the weights)
    __device__ __forceinline__ double k(double d) const
    {
        const double ev = e(d);
        if constexpr (KID == GPX_KERNEL_MATERN32)
            return ev * fma(d, s, 1.0);
        else if constexpr (KID == GPX_KERNEL_MATERN52)
            return ev * fma(d, fma(d, s2_3, s), 1.0);
        else
            return ev;
    }
    // k(d) and the reference's "computediff" (cov_k_diff below), both without the amplitude
    __device__ __forceinline__ void k_diff(double d, double &kv, double &kd) const
    {
        const double ev = e(d);
        if constexpr (KID == GPX_KERNEL_MATERN32) {
            kv = ev * fma(d, s, 1.0);
            kd = -(s * s) * ev;
        } else if constexpr (KID == GPX_KERNEL_MATERN52) {
            const double t1 = fma(d, s, 1.0);
            kv = ev * fma(d * d, s2_3, t1);
            kd = -s2_3 * t1 * ev;
        } else {
            kv = ev;
            kd = -s * ev;
        }
    }
};

// k(d) given the squared distance d2.  UNIT_A: without the amplitude c.a of the exponential kernels (the caller has
// folded it into the weights the values are multiplied with); thin-plate has none.
template <typename T, int KID, typename M = MathAcc, bool UNIT_A = false>
__device__ __forceinline__ T cov_k(const Cov<T> &c, T d2)
{
    if constexpr (KID == GPX_KERNEL_THINPLATE) {
        // 2d^3 - 3R d^2 + R^3 == (d - R)^2 (2d + R): the factored form has no cancellation between
        // O(R^3) terms, which matters in fp32 for d > R (k is small there, the monomials are not).
        T d = M::sqrt_(d2);
        T e = d - c.R;
        return e * e * (T(2) * d + c.R);
    } else {
        T d = M::sqrt_(d2);
        T t = c.s * d;
        T e = UNIT_A ? M::exp_(-t) : c.a * M::exp_(-t);
        if constexpr (KID == GPX_KERNEL_MATERN32)
            return e * (T(1) + t);
        else if constexpr (KID == GPX_KERNEL_MATERN52)
            return e * (T(1) + t + t * t * T(1.0 / 3.0));
        else
            return e;  // Gaussian: a = sigma^2, s = 1/l^2 ; Laplace: a = 2 sigma, s = 1/l
    }
}

// k(d) and the reference's "computediff" (multiplies (q - p) in the gradient).
template <typename T, int KID, typename M = MathAcc, bool UNIT_A = false>
__device__ __forceinline__ void cov_k_diff(const Cov<T> &c, T d2, T &k, T &kd)
{
    T d = M::sqrt_(d2);
    if constexpr (KID == GPX_KERNEL_THINPLATE) {
        T e = d - c.R;
        k = e * e * (T(2) * d + c.R);
        kd = T(6) * e;  // -6 (R - d)
    } else {
        T t = c.s * d;
        T e = UNIT_A ? M::exp_(-t) : c.a * M::exp_(-t);
        if constexpr (KID == GPX_KERNEL_MATERN32) {
            k = e * (T(1) + t);
            kd = -(c.s * c.s) * e;  // -3 sigma^2 / l^2 e^-t
        } else if constexpr (KID == GPX_KERNEL_MATERN52) {
            k = e * (T(1) + t + t * t * T(1.0 / 3.0));
            kd = -(c.s * c.s * T(1.0 / 3.0)) * (T(1) + t) * e;
        } else {
            k = e;
            kd = -c.s * e;  // -(1/l^2) k  |  -(1/l) k
        }
    }
}

// Dispatch a functor templated on <T, KID> from run-time (prec, id).
#define GPX_DISPATCH_KID(ID, ...)                                              \
    switch (ID) {                                                              \
    case GPX_KERNEL_GAUSSIAN:                                                  \
    case GPX_KERNEL_LAPLACE: {                                                 \
        constexpr int KID = GPX_KERNEL_GAUSSIAN;                               \
        __VA_ARGS__;                                                           \
    } break;                                                                   \
    case GPX_KERNEL_THINPLATE: {                                               \
        constexpr int KID = GPX_KERNEL_THINPLATE;                              \
        __VA_ARGS__;                                                           \
    } break;                                                                   \
    case GPX_KERNEL_MATERN32: {                                                \
        constexpr int KID = GPX_KERNEL_MATERN32;                               \
        __VA_ARGS__;                                                           \
    } break;                                                                   \
    default: {                                                                 \
        constexpr int KID = GPX_KERNEL_MATERN52;                               \
        __VA_ARGS__;                                                           \
    } break;                                                                   \
    }

}  // namespace gpx
