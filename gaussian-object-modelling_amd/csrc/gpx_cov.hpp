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

// k(d) given the squared distance d2.
template <typename T, int KID>
__device__ __forceinline__ T cov_k(const Cov<T> &c, T d2)
{
    if constexpr (KID == GPX_KERNEL_THINPLATE) {
        // 2d^3 - 3R d^2 + R^3 == (d - R)^2 (2d + R): the factored form has no cancellation between
        // O(R^3) terms, which matters in fp32 for d > R (k is small there, the monomials are not).
        T d = dev_sqrt<T>(d2);
        T e = d - c.R;
        return e * e * (T(2) * d + c.R);
    } else {
        T d = dev_sqrt<T>(d2);
        T t = c.s * d;
        T e = c.a * dev_exp<T>(-t);
        if constexpr (KID == GPX_KERNEL_MATERN32)
            return e * (T(1) + t);
        else if constexpr (KID == GPX_KERNEL_MATERN52)
            return e * (T(1) + t + t * t * T(1.0 / 3.0));
        else
            return e;  // Gaussian: a = sigma^2, s = 1/l^2 ; Laplace: a = 2 sigma, s = 1/l
    }
}

// k(d) and the reference's "computediff" (multiplies (q - p) in the gradient).
template <typename T, int KID>
__device__ __forceinline__ void cov_k_diff(const Cov<T> &c, T d2, T &k, T &kd)
{
    T d = dev_sqrt<T>(d2);
    if constexpr (KID == GPX_KERNEL_THINPLATE) {
        T e = d - c.R;
        k = e * e * (T(2) * d + c.R);
        kd = T(6) * e;  // -6 (R - d)
    } else {
        T t = c.s * d;
        T e = c.a * dev_exp<T>(-t);
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

// (a_q, b_q) of the fit k(d) ~ a_q + b_q d^2 for one query (see gpx_internal.hpp): mean and variance of
// u = |q - p|^2 over the training points from the cloud's moments (double: E[u^2] - E[u]^2 cancels), then the
// least-squares line through k at u_lo, u_mid, u_hi = mean -+ sqrt(3) sigma (clamped at 0).
template <typename T, int KID>
__device__ __forceinline__ void var_fit_query(const Cov<T> &cov, const double *__restrict__ mom, T qx, T qy, T qz,
                                              T &fa, T &fb)
{
    const double x = (double)qx, y = (double)qy, z = (double)qz;
    const double a = x * x + y * y + z * z;
    const double bx = -2.0 * x, by = -2.0 * y, bz = -2.0 * z;
    const double bm1 = bx * mom[0] + by * mom[1] + bz * mom[2];
    const double s2 = mom[9];
    const double eu = a + bm1 + s2;
    const double bMb = bx * (bx * mom[3] + 2.0 * (by * mom[4] + bz * mom[5])) + by * (by * mom[6] + 2.0 * bz * mom[7]) +
                       bz * bz * mom[8];
    const double eu2 = a * a + 2.0 * a * (bm1 + s2) + bMb + 2.0 * (bx * mom[10] + by * mom[11] + bz * mom[12]) + mom[13];
    const double var = fmax(eu2 - eu * eu, 0.0);
    const double hw = 1.7320508075688772 * sqrt(var);
    const double ulo = fmax(eu - hw, 0.0), uhi = fmax(eu + hw, 0.0), umid = 0.5 * (ulo + uhi);
    const T k0 = cov_k<T, KID>(cov, (T)ulo), k1 = cov_k<T, KID>(cov, (T)umid), k2 = cov_k<T, KID>(cov, (T)uhi);
    const double du = uhi - ulo;
    const double b = du > 1e-12 * (1.0 + uhi) ? ((double)k2 - (double)k0) / du : 0.0;
    fb = (T)b;
    fa = (T)(((double)k0 + (double)k1 + (double)k2) * (1.0 / 3.0) - (double)fb * umid);
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
