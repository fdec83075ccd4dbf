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
