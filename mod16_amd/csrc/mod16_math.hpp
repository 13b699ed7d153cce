// Device arithmetic primitives for the MOD16 pixel kernels (gfx950).
//
// Two policies share one interface:
//   ExactMath<T>  IEEE divide, ocml exp/pow -- used by the kernel variant that
//                 keeps the reference's operation order (MOD16_MATH_EXACT);
//   FastMath<T>   hardware reciprocal / reciprocal-square-root seeds refined by
//                 Newton steps, range-reduced polynomial exp and log. Relative
//                 error of each primitive is ~1e-14 (f64), far inside the
//                 1e-5 parity budget; the f64 VALU rate, not HBM, is what a
//                 straight IEEE transcription of this pixel stack is bound by
//                 (DESIGN.md "Arithmetic budget").
#pragma once
#include <hip/hip_runtime.h>

namespace mod16 {

template <typename T> struct ExactMath {
    static __device__ __forceinline__ T div(T a, T b) { return a / b; }
    static __device__ __forceinline__ T exp(T x);
    static __device__ __forceinline__ T pow(T x, T y);
};
template <> __device__ __forceinline__ double ExactMath<double>::exp(double x) { return ::exp(x); }
template <> __device__ __forceinline__ float ExactMath<float>::exp(float x) { return ::expf(x); }
template <> __device__ __forceinline__ double ExactMath<double>::pow(double x, double y) { return ::pow(x, y); }
template <> __device__ __forceinline__ float ExactMath<float>::pow(float x, float y) { return ::powf(x, y); }

template <typename T> struct FastMath;

// ---------------------------------------------------------------- float64
template <> struct FastMath<double> {
    typedef double T;

    // 1/x: v_rcp_f64 (about 2^-23 relative) + one Newton step -> ~2^-45.
    static __device__ __forceinline__ T rcp(T x) {
        T r = __builtin_amdgcn_rcp(x);
        T e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(r, e, r);
    }
    // Newton steps make NaN out of rcp(0) = inf and rcp(inf) = 0; this form
    // keeps the IEEE results 1/0 = inf, 1/inf = 0 where a mask depends on them.
    static __device__ __forceinline__ T rcp_safe(T x) {
        T r0 = __builtin_amdgcn_rcp(x);
        T e = __builtin_fma(-x, r0, 1.0);
        T r1 = __builtin_fma(r0, e, r0);
        return (r1 == r1) ? r1 : r0;
    }
    static __device__ __forceinline__ T div(T a, T b) { return a * rcp(b); }

    // x^(-7/4) for the r_corr term (mod16/__init__.py:771). Seed y = x^(-1/4)
    // from v_rsq_f64 + v_sqrt_f64, two Newton steps on y^-4 = x, then y^7.
    static __device__ __forceinline__ T pow_m1p75(T x) {
        T y = __builtin_amdgcn_sqrt(__builtin_amdgcn_rsq(x));
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            T y2 = y * y;
            T t = x * (y2 * y2);
            y = y * __builtin_fma(-0.25, t, 1.25);
        }
        T y2 = y * y;
        T y4 = y2 * y2;
        return (y4 * y2) * y;
    }

    // e^x, |rel err| < 1e-15 over the finite range; exp(-inf) = 0,
    // exp(+inf) = inf, exp(NaN) = NaN.
    static __device__ __forceinline__ T exp(T x) {
        T xc = __builtin_fmin(__builtin_fmax(x, -746.0), 710.0);
        T k = __builtin_rint(xc * 1.4426950408889634);
        T r = __builtin_fma(k, -6.93147180369123816490e-01, xc);
        r = __builtin_fma(k, -1.90821492927058770002e-10, r);
        // Taylor to r^12 on |r| <= ln2/2: truncation 1.7e-16
        T p = 1.0 / 479001600.0;
        p = __builtin_fma(p, r, 1.0 / 39916800.0);
        p = __builtin_fma(p, r, 1.0 / 3628800.0);
        p = __builtin_fma(p, r, 1.0 / 362880.0);
        p = __builtin_fma(p, r, 1.0 / 40320.0);
        p = __builtin_fma(p, r, 1.0 / 5040.0);
        p = __builtin_fma(p, r, 1.0 / 720.0);
        p = __builtin_fma(p, r, 1.0 / 120.0);
        p = __builtin_fma(p, r, 1.0 / 24.0);
        p = __builtin_fma(p, r, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        T v = __builtin_amdgcn_ldexp(p, (int)k);
        return (x == x) ? v : x;
    }

    // ln(x) for x >= 0 (x is a relative humidity in [0, 1] here):
    // log(0) = -inf, log(NaN) = NaN. |rel err| ~ 1e-15.
    static __device__ __forceinline__ T log(T x) {
        T m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
        int e = __builtin_amdgcn_frexp_exp(x);
        bool lo = m < 0.70710678118654752440;
        m = lo ? m + m : m;                             // [sqrt(.5), sqrt(2))
        e = lo ? e - 1 : e;
        T f = m - 1.0;
        T s = f * rcp(2.0 + f);
        T z = s * s;
        // atanh series: log(m) = 2s (1 + z/3 + z^2/5 + ...), z <= 0.0295
        T p = 1.0 / 21.0;
        p = __builtin_fma(p, z, 1.0 / 19.0);
        p = __builtin_fma(p, z, 1.0 / 17.0);
        p = __builtin_fma(p, z, 1.0 / 15.0);
        p = __builtin_fma(p, z, 1.0 / 13.0);
        p = __builtin_fma(p, z, 1.0 / 11.0);
        p = __builtin_fma(p, z, 1.0 / 9.0);
        p = __builtin_fma(p, z, 1.0 / 7.0);
        p = __builtin_fma(p, z, 1.0 / 5.0);
        p = __builtin_fma(p, z, 1.0 / 3.0);
        T sz = s * z;
        T lm = __builtin_fma(sz, p, s);                 // s + s z p
        lm = lm + lm;
        T ed = (T)e;
        T v = __builtin_fma(ed, 6.93147180369123816490e-01, lm);
        v = __builtin_fma(ed, 1.90821492927058770002e-10, v);
        return (x == 0.0) ? -__builtin_huge_val() : v;
    }

    // x^y for x in [0, 1] (rh ** (vpd / beta), mod16/__init__.py:861)
    static __device__ __forceinline__ T pow01(T x, T y) {
        T v = exp(y * log(x));
        // 1 ** y = 1 for every y incl. inf and NaN (C pow); y == 0 -> 1
        return (x == 1.0 || y == 0.0) ? 1.0 : v;
    }
};

// ---------------------------------------------------------------- float32
template <> struct FastMath<float> {
    typedef float T;
    static __device__ __forceinline__ T rcp(T x) {
        T r = __builtin_amdgcn_rcpf(x);
        T e = __builtin_fmaf(-x, r, 1.0f);
        return __builtin_fmaf(r, e, r);
    }
    static __device__ __forceinline__ T rcp_safe(T x) {
        T r0 = __builtin_amdgcn_rcpf(x);
        T e = __builtin_fmaf(-x, r0, 1.0f);
        T r1 = __builtin_fmaf(r0, e, r0);
        return (r1 == r1) ? r1 : r0;
    }
    static __device__ __forceinline__ T div(T a, T b) { return a * rcp(b); }
    static __device__ __forceinline__ T pow_m1p75(T x) {
        T y = __builtin_amdgcn_sqrtf(__builtin_amdgcn_rsqf(x));
        T y2 = y * y;
        T t = x * (y2 * y2);
        y = y * __builtin_fmaf(-0.25f, t, 1.25f);
        y2 = y * y;
        T y4 = y2 * y2;
        return (y4 * y2) * y;
    }
    static __device__ __forceinline__ T exp(T x) { return ::expf(x); }
    static __device__ __forceinline__ T log(T x) { return ::logf(x); }
    static __device__ __forceinline__ T pow01(T x, T y) {
        T v = ::expf(y * ::logf(x));
        return (x == 1.0f || y == 0.0f) ? 1.0f : v;
    }
};

}  // namespace mod16
