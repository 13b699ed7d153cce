// Device arithmetic primitives for the MOD16 pixel kernels (gfx950).
//
// Two policies share one interface:
//   ExactMath<T>  IEEE divide, ocml exp/pow -- used by the kernel variant that
//                 keeps the reference's operation order (MOD16_MATH_EXACT);
//   FastMath<T>   hardware reciprocal / reciprocal-square-root seeds refined by
//                 one Newton step, table-driven exp and log (tables in LDS).
//                 Relative error of each primitive is ~1e-14 (f64), far inside
//                 the 1e-5 parity budget; the f64 VALU rate, not HBM, is what a
//                 straight IEEE transcription of this pixel stack is bound by
//                 (DESIGN.md "Arithmetic budget").
#pragma once
// Measurement and mapping switches (tools/README.md) exist only in builds made with
// -DMOD16_EXPERIMENTS, which mod16_amd/csrc/build.py passes to libmod16hip_exp.so alone: the
// shipped library has one arithmetic and one launch geometry, and reads no environment variable
// that changes either (tests/test_abi.py checks its strings).
#ifndef MOD16_EXPERIMENTS
#if defined(MOD16_TRIVIAL_BODY) || defined(MOD16_NO_GUARD) || defined(MOD16_EXPERIMENT_SEED_RCP) || \
    defined(MOD16_PRIO) || defined(MOD16_REPRO_V4) || defined(MOD16_NO_FUSED_FINAL) ||                \
    defined(MOD16_NO_REDO_LAUNCH) || defined(MOD16_KK_M) || defined(MOD16_NO_FMA_KK) ||               \
    defined(MOD16_DYN_RUN) || defined(MOD16_MIXED_NO_CANCEL) || defined(MOD16_MIXED_CANCEL) || defined(MOD16_F64_CAND)
#error "MOD16_* measurement switches need -DMOD16_EXPERIMENTS (they are not part of the product build)"
#endif
#endif
#include <hip/hip_runtime.h>

namespace mod16 {

// The domain guards (mod16_physics.hpp, mod16_mixed.hpp) are chains of v_max_f64 / v_max3_f32,
// which ignore NaN operands -- QUIET ones. Compute waves start with MODE.IEEE = 1, where a
// SIGNALLING NaN operand makes v_max return the quieted NaN instead, and the next link of the
// chain then drops the running maximum: an sNaN bit pattern in one driver would hide an infinity
// in another. With MODE.IEEE = 0 v_min / v_max treat signalling NaNs like quiet
// ones. Nothing else in these kernels depends on the bit (it only governs sNaN quieting; every
// comparison of the pixel functions is an ordinary v_cmp), so the kernels that evaluate a guard
// clear it for their waves first thing: one scalar instruction per wave.
__device__ __forceinline__ void ignore_signalling_nans() {
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0\n\ts_nop 3" ::: "memory");
}

// A constant in a VECTOR register pair: the production loops fill the scalar register file with
// the pixel function's constants (v_fma_f64 takes no literal on gfx9: every float64 constant that is
// not one of the inline ones occupies a scalar pair), and what does not fit is rebuilt from two
// s_mov_b32 at every use; vector registers are to spare at two waves per SIMD. The compiler
// hoists the (side-effect-free) statement in front of a loop.
__device__ __forceinline__ double in_vgpr(double k) {
    asm("" : "+v"(k));
    return k;
}

template <typename T> struct ExactMath {
    static __device__ __forceinline__ T exp(T x);
    static __device__ __forceinline__ T pow(T x, T y);
};
template <> __device__ __forceinline__ double ExactMath<double>::exp(double x) { return ::exp(x); }
template <> __device__ __forceinline__ float ExactMath<float>::exp(float x) { return ::expf(x); }
template <> __device__ __forceinline__ double ExactMath<double>::pow(double x, double y) { return ::pow(x, y); }
template <> __device__ __forceinline__ float ExactMath<float>::pow(float x, float y) { return ::powf(x, y); }

template <typename T> struct FastMath;   // float64 only: FAST kernels always compute in float64

// ---------------------------------------------------------------- float64
template <> struct FastMath<double> {
    typedef double T;

    // x * m + c with two CONSTANTS m and c: one v_fma_f64 with m and c in registers of their own. For this shape hipcc 7.2 emits v_mov_b64
    // (copy c) + v_fmac_f64 -- two vector instructions; 28 of them per pair of pixels.
    // (m in a VECTOR register pair as well since round 3: the domain guard's few extra scalar
    // values made the pipeline's loop spill scalar registers -- every "s" constant occupies a
    // pair all through the loop, v_fma_f64 takes no literal on gfx9 -- while vector registers
    // are to spare at two waves per SIMD. -DMOD16_KK_M='"s"' restores the scalar form.)
#ifndef MOD16_KK_M
#define MOD16_KK_M "v"
#endif
    static __device__ __forceinline__ T fma_kk(T x, T m, T c) {
#ifdef MOD16_NO_FMA_KK
        return __builtin_fma(x, m, c);
#else
        T d;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(x), MOD16_KK_M(m), "v"(c));
        return d;
#endif
    }

    // maxNum / minNum as ONE v_max_f64 / v_min_f64. __builtin_fmax makes hipcc quiet a
    // possible signalling NaN first (v_max_f64 x, x, x in front of every use: 20 extra
    // vector instructions per pair of pixels); every value that reaches these is the
    // result of arithmetic, i.e. already quiet. `k`: a constant (scalar register pair).
    static __device__ __forceinline__ T vmax(T a, T b) {
        T d;
        asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    static __device__ __forceinline__ T vmax_k(T a, T k) {
        T d;
        asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
        return d;
    }
    static __device__ __forceinline__ T vmin_k(T a, T k) {
        T d;
        asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "s"(k));
        return d;
    }

    // min(max(x, 0), 1) as v_max_f64 + v_min_f64 (a NaN x gives 0: v_max passes it over)
    static __device__ __forceinline__ T clamp01(T x) {
        T d;
        asm("v_max_f64 %0, %1, 0\n\tv_min_f64 %0, %0, 1.0" : "=&v"(d) : "v"(x));
        return d;
    }

    // 1/x: v_rcp_f64 (about 2^-23 relative) + one Newton step -> ~2^-45.
    static __device__ __forceinline__ T rcp(T x) {
        T r = __builtin_amdgcn_rcp(x);
        T e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(r, e, r);
    }

    // (measurement builds only, -DMOD16_EXPERIMENT_SEED_RCP: the reciprocals of the three
    // Penman-Monteith quotients without their Newton step -- how the step time answers to
    // 12 float64 fma fewer per pixel; DESIGN.md section 6. Not the product: the step also makes
    // 1/0 and 1/inf NaN, as the reference's 0/0 and inf/inf are.)
    static __device__ __forceinline__ T rcp_quotient(T x) {
#ifdef MOD16_EXPERIMENT_SEED_RCP
        return __builtin_amdgcn_rcp(x);
#else
        return rcp(x);
#endif
    }

    // x^(-7/4) for the r_corr term (mod16/__init__.py:771). Seed y = x^(-1/4)
    // from v_rsq_f64 + v_sqrt_f64 (relative error <= 5.3e-8, measured seeds
    // 2^-24.2 and 2^-25.3), one Newton step on y^-4 = x (error 2.5 e^2 = 7e-15),
    // then y^7 (5e-14).
    static __device__ __forceinline__ T pow_m1p75(T x) {
        T y = __builtin_amdgcn_sqrt(__builtin_amdgcn_rsq(x));
        T y2 = y * y;
        T t = x * (y2 * y2);
        y = y * fma_kk(t, -0.25, 1.25);
        y2 = y * y;
        T y4 = y2 * y2;
        return (y4 * y2) * y;
    }

    // x^(-7/4) and x^-1 from the same seed and Newton step (round 5): after the step y = x^(-1/4)
    // to 7e-15, so y^4 is 1 / x to 2.8e-14 -- the forward run's period needs 1 / t for the air
    // density next to t^(-7/4) for r_corr, and gets it here for nothing (a reciprocal with its
    // Newton step and two products less per period than 1 / (N t) did).
    static __device__ __forceinline__ T pow_m1p75_rcp(T x, T& rx) {
        T y = __builtin_amdgcn_sqrt(__builtin_amdgcn_rsq(x));
        T y2 = y * y;
        T t = x * (y2 * y2);
        y = y * fma_kk(t, -0.25, 1.25);
        y2 = y * y;
        T y4 = y2 * y2;
        rx = y4;
        return (y4 * y2) * y;
    }

    // ---- table-driven exp / log (tables in LDS, filled by the library)
    // tb[0..63]      = 2^(j/64)
    // tb[64 + 2j..]  = { 1/c_j rounded, -log(1/c_j) },  c_j = 1 + (j + 1/2)/128
    // tb[320 .. 335]   constants of the raw-driver pre-processing (mod16_physics.hpp, raw_to_pixel_fast):
    //                  [0..9] air pressure from elevation as a polynomial in u = (z - 5000) / 7000
    static constexpr int kTabRaw = 64 + 2 * 128;
    static constexpr int kTabDoubles = kTabRaw + 16;
    static constexpr double kRintShift = 6755399441055744.0;       // 1.5 * 2^52

    // e^x for finite x (no clamp: a huge |x| saturates through v_cvt_i32 and
    // v_ldexp to 0 or inf; NaN -> NaN; +-inf -> NaN, callers clamp where -inf
    // can occur). 12 f64-rate operations, relative error ~2e-16 (esat wants all of it: rh =
    // (esat - vpd) / esat amplifies its error by 1 / rh in dry air).
    static __device__ __forceinline__ T exp_tab(T x, const T* tb) {
        // k = rint(x 64 / ln 2) by the 1.5 * 2^52 shift: the integer is then the low word of
        // the shifted sum (no v_cvt_i32_f64). |x| beyond 2^31 ln2 / 64 = 2.3e7 is not reduced
        // properly any more (the callers' arguments are bounded: [-746, 0], or 17.3 tc / (tc + 237)
        // of a temperature)
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);     // 64 / ln 2
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0x1.62e42fee00000p-7, x);          // ln2/64, 32-bit head
        r = __builtin_fma(kf, -0x1.a39ef35793c76p-39, r);
        T p = fma_kk(r, 1.0 / 120.0, 1.0 / 24.0);                   // |r| <= ln2/128
        p = __builtin_fma(p, r, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // the same with a quartic: the first term left out is r^5/120 <= 4e-14 relative -- for
    // results that are not differenced afterwards (rh^(vpd/beta))
    static __device__ __forceinline__ T exp_tab4(T x, const T* tb) {
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0x1.62e42fee00000p-7, x);
        r = __builtin_fma(kf, -0x1.a39ef35793c76p-39, r);
        T p = fma_kk(r, 1.0 / 24.0, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // quartic, and the argument reduced with ONE constant (round 5): r = x - k ln2/64 with ln2/64
    // rounded to float64 is off by |k| 1.2e-18 -- for the forward run's two exponentials (the
    // saturation pressure: |x| < 20, k < 2000: 2e-15 relative; rh^(vpd/beta): x in [-746, 0], where
    // the error grows only as the value vanishes) that is the size of the result's own rounding.
    // Two float64 operations less than exp_tab.
    static __device__ __forceinline__ T exp_tab4s(T x, const T* tb) {
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0.010830424696249145, x);          // ln 2 / 64
        T p = fma_kk(r, 1.0 / 24.0, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // exp_tab with the one-constant reduction (quintic as there: the saturation pressure feeds
    // differences -- esat - vpd, s A + rho Cp vpd / r_a at night -- that amplify its error 1e5-fold
    // in the worst pixels of a global grid; a quartic's 4e-14 showed there as 1.3e-9 against the
    // reference-order kernel where 1e-9 is promised)
    static __device__ __forceinline__ T exp_tab5s(T x, const T* tb) {
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0.010830424696249145, x);          // ln 2 / 64
        T p = fma_kk(r, 1.0 / 120.0, 1.0 / 24.0);
        p = __builtin_fma(p, r, 1.0 / 6.0);
        p = __builtin_fma(p, r, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // the same to 4e-11 (cubic): for results that end up in float32 (mod16_mixed.hpp)
    static __device__ __forceinline__ T exp_tab3(T x, const T* tb) {
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0x1.62e42fee00000p-7, x);
        r = __builtin_fma(kf, -0x1.a39ef35793c76p-39, r);
        T p = fma_kk(r, 1.0 / 6.0, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // ... and with the one-constant reduction (mixed form, round 5)
    static __device__ __forceinline__ T exp_tab3s(T x, const T* tb) {
        T km = __builtin_fma(x, 92.33248261689366, kRintShift);
        T kf = km - kRintShift;
        T r = __builtin_fma(kf, -0.010830424696249145, x);
        T p = fma_kk(r, 1.0 / 6.0, 0.5);
        p = __builtin_fma(p, r, 1.0);
        p = __builtin_fma(p, r, 1.0);
        int ki = __double2loint(km);
        return __builtin_amdgcn_ldexp(tb[ki & 63] * p, ki >> 6);
    }

    // ln(x) for x > 0 normal (x is a relative humidity in (0, 1] here);
    // log(1) is exactly 0 by construction of table entry 0; x = 0 -> -inf.
    // Absolute error ~1e-15 (what matters: the result feeds exp(y log x)).
    static __device__ __forceinline__ T log_tab(T x, const T* tb) {
        const unsigned hi = (unsigned)__double2hiint(x);
        const int e = (int)(hi >> 20) - 1023;
        const unsigned j = (hi >> 13) & 127u;
        const T m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(x));
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 ent = *reinterpret_cast<const d2*>(tb + 64 + 2 * j);
        T r = __builtin_fma(m, ent[0], -1.0);                        // |r| <= 1/256
        T l1p = log1p_poly(r);
        T ed = (T)e;
        T v = __builtin_fma(ed, 0x1.62e42fefa2000p-1, ent[1]);      // ln2, 40-bit head
        v = v + l1p;
        v = __builtin_fma(ed, 7.371002565167799e-13, v);
        return (x == 0.0) ? -__builtin_huge_val() : v;
    }
    // the same with ln 2 as ONE constant: e (ln 2 - fl(ln 2)) = e 2.3e-17 absolute -- for the
    // relative humidities of the forward run (e = -1 ... -10; below, rh^y is vanishing anyway)
    static __device__ __forceinline__ T log_tab1(T x, const T* tb) {
        const unsigned hi = (unsigned)__double2hiint(x);
        const int e = (int)(hi >> 20) - 1023;
        const unsigned j = (hi >> 13) & 127u;
        const T m = __hiloint2double((int)((hi & 0x000fffffu) | 0x3ff00000u), __double2loint(x));
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 ent = *reinterpret_cast<const d2*>(tb + 64 + 2 * j);
        T r = __builtin_fma(m, ent[0], -1.0);
        T v = __builtin_fma((T)e, 0.6931471805599453, ent[1]) + log1p_poly(r);
        return (x == 0.0) ? -__builtin_huge_val() : v;
    }
    // r - r^2/2 + ... + r^5/5 on |r| <= 1/256: the next term, r^6/6, is below 6e-16 (the
    // exponent vpd / beta reaches thousands on the calibration path, where beta is sampled
    // down to ~1: a quartic's 1.9e-13 showed as 1e-9 there); the host evaluates the same
    // sequence (std::fma) to make table entry 0 cancel exactly at x = 1
    static __host__ __device__ __forceinline__ T log1p_poly(T r) {
#if defined(__HIP_DEVICE_COMPILE__)
        T p = fma_kk(r, 0.2, -0.25);
#else
        T p = __builtin_fma(r, 0.2, -0.25);
#endif
        p = __builtin_fma(p, r, 1.0 / 3.0);
        p = __builtin_fma(p, r, -0.5);
        return __builtin_fma(r * r, p, r);
    }

    // x^y for x in [0, 1] (rh ** (vpd / beta), mod16/__init__.py:861).
    // C pow semantics kept for: 0^y = 0 (y > 0), 1^y = 1 for every y incl. inf
    // and NaN (log_tab(1) = 0 exactly and y is clamped finite), x^0 = 1.
    static __device__ __forceinline__ T pow01_tab(T x, T y, const T* tb) {
        T yc = vmin_k(y, 1e300);
        T t = vmax_k(yc * log_tab(x, tb), -746.0);
        return exp_tab4(t, tb);
    }
    // the forward run's form (round 5): one-constant reductions in the log and the exp
    static __device__ __forceinline__ T pow01_tab1(T x, T y, const T* tb) {
        T yc = vmin_k(y, 1e300);
        T t = vmax_k(yc * log_tab1(x, tb), -746.0);
        return exp_tab4s(t, tb);
    }
};

}  // namespace mod16
