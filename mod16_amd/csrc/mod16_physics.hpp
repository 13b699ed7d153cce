// Per-pixel MOD16 Penman-Monteith stack as device functions (gfx950).
//
// Replaces the reference's instance path MOD16.evapotranspiration()
// (reference mod16/__init__.py:675-793) and its callees. Two formulations:
//
//   et_pixel_exact<T>  keeps the reference's operation order, one IEEE
//                      operation per numpy ufunc, contraction off. Used by
//                      MOD16_MATH_EXACT and as the on-device cross-check of
//                      the fast form at full raster sizes.
//   et_pixel_fast<T>   the production form: every intermediate shared by the
//                      three components is computed once per period, parallel
//                      resistances are carried as conductances so each
//                      component costs one reciprocal, pow/exp/log are the
//                      FastMath forms. All predicates (np.where masks) are
//                      kept with the reference's comparison semantics so that
//                      zero/NaN masks are identical.
#pragma once
#include "mod16_math.hpp"

namespace mod16 {

template <typename T> struct PixelIn {
    T lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_ann, tmin, vpd_d, vpd_n, pa,
        fpar, lai;
};

// The 11 reference parameters (mod16/__init__.py:152-155) + derived terms.
template <typename T> struct ClassPar {
    T tmin_close, tmin_open, vpd_open, vpd_close, gl_sh, gl_wv, g_cut, csl,
        rbl_min, rbl_max, beta;
    // derived (filled by derive()): reciprocal ramp widths, rbl slope, 1/beta
    T inv_dtmin, inv_dvpd, rbl_slope, inv_beta;
    __device__ __forceinline__ void derive() {
        inv_dtmin = T(1) / (tmin_open - tmin_close);
        inv_dvpd = T(1) / (vpd_close - vpd_open);
        rbl_slope = (rbl_max - rbl_min) / (vpd_close - vpd_open);
        inv_beta = T(1) / beta;
    }
};
#define MOD16_LUT_ROWS 16  // 11 parameters + 5 derived, LDS layout [row][16]

template <typename T> struct PixelOut {
    T canopy_d, soil_d, trans_d, canopy_n, soil_n, trans_n;
    T pet_d, pet_n;   // potential ET (only filled by the *_pet forms)
};
constexpr double kPriestleyTaylorAlpha = 1.26;   // mod16/__init__.py:550

// Module constants, mod16/__init__.py:106-118, :869, :1157
template <typename T> struct K {
    static constexpr T sigma4 = T(4 * 5.67e-8);       // 4 * STEFAN_BOLTZMANN
    static constexpr T cp = T(1013);                  // SPECIFIC_HEAT_CAPACITY_AIR
    static constexpr T eps = T(0.622);                // MOL_WEIGHT_WET_DRY_RATIO_AIR
    static constexpr T tiny = T(1e-7);
    static constexpr T t0 = T(273.15);
};

// ===================================================================== exact
// One function per reference method; `#pragma clang fp contract(off)` keeps
// mul and add separate as numpy does.

template <typename T> __device__ __forceinline__ T lhv_exact(T t) {
#pragma clang fp contract(off)
    return (T(2.501) - T(0.002361) * (t - K<T>::t0)) * T(1e6);  // :121
}
template <typename T> __device__ __forceinline__ T svp_exact(T t) {
#pragma clang fp contract(off)
    T tc = t - K<T>::t0;                                          // :1365-1367
    return T(1e3 * 0.6108) * ExactMath<T>::exp((T(17.27) * tc) / (tc + T(237.3)));
}
template <typename T> __device__ __forceinline__ T svp_slope_exact(T t) {
#pragma clang fp contract(off)
    T s = svp_exact(t);                                           // :1395-1397
    T d = (T(239.0) + t) - K<T>::t0;
    return (T(17.38 * 239.0) * s) / (d * d);
}
template <typename T> __device__ __forceinline__ T gamma_exact(T pa, T t) {
#pragma clang fp contract(off)
    return (K<T>::cp * pa) / (lhv_exact(t) * K<T>::eps);          // :1288-1290
}
template <typename T> __device__ __forceinline__ T rho_exact(T t, T pa, T rh) {
#pragma clang fp contract(off)
    return (T(0.348444) * (pa / T(100)) -                         // :408-412
            (rh * T(100)) * (T(0.00252) * (t - K<T>::t0) - T(0.020582))) / t;
}
template <typename T> __device__ __forceinline__ T rh_exact(T t, T vpd) {
#pragma clang fp contract(off)
    T esat = svp_exact(t);                                        // :669-673
    T avp = esat - vpd;
    T rh = avp / esat;
    return (avp < T(0)) ? T(0) : ((rh > T(1)) ? T(1) : rh);
}
template <typename T> __device__ __forceinline__ T fwet_exact(T rh) {
    return (rh < T(0.7)) ? T(0) : ExactMath<T>::pow(rh, T(4));    // :764
}
template <typename T> __device__ __forceinline__ T rcorr_exact(T pa, T t) {
#pragma clang fp contract(off)
    return (T(101300) / pa) * ExactMath<T>::pow(t / T(293.15), T(1.75));  // :771
}
template <typename T> __device__ __forceinline__ T rr_exact(T rho, T t) {
#pragma clang fp contract(off)
    return (rho * K<T>::cp) / (K<T>::sigma4 * ExactMath<T>::pow(t, T(3)));  // :947
}

// soil_heat_flux + radiation_soil, :963-1119
template <typename T>
__device__ __forceinline__ void rad_soil_exact(const PixelIn<T>& x, const ClassPar<T>& p,
                                               T& rs_d, T& rs_n) {
#pragma clang fp contract(off)
    T a_d = x.sw_d * (T(1) - x.alb) + x.lw_d;                     // :1033
    T a_n = x.lw_n;                                               // :1034
    bool cond = (x.t_ann < T(273.15 + 25.0)) &&                   // :1103-1107
                (x.t_ann >= (K<T>::t0 + p.tmin_close)) && ((x.t_d - x.t_n) >= T(5));
    T g_d = cond ? (T(4.73) * (x.t_d - K<T>::t0)) - T(20.87) : T(0);  // :1110
    g_d = (__builtin_fabs(g_d) > (T(0.39) * __builtin_fabs(a_d))) ? T(0.39) * a_d : g_d;
    T g_n = cond ? (T(4.73) * (x.t_n - K<T>::t0)) - T(20.87) : T(0);
    g_n = (__builtin_fabs(g_n) > (T(0.39) * __builtin_fabs(a_n))) ? T(0.39) * a_n : g_n;
    g_d = ((a_d - g_d < T(0)) && (a_d > T(0))) ? a_d : g_d;       // :1039-1041
    g_n = ((a_d > T(0)) && ((a_n - g_n) < (T(-0.5) * a_d)))       // :1042-1046
              ? a_n + (T(0.5) * a_d) : g_n;
    rs_d = (T(1) - x.fpar) * (a_d - g_d);                         // :1051-1052
    rs_n = (T(1) - x.fpar) * (a_n - g_n);
}

// evaporation_wet_canopy, :866-961
template <typename T>
__device__ __forceinline__ T wet_canopy_exact(const ClassPar<T>& p, T pa, T t, T vpd, T lai,
                                              T fpar, T rad_canopy, T lhv, T rh, T fwet,
                                              T tiny = K<T>::tiny) {
#pragma clang fp contract(off)
    fwet = (fwet == T(0)) ? fwet + tiny : fwet;                   // :934-935
    lai = (lai == T(0)) ? lai + tiny : lai;
    T s = svp_slope_exact(t);
    T rho = rho_exact(t, pa, rh);
    T r_h = T(1) / (p.gl_sh * lai * fwet);                        // :943
    T r_e = T(1) / (p.gl_wv * lai * fwet);                        // :945
    T r_r = rr_exact(rho, t);
    T r_a = (r_h * r_r) / (r_h + r_r);                            // :951
    T numer = fwet * ((s * rad_canopy) + (rho * K<T>::cp * fpar * vpd * T(1) / r_a));
    T denom = s + ((pa * K<T>::cp * r_e) * T(1) / (lhv * K<T>::eps * r_a));
    T evap = (numer < T(0)) ? T(0) : (numer / denom) / lhv;       // :959
    return ((fwet <= tiny) || (lai <= tiny)) ? T(0) : evap;       // :961
}

// potential_soil_evaporation, :449-544
template <typename T>
__device__ __forceinline__ void pot_soil_exact(const ClassPar<T>& p, T pa, T t, T vpd, T fpar,
                                               T rad_soil, T r_corr, T rh, T fwet, T& sat,
                                               T& unsat) {
#pragma clang fp contract(off)
    T s = svp_slope_exact(t);
    T rho = rho_exact(t, pa, rh);
    T gamma = gamma_exact(pa, t);
    T r_r = rr_exact(rho, t);
    T r_tot = (vpd <= p.vpd_open) ? p.rbl_min                      // :527-531
              : ((vpd >= p.vpd_close) ? p.rbl_max
                 : p.rbl_max - ((p.rbl_max - p.rbl_min) * (p.vpd_close - vpd)) /
                                   (p.vpd_close - p.vpd_open));
    r_tot = r_tot / r_corr;                                       // :533
    T r_as = (r_tot * r_r) / (r_tot + r_r);                       // :535
    T numer = (s * rad_soil) + (rho * K<T>::cp * (T(1) - fpar) * (vpd / r_as));
    T denom = s + gamma * (r_tot / r_as);
    sat = (numer * fwet) / denom;                                 // :541-543
    unsat = (numer * (T(1) - fwet)) / denom;
}

// evaporation_soil, :795-864
template <typename T>
__device__ __forceinline__ T soil_exact(const ClassPar<T>& p, T pa, T t, T vpd, T fpar,
                                        T rad_soil, T r_corr, T lhv, T rh, T fwet) {
#pragma clang fp contract(off)
    T sat, unsat;
    pot_soil_exact(p, pa, t, vpd, fpar, rad_soil, r_corr, rh, fwet, sat, unsat);
    T e = (sat < T(0)) ? T(0) : sat;                              // :858-861
    e = e + ((unsat < T(0)) ? T(0) : unsat * ExactMath<T>::pow(rh, vpd / p.beta));
    return e / lhv;                                               // :864
}

// soil_heat_flux, :1055-1119
template <typename T>
__device__ __forceinline__ void soil_heat_flux_exact(const ClassPar<T>& p, T a_d, T a_n, T t_d,
                                                     T t_n, T t_ann, T& g_d, T& g_n) {
#pragma clang fp contract(off)
    bool cond = (t_ann < T(273.15 + 25.0)) && (t_ann >= (K<T>::t0 + p.tmin_close)) &&
                ((t_d - t_n) >= T(5));
    g_d = cond ? (T(4.73) * (t_d - K<T>::t0)) - T(20.87) : T(0);
    g_d = (__builtin_fabs(g_d) > (T(0.39) * __builtin_fabs(a_d))) ? T(0.39) * a_d : g_d;
    g_n = cond ? (T(4.73) * (t_n - K<T>::t0)) - T(20.87) : T(0);
    g_n = (__builtin_fabs(g_n) > (T(0.39) * __builtin_fabs(a_n))) ? T(0.39) * a_n : g_n;
}

// MOD16.potential_transpiration, :546-602
template <typename T>
__device__ __forceinline__ T pot_transpiration_exact(T lw, T sw, T alb, T pa, T t, T fpar,
                                                     T fwet, T alpha) {
#pragma clang fp contract(off)
    T rad_c = fpar * (sw * (T(1) - alb) + lw);
    T s = svp_slope_exact(t);
    T gamma = gamma_exact(pa, t);
    return (alpha * (s * rad_c) * (T(1) - fwet)) / (s + gamma);
}

// MOD16.vpd, :604-644 (note: its own SVP constants)
template <typename T> __device__ __forceinline__ T vpd_exact(T qv, T pa, T tmean) {
#pragma clang fp contract(off)
    T tc = tmean - K<T>::t0;
    T avp = (qv * pa) / (T(0.622) + (T(0.379) * qv));
    T sv = T(610.7) * ExactMath<T>::exp((T(17.38) * tc) / (T(239) + tc));
    return sv - avp;
}

// MOD16.air_pressure, :414-447
template <typename T> __device__ __forceinline__ T air_pressure_exact(T elev) {
#pragma clang fp contract(off)
    T ratio = T(1) - ((T(0.0065) * elev) / T(288.15));
    return T(101325.0) * ExactMath<T>::pow(ratio, T(9.80665 / (0.0065 * (8.3143 / 28.9644e-3))));
}

// radiation_net (deprecated), :1293-1337
template <typename T> __device__ __forceinline__ T radiation_net_exact(T sw, T alb, T t) {
#pragma clang fp contract(off)
    T tc = t - K<T>::t0;
    T emis_a = T(1) - T(0.26) * ExactMath<T>::exp(T(-7.77e-4) * (tc * tc));
    return sw * (T(1) - alb) + T(5.67e-8) * (emis_a - T(0.97)) * ExactMath<T>::pow(t, T(4));
}

// mod17.linear_constraint (reference README.md:351-369)
template <typename T> __device__ __forceinline__ T ramp_up_exact(T x, T lo, T hi) {
#pragma clang fp contract(off)
    return (x >= hi) ? T(1) : ((x < lo) ? T(0) : (x - lo) / (hi - lo));
}
template <typename T> __device__ __forceinline__ T ramp_down_exact(T x, T lo, T hi) {
#pragma clang fp contract(off)
    return (x >= hi) ? T(0) : ((x < lo) ? T(1) : T(1) - (x - lo) / (hi - lo));
}

// surface_conductance + transpiration, :1121-1258
template <typename T, bool DAY>
__device__ __forceinline__ T transpiration_exact(const ClassPar<T>& p, T pa, T t, T vpd, T lai,
                                                 T fpar, T rad_canopy, T tmin, T r_corr,
                                                 T lhv, T rh, T fwet, T tiny = K<T>::tiny) {
#pragma clang fp contract(off)
    T s = svp_slope_exact(t);
    T rho = rho_exact(t, pa, rh);
    T gamma = gamma_exact(pa, t);
    T r_r = rr_exact(rho, t);
    T g_surf = T(0);
    if (DAY) {
        T gs = p.csl * ramp_up_exact(tmin - K<T>::t0, p.tmin_close, p.tmin_open) *
               ramp_down_exact(vpd, p.vpd_open, p.vpd_close);    // :1148-1150
        g_surf = gs / r_corr;                                     // :1237
    }
    T g_cut = p.g_cut / r_corr;                                   // :1238
    T gl_sh = p.gl_sh * lai * (T(1) - fwet);                      // :1242
    T g = (gl_sh * (g_surf + g_cut)) / (gl_sh + g_surf + g_cut);  // :1243
    T g_canopy = ((lai > T(0)) && ((T(1) - fwet) > T(0))) ? g : tiny;  // :1245
    T r_dry = (T(1) / p.gl_sh * r_r) / (T(1) / p.gl_sh + r_r);    // :1248
    rad_canopy = (rad_canopy < T(0)) ? T(0) : rad_canopy;         // :1251
    T tr = (T(1) - fwet) * ((s * rad_canopy) + (rho * K<T>::cp * fpar * (vpd / r_dry)));
    tr = tr / (s + gamma * (T(1) + (T(1) / g_canopy) / r_dry));   // :1255
    return (g_canopy <= tiny) ? T(0) : tr / lhv;                  // :1258
}

// potential ET of one period from the reference's own component functions
// (README.md:404-424): canopy + (sat + unsat, each clamped at 0 as in :858-861,
// no moisture constraint) / lhv + potential_transpiration / lhv
template <typename T>
__device__ __forceinline__ T pet_exact(const ClassPar<T>& p, const PixelIn<T>& x, T t, T vpd,
                                       T lw, T sw, T rad_soil, T canopy, T r_corr, T lhv, T rh,
                                       T fw) {
#pragma clang fp contract(off)
    T sat, unsat;
    pot_soil_exact(p, x.pa, t, vpd, x.fpar, rad_soil, r_corr, rh, fw, sat, unsat);
    T e = (sat < T(0)) ? T(0) : sat;
    e = e + ((unsat < T(0)) ? T(0) : unsat);
    T ptr = pot_transpiration_exact(lw, sw, x.alb, x.pa, t, x.fpar, fw, T(kPriestleyTaylorAlpha));
    return (canopy + e / lhv) + ptr / lhv;
}

// SERIAL: scheduling fences between the components, so that the code is laid out (and its
// registers allocated) one component at a time -- for the slow branch of the production
// kernels (domain guard below), where this function shares the register file with the
// state of a pipeline that wants two waves per SIMD. Same operations, same results.
template <typename T, bool PET = false, bool SERIAL = false>
__device__ __forceinline__ PixelOut<T> et_pixel_exact(const PixelIn<T>& x, const ClassPar<T>& p) {
#pragma clang fp contract(off)
    PixelOut<T> o;
    T rs_d, rs_n;
    auto fence = [] { if (SERIAL) __builtin_amdgcn_sched_barrier(0); };
    rad_soil_exact(x, p, rs_d, rs_n);                             // :737
    fence();
    {   // day, :751-792
        T rad_net = x.sw_d * (T(1) - x.alb) + x.lw_d;
        T rad_c = x.fpar * rad_net;
        T rh = rh_exact(x.t_d, x.vpd_d);
        T fw = fwet_exact(rh);
        T lhv = lhv_exact(x.t_d);
        T rc = rcorr_exact(x.pa, x.t_d);
        fence();
        o.canopy_d = wet_canopy_exact(p, x.pa, x.t_d, x.vpd_d, x.lai, x.fpar, rad_c, lhv, rh, fw);
        fence();
        o.soil_d = soil_exact(p, x.pa, x.t_d, x.vpd_d, x.fpar, rs_d, rc, lhv, rh, fw);
        fence();
        o.trans_d = transpiration_exact<T, true>(p, x.pa, x.t_d, x.vpd_d, x.lai, x.fpar, rad_c,
                                                 x.tmin, rc, lhv, rh, fw);
        fence();
        if (PET) o.pet_d = pet_exact(p, x, x.t_d, x.vpd_d, x.lw_d, x.sw_d, rs_d, o.canopy_d, rc, lhv, rh, fw);
    }
    fence();
    {   // night
        T rad_net = x.sw_n * (T(1) - x.alb) + x.lw_n;
        T rad_c = x.fpar * rad_net;
        T rh = rh_exact(x.t_n, x.vpd_n);
        T fw = fwet_exact(rh);
        T lhv = lhv_exact(x.t_n);
        T rc = rcorr_exact(x.pa, x.t_n);
        fence();
        o.canopy_n = wet_canopy_exact(p, x.pa, x.t_n, x.vpd_n, x.lai, x.fpar, rad_c, lhv, rh, fw);
        fence();
        o.soil_n = soil_exact(p, x.pa, x.t_n, x.vpd_n, x.fpar, rs_n, rc, lhv, rh, fw);
        fence();
        o.trans_n = transpiration_exact<T, false>(p, x.pa, x.t_n, x.vpd_n, x.lai, x.fpar, rad_c,
                                                  x.tmin, rc, lhv, rh, fw);
        fence();
        if (PET) o.pet_n = pet_exact(p, x, x.t_n, x.vpd_n, x.lw_n, x.sw_n, rs_n, o.canopy_n, rc, lhv, rh, fw);
    }
    return o;
}

// ============================================================ static (N2)
// MOD16._evapotranspiration, mod16/__init__.py:195-382: the vectorised
// calibration path. A different algorithm from the instance path (W m-2, no
// upper RH clamp, tmin_open in the G condition, no negative clamps, and a
// whole-array branch on any(g_surf > 0)), kept in its own operation order.

// day-time surface conductance before the division by r_corr, :325-327
template <typename T>
__device__ __forceinline__ T gsurf_static(const ClassPar<T>& p, T tmin, T vpd) {
#pragma clang fp contract(off)
    return p.csl * ramp_up_exact(tmin - K<T>::t0, p.tmin_close, p.tmin_open) *
           ramp_down_exact(vpd, p.vpd_open, p.vpd_close);
}

template <typename T, bool DAY>
__device__ __forceinline__ T period_static(const PixelIn<T>& x, const ClassPar<T>& p, T t, T vpd,
                                           T sw, T lw, T rad_soil, bool has_rc, T rc_in,
                                           bool any_gs, T tiny = K<T>::tiny) {
#pragma clang fp contract(off)
    T rad_net = sw * (T(1) - x.alb) + lw;                          // :272-274
    T rad_c = x.fpar * rad_net;
    T sv = svp_exact(t);
    T rh = (sv - vpd) / sv;                                        // :280-281
    rh = (rh < T(0)) ? T(0) : rh;
    T fw = (rh < T(0.7)) ? T(0) : ExactMath<T>::pow(rh, T(4));
    T d = (T(239.0) + t) - K<T>::t0;
    T s = (T(17.38 * 239.0) * sv) / (d * d);
    T lhv = lhv_exact(t);
    T gamma = gamma_exact(x.pa, t);
    T rc = has_rc ? rc_in : rcorr_exact(x.pa, t);                  // :291-294
    T rho = rho_exact(t, x.pa, rh);
    T r_r = rr_exact(rho, t);
    T r_h = T(1) / (p.gl_sh * x.lai * fw);                         // :305-311
    T r_e = T(1) / (p.gl_wv * x.lai * fw);
    T r_a = (r_h * r_r) / (r_h + r_r);
    T e = (fw * ((s * rad_c) + (rho * K<T>::cp * x.fpar * vpd * T(1) / r_a))) /
          (s + ((x.pa * K<T>::cp * r_e) * T(1) / (lhv * K<T>::eps * r_a)));
    T e_canopy = (x.lai * fw <= tiny) ? T(0) : e;                  // :320
    T g_surf = DAY ? gsurf_static(p, x.tmin, vpd) : T(0);
    g_surf = g_surf / rc;                                          // :328
    T g_cut = p.g_cut / rc;
    T gl = p.gl_sh * x.lai * (T(1) - fw);
    T g = (gl * (g_surf + g_cut)) / (gl + g_surf + g_cut);
    T g_can = ((x.lai > T(0)) && ((T(1) - fw) > T(0))) ? g : tiny;
    T r_dry = (T(1) / p.gl_sh * r_r) / (T(1) / p.gl_sh + r_r);
    T tr = T(0);                                                   // :343-348
    if (any_gs) {
        tr = (T(1) - fw) * ((s * rad_c) + (rho * K<T>::cp * x.fpar * (vpd / r_dry)));
        tr = tr / (s + gamma * (T(1) + (T(1) / g_can) / r_dry));
    }
    T r_tot = (vpd <= p.vpd_open) ? p.rbl_min
              : ((vpd >= p.vpd_close) ? p.rbl_max
                 : p.rbl_max - ((p.rbl_max - p.rbl_min) * (p.vpd_close - vpd)) /
                                   (p.vpd_close - p.vpd_open));
    r_tot = r_tot / rc;
    T r_as = (r_tot * r_r) / (r_tot + r_r);
    T numer = (s * rad_soil) + (rho * K<T>::cp * (T(1) - x.fpar) * (vpd / r_as));
    T denom = s + gamma * (r_tot / r_as);
    T sat = (numer * fw) / denom;
    T unsat = (numer * (T(1) - fw)) / denom;
    T e_soil = sat + unsat * ExactMath<T>::pow(rh, vpd / p.beta);  // :376
    return (tr + e_canopy) + e_soil;                               // :380
}

template <typename T>
__device__ __forceinline__ void et_static_pixel(const PixelIn<T>& x, const ClassPar<T>& p,
                                                bool has_rc, T rc_d, T rc_n, bool any_gs_day,
                                                T& day, T& night, T tiny = K<T>::tiny) {
#pragma clang fp contract(off)
    T a_d = x.sw_d * (T(1) - x.alb) + x.lw_d;                      // :225-226
    T a_n = x.lw_n;
    bool cond = (x.t_ann < T(273.15 + 25.0)) && (x.t_ann > (K<T>::t0 + p.tmin_open)) &&
                ((x.t_d - x.t_n) >= T(5));                         // :230-234 (tmin_open, strict)
    T g_d = cond ? (T(4.73) * (x.t_d - K<T>::t0)) - T(20.87) : T(0);
    g_d = (__builtin_fabs(g_d) > (T(0.39) * __builtin_fabs(a_d))) ? T(0.39) * a_d : g_d;
    T g_n = cond ? (T(4.73) * (x.t_n - K<T>::t0)) - T(20.87) : T(0);
    g_n = (__builtin_fabs(g_n) > (T(0.39) * __builtin_fabs(a_n))) ? T(0.39) * a_n : g_n;
    g_d = ((a_d - g_d < T(0)) && (a_d > T(0))) ? a_d : g_d;
    g_n = ((a_d > T(0)) && ((a_n - g_n) < (T(-0.5) * a_d))) ? a_n + (T(0.5) * a_d) : g_n;
    T rs_d = (T(1) - x.fpar) * (a_d - g_d);
    T rs_n = (T(1) - x.fpar) * (a_n - g_n);
    day = period_static<T, true>(x, p, x.t_d, x.vpd_d, x.sw_d, x.lw_d, rs_d, has_rc, rc_d, any_gs_day, tiny);
    // at night g_surf = 0 / r_corr, so any(g_surf > 0) is False: t = 0 (:343-348)
    night = period_static<T, false>(x, p, x.t_n, x.vpd_n, x.sw_n, x.lw_n, rs_n, has_rc, rc_n, false, tiny);
}

// ---- static path, strength-reduced and split for batching over parameter
// vectors (mod16_et_static_batch_*, MOD16_MATH_FAST): what does not depend on
// the parameters is prepared once per pixel (StaticPeriod / StaticPixel), the
// rest is evaluated per parameter vector with the arithmetic of the FAST
// forward run (reciprocal + Newton step, table exp / log, conductances instead
// of parallel resistances). Same predicates as period_static.
struct StaticPeriod {
    double vpd, rad_c, s, gamma, inv_rc, g_rr, rcfv, rh, fw, omw, logrh;
};
struct StaticPixel {
    StaticPeriod d, n;
    double fpar, omf, lai, tm, t_ann;
    double rs_d[2], rs_n[2];     // radiation received by the soil without / with soil heat flux
    bool base_cond, lai_pos;
};

__device__ __forceinline__ StaticPeriod static_period_prep(double t, double vpd, double sw, double lw,
                                                           double alb, double fpar, double pa,
                                                           const double* tb) {
    typedef FastMath<double> M;
    StaticPeriod c;
    c.vpd = vpd;
    c.rad_c = fpar * __builtin_fma(sw, 1.0 - alb, lw);                       // :272-275
    const double tc = t - 273.15;
    const double sv = __builtin_fma(1e3 * 0.6108, M::exp_tab((17.27 * tc) * M::rcp(tc + 237.3), tb), tc * 0.0);
    const double rsv = M::rcp(sv), avp = sv - vpd;
    double rh = avp * rsv;
    rh = __builtin_fma(__builtin_fma(-rh, sv, avp), rsv, rh);                // finished like a division
    rh = (rh < 0.0) ? 0.0 : rh;                                              // :280-281, no upper clamp
    const double rh2 = rh * rh;
    c.rh = rh;
    c.fw = (rh < 0.7) ? 0.0 : rh2 * rh2;
    c.omw = 1.0 - c.fw;
    const double rd = M::rcp((239.0 + t) - 273.15);
    c.s = ((17.38 * 239.0) * sv) * (rd * rd);
    const double lhv = (2.501 - 0.002361 * tc) * 1e6;
    c.gamma = (1013.0 * pa) * M::rcp(lhv * 0.622);
    c.inv_rc = (pa * (1.0 / 101300.0)) * M::pow_m1p75(t * (1.0 / 293.15));   // 1 / r_corr
    const double nn = 0.348444 * (pa * 0.01) - (rh * 100.0) * (0.00252 * tc - 0.020582);
    const double rho_cp = 1013.0 * (nn * M::rcp(t));
    c.g_rr = (K<double>::sigma4 * ((t * t) * t)) * M::rcp(rho_cp);           // 1 / r_r
    c.rcfv = rho_cp * vpd;
    c.logrh = M::log_tab(rh, tb);
    return c;
}

__device__ __forceinline__ StaticPixel static_pixel_prep(const PixelIn<double>& x, const double* tb) {
    StaticPixel c;
    c.d = static_period_prep(x.t_d, x.vpd_d, x.sw_d, x.lw_d, x.alb, x.fpar, x.pa, tb);
    c.n = static_period_prep(x.t_n, x.vpd_n, x.sw_n, x.lw_n, x.alb, x.fpar, x.pa, tb);
    c.fpar = x.fpar; c.omf = 1.0 - x.fpar; c.lai = x.lai; c.lai_pos = x.lai > 0.0;
    c.tm = x.tmin - 273.15; c.t_ann = x.t_ann;
    // soil heat flux, :225-252: the condition's parameter-dependent part is applied per draw
    c.base_cond = (x.t_ann < 273.15 + 25.0) && ((x.t_d - x.t_n) >= 5.0);
    const double a_d = __builtin_fma(x.sw_d, 1.0 - x.alb, x.lw_d), a_n = x.lw_n;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double g_d = k ? (4.73 * (x.t_d - 273.15)) - 20.87 : 0.0;
        g_d = (__builtin_fabs(g_d) > (0.39 * __builtin_fabs(a_d))) ? 0.39 * a_d : g_d;
        double g_n = k ? (4.73 * (x.t_n - 273.15)) - 20.87 : 0.0;
        g_n = (__builtin_fabs(g_n) > (0.39 * __builtin_fabs(a_n))) ? 0.39 * a_n : g_n;
        g_d = ((a_d - g_d < 0.0) && (a_d > 0.0)) ? a_d : g_d;
        g_n = ((a_d > 0.0) && ((a_n - g_n) < (-0.5 * a_d))) ? a_n + (0.5 * a_d) : g_n;
        c.rs_d[k] = c.omf * (a_d - g_d);
        c.rs_n[k] = c.omf * (a_n - g_n);
    }
    return c;
}

// per-draw terms shared by both periods
struct StaticDraw {
    double m_tmin, inv_dvpd, rbl_slope, inv_beta;
    bool cond;
};
__device__ __forceinline__ StaticDraw static_draw_prep(const StaticPixel& c, const ClassPar<double>& p) {
    typedef FastMath<double> M;
    StaticDraw d;
    d.m_tmin = (c.tm >= p.tmin_open) ? 1.0
               : ((c.tm < p.tmin_close) ? 0.0 : (c.tm - p.tmin_close) * M::rcp(p.tmin_open - p.tmin_close));
    d.inv_dvpd = M::rcp(p.vpd_close - p.vpd_open);
    d.rbl_slope = (p.rbl_max - p.rbl_min) * d.inv_dvpd;
    d.inv_beta = M::rcp(p.beta);
    d.cond = c.base_cond && (c.t_ann > (273.15 + p.tmin_open));              // :230-234 (tmin_open, strict)
    return d;
}
// day-time surface conductance / r_corr (the quantity behind any(g_surf > 0))
__device__ __forceinline__ double static_gsurf(const StaticPeriod& c, const StaticDraw& d,
                                               const ClassPar<double>& p) {
    const double m_vpd = (c.vpd >= p.vpd_close) ? 0.0
                         : ((c.vpd < p.vpd_open) ? 1.0 : 1.0 - (c.vpd - p.vpd_open) * d.inv_dvpd);
    return ((p.csl * d.m_tmin) * m_vpd) * c.inv_rc;
}

template <bool DAY>
__device__ __forceinline__ double static_period_eval(const StaticPixel& px, const StaticPeriod& c,
                                                     const StaticDraw& d, const ClassPar<double>& p,
                                                     double rad_soil, bool any_gs, const double* tb) {
    typedef FastMath<double> M;
    const double tiny = 1e-7;
    const double sx = c.s * c.rad_c, rf = c.rcfv * px.fpar;
    // wet canopy, :305-320
    const double lf = px.lai * c.fw;
    const double g_e = p.gl_wv * lf, g_a = p.gl_sh * lf + c.g_rr;
    double e_can = ((c.fw * __builtin_fma(rf, g_a, sx)) * g_e) * M::rcp(__builtin_fma(c.s, g_e, c.gamma * g_a));
    e_can = (lf <= tiny) ? 0.0 : e_can;
    // transpiration, :325-348
    double tr = 0.0;
    if (any_gs) {
        const double gsc = (DAY ? static_gsurf(c, d, p) : 0.0) + p.g_cut * c.inv_rc;
        const double gl = (p.gl_sh * px.lai) * c.omw;
        const bool open = px.lai_pos && (c.omw > 0.0);
        const double p1 = open ? gl * gsc : tiny, s1 = open ? gl + gsc : 1.0;   // g_canopy = p1 / s1
        const double g_d = p.gl_sh + c.g_rr;                                    // 1 / r_dry
        const double num = (c.omw * __builtin_fma(rf, g_d, sx)) * p1;
        const double den = __builtin_fma(c.s, p1, c.gamma * __builtin_fma(g_d, s1, p1));
        tr = num * M::rcp(den);
    }
    // bare soil, :350-376
    const double r0 = (c.vpd <= p.vpd_open) ? p.rbl_min
                      : ((c.vpd >= p.vpd_close) ? p.rbl_max
                         : __builtin_fma(-(p.vpd_close - c.vpd), d.rbl_slope, p.rbl_max));
    const double r_tot = r0 * c.inv_rc;
    const double w = __builtin_fma(r_tot, c.g_rr, 1.0);
    const double num = __builtin_fma(c.s * rad_soil, r_tot, (c.rcfv * px.omf) * w);
    const double q = num * M::rcp(r_tot * __builtin_fma(c.gamma, w, c.s));
    const double yc = __builtin_fmin(c.vpd * d.inv_beta, 1e300);
    const double pw = M::exp_tab4(__builtin_fmax(yc * c.logrh, -746.0), tb);  // rh ** (vpd / beta), :376
    const double e_soil = q * __builtin_fma(c.omw, pw, c.fw);
    return (tr + e_can) + e_soil;                                            // :380
}

// ====================================================================== fast
// Where the constants of the fast pixel function live. The production loop fills the scalar
// register file (v_fma_f64 takes no literal on gfx9: every float64 constant that is not an inline
// one occupies a scalar pair), and what does not fit is rebuilt from two s_mov_b32 at every use --
// 57 of them per iteration of the float64 totals loop, plus 15 v_readlane / v_writelane of scalar
// spills (round 4 listing). KPin<LEVEL> holds the constants that are used most in VECTOR register
// pairs through the loop instead (the totals instances have the registers to spare at two waves per
// SIMD; the other forms and the plain kernels do not: KLit, literals as before). Same values, same
// operations: results are bit-identical either way.
struct KLit {
    static __device__ __forceinline__ double per_period(double c) { return c; }
    static __device__ __forceinline__ double per_pixel(double c) { return c; }
};
template <int LEVEL> struct KPin {
    static __device__ __forceinline__ double per_period(double c) { return LEVEL >= 1 ? in_vgpr(c) : c; }
    static __device__ __forceinline__ double per_pixel(double c) { return LEVEL >= 2 ? in_vgpr(c) : c; }
};
// Round 6, fewer float64 instructions per pixel (three candidates, A/B on one device:
// profiles/r06_ab_f64_candidates.txt; 1018 -> 987 vector instructions per pair of pixels in the
// totals loop). MOD16_F64_CAND (measurement builds) keeps the first n of the two that stayed:
//   1  LAI = 0 is not replaced by `tiny` (:935): nothing divides by LAI in conductances, and such a
//      pixel's canopy is the 0 of :961 whatever its quotient (as fwet = 0 since round 5)
//   2  the two VPD ramps (:527-531 r_tot, :1148-1150 m_vpd) from ONE clamp of (vpd - vpd_open) /
//      (vpd_close - vpd_open) to [0, 1] (v_max_f64 + v_min_f64) instead of two compares and two
//      64-bit selects each
// (the third, the Tetens exponent as one fma behind the shared reciprocal, gained nothing and cost
// accuracy: see period_fast)
#ifndef MOD16_F64_CAND
#define MOD16_F64_CAND 2
#endif
// Quantities that do not depend on the period (day / night).
template <typename T> struct PixelShared {
    T oma;        // 1 - albedo
    T omf;        // 1 - fpar
    T p_rel;      // pressure 293.15^1.75 / 101300  (1 / r_corr = p_rel t^-1.75)
    T k_p;        // Cp * pressure / eps  (= gamma * lhv)
    T p_mbar_k;   // Cp 0.348444 * pressure / 100  (round 5: the air-density numerator N carries Cp)
    T l_wet;      // lai with 0 -> tiny (wet canopy, :935)
    T glsh_l, glwv_l;   // gl_sh * l_wet, gl_wv * l_wet
    T glsh_lai;   // gl_sh * lai (transpiration, :1242)
    T m_tmin;     // Tmin ramp (day only)
    T drbl;       // rbl_max - rbl_min
    bool lai_pos, lai_tiny;
};

// PET: also the potential ET of the period (reference README.md:404-424):
// wet-canopy evaporation + saturated-soil evaporation + unsaturated-soil
// evaporation without the soil-moisture constraint + Priestley-Taylor
// potential transpiration (MOD16.potential_transpiration, :546-602), as a mass
// flux like the other outputs.
// RAW (the raw-driver forms, round 5): `vpd` arrives as the humidity quotient's numerator q p
// (specific humidity x surface pressure) and `dav` as its denominator 0.622 + 0.379 q; the period
// forms MOD16.vpd (mod16/__init__.py:604-644) itself, with the exponential it SHARES with the
// saturation pressure: 610.7 exp(17.38 tc / (239 + tc)) = 610.7 exp(17.27 tc / (237.3 + tc)) exp(delta),
// delta = tc (0.11 tc - 3.256) / ((tc + 239)(tc + 237.3)) -- |delta| < 0.043 on 190 K .. 360 K (the
// raw forms' temperature domain, raw_guard_value), so exp(delta) is six fused multiply-adds and
// the second table exponential of the period, its range reduction and its reciprocal are gone.
template <typename T, bool DAY, bool PET = false, typename KP = KLit, bool RAW = false>
__device__ __forceinline__ void period_fast(const PixelIn<T>& x, const ClassPar<T>& p,
                                            const PixelShared<T>& sh, const T* tb, T t, T vpd,
                                            T rad_net, T rad_soil, T& canopy, T& soil,
                                            T& trans, T* pet = nullptr, T dav = T(1)) {
    // No implicit contraction: every fma of this function is written out, so that all
    // kernels instantiated from it round alike (hipcc contracts a * b + c by context).
#pragma clang fp contract(off)
    typedef FastMath<T> M;
    const T tiny = K<T>::tiny;
    // -- humidity, :646-673 and :763-764
    T tc = t - K<T>::t0;
    // the two reciprocals of the temperature terms -- 1 / (tc + 237.3) for esat and
    // 1 / (tc + 239) for the slope of the curve -- from ONE v_rcp_f64 of their product
    // (a quarter-rate instruction and its Newton step against two multiplications)
    T d_es = tc + KP::per_period(237.3);
    T ta = tc + KP::per_period(239.0);                                          // (239 + T) - 273.15 to 1 ulp, :1395
    T dd = d_es * ta;
    T r_both, r_av = T(0);
    if constexpr (RAW) {
        // ... and the humidity quotient's reciprocal from the same v_rcp_f64 (0.243 < dav < 1.001
        // inside the raw forms' domain)
        T r3 = M::rcp(dd * dav);
        r_both = r3 * dav;
        r_av = r3 * dd;
    } else {
        r_both = M::rcp(dd);
    }
    T r_es = r_both * ta, rta = r_both * d_es;
    // (a NaN temperature stays NaN through the table exp: rint, the fmas and the table product
    // all propagate it; an infinite one gives inf * 0 in r_es)
    // (round 6 tried 17.27 - 17.27 * 237.3 r_es, one fma: no time gained, and the exponent's ABSOLUTE
    // error near 0 C -- 2e-15 instead of 2e-16 |x| -- showed as 2e-9 in the worst pixels of the grid)
    T e_es = M::exp_tab5s((KP::per_period(17.27) * tc) * r_es, tb);
    T esat = T(1e3 * 0.6108) * e_es;
    if constexpr (RAW) {
        // 17.38 * 237.3 - 17.27 * 239 and 17.38 - 17.27, exact differences of the float64 constants
        T delta = (tc * M::fma_kk(tc, 0x1.c28f5c28f5c00p-4, -0x1.a0c49ba5e34b1p+1)) * r_both;
        T q = M::fma_kk(delta, 1.0 / 720.0, 1.0 / 120.0);
        q = __builtin_fma(q, delta, 1.0 / 24.0);
        q = __builtin_fma(q, delta, 1.0 / 6.0);
        q = __builtin_fma(q, delta, 0.5);
        q = __builtin_fma(q, delta, 1.0);
        q = __builtin_fma(q, delta, 1.0);
        T sv = (T(610.7) * e_es) * q;                              // :640-642
        vpd = __builtin_fma(-vpd, r_av, sv);                       // sv - q p / (0.622 + 0.379 q)
        if constexpr (!DAY) vpd = (vpd < T(0)) ? T(0) : vpd;       // calibration.py:401
    }
    T avp = esat - vpd;
    // rh drives the thresholds (rh < 0.7, 1 - fwet > 0), so this one quotient is
    // finished like an IEEE division (residual correction): x / x = 1 exactly
    T resat = M::rcp(esat);
    T rh = avp * resat;
    rh = __builtin_fma(__builtin_fma(-rh, esat, avp), resat, rh);
    // flat selects, innermost first: hipcc turns a NESTED conditional into exec-masked
    // branches (s_and_saveexec / s_cbranch / v_mov / s_or: ~8 instructions a level)
    rh = (rh > T(1)) ? T(1) : rh;
    rh = (avp < T(0)) ? T(0) : rh;                                 // :670-673
    T rh2 = rh * rh;
    const bool dry = rh < T(0.7);                                  // :764 (NaN compares false)
    T fwet = dry ? T(0) : rh2 * rh2;
    T omw = T(1) - fwet;
    // -- slope of the SVP curve (:1395-1397), latent heat (:121)
    T s = (KP::per_period(17.38 * 239.0) * esat) * (rta * rta);
    T lhv = M::fma_kk(tc, T(-0.002361e6), T(2.501e6));         // (2.501 - 0.002361 tc) 1e6, :121
    T slhv = s * lhv;
    // -- 1 / r_corr = (P / 101300) (T / 293.15)^-1.75, :771; 1 / T comes with it
    T rt;
    T inv_rcorr = sh.p_rel * M::pow_m1p75_rcp(t, rt);
    // -- air density (:408-412) and radiative conductance 1/r_r (:947):
    //    rho = N / T, 1/r_r = 4 sigma T^4 / (Cp N)
    // p_mbar_k - (rh 100)(0.00252 tc - 0.020582); the contraction is written out so that
    // every kernel built from this function rounds alike
    // (N carries the factor Cp: rho Cp = N / T and 1/r_r = 4 sigma T^4 / N without a product by Cp)
    T nn = __builtin_fma(-rh, M::fma_kk(tc, T(0.252 * 1013.0), T(-2.0582 * 1013.0)), sh.p_mbar_k);
    T rho_cp = nn * rt;
    T t2 = t * t;
    T g_rr = (K<T>::sigma4 * (t2 * t2)) * M::rcp(nn);
    T rcfv = rho_cp * vpd;             // rho Cp vpd
#if MOD16_F64_CAND >= 2
    T vramp = M::clamp01((vpd - p.vpd_open) * p.inv_dvpd);
#endif

    // -- wet canopy, :866-961 in conductances:
    //    1/r_a = g_h + 1/r_r ; evap = numer g_e / ((s lhv) g_e + k_p / r_a)
    {
        // fwet is 0 exactly when rh < 0.7 (else rh^4 >= 0.24, or NaN), so the three
        // tests on it -- fwet == 0 (:934), fw <= tiny (:961) -- are that one mask; the reference
        // replaces a zero by `tiny` only to keep 1 / (gl fwet) finite (:934), and the dry pixel's
        // result is the 0 of :961 whatever its quotient -- in conductances nothing divides by
        // fwet, so the replacement itself is not needed (two selects per period less, round 5)
        T fw = fwet;
        T g_e = sh.glwv_l * fw;
        T g_a = __builtin_fma(sh.glsh_l, fw, g_rr);                // g_h + 1/r_r
        T numer = fw * __builtin_fma(rcfv * x.fpar, g_a, s * (x.fpar * rad_net));
        T den = __builtin_fma(slhv, g_e, sh.k_p * g_a);
        T evap = (numer * g_e) * M::rcp_quotient(den);
        // numer < 0 -> 0 (:959), then fw <= tiny or lai <= tiny -> 0 (:961): one select
        canopy = ((numer < T(0)) | dry | sh.lai_tiny) ? T(0) : evap;
    }
    // -- bare soil, :449-544 and :795-864
    {
#if MOD16_F64_CAND >= 2
        // rbl_min + (rbl_max - rbl_min) clamp01((vpd - vpd_open) / (vpd_close - vpd_open)), :527-531:
        // exactly rbl_min up to vpd_open, rbl_max to an ulp from vpd_close on. A NaN vpd leaves the
        // clamp as 0 (v_max_f64 passes a NaN over) -- and the period as NaN through rho Cp vpd.
        T r0 = __builtin_fma(vramp, sh.drbl, p.rbl_min);
#else
        T r0 = __builtin_fma(-(p.vpd_close - vpd), p.rbl_slope, p.rbl_max);   // :527-531
        r0 = (vpd >= p.vpd_close) ? p.rbl_max : r0;
        r0 = (vpd <= p.vpd_open) ? p.rbl_min : r0;
#endif
        T r_tot = r0 * inv_rcorr;                                  // :533
        T w = __builtin_fma(r_tot, g_rr, T(1));                    // r_tot / r_as
        T num = __builtin_fma((s * rad_soil), r_tot, (rcfv * sh.omf) * w);
        T den = r_tot * __builtin_fma(sh.k_p, w, slhv);
        T q = num * M::rcp_quotient(den);                                   // numer/denom/lhv
        T pw = M::pow01_tab1(rh, vpd * p.inv_beta, tb);            // :861
        // sat = q fwet and unsat = q (1 - fwet) with 0 <= fwet <= 1: both
        // clamps of :858-861 fire exactly when q < 0 (NaN falls through)
        T e = q * __builtin_fma(omw, pw, fwet);
        soil = (q < T(0)) ? T(0) : e;
        if (PET) {
            // sat + unsat without the rh^(vpd/beta) factor
            // two products as in :541-543, so that inf * 0 is NaN as it is there
            T pot_soil = (q < T(0)) ? T(0) : __builtin_fma(q, fwet, q * omw);
            // alpha s A_c (1 - fwet) / (s + gamma) / lhv, gamma lhv = k_p
            T pot_tr = (T(kPriestleyTaylorAlpha) * (s * (x.fpar * rad_net)) * omw) *
                       M::rcp(__builtin_fma(s, lhv, sh.k_p));
            *pet = (canopy + pot_soil) + pot_tr;
        }
    }
    // -- transpiration, :1152-1258, with g_canopy = P1 / S1 kept as a ratio
    {
        T g_s = T(0);
        if (DAY) {
#if MOD16_F64_CAND >= 2
            T m_vpd = T(1) - vramp;                                // :1148-1150 (exactly 1 / 0 outside the ramp)
#else
            T m_vpd = __builtin_fma(-(vpd - p.vpd_open), p.inv_dvpd, T(1));
            m_vpd = (vpd < p.vpd_open) ? T(1) : m_vpd;
            m_vpd = (vpd >= p.vpd_close) ? T(0) : m_vpd;
#endif
            g_s = ((p.csl * sh.m_tmin) * m_vpd) * inv_rcorr;       // :1237
        }
        T gsc = __builtin_fma(p.g_cut, inv_rcorr, g_s);            // :1238
        T g_bl = sh.glsh_lai * omw;                                // :1242
        T p1 = g_bl * gsc;
        T s1 = __builtin_fma(sh.glsh_lai, omw, gsc);               // g_bl + gsc
        bool open = sh.lai_pos & (omw > T(0));                     // :1245
        // g_canopy <= tiny  <=>  P1 <= tiny S1 (S1 > 0); NaN compares false
        bool shut = !open | (p1 <= tiny * s1);                     // :1258
        T g_d = p.gl_sh + g_rr;                                    // 1 / r_a_dry, :1248
        T rad_c = x.fpar * rad_net;
        rad_c = (rad_c < T(0)) ? T(0) : rad_c;                     // :1251
        T num = (omw * __builtin_fma(rcfv * x.fpar, g_d, s * rad_c)) * p1;
        T den = __builtin_fma(slhv, p1, sh.k_p * __builtin_fma(g_d, s1, p1));
        T tr = num * M::rcp_quotient(den);
        trans = shut ? T(0) : tr;
    }
}

// RAW: x.vpd_d / x.vpd_n hold q p of the period and dav_d / dav_n its 0.622 + 0.379 q (see period_fast)
template <typename T, bool PET = false, typename KP = KLit, bool RAW = false>
__device__ __forceinline__ PixelOut<T> et_pixel_fast(const PixelIn<T>& x, const ClassPar<T>& p,
                                                     const T* tb, T dav_d = T(1), T dav_n = T(1)) {
#pragma clang fp contract(off)
    PixelOut<T> o;
#ifdef MOD16_TRIVIAL_BODY   // measurement aid (-DMOD16_TRIVIAL_BODY): memory pattern only, no arithmetic
    o.canopy_d = x.lw_d + x.sw_d + x.alb + x.t_d + x.t_ann + x.vpd_d + x.pa;
    o.soil_d = p.beta; o.trans_d = x.fpar;
    o.canopy_n = x.lw_n + x.sw_n + x.t_n + x.tmin + x.vpd_n + x.lai;
    o.soil_n = p.csl; o.trans_n = p.gl_sh;
    return o;
#endif
    PixelShared<T> sh;
    sh.oma = T(1) - x.alb;
    sh.omf = T(1) - x.fpar;
    // -- radiation received by the soil, :963-1119 (predicates verbatim)
    T a_d = __builtin_fma(x.sw_d, sh.oma, x.lw_d);
    T a_n = x.lw_n;
    // `&` and `|`, not `&&` and `||`: the short-circuit forms make hipcc evaluate the
    // right-hand comparison inside an exec-masked branch (saveexec / cbranch / restore)
    const T k473 = KP::per_pixel(4.73), k2087 = KP::per_pixel(-20.87), k039 = KP::per_pixel(0.39);
    const T t_ann_max = KP::per_pixel(273.15 + 25.0), dt_min = KP::per_pixel(5.0);
    bool cond = (x.t_ann < t_ann_max) & (x.t_ann >= (K<T>::t0 + p.tmin_close)) & ((x.t_d - x.t_n) >= dt_min);
    // (0.39 |A| = |0.39 A| exactly: one product serves the test and the cap, :1112)
    const T cap_d = k039 * a_d, cap_n = k039 * a_n;
    T g_d = cond ? __builtin_fma(k473, x.t_d - K<T>::t0, k2087) : T(0);
    g_d = (__builtin_fabs(g_d) > __builtin_fabs(cap_d)) ? cap_d : g_d;
    T g_n = cond ? __builtin_fma(k473, x.t_n - K<T>::t0, k2087) : T(0);
    g_n = (__builtin_fabs(g_n) > __builtin_fabs(cap_n)) ? cap_n : g_n;
    g_d = ((a_d - g_d < T(0)) & (a_d > T(0))) ? a_d : g_d;
    g_n = ((a_d > T(0)) & ((a_n - g_n) < (T(-0.5) * a_d))) ? __builtin_fma(T(0.5), a_d, a_n) : g_n;
    T rs_d = sh.omf * (a_d - g_d);
    T rs_n = sh.omf * (a_n - g_n);
    // -- period-independent terms
    sh.p_rel = x.pa * KP::per_pixel(0.2050207779207528);        // 293.15^1.75 / 101300
    sh.k_p = x.pa * KP::per_pixel(1013.0 / 0.622);
    sh.p_mbar_k = x.pa * KP::per_pixel(1013.0 * 0.348444 / 100.0);
#if MOD16_F64_CAND >= 1
    sh.l_wet = x.lai;
    sh.lai_tiny = x.lai <= K<T>::tiny;                 // (:935 made a LAI of 0 `tiny`: one of them)
    sh.lai_pos = x.lai > T(0);
    sh.glsh_lai = p.gl_sh * x.lai;
    sh.glsh_l = sh.glsh_lai;
    sh.glwv_l = p.gl_wv * x.lai;
#else
    sh.l_wet = (x.lai == T(0)) ? K<T>::tiny : x.lai;
    sh.lai_tiny = sh.l_wet <= K<T>::tiny;
    sh.lai_pos = x.lai > T(0);
    sh.glsh_l = p.gl_sh * sh.l_wet;
    sh.glwv_l = p.gl_wv * sh.l_wet;
    sh.glsh_lai = p.gl_sh * x.lai;
#endif
    sh.drbl = p.rbl_max - p.rbl_min;
    T tm = x.tmin - K<T>::t0;
    sh.m_tmin = (tm - p.tmin_close) * p.inv_dtmin;
    sh.m_tmin = (tm < p.tmin_close) ? T(0) : sh.m_tmin;
    sh.m_tmin = (tm >= p.tmin_open) ? T(1) : sh.m_tmin;
    period_fast<T, true, PET, KP, RAW>(x, p, sh, tb, x.t_d, x.vpd_d, a_d, rs_d, o.canopy_d, o.soil_d, o.trans_d,
                                       &o.pet_d, dav_d);
    T rn_n = __builtin_fma(x.sw_n, sh.oma, x.lw_n);
    period_fast<T, false, PET, KP, RAW>(x, p, sh, tb, x.t_n, x.vpd_n, rn_n, rs_n, o.canopy_n, o.soil_n, o.trans_n,
                                        &o.pet_n, dav_n);
    return o;
}

// =============================================================== domain guard
// The strength-reduced arithmetic above is the reference's arithmetic rearranged, and a
// rearrangement is only the same function where nothing overflows, no reciprocal meets a
// zero and every sign is the physical one. Outside that domain -- a fill value left in a
// temperature or the pressure, an infinity -- the reference (plain IEEE numpy,
// mod16/__init__.py:646-673, :121, :795-864, :1340-1367) still returns definite NaN / zero /
// inf / finite results, and the default arithmetic must return the same. The domain was
// mapped on the GPU with a ladder of 56 magnitudes in every driver (tests/fuzz_domain.py,
// profiles/r03_fuzz_domain_*.txt); et_pixel_fast differs from the reference exactly for
//   - an infinite lw_net_day / lw_net_night / sw_rad_day / sw_albedo / fpar / vpd (inf * 0 or
//     inf - inf where the reference has separate operations; vpd = -inf shows in the components
//     only), |lai| or |pressure| >= 1e200 (products overflow),
//   - a negative pressure (the merged clamps assume rho, r_corr > 0); pairs of special values
//     add: a pressure of 1e-300 next to a large value (underflow), sw_rad_night = 1e300 next to
//     a large LAI / pressure (overflow),
//   - a temperature above 1332.4 K (latent heat <= 0: the reference's soil evaporation turns
//     negative where the merged clamp gives 0) or within 2e-4 K of 35.85 K (the pole of the
//     Tetens formula: the table exp is not reduced for |x| > 2.3e7);
// NaN anywhere, zeros, negative or huge values elsewhere are inside the domain. The guard
// below is wider than that map (1e50, everything below 36 K, pressures below 1 Pa) and costs 14 vector
// instructions per pixel; a pixel it flags is computed again by et_pixel_exact -- the
// reference's own operation order -- in a branch that a wave enters only if one of its
// lanes holds such a pixel (stream kernels: mod16_stream.hpp; plain kernels: et_kernel).
// NaN compares false everywhere here: a NaN driver never sends a pixel to the slow branch.
__device__ __forceinline__ double max_abs(double a, double b) {   // maxNum(|a|, |b|): ignores a NaN
    double d;
    asm("v_max_f64 %0, |%1|, |%2|" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ double max_abs_signed(double a, double b) {   // maxNum(|a|, b)
    double d;
    asm("v_max_f64 %0, |%1|, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
constexpr double kGuardHuge = 1e50;           // |x| at or beyond this (or infinite): reference order. Single
                                              // drivers pass the fast form up to 1e100 and beyond, but its
                                              // products take up to four of them: 1e50 keeps every product of
                                              // guarded values inside float64 (tests/fuzz_domain.py, pairs)
constexpr double kGuardTmin = 36.0;           // K; 35.85 K is the pole of the Tetens formula
constexpr double kGuardTmax = 1332.0;         // K; latent heat of vaporization <= 0 from 1332.4 K

// One chain of v_max_f64 with |.| modifiers over everything that has a bound -- NaN operands
// are ignored, so a NaN driver never flags a pixel -- and ONE comparison at the end (one lane
// mask: the pixel function itself already keeps the scalar registers full). The temperatures
// enter as (t - 684) * (1e100 / 648): 36 < t < 1332 <=> |t - 684| < 648.
// (The guard's constants live in VECTOR registers: the pixel function fills the scalar register
// file with its own -- v_fma_f64 takes no literal on gfx9 -- and four more pairs there made the
// loop spill scalar registers; vector registers are to spare at two waves per SIMD.)
__device__ __forceinline__ double guard_temperature(double t) {
    constexpr double mid = 0.5 * (kGuardTmin + kGuardTmax), k = kGuardHuge / (0.5 * (kGuardTmax - kGuardTmin));
    return __builtin_fma(t, in_vgpr(k), in_vgpr(-mid * k));
}
// The guard's three constants, in vector registers (in_vgpr: the compiler hoists them in front of a loop)
struct GuardConsts {
    double t_scale, t_shift, huge;
};
__device__ __forceinline__ GuardConsts guard_consts() {
    constexpr double mid = 0.5 * (kGuardTmin + kGuardTmax), k = kGuardHuge / (0.5 * (kGuardTmax - kGuardTmin));
    return GuardConsts{in_vgpr(k), in_vgpr(-mid * k), in_vgpr(kGuardHuge)};
}
// The guard's magnitude: >= kGuardHuge exactly for the pixels outside the domain. The pipeline keeps
// it in a VECTOR register and tests it where it is used (a compare into VCC next to its select;
// the maximum over a thread's pixels once per iteration for the flag record): as a lane mask
// carried across the pixel function it cost the loop -- which fills the scalar register file on its
// own -- 46 instructions of scalar-register relief per iteration (28 s_mov re-materialisations, 18
// v_readlane / v_writelane), more than the guard's own 26 float64 operations (round 4).
__device__ __forceinline__ double fast_guard_value(const PixelIn<double>& x, const GuardConsts& gc) {
#ifdef MOD16_NO_GUARD      // measurement / mapping builds only (tests/fuzz_domain.py)
    return 0.0;
#else
    // lw_net_day, sw_rad_day and sw_albedo through the day's net radiation A = sw (1 - albedo) +
    // lw, which the pixel function needs anyway: an infinite one of them makes A infinite -- or
    // NaN (inf * 0, inf - inf), and then it is NaN in the reference as well and behaves like a
    // NaN driver, which is inside the domain. (Finite huge values of these three are inside it.)
    // sw_rad_night likewise through the night's net radiation (next to lw_net_night itself,
    // which also enters the soil's balance on its own).
    const double oma = 1.0 - x.alb;
    const double a_d = __builtin_fma(x.sw_d, oma, x.lw_d), rn_n = __builtin_fma(x.sw_n, oma, x.lw_n);
    double m = max_abs(a_d, x.lw_n);
    m = max_abs(m, rn_n);
    m = max_abs(m, x.fpar);
    m = max_abs(m, x.lai);
    m = max_abs(m, x.pa);
    m = max_abs(m, x.vpd_d);
    m = max_abs(m, x.vpd_n);
    m = max_abs(m, __builtin_fma(x.t_d, gc.t_scale, gc.t_shift));
    m = max_abs(m, __builtin_fma(x.t_n, gc.t_scale, gc.t_shift));
    // (a pressure below 1 Pa -- zero, negative, 1e-300 -- as well: its products with the other
    // drivers underflow where the reference's do not. Folded into the magnitude through its high
    // word: 0x7fe00000'xxxxxxxx is at least 2^1023; a NaN pressure compares false)
    return __hiloint2double((x.pa < 1.0) ? 0x7fe00000 : __double2hiint(m), __double2loint(m));
#endif
}
__device__ __forceinline__ bool fast_out_of_domain(const PixelIn<double>& x) {
    const GuardConsts gc = guard_consts();
    return fast_guard_value(x, gc) >= gc.huge;
}

// ================================================================ raw drivers
// the driver pre-processing the reference does in
// front of the forward run (mod16/calibration.py:380-423) folded into the
// pixel function: VPD from 10-m specific humidity and surface pressure
// (MOD16.vpd :604-644; the night value clamped at 0, calibration.py:401),
// air pressure from elevation (MOD16.air_pressure :414-447), fPAR in percent
// and LAI x 10 as the MODIS uint8 encodings (calibration.py:422-423; codes
// >= 249 are the MOD15 fill values -> NaN).
template <typename T> struct RawIn {
    T lw_d, lw_n, sw_d, sw_n, alb, t_d, t_n, t_ann, tmin, qv_d, qv_n, ps_d, ps_n, elev;
    unsigned fpar_pct, lai_x10;
};

template <typename T, bool SERIAL = false>
__device__ __forceinline__ PixelIn<T> raw_to_pixel_exact(const RawIn<T>& r) {
#pragma clang fp contract(off)
    PixelIn<T> x;
    auto fence = [] { if (SERIAL) __builtin_amdgcn_sched_barrier(0); };   // see et_pixel_exact
    x.lw_d = r.lw_d; x.lw_n = r.lw_n; x.sw_d = r.sw_d; x.sw_n = r.sw_n; x.alb = r.alb;
    x.t_d = r.t_d; x.t_n = r.t_n; x.t_ann = r.t_ann; x.tmin = r.tmin;
    x.vpd_d = vpd_exact(r.qv_d, r.ps_d, r.t_d);
    fence();
    T vn = vpd_exact(r.qv_n, r.ps_n, r.t_n);
    x.vpd_n = (vn < T(0)) ? T(0) : vn;
    fence();
    x.pa = air_pressure_exact(r.elev);
    fence();
    const T nan = __builtin_nan("");
    x.fpar = (r.fpar_pct >= 249u) ? nan : T(r.fpar_pct) / T(100);
    x.lai = (r.lai_x10 >= 249u) ? nan : T(r.lai_x10) / T(10);
    return x;
}

constexpr double kElevMid = 5000.0, kElevHalf = 7000.0;   // the fast form's elevations: -2000 m .. 12000 m
// -> the pixel of the fast forward run, with the humidity left in its raw terms: x.vpd_d / x.vpd_n
// hold q p (specific humidity x surface pressure) and dav_d / dav_n the quotient's denominator
// 0.622 + 0.379 q -- period_fast<..., RAW> forms MOD16.vpd from them next to the saturation
// pressure, whose exponential and reciprocal it shares (round 5).
__device__ __forceinline__ PixelIn<double> raw_to_pixel_fast(const RawIn<double>& r,
                                                             const double* tb, double& dav_d, double& dav_n) {
    typedef FastMath<double> M;
    PixelIn<double> x;
    x.lw_d = r.lw_d; x.lw_n = r.lw_n; x.sw_d = r.sw_d; x.sw_n = r.sw_n; x.alb = r.alb;
    x.t_d = r.t_d; x.t_n = r.t_n; x.t_ann = r.t_ann; x.tmin = r.tmin;
    x.vpd_d = r.qv_d * r.ps_d;                                     // :638-639
    x.vpd_n = r.qv_n * r.ps_n;
    dav_d = M::fma_kk(r.qv_d, 0.379, 0.622);
    dav_n = M::fma_kk(r.qv_n, 0.379, 0.622);
    // 101325 (1 - 0.0065 z / 288.15)^5.2559 as a polynomial of degree 9 in u = (z - 5000) / 7000
    // (coefficients in LDS behind the exp / log tables: 3.7e-14 relative on -2000 m .. 12000 m, the
    // interval the guard below admits; a log + an exp before -- 20 float64 instructions more). A NaN
    // elevation stays NaN.
    {
        const double* c = tb + M::kTabRaw;
        const double u = __builtin_fma(r.elev, 1.0 / kElevHalf, -kElevMid / kElevHalf);
        double pa = __builtin_fma(c[9], u, c[8]);
#pragma unroll
        for (int k = 7; k >= 0; --k) pa = __builtin_fma(pa, u, c[k]);
        x.pa = pa;
    }
    const double nan = __builtin_nan("");
    x.fpar = (r.fpar_pct >= 249u) ? nan : (double)r.fpar_pct * 0.01;
    x.lai = (r.lai_x10 >= 249u) ? nan : (double)r.lai_x10 * 0.1;
    return x;
}

// The raw forms' temperature interval: 190 K .. 360 K (-83 C .. +87 C; the coldest 10-m air of a
// reanalysis is ~195 K, the hottest ~330 K). On it the two saturation formulas' exponents differ by
// |delta| < 0.043 and exp(delta) is a degree-6 polynomial to 5e-14 (period_fast<..., RAW>); a
// temperature outside goes the reference's way like every other pixel outside the domain.
constexpr double kRawTmin = 190.0, kRawTmax = 360.0;
__device__ __forceinline__ GuardConsts raw_guard_consts() {
    constexpr double mid = 0.5 * (kRawTmin + kRawTmax), k = kGuardHuge / (0.5 * (kRawTmax - kRawTmin));
    return GuardConsts{in_vgpr(k), in_vgpr(-mid * k), in_vgpr(kGuardHuge)};
}
// Domain of raw_to_pixel_fast + et_pixel_fast on raw drivers: as above for the fields that
// pass through, plus what the fused pre-processing assumes: a specific humidity below 1 kg/kg
// (0.379 qv + 0.622 > 0: one reciprocal serves both quotients), a finite surface pressure, an
// elevation between -2000 m and 12000 m (the interval of the air-pressure polynomial; the Dead Sea
// shore lies at -430 m, Mount Everest at 8849 m; a DEM's fill value goes the reference's way).
// fPAR, LAI (byte decodings) and the air pressure (from the bounded elevation) cannot leave it.
__device__ __forceinline__ double raw_guard_value(const RawIn<double>& r, const GuardConsts& gc) {
#ifdef MOD16_NO_GUARD
    return 0.0;
#else
    const double oma = 1.0 - r.alb;                                     // see fast_guard_value
    const double a_d = __builtin_fma(r.sw_d, oma, r.lw_d), rn_n = __builtin_fma(r.sw_n, oma, r.lw_n);
    double m = max_abs(a_d, r.lw_n);
    m = max_abs(m, rn_n);
    m = max_abs(m, r.ps_d);
    m = max_abs(m, r.ps_n);
    m = max_abs(m, r.qv_d * gc.huge);          // |qv| < 1
    m = max_abs(m, r.qv_n * gc.huge);
    // -2000 m < z < 12000 m: the interval of the pressure polynomial (19.4 kPa .. 127.8 kPa)
    m = max_abs(m, __builtin_fma(r.elev, in_vgpr(kGuardHuge / kElevHalf), in_vgpr(-kElevMid * (kGuardHuge / kElevHalf))));
    m = max_abs(m, __builtin_fma(r.t_d, gc.t_scale, gc.t_shift));
    m = max_abs(m, __builtin_fma(r.t_n, gc.t_scale, gc.t_shift));
    return m;
#endif
}
__device__ __forceinline__ bool raw_out_of_domain(const RawIn<double>& r) {
    const GuardConsts gc = raw_guard_consts();
    return raw_guard_value(r, gc) >= gc.huge;
}

}  // namespace mod16
