// The sub-methods of the reference's MOD16 class surface as one generic gfx950
// kernel (reference operation order, IEEE divide / ocml exp, pow): the public
// methods mod16/__init__.py:384-673, :795-1258 and module functions :1261-1397
// that user code and the reference's own tests call directly. These are not
// the hot path (the fused kernels are); they exist so that the whole class
// surface computes on the GPU and shares its device functions with the EXACT
// fused kernel.
#pragma once
#include "mod16_kernels.hpp"

namespace mod16 {

constexpr int kMethodMaxIn = 13;

template <typename T> struct MethodArgs {
    const T* in[kMethodMaxIn];
    const T* par[11];
    T* out[2];
    int64_t n;
    uint32_t dense_in, present_in, dense_par;
    int method;
    T alpha;
    T tiny;                  // the reference's `tiny` argument (:869, :1157), default 1e-7
};

// Input slots per method (order of the reference signatures; "?" = optional,
// absent -> computed as the reference does):
//  0 SVP                 temp_k
//  1 SVP_SLOPE           temp_k, s?
//  2 LHV                 temp_k
//  3 PSYCHROMETRIC       pressure, temp_k
//  4 RADIATION_NET       sw_rad, sw_albedo, temp_k
//  5 AIR_DENSITY         temp_k, pressure, rhumidity
//  6 AIR_PRESSURE        elevation_m
//  7 VPD                 qv10m, pressure, tmean
//  8 RHUMIDITY           temp_k, vpd
//  9 POT_SOIL_EVAP       pressure, temp_k, vpd, fpar, rad_soil, r_corr?, lhv?, rh?, f_wet?   -> sat, unsat
// 10 POT_TRANSPIRATION   lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd, fpar, rh?, f_wet?   (alpha)
// 11 EVAP_SOIL           pressure, temp_k, vpd, fpar, rad_soil, r_corr?, lhv?, rh?, f_wet?
// 12 EVAP_WET_CANOPY     pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv?, rh?, f_wet?
// 13 RADIATION_SOIL      lw_d, lw_n, sw_d, sw_n, albedo, t_d, t_n, t_annual, fpar           -> day, night
// 14 SOIL_HEAT_FLUX      rad_net_day, rad_net_night, t_d, t_n, t_annual                      -> day, night
// 15 SURFACE_CONDUCTANCE tmin, vpd_day
// 16 TRANSPIRATION_DAY   pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin, r_corr?, lhv?, rh?, f_wet?
// 17 TRANSPIRATION_NIGHT same
template <typename T>
__global__ void __launch_bounds__(kBlock) method_kernel(const MethodArgs<T> a) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        auto has = [&](int k) { return (a.present_in >> k) & 1u; };
        auto in = [&](int k) { return ((a.dense_in >> k) & 1u) ? a.in[k][i] : a.in[k][0]; };
        auto par = [&](int k) { return ((a.dense_par >> k) & 1u) ? a.par[k][i] : a.par[k][0]; };
        ClassPar<T> p;
        p.tmin_close = par(0); p.tmin_open = par(1); p.vpd_open = par(2); p.vpd_close = par(3);
        p.gl_sh = par(4); p.gl_wv = par(5); p.g_cut = par(6); p.csl = par(7);
        p.rbl_min = par(8); p.rbl_max = par(9); p.beta = par(10);
        T o0 = T(0), o1 = T(0);
        // shared defaults of the optional arguments (:502-509, :926-931, :1216-1223)
        auto rh_or = [&](int k, T t, T vpd) { return has(k) ? in(k) : rh_exact(t, vpd); };
        auto fwet_or = [&](int k, T rh) { return has(k) ? in(k) : fwet_exact(rh); };
        auto lhv_or = [&](int k, T t) { return has(k) ? in(k) : lhv_exact(t); };
        auto rcorr_or = [&](int k, T pa, T t) { return has(k) ? in(k) : rcorr_exact(pa, t); };
        switch (a.method) {
            case 0: o0 = svp_exact(in(0)); break;
            case 1: {
#pragma clang fp contract(off)
                T t = in(0);
                T s = has(1) ? in(1) : svp_exact(t);
                T d = (T(239.0) + t) - K<T>::t0;
                o0 = (T(17.38 * 239.0) * s) / (d * d);
                break;
            }
            case 2: o0 = lhv_exact(in(0)); break;
            case 3: o0 = gamma_exact(in(0), in(1)); break;
            case 4: o0 = radiation_net_exact(in(0), in(1), in(2)); break;
            case 5: o0 = rho_exact(in(0), in(1), in(2)); break;
            case 6: o0 = air_pressure_exact(in(0)); break;
            case 7: o0 = vpd_exact(in(0), in(1), in(2)); break;
            case 8: o0 = rh_exact(in(0), in(1)); break;
            case 9: {
                T pa = in(0), t = in(1), vpd = in(2);
                T rh = rh_or(7, t, vpd);
                pot_soil_exact(p, pa, t, vpd, in(3), in(4), rcorr_or(5, pa, t), rh, fwet_or(8, rh),
                               o0, o1);
                break;
            }
            case 10: {
                T t = in(4), vpd = in(5);
                T rh = rh_or(7, t, vpd);
                o0 = pot_transpiration_exact(in(0), in(1), in(2), in(3), t, in(6), fwet_or(8, rh),
                                             a.alpha);
                break;
            }
            case 11: {
                T pa = in(0), t = in(1), vpd = in(2);
                T rh = rh_or(7, t, vpd);
                o0 = soil_exact(p, pa, t, vpd, in(3), in(4), rcorr_or(5, pa, t), lhv_or(6, t), rh,
                                fwet_or(8, rh));
                break;
            }
            case 12: {
                T pa = in(0), t = in(1), vpd = in(2);
                T rh = rh_or(7, t, vpd);
                o0 = wet_canopy_exact(p, pa, t, vpd, in(3), in(4), in(5), lhv_or(6, t), rh,
                                      fwet_or(8, rh), a.tiny);
                break;
            }
            case 13: {
                PixelIn<T> x = {in(0), in(1), in(2), in(3), in(4), in(5), in(6), in(7), T(0), T(0),
                                T(0), T(0), in(8), T(0)};
                rad_soil_exact(x, p, o0, o1);
                break;
            }
            case 14: soil_heat_flux_exact(p, in(0), in(1), in(2), in(3), in(4), o0, o1); break;
            case 15: {
#pragma clang fp contract(off)
                o0 = p.csl * ramp_up_exact(in(0) - K<T>::t0, p.tmin_close, p.tmin_open) *
                     ramp_down_exact(in(1), p.vpd_open, p.vpd_close);
                break;
            }
            case 16:
            case 17: {
                T pa = in(0), t = in(1), vpd = in(2);
                T rh = rh_or(9, t, vpd);
                T rc = rcorr_or(7, pa, t), lhv = lhv_or(8, t), fw = fwet_or(10, rh);
                o0 = (a.method == 16)
                         ? transpiration_exact<T, true>(p, pa, t, vpd, in(3), in(4), in(5), in(6), rc, lhv, rh, fw, a.tiny)
                         : transpiration_exact<T, false>(p, pa, t, vpd, in(3), in(4), in(5), in(6), rc, lhv, rh, fw, a.tiny);
                break;
            }
            default: break;
        }
        if (a.out[0]) a.out[0][i] = o0;
        if (a.out[1]) a.out[1][i] = o1;
    }
}

// ---- the vectorised calibration path, MOD16._evapotranspiration (:195-382)
template <typename T> struct StaticArgs {
    const T* drv[14];
    const T* par[11];
    const T* rc[2];          // optional r_corr_list (day, night), NULL = compute
    T* out[2];
    int64_t n;
    uint32_t dense_drv, dense_par, dense_rc;
    unsigned* flag;          // device word: bit 0 = any(g_surf > 0) in the day period
    T tiny;                  // the reference's `tiny` argument (:199), default 1e-7
};

template <typename T>
__device__ __forceinline__ void static_load(const StaticArgs<T>& a, int64_t i, PixelIn<T>& x,
                                            ClassPar<T>& p) {
    auto d = [&](int k) { return ((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]; };
    auto q = [&](int k) { return ((a.dense_par >> k) & 1u) ? a.par[k][i] : a.par[k][0]; };
    x = {d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), d(12), d(13)};
    p.tmin_close = q(0); p.tmin_open = q(1); p.vpd_open = q(2); p.vpd_close = q(3);
    p.gl_sh = q(4); p.gl_wv = q(5); p.g_cut = q(6); p.csl = q(7);
    p.rbl_min = q(8); p.rbl_max = q(9); p.beta = q(10);
}

// pass 1: does any pixel have g_surf > 0 (the reference's whole-array branch)?
template <typename T>
__global__ void __launch_bounds__(kBlock) static_flag_kernel(const StaticArgs<T> a) {
    const int64_t step = (int64_t)gridDim.x * kBlock;
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
#pragma clang fp contract(off)
        PixelIn<T> x;
        ClassPar<T> p;
        static_load(a, i, x, p);
        T rc = a.rc[0] ? (((a.dense_rc >> 0) & 1u) ? a.rc[0][i] : a.rc[0][0]) : rcorr_exact(x.pa, x.t_d);
        any = any || ((gsurf_static(p, x.tmin, x.vpd_d) / rc) > T(0));
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(a.flag, 1u);
}

template <typename T>
__global__ void __launch_bounds__(kBlock) static_kernel(const StaticArgs<T> a) {
    const bool any_gs = (*a.flag & 1u) != 0;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        PixelIn<T> x;
        ClassPar<T> p;
        static_load(a, i, x, p);
        const bool has_rc = a.rc[0] != nullptr;
        T rc_d = has_rc ? (((a.dense_rc >> 0) & 1u) ? a.rc[0][i] : a.rc[0][0]) : T(0);
        T rc_n = has_rc ? (((a.dense_rc >> 1) & 1u) ? a.rc[1][i] : a.rc[1][0]) : T(0);
        T day, night;
        et_static_pixel(x, p, has_rc, rc_d, rc_n, any_gs, day, night, a.tiny);
        a.out[0][i] = day;
        a.out[1][i] = night;
    }
}

// ---- the calibration path batched over parameter vectors:
// the reference's MCMC / Sobol sampling hammer (reference calibration.py:907,
// sensitivity.py:95) is MOD16._et(params, *drivers) for thousands of parameter
// draws over the same few 10^4..10^6 tower-days. One launch evaluates every
// (draw, pixel) pair: a thread owns one pixel and walks kBatchDraws
// consecutive draws (blockIdx.y = chunk of draws; the 11 parameters of a draw
// are block-uniform scalar loads), so everything that does not depend on the
// parameters -- svp, its slope, rh, fwet, air density, r_corr with its pow() --
// is loop-invariant and computed once per chunk. Same pixel function as
// static_kernel in the same operation order (fp contraction off), so row d of
// the result is bit-identical to a single-draw call with params[d].
constexpr int kBatchDraws = 32;
template <typename T> struct StaticBatchArgs {
    const T* drv[14];
    const T* params;         // [ndraw][11], MOD16.required_parameters order
    T* out[3];               // day, night, day + night: [ndraw][n], any may be NULL
    int64_t n;
    int64_t draw0;           // first draw of this launch (gridDim.y chunks of kBatchDraws per launch)
    int64_t ndraw;
    const double* tab;       // exp / log tables of FastMath<double> (FAST kernels)
    uint32_t dense_drv;
    unsigned* flags;         // [ndraw] words: bit 0 = any(g_surf > 0) for that draw
};

template <typename T>
__device__ __forceinline__ void batch_load(const StaticBatchArgs<T>& a, int64_t draw, int64_t i,
                                           PixelIn<T>& x, ClassPar<T>& p) {
    auto d = [&](int k) { return ((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]; };
    const T* q = a.params + draw * 11;
    x = {d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), d(12), d(13)};
    p.tmin_close = q[0]; p.tmin_open = q[1]; p.vpd_open = q[2]; p.vpd_close = q[3];
    p.gl_sh = q[4]; p.gl_wv = q[5]; p.g_cut = q[6]; p.csl = q[7];
    p.rbl_min = q[8]; p.rbl_max = q[9]; p.beta = q[10];
}

template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_flag_kernel(const StaticBatchArgs<T> a) {
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        for (int64_t draw = d0; draw < d1; ++draw) {
#pragma clang fp contract(off)
            PixelIn<T> x;
            ClassPar<T> p;
            batch_load(a, draw, i, x, p);
            const bool any = (gsurf_static(p, x.tmin, x.vpd_d) / rcorr_exact(x.pa, x.t_d)) > T(0);
            if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(a.flags + draw, 1u);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_kernel(const StaticBatchArgs<T> a) {
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        for (int64_t draw = d0; draw < d1; ++draw) {
            const bool any_gs = (a.flags[draw] & 1u) != 0;
            PixelIn<T> x;
            ClassPar<T> p;
            batch_load(a, draw, i, x, p);
            T day, night;
            et_static_pixel(x, p, false, T(0), T(0), any_gs, day, night);
            const int64_t row = draw * a.n;
            if (a.out[0]) a.out[0][row + i] = day;
            if (a.out[1]) a.out[1][row + i] = night;
            if (a.out[2]) a.out[2][row + i] = day + night;      // MOD16._et, :193
        }
    }
}

// ---- the same with the strength-reduced arithmetic (MOD16_MATH_FAST): the
// parameter-independent part of the pixel function is prepared explicitly
// (static_pixel_prep), each draw costs ~100 float64 instructions per period
// instead of ~25 IEEE divisions and a pow(). float32 data are widened, computed
// in float64 and rounded once.
template <typename T>
__device__ __forceinline__ void batch_load_fast(const StaticBatchArgs<T>& a, int64_t i, PixelIn<double>& x) {
    auto d = [&](int k) { return (double)(((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]); };
    x = {d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11), d(12), d(13)};
}
template <typename T>
__device__ __forceinline__ ClassPar<double> batch_params_fast(const StaticBatchArgs<T>& a, int64_t draw) {
    const T* q = a.params + draw * 11;
    ClassPar<double> p;
    p.tmin_close = q[0]; p.tmin_open = q[1]; p.vpd_open = q[2]; p.vpd_close = q[3];
    p.gl_sh = q[4]; p.gl_wv = q[5]; p.g_cut = q[6]; p.csl = q[7];
    p.rbl_min = q[8]; p.rbl_max = q[9]; p.beta = q[10];
    return p;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_flag_fast_kernel(const StaticBatchArgs<T> a) {
    constexpr int kTab = FastMath<double>::kTabDoubles;
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        PixelIn<double> x;
        batch_load_fast(a, i, x);
        const StaticPixel c = static_pixel_prep(x, tab);
        for (int64_t draw = d0; draw < d1; ++draw) {
            const ClassPar<double> p = batch_params_fast(a, draw);
            const StaticDraw d = static_draw_prep(c, p);
            const bool any = static_gsurf(c.d, d, p) > 0.0;
            if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(a.flags + draw, 1u);
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_fast_kernel(const StaticBatchArgs<T> a) {
    constexpr int kTab = FastMath<double>::kTabDoubles;
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        PixelIn<double> x;
        batch_load_fast(a, i, x);
        const StaticPixel c = static_pixel_prep(x, tab);
        for (int64_t draw = d0; draw < d1; ++draw) {
            const bool any_gs = (a.flags[draw] & 1u) != 0;
            const ClassPar<double> p = batch_params_fast(a, draw);
            const StaticDraw d = static_draw_prep(c, p);
            const int k = d.cond ? 1 : 0;
            const double day = static_period_eval<true>(c, c.d, d, p, c.rs_d[k], any_gs, tab);
            // at night g_surf = 0 / r_corr, so any(g_surf > 0) is False: no transpiration (:343-348)
            const double night = static_period_eval<false>(c, c.n, d, p, c.rs_n[k], false, tab);
            const int64_t row = draw * a.n;
            if (a.out[0]) a.out[0][row + i] = (T)day;
            if (a.out[1]) a.out[1][row + i] = (T)night;
            if (a.out[2]) a.out[2][row + i] = (T)(day + night);
        }
    }
}

// Weighted sum of squared residuals of each draw against observations, NaN
// pairs skipped: sse[d] = sum_i (w_i (total[d][i] - obs_i))^2, cnt[d] = pairs
// used. One block per draw, fixed order (deterministic).
template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_sse_kernel(const T* total, const T* obs,
                                                                  const T* weights, int64_t n,
                                                                  double* sse, double* cnt) {
    const T* row = total + (int64_t)blockIdx.x * n;
    double s = 0.0, c = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
        const double r = ((double)row[i] - (double)obs[i]) * (weights ? (double)weights[i] : 1.0);
        const bool ok = r == r;
        s += ok ? r * r : 0.0;
        c += ok ? 1.0 : 0.0;
    }
    __shared__ double sm[2][kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s; sm[1][threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) { s += sm[0][w]; c += sm[1][w]; }
        sse[blockIdx.x] = s;
        cnt[blockIdx.x] = c;
    }
}

// ---- forward run on raw drivers
template <typename T> struct RawArgs {
    const T* drv[14];         // enum mod16_raw_driver order
    const uint8_t* fpar_pct;  // fPAR in percent
    const uint8_t* lai_x10;   // LAI x 10
    const uint8_t* cls;
    const T* day_hours;       // optional: hours of daylight -> 8-day total output
    const T* lut;
    const double* lut64;
    const double* tab;
    T* out[3];                // day, night, 8-day total [kg m-2 (8 d)-1]
    int64_t n;
    unsigned* status;
    uint32_t dense_drv;
    uint32_t dense_hours;
};

template <typename T, bool FAST>
__global__ void __launch_bounds__(kBlock) et_raw_kernel(const RawArgs<T> a) {
    typedef typename std::conditional<FAST, double, T>::type C;
    constexpr int kTab = FAST ? FastMath<double>::kTabDoubles : 1;
    __shared__ C lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    if constexpr (FAST) ignore_signalling_nans();       // the domain guard's NaN-ignoring chain
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) {
        if constexpr (FAST) lut[i] = a.lut64[i];
        else lut[i] = a.lut[i];
    }
    if (FAST)
        for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        auto d = [&](int k) { return (C)(((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]); };
        RawIn<C> r = {d(0), d(1), d(2), d(3), d(4), d(5), d(6), d(7), d(8), d(9), d(10), d(11),
                      d(12), d(13), a.fpar_pct[i], a.lai_x10[i]};
        unsigned c = a.cls[i];
        if (c >= 13u) {
            atomicOr(a.status, kStatusClassRange);
            c = 13u;
        }
        const C* l = lut + c;
        ClassPar<C> p;
        p.tmin_close = l[0 * kLutCols]; p.tmin_open = l[1 * kLutCols];
        p.vpd_open = l[2 * kLutCols]; p.vpd_close = l[3 * kLutCols];
        p.gl_sh = l[4 * kLutCols]; p.gl_wv = l[5 * kLutCols];
        p.g_cut = l[6 * kLutCols]; p.csl = l[7 * kLutCols];
        p.rbl_min = l[8 * kLutCols]; p.rbl_max = l[9 * kLutCols];
        p.beta = l[10 * kLutCols];
        p.inv_dtmin = l[11 * kLutCols]; p.inv_dvpd = l[12 * kLutCols];
        p.rbl_slope = l[13 * kLutCols]; p.inv_beta = l[14 * kLutCols];
        PixelOut<C> o;
        if constexpr (FAST) {
            o = et_pixel_fast<double>(raw_to_pixel_fast(r, tab), p, tab);
            // outside the domain of the fast forms: the reference's operation order
            // (mod16_physics.hpp, "domain guard")
            if (raw_out_of_domain(r)) o = et_pixel_exact<double, false, true>(raw_to_pixel_exact<double, true>(r), p);
        } else {
            o = et_pixel_exact<T>(raw_to_pixel_exact<T>(r), p);
        }
        C day = (o.canopy_d + o.soil_d) + o.trans_d;
        C night = (o.canopy_n + o.soil_n) + o.trans_n;
        if (a.out[0]) a.out[0][i] = (T)day;
        if (a.out[1]) a.out[1][i] = (T)night;
        if (a.out[2]) {   // tests/verification/verify2.py:113-115
#pragma clang fp contract(off)
            C h = (C)((a.dense_hours & 1u) ? a.day_hours[i] : a.day_hours[0]);
            a.out[2][i] = (T)((day * h * C(8) * C(60) * C(60)) +
                              (night * (C(24) - h) * C(8) * C(60) * C(60)));
        }
    }
}

}  // namespace mod16
