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
    const uint8_t* skip;     // FAST kernels: [n] or NULL, 1 = pixel outside the FAST domain (left to the
                             // reference-order kernels static_batch_flag_list_kernel / static_batch_redo_rows_kernel)
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
        const bool skipped = a.skip && a.skip[i];
        const StaticPixel c = static_pixel_prep(x, tab);
        for (int64_t draw = d0; draw < d1; ++draw) {
            const ClassPar<double> p = batch_params_fast(a, draw);
            const StaticDraw d = static_draw_prep(c, p);
            const bool any = !skipped && static_gsurf(c.d, d, p) > 0.0;
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
        if (a.skip && a.skip[i]) continue;          // static_batch_redo_rows_kernel writes this pixel's rows
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

// ---- the calibration objective on RESIDENT drivers (mod16_static_batch_bind_*):
// what an MCMC chain or a Sobol analysis (reference calibration.py:907-909, sensitivity.py:94-96)
// evaluates thousands of times on the same drivers. Per evaluation only the parameter vectors go
// up and (sse, count) per draw come back; nothing of size [ndraw][n] exists. One launch
// evaluates every (draw, pixel) pair with the FAST arithmetic and reduces the residuals in a
// fixed order:
//   static_obj_params_kernel   the 11 parameters of a draw + what depends on them alone (three
//                              reciprocals, the rbl slope, the G threshold) -> 16 doubles per draw
//   static_obj_kernel<TR>      thread = pixel, 32 consecutive draws; residual r = w (et - obs),
//                              per-draw sums over the wave's 64 pixels through LDS (lane order),
//                              over the block's waves in order -> one partial per (draw, block)
//   static_obj_redo_kernel     the pixels outside the domain of the FAST arithmetic (marked once,
//                              at bind time), per draw in the reference's operation order
//   static_obj_any_kernel      any(g_surf > 0) per draw over all blocks (the reference's
//                              whole-array branch, mod16/__init__.py:343-348)
//   static_obj_kernel<false>   the draws without it, evaluated again without transpiration (normally
//                              none: every block reads 32 flags and ends)
//   static_obj_final_kernel    partials of a draw added up in block order -> sse[d], count[d]
constexpr int kObjDraws = 32;
constexpr int kObjPass = 16;     // draws reduced at a time (LDS: 16 x 64 doubles per wave)
constexpr int kPar16 = 16;       // doubles per draw: 11 parameters, 1 / (tmin_open - tmin_close), 1 / (vpd_close -
                                 // vpd_open), rbl slope, 1 / beta, 273.15 + tmin_open

template <typename T>
__global__ void __launch_bounds__(kBlock) static_obj_params_kernel(const T* params, int64_t ndraw, double* par16) {
    const int64_t d = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (d >= ndraw) return;
    typedef FastMath<double> M;
    const T* q = params + d * 11;
    double* o = par16 + d * kPar16;
    double p[11];
    for (int k = 0; k < 11; ++k) o[k] = p[k] = (double)q[k];
    // the same operations static_draw_prep performs (so that rows and objective share their bits)
    o[11] = M::rcp(p[1] - p[0]);
    o[12] = M::rcp(p[3] - p[2]);
    o[13] = (p[9] - p[8]) * o[12];
    o[14] = M::rcp(p[10]);
    o[15] = 273.15 + p[1];
}

template <typename T> struct StaticObjArgs {
    const T* drv[14];
    uint32_t dense_drv;
    int64_t n;
    const T* observed;          // [n]
    const T* weights;           // [n] or NULL
    const uint8_t* skip;        // [n] or NULL: 1 = pixel outside the FAST domain (static_obj_redo_kernel takes it)
    const double* par16;        // [ndraw][kPar16]
    const double* tab;
    int64_t ndraw;
    const unsigned* any_draw;   // TR = false: [ndraw], only the draws with 0 here are evaluated
    double* partial;            // [gridDim.x][ndraw][2]: sse, count
    unsigned* any_gs;           // TR = true: [gridDim.x][ndraw], 1 = a pixel of the block has g_surf > 0
};

// TR: evaluate with the transpiration term (any(g_surf > 0) assumed true; the block also reports
// whether one of ITS pixels has g_surf > 0). !TR: only the draws for which no block reported one,
// transpiration = 0 (mod16/__init__.py:343-348) -- normally none: the block reads 32 flags and ends.
template <typename T, bool TR>
__global__ void __launch_bounds__(kBlock) static_obj_kernel(const StaticObjArgs<T> a) {
    constexpr int kTab = FastMath<double>::kTabDoubles;
    constexpr int kWaves = kBlock / 64;
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    // (rows of 65: the column sums below read 16 rows at one lane offset -- with 64 doubles per row all
    // sixteen would hit the same banks)
    __shared__ double red[kWaves][kObjPass][65];     // 33 KiB a block
    __shared__ double wsum[kWaves][kObjDraws][2];
    __shared__ unsigned wany[kWaves][kObjDraws];
    __shared__ unsigned todo;
    const int64_t c0 = (int64_t)blockIdx.y * kObjDraws;
    const int64_t c1 = c0 + kObjDraws < a.ndraw ? c0 + kObjDraws : a.ndraw;
    const int nd = (int)(c1 - c0);
    unsigned want = nd >= 32 ? 0xffffffffu : ((1u << nd) - 1u);     // bit j: evaluate draw c0 + j
    if (!TR) {
        if (threadIdx.x == 0) todo = 0u;
        __syncthreads();
        if ((int)threadIdx.x < nd && !a.any_draw[c0 + threadIdx.x]) atomicOr(&todo, 1u << threadIdx.x);
        __syncthreads();
        want = todo;
        if (want == 0u) return;
    }
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const bool live = i < a.n && !(a.skip && a.skip[i]);
    StaticPixel c;
    double obs = 0.0, w = 1.0;
    if (live) {
        auto dv = [&](int k) { return (double)(((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]); };
        const PixelIn<double> x = {dv(0), dv(1), dv(2), dv(3), dv(4), dv(5), dv(6), dv(7), dv(8), dv(9), dv(10), dv(11), dv(12), dv(13)};
        c = static_pixel_prep(x, tab);
        obs = (double)a.observed[i];
        w = a.weights ? (double)a.weights[i] : 1.0;
    }
    unsigned cnt = 0;            // bit j: draw c0 + j has a pair (the residual is a number) at this pixel
    unsigned gs = 0;             // bit j: g_surf > 0 at this pixel for draw c0 + j
#pragma unroll 1
    for (int h = 0; h < kObjDraws; h += kObjPass) {
#pragma unroll 1
        for (int jj = 0; jj < kObjPass; ++jj) {
            const int j = h + jj;
            double r2 = 0.0;
            if (live && ((want >> j) & 1u)) {
                const double* q = a.par16 + (c0 + j) * kPar16;       // block-uniform: scalar loads
                ClassPar<double> p;
                p.tmin_close = q[0]; p.tmin_open = q[1]; p.vpd_open = q[2]; p.vpd_close = q[3];
                p.gl_sh = q[4]; p.gl_wv = q[5]; p.g_cut = q[6]; p.csl = q[7];
                p.rbl_min = q[8]; p.rbl_max = q[9]; p.beta = q[10];
                StaticDraw d;
                d.m_tmin = (c.tm >= p.tmin_open) ? 1.0
                           : ((c.tm < p.tmin_close) ? 0.0 : (c.tm - p.tmin_close) * q[11]);
                d.inv_dvpd = q[12];
                d.rbl_slope = q[13];
                d.inv_beta = q[14];
                d.cond = c.base_cond && (c.t_ann > q[15]);                    // :230-234 (tmin_open, strict)
                const int k = d.cond ? 1 : 0;
                if (TR) gs |= (static_gsurf(c.d, d, p) > 0.0) ? 1u << j : 0u;
                const double day = static_period_eval<true>(c, c.d, d, p, c.rs_d[k], TR, tab);
                const double night = static_period_eval<false>(c, c.n, d, p, c.rs_n[k], false, tab);
                const double r = ((day + night) - obs) * w;                    // MOD16._et, :193
                const bool ok = r == r;
                r2 = ok ? r * r : 0.0;
                cnt |= ok ? 1u << j : 0u;
            }
            red[wave][jj][lane] = r2;
        }
        // per-draw sums over the wave's 64 pixels, in lane order (lane l of quarter q adds lanes
        // 16 q .. 16 q + 15 of draw l & 15; then the quarters in order)
        __builtin_amdgcn_wave_barrier();
        const int dj = lane & 15, qt = lane >> 4;
        double s = red[wave][dj][qt * 16];
#pragma unroll
        for (int l = 1; l < 16; ++l) s += red[wave][dj][qt * 16 + l];
        const double s1 = __shfl(s, dj + 16, 64), s2 = __shfl(s, dj + 32, 64), s3 = __shfl(s, dj + 48, 64);
        if (lane < 16) wsum[wave][h + lane][0] = ((s + s1) + s2) + s3;
        __builtin_amdgcn_wave_barrier();
    }
    // pairs per draw: population counts of the lanes' bits
#pragma unroll 1
    for (int j = 0; j < kObjDraws; ++j) {
        const unsigned long long m = __ballot((cnt >> j) & 1u);
        const unsigned long long g = __ballot((gs >> j) & 1u);
        if (lane == 0) {
            wsum[wave][j][1] = (double)__builtin_popcountll(m);
            wany[wave][j] = g != 0ull;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nd && ((want >> threadIdx.x) & 1u)) {
        const int j = threadIdx.x;
        double s = wsum[0][j][0], n = wsum[0][j][1];
        unsigned any = wany[0][j];
#pragma unroll
        for (int wv = 1; wv < kWaves; ++wv) {
            s += wsum[wv][j][0];
            n += wsum[wv][j][1];
            any |= wany[wv][j];
        }
        double* out = a.partial + ((int64_t)blockIdx.x * a.ndraw + (c0 + j)) * 2;
        out[0] = s;
        out[1] = n;
        if (TR) a.any_gs[(int64_t)blockIdx.x * a.ndraw + (c0 + j)] = any;
    }
}

// The pixels outside the FAST domain (list made at bind time, ascending), per draw in the
// reference's operation order -- with and without transpiration, since whether any pixel of
// the draw has g_surf > 0 is only known once every block has reported. One block per draw.
template <typename T> struct StaticObjRedoArgs {
    const T* drv[14];
    uint32_t dense_drv;
    const T* params;            // [ndraw][11]
    const T* observed;
    const T* weights;
    const int64_t* list;        // flagged pixels
    int64_t nlist;
    double* redo;               // [ndraw][5]: sse with t, count with t, sse without, count without, pixels with g_surf > 0
};
template <typename T>
__global__ void __launch_bounds__(kBlock) static_obj_redo_kernel(const StaticObjRedoArgs<T> a) {
    const int64_t draw = blockIdx.x;
    double acc[5] = {0, 0, 0, 0, 0};
    for (int64_t u = threadIdx.x; u < a.nlist; u += kBlock) {
#pragma clang fp contract(off)
        const int64_t i = a.list[u];
        auto dv = [&](int k) { return ((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]; };
        PixelIn<T> x = {dv(0), dv(1), dv(2), dv(3), dv(4), dv(5), dv(6), dv(7), dv(8), dv(9), dv(10), dv(11), dv(12), dv(13)};
        const T* q = a.params + draw * 11;
        ClassPar<T> p;
        p.tmin_close = q[0]; p.tmin_open = q[1]; p.vpd_open = q[2]; p.vpd_close = q[3];
        p.gl_sh = q[4]; p.gl_wv = q[5]; p.g_cut = q[6]; p.csl = q[7];
        p.rbl_min = q[8]; p.rbl_max = q[9]; p.beta = q[10];
        const bool any = (gsurf_static(p, x.tmin, x.vpd_d) / rcorr_exact(x.pa, x.t_d)) > T(0);
        const double obs = (double)a.observed[i], w = a.weights ? (double)a.weights[i] : 1.0;
        T day, night;
        et_static_pixel(x, p, false, T(0), T(0), true, day, night);
        double r = ((double)(T)(day + night) - obs) * w;
        bool ok = r == r;
        acc[0] += ok ? r * r : 0.0;
        acc[1] += ok ? 1.0 : 0.0;
        et_static_pixel(x, p, false, T(0), T(0), false, day, night);
        r = ((double)(T)(day + night) - obs) * w;
        ok = r == r;
        acc[2] += ok ? r * r : 0.0;
        acc[3] += ok ? 1.0 : 0.0;
        acc[4] += any ? 1.0 : 0.0;
    }
    __shared__ double sm[5][kBlock / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += __shfl_down(acc[k], off, 64);
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 5; ++k) sm[k][threadIdx.x >> 6] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 5; ++k) {
            double s = sm[k][0];
            for (int wv = 1; wv < kBlock / 64; ++wv) s += sm[k][wv];
            a.redo[draw * 5 + k] = s;
        }
}

// any(g_surf > 0) of every draw over the blocks (and the flagged pixels), and the final sums. A
// block handles 32 draws x 8 slices of the pixel blocks (thread: draw d, blocks slice, slice + 8,
// ...; the per-block values are laid out [block][draw], so 32 lanes read 32 consecutive words),
// several loads in flight per thread; the slices are combined in order through LDS. (One thread per
// draw walking all 391 blocks was 92 us per kernel -- two of them 13 % of an evaluation.)
constexpr int kObjSlices = 8, kObjPerBlock = kBlock / kObjSlices;
static __global__ void __launch_bounds__(kBlock) static_obj_any_kernel(const unsigned* any_gs, const double* redo, int64_t ndraw,
                                                                int gx, unsigned* any_draw) {
    __shared__ unsigned sm[kObjSlices][kObjPerBlock];
    const int dl = threadIdx.x % kObjPerBlock, slice = threadIdx.x / kObjPerBlock;
    const int64_t d = (int64_t)blockIdx.x * kObjPerBlock + dl;
    unsigned any = 0;
    if (d < ndraw) {
#pragma unroll 4
        for (int b = slice; b < gx; b += kObjSlices) any |= any_gs[(int64_t)b * ndraw + d];
    }
    sm[slice][dl] = any;
    __syncthreads();
    if (slice == 0 && d < ndraw) {
#pragma unroll
        for (int k = 1; k < kObjSlices; ++k) any |= sm[k][dl];
        if (redo && redo[d * 5 + 4] > 0.0) any = 1u;
        any_draw[d] = any;
    }
}

// sse[d], count[d]: the block partials of the draw -- slice k adds blocks k, k + 8, ... in order,
// the slices are added in order -- then the flagged pixels'. A fixed order: the same bits on every
// launch. (By now the partials of a draw without g_surf > 0 anywhere hold the pass without
// transpiration.)
static __global__ void __launch_bounds__(kBlock) static_obj_final_kernel(const double* partial, const double* redo,
                                                                  const unsigned* any_draw, int64_t ndraw, int gx,
                                                                  double* sse, double* count) {
    __shared__ double sm[kObjSlices][kObjPerBlock][2];
    const int dl = threadIdx.x % kObjPerBlock, slice = threadIdx.x / kObjPerBlock;
    const int64_t d = (int64_t)blockIdx.x * kObjPerBlock + dl;
    double s = 0.0, n = 0.0;
    if (d < ndraw) {
        typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll 4
        for (int b = slice; b < gx; b += kObjSlices) {
            const d2 v = *reinterpret_cast<const d2*>(partial + ((int64_t)b * ndraw + d) * 2);
            s += v[0];
            n += v[1];
        }
    }
    sm[slice][dl][0] = s;
    sm[slice][dl][1] = n;
    __syncthreads();
    if (slice == 0 && d < ndraw) {
#pragma unroll
        for (int k = 1; k < kObjSlices; ++k) {
            s += sm[k][dl][0];
            n += sm[k][dl][1];
        }
        if (redo) {
            const int o = any_draw[d] ? 0 : 2;
            s += redo[d * 5 + o];
            n += redo[d * 5 + o + 1];
        }
        sse[d] = s;
        count[d] = n;
    }
}

// bind time: which pixels lie outside the domain of the FAST arithmetic (mod16_physics.hpp,
// fast_out_of_domain: the static path rearranges the same terms)
template <typename T>
__global__ void __launch_bounds__(kBlock) static_domain_kernel(const StaticBatchArgs<T> a, uint8_t* skip) {
    ignore_signalling_nans();
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= a.n) return;
    auto dv = [&](int k) { return (double)(((a.dense_drv >> k) & 1u) ? a.drv[k][i] : a.drv[k][0]); };
    const PixelIn<double> x = {dv(0), dv(1), dv(2), dv(3), dv(4), dv(5), dv(6), dv(7), dv(8), dv(9), dv(10), dv(11), dv(12), dv(13)};
    // the forward run's test, and -- this path has no upper clamp on the relative humidity
    // (mod16/__init__.py:280-281), so rh^4 = ((svp - vpd) / svp)^4 grows without bound for a very
    // negative VPD and the conductance forms overflow where the reference's quotients do not -- a
    // VPD beyond 1e18 Pa in magnitude (tests/fuzz_domain.py: -1e15 is inside, -1e30 was not)
    const bool wild_vpd = (__builtin_fabs(x.vpd_d) >= 1e18) | (__builtin_fabs(x.vpd_n) >= 1e18);
    skip[i] = (fast_out_of_domain(x) | wild_vpd) ? 1 : 0;
}

// The pixels the FAST kernels skip (a.skip), in the reference's operation order: their part of
// any(g_surf > 0) per draw, then their rows. Same grid as the FAST kernels; a thread whose pixel is
// inside the domain has nothing to do (one byte read).
template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_flag_skipped_kernel(const StaticBatchArgs<T> a) {
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        if (!a.skip[i]) continue;
        for (int64_t draw = d0; draw < d1; ++draw) {
#pragma clang fp contract(off)
            PixelIn<T> x;
            ClassPar<T> p;
            batch_load(a, draw, i, x, p);
            if ((gsurf_static(p, x.tmin, x.vpd_d) / rcorr_exact(x.pa, x.t_d)) > T(0)) atomicOr(a.flags + draw, 1u);
        }
    }
}
template <typename T>
__global__ void __launch_bounds__(kBlock) static_batch_redo_rows_kernel(const StaticBatchArgs<T> a) {
    const int64_t d0 = a.draw0 + (int64_t)blockIdx.y * kBatchDraws;
    const int64_t d1 = (d0 + kBatchDraws < a.ndraw) ? d0 + kBatchDraws : a.ndraw;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        if (!a.skip[i]) continue;
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
            if (a.out[2]) a.out[2][row + i] = day + night;
        }
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
            double dav_d, dav_n;
            const PixelIn<double> x = raw_to_pixel_fast(r, tab, dav_d, dav_n);
            o = et_pixel_fast<double, false, KLit, true>(x, p, tab, dav_d, dav_n);
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
