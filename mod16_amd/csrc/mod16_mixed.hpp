// Mixed-precision pixel function for float32 rasters (MOD16_MATH_MIXED,
// BASELINE.json configs[4]).
//
// The FAST form computes everything in float64 whatever the storage type; on
// float32 data that makes the kernel VALU-bound at 45 % of the HBM rate. gfx950
// issues a wave64 float32 instruction at the float64 rate unless it is PACKED
// (v_pk_fma_f32 & co: two values per lane per instruction), so this form works
// on PAIRS of pixels (float2) and keeps float64 only where it decides something
// or where a difference of nearly equal numbers feeds a quotient:
//
//   float64  per period esat -> avp = esat - vpd -> rh with the rh < 0.7 and
//            1 - fwet > 0 decisions (mod16/__init__.py:646-673, :763-764, :1245); and the
//            radiation balance with its discontinuous soil-heat-flux clamps (:1033-1052,
//            :1103-1112) for the waves in which some pixel sits within 1e-3 W m-2 of one of
//            those clamps (about 1 wave-iteration in 300).
//   float32  that radiation balance everywhere else (the comparisons whose operands are
//            inputs are made exact by rounding the threshold the right way, the computed
//            ones are the near-tie test), the Tmin ramp, and everything downstream
//            (slope, densities, conductances, the three Penman-Monteith quotients),
//            packed; reciprocals, 2^x and log2 x are the hardware ones (v_rcp_f32,
//            v_exp_f32, v_log_f32, 1 ulp).
// Every decision behind a NaN or an exact zero is made as in the FAST form.
//
// Accuracy against the float64 arithmetic on the same float32 inputs, all 933 M
// pixels of the global grid x 2 outputs (tools/mixedbench.py; bench.py configs[4]):
// NaN masks identical, exact-zero masks identical, relative error above 1e-5 for
// 0.03 % of the values, above 1e-4 for 1181 of 1.87 G, none above 2.0e-4 (round 5:
// 47268 above 1e-4, 2047 above 1e-3, the largest 1.29 -- the cancellation
// s*A + rho*Cp*vpd/r_a with A < 0, which float32 factors cannot resolve better than
// 1e-7 * |s*A|: such values are now found and computed in float64, see period_mixed).
// tests/test_gpu_mixed.py holds the masks, the median (< 2e-7), the 99th percentile
// (< 3e-6) and the absolute bound, runs the reference's edge cases through this form
// and compares it with the reference's own float32 run (fixture F5).
#pragma once
#include "mod16_physics.hpp"

namespace mod16 {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float x) { return f2{x, x}; }
// 1/x: v_rcp_f32 + one Newton step. The step is there for the degenerate
// inputs, not the last bit: like FastMath<double>::rcp it turns 1/0 and 1/inf
// into NaN (0 * inf), which is what the reference's 0/0 and inf/inf give at
// pressure = 0 and the like.
__device__ __forceinline__ f2 rcp2(f2 x) {
    f2 r = f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
    f2 e = __builtin_elementwise_fma(-x, r, f2{1.f, 1.f});
    return __builtin_elementwise_fma(r, e, r);
}
// 1/x of a value that cannot be zero or infinite inside the form's domain (a temperature offset,
// N t of the air density): the hardware reciprocal alone, 1 ulp (round 6)
__device__ __forceinline__ f2 rcp2_plain(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
// min(max(x, 0), 1) as ONE v_med3_f32 per value (round 6: the three ramps were two compares and two
// selects per value). A NaN x gives 0 (v_med3 returns the minimum then): see the callers.
__device__ __forceinline__ f2 clamp01(f2 x) {
    return f2{__builtin_amdgcn_fmed3f(x.x, 0.f, 1.f), __builtin_amdgcn_fmed3f(x.y, 0.f, 1.f)};
}
__device__ __forceinline__ f2 exp2_2(f2 x) { return f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
__device__ __forceinline__ f2 log2_2(f2 x) { return f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
// Predicates of the two pixels are pairs of plain bools (x_0, x_1), not an int vector: each
// stays a lane mask in scalar registers (v_cmp -> s[..]), combines with s_or / s_and and
// feeds v_cndmask directly. A vector mask costs two v_cndmask to materialise, two v_or /
// v_and per combination and two more v_cmp to test, all on the vector pipe that bounds this
// kernel. (Not a struct of two bools passed by value either: that is coerced to an i16 and
// the bits are shifted in and out of it.)
__device__ __forceinline__ f2 sel(bool m0, bool m1, f2 a, f2 b) { return f2{m0 ? a.x : b.x, m1 ? a.y : b.y}; }

// class parameters of the two pixels, float32 (the LDS table is float64:
// rounded on the way in, 15 conversions per pixel)
// One float32 of the table in LDS as ONE ds_read_b32 into a register of its own: read as plain
// loads hipcc pairs the rows of a pixel into ds_read2_b32 and then moves every value into the
// (pixel 0, pixel 1) register pair the packed arithmetic wants -- 18 v_mov_b32 per pair of pixels
// (round 6). volatile keeps the loads apart; the LDS address space is spelled out (a volatile
// generic pointer becomes a flat load, which would also count against the pipeline's vmcnt).
__device__ __forceinline__ float lds_f32(const float* p) {
    return *(const volatile __attribute__((address_space(3))) float*)p;
}
constexpr int kLutDrbl = 13;     // row of the float32 table that holds rbl_max - rbl_min (mod16_stream.hpp fills it)
struct ClassPar2 {
    f2 vpd_open, gl_sh, gl_wv, g_cut, rbl_min, inv_dvpd, drbl, inv_beta;     // drbl = rbl_max - rbl_min
};

struct Shared2 {
    f2 fpar, omf, p_rel, k_p, p_mbar_k, glwv_l, glsh_lai, csl_m;     // csl_m = csl x the Tmin ramp
    bool lai_pos[2], lai_tiny[2];
};

// what the float64 section hands to one period
struct Humid2 {
    f2 esat, rh, fwet, omw;
    bool dry[2];       // rh < 0.7  (fwet = 0)
    bool open_w[2];    // 1 - fwet > 0
};

// esat, rh, fwet, 1 - fwet of one pixel and period in float64, as in period_fast
__device__ __forceinline__ void humid64_tail(double esat, double vpd, float& esat_f, float& rh_f,
                                             float& fwet_f, float& omw_f, bool& dry, bool& open_w);
__device__ __forceinline__ void humid64(double t, double vpd, const double* tb, float& esat_f,
                                        float& rh_f, float& fwet_f, float& omw_f, bool& dry,
                                        bool& open_w) {
    typedef FastMath<double> M;
    double tc = t - K<double>::t0;
    // exp to 4e-11 (cubic on the table's |r| <= ln2/128): the float32 results below keep 6e-8.
    // (A NaN temperature stays NaN through exp_tab3; an infinite one through the reciprocal.)
    // 17.27 tc / (tc + 237.3) = 17.27 - 17.27 * 237.3 / (tc + 237.3): one fma behind the reciprocal
    double esat = (1e3 * 0.6108) * M::exp_tab3s(__builtin_fma(M::rcp(tc + 237.3), -(17.27 * 237.3), 17.27), tb);
    humid64_tail(esat, vpd, esat_f, rh_f, fwet_f, omw_f, dry, open_w);
}
// Raw drivers (round 5): MOD16.vpd (mod16/__init__.py:604-644; the night value clamped at 0,
// calibration.py:401) and the humidity of one pixel and period in ONE float64 section -- the two
// saturation formulas share their exponential (exp(delta) as a quartic: |delta| < 0.043 on the raw
// forms' 190 K .. 360 K, 1e-9 at the cold end, 1e-13 in ordinary air) and the three quotients one
// reciprocal, as in period_fast<..., RAW>.
template <bool NIGHT>
__device__ __forceinline__ double humid64_raw(double t, double qv, double ps, const double* tb, float& esat_f,
                                              float& rh_f, float& fwet_f, float& omw_f, bool& dry,
                                              bool& open_w) {
    typedef FastMath<double> M;
    const double tc = t - K<double>::t0;
    const double d_es = tc + 237.3, ta = tc + 239.0, dav = __builtin_fma(0.379, qv, 0.622);
    const double dd = d_es * ta;
    const double r3 = M::rcp(dd * dav);
    const double r_both = r3 * dav, r_av = r3 * dd;
    const double e_es = M::exp_tab3s((17.27 * tc) * (r_both * ta), tb);
    const double esat = (1e3 * 0.6108) * e_es;
    const double delta = (tc * __builtin_fma(tc, 0x1.c28f5c28f5c00p-4, -0x1.a0c49ba5e34b1p+1)) * r_both;
    double q = __builtin_fma(delta, 1.0 / 24.0, 1.0 / 6.0);
    q = __builtin_fma(q, delta, 0.5);
    q = __builtin_fma(q, delta, 1.0);
    q = __builtin_fma(q, delta, 1.0);
    double vpd = __builtin_fma(-(qv * ps), r_av, (610.7 * e_es) * q);
    if (NIGHT) vpd = (vpd < 0.0) ? 0.0 : vpd;
    humid64_tail(esat, vpd, esat_f, rh_f, fwet_f, omw_f, dry, open_w);
    return vpd;
}
__device__ __forceinline__ void humid64_tail(double esat, double vpd, float& esat_f, float& rh_f,
                                             float& fwet_f, float& omw_f, bool& dry, bool& open_w) {
    typedef FastMath<double> M;
    // rh = avp / esat = 1 - vpd / esat in one fma: exactly 1 for vpd = 0 (fwet = 1 and the
    // 1 - fwet > 0 decision depend on it), 3e-14 absolute otherwise (the reciprocal's) -- the
    // results are float32. The two clamps test the inputs themselves, as the reference's
    // avp < 0 and rh > 1 do in exact arithmetic.
    double rh = __builtin_fma(-vpd, M::rcp(esat), 1.0);
    rh = (vpd < 0.0) ? 1.0 : rh;                // rh > 1 -> 1 (flat selects: a nested conditional becomes a branch)
    rh = (vpd > esat) ? 0.0 : rh;               // avp < 0 -> 0
    dry = rh < 0.7;
    double rh2 = rh * rh;
    double fwet = dry ? 0.0 : rh2 * rh2;
    double omw = 1.0 - fwet;
    open_w = omw > 0.0;
    esat_f = (float)esat; rh_f = (float)rh; fwet_f = (float)fwet; omw_f = (float)omw;
}

// the three components of one period [kg m-2 s-1], their sum (mod16/__init__.py:792) and, with
// PET, the potential ET of the period (as period_fast, reference README.md:404-424)
struct Parts2 {
    f2 canopy, soil, trans, total, pet;
    f2 zc, zs;     // the cancellation class: negative where the canopy's / the soil's budget exceeds the period's total
};

// ---- the cancellation class (round 6). Two numerators of a period are DIFFERENCES of products of
// like size wherever the net radiation is negative (every night): the wet canopy's
// rho Cp fpar vpd / r_a + s A_c (:952) and the bare soil's s A_soil r_tot + rho Cp (1 - fpar) vpd (r_tot /
// r_as) (:537). float32 factors resolve such a sum to ~3e-7 of its TERMS, so where it cancels to a
// hundredth of them the component is right to 3e-5 only -- and where the period's total is that small
// as well (a dry night over bare ground: the soil term is all there is) the total's relative error
// has no bound: 2047 values of the global grid were off by more than 1e-3 of themselves, one by
// 1.29 (profiles/r05d_bench_line.json; r03_mixed_tail_probe.txt). The period therefore carries the
// size of the radiative term of each numerator through the component's own quotient -- what a
// relative error of that term does to the component -- and a value whose budget exceeds
// kMixedCancel x the period's total is computed again in float64 -- not here: inside the pipeline's
// loop the float64 pixel function's constants cost the loop its scalar registers (+27 v_readlane /
// v_writelane per iteration, measured on the listing); the pixel is marked (its first output holds
// kCancelPoison, a NaN no arithmetic produces, and it counts as NaN in the run's diagnostics) and
// what runs behind the loop -- the machinery of the domain guard, mod16_stream.hpp::redo_piece --
// puts the FAST form's float64 result in its place. kMixedCancel = 320 (1 pixel in 700 of the
// synthetic grid): of 1.87 G values none is then off by more than 2.0e-4 of itself and 1181 by more
// than 1e-4 (47268 before; 128 leaves none above 1e-4 and costs the step 2 % more, 512 leaves 7468:
// profiles/r06_mixed_cancel_threshold.txt). Masked components (a 0 of :959 / :961 / :858-861) carry
// no budget.
#ifndef MOD16_MIXED_CANCEL
#define MOD16_MIXED_CANCEL 320.0
#endif
constexpr float kMixedCancel = (float)(MOD16_MIXED_CANCEL);
constexpr unsigned kCancelPoison = 0x7fc16a5du;      // a quiet NaN with a payload

template <bool DAY, bool PET = false>
__device__ __forceinline__ Parts2 period_mixed(const ClassPar2& p, const Shared2& sh, const Humid2& h,
                                               f2 t, f2 vpd, f2 rad_net, f2 rad_soil) {
    const f2 zero = splat(0.f), one = splat(1.f), tiny = splat(1e-7f);
    f2 tc = t - splat(273.15f);
    f2 ta = t - splat(34.15f);                                              // (239 + T) - 273.15, :1395
    f2 rta = rcp2_plain(ta);                                                // (t > 90 K inside the domain)
    f2 s = (splat((float)(17.38 * 239.0)) * h.esat) * (rta * rta);          // :1395-1397
    f2 lhv = __builtin_elementwise_fma(tc, splat(-0.002361e6f), splat(2.501e6f));   // (2.501 - 0.002361 tc) 1e6, :121
    f2 slhv = s * lhv;
    // 1 / r_corr = (P / 101300) (T / 293.15)^-1.75, :771
    // (the argument near 1: v_log_f32's error is absolute, ~1 ulp of the RESULT -- log2 of t itself,
    // ~8, would cost 1e-6 relative in the power)
    f2 inv_rcorr = sh.p_rel * exp2_2(splat(-1.75f) * log2_2(t * splat((float)(1.0 / 293.15))));
    // rho Cp and 4 sigma T^3 / (rho Cp) from one reciprocal, :408-412, :947
    // (N carries the factor Cp, as in period_fast: rho Cp = N / T, 1/r_r = 4 sigma T^4 / N)
    f2 nn = sh.p_mbar_k - h.rh * __builtin_elementwise_fma(tc, splat((float)(0.252 * 1013.0)), splat((float)(-2.0582 * 1013.0)));   // Cp (rh 100)(0.00252 tc - 0.020582)
    f2 u = rcp2_plain(nn * t);
    f2 rho_cp = (nn * nn) * u;
    f2 t2 = t * t;
    f2 g_rr = splat((float)(4.0 * 5.67e-8)) * ((t2 * t2) * t) * u;
    f2 rcfv = rho_cp * vpd;
    f2 rf = rcfv * sh.fpar;
    f2 radc_raw = sh.fpar * rad_net;
    f2 s_radc = s * radc_raw;

    // wet canopy, :866-961
    // (:934-935 replace fwet = 0 and lai = 0 by `tiny` to keep 1 / (gl lai fwet) finite; nothing
    // divides by either here and such a pixel's result is the 0 of :961 whatever its quotient:
    // no replacement, as in period_fast)
    f2 fw = h.fwet;
    f2 g_e = sh.glwv_l * fw, g_a = __builtin_elementwise_fma(sh.glsh_lai, fw, g_rr);
    f2 csum = __builtin_elementwise_fma(rf, g_a, s_radc);              // numer / fwet
    f2 den = __builtin_elementwise_fma(slhv, g_e, sh.k_p * g_a);
    f2 k_c = (fw * g_e) * rcp2(den);
    // numer < 0 -> 0 (:959; fwet > 0 where it counts, and a NaN fwet comes with a NaN sum), fw <=
    // tiny (<=> dry) or lai <= tiny -> 0 (:961): one select
    const bool cm0 = (csum.x < 0.f) | h.dry[0] | sh.lai_tiny[0];
    const bool cm1 = (csum.y < 0.f) | h.dry[1] | sh.lai_tiny[1];
    f2 ev = csum * k_c;
    f2 canopy = sel(cm0, cm1, zero, ev);

    // bare soil, :449-544, :795-864
    // r_tot of :527-531 as rbl_min + (rbl_max - rbl_min) clamp01((vpd - vpd_open) / (vpd_close -
    // vpd_open)): exactly rbl_min up to vpd_open, rbl_max to an ulp from vpd_close on (a continuous
    // ramp: nothing is decided here). A NaN vpd leaves the clamp as 0 -- and the period as NaN
    // through rho Cp vpd.
    f2 ramp = clamp01((vpd - p.vpd_open) * p.inv_dvpd);
    f2 r0 = __builtin_elementwise_fma(ramp, p.drbl, p.rbl_min);
    f2 r_tot = r0 * inv_rcorr;
    f2 w = __builtin_elementwise_fma(r_tot, g_rr, one);
    f2 s_t1 = (s * rad_soil) * r_tot;
    f2 num = __builtin_elementwise_fma(rcfv * sh.omf, w, s_t1);
    f2 rdens = rcp2(r_tot * __builtin_elementwise_fma(sh.k_p, w, slhv));
    f2 q = num * rdens;
    f2 pw = exp2_2((vpd * p.inv_beta) * log2_2(h.rh));                      // rh ** (vpd / beta), :861
    f2 wet = __builtin_elementwise_fma(h.omw, pw, h.fwet);
    const bool sm0 = q.x < 0.f, sm1 = q.y < 0.f;
    f2 es = q * wet;
    f2 soil = sel(sm0, sm1, zero, es);

    // transpiration, :1152-1258
    f2 g_s = zero;
    if (DAY) g_s = (sh.csl_m * (one - ramp)) * inv_rcorr;                   // :1148-1150, :1237
    f2 gsc = g_s + p.g_cut * inv_rcorr;
    f2 g_bl = sh.glsh_lai * h.omw;
    f2 p1 = g_bl * gsc, s1 = g_bl + gsc;
    const f2 lim = tiny * s1;
    // !(lai > 0 and 1 - fwet > 0) (:1245) or the conductances vanish (:1258)
    const bool shut0 = !(sh.lai_pos[0] & h.open_w[0]) | (p1.x <= lim.x);
    const bool shut1 = !(sh.lai_pos[1] & h.open_w[1]) | (p1.y <= lim.y);
    f2 g_d = p.gl_sh + g_rr;
    // s max(A_c, 0), :1251 (a NaN A_c stays; a NaN s comes with a NaN rho Cp)
    f2 s_radp = sel(radc_raw.x < 0.f, radc_raw.y < 0.f, zero, s_radc);
    f2 numt = (h.omw * __builtin_elementwise_fma(rf, g_d, s_radp)) * p1;
    f2 dent = slhv * p1 + sh.k_p * (g_d * s1 + p1);
    f2 tr = numt * rcp2(dent);
    Parts2 o;
    o.canopy = canopy;
    o.soil = soil;
    o.trans = sel(shut0, shut1, zero, tr);
    o.total = (o.canopy + o.soil) + o.trans;                                // :792
    o.pet = zero;
    if (PET) {
        // sat + unsat without the rh^(vpd/beta) factor (two products, so that inf * 0 is
        // NaN as in :541-543) + Priestley-Taylor potential transpiration (:546-602)
        f2 pot_soil = sel(sm0, sm1, zero, __builtin_elementwise_fma(q, h.fwet, q * h.omw));
        f2 pot_tr = (splat((float)kPriestleyTaylorAlpha) * s_radc * h.omw) * rcp2(slhv + sh.k_p);
        o.pet = (canopy + pot_soil) + pot_tr;
    }
    // the cancellation budgets against the total, all in the vector pipe (as lane masks the tests
    // cost six scalar instructions per value, and at two waves per SIMD a scalar instruction is worth
    // two thirds of a vector one: measured). z = budget + kMixedCancel x (total - (value before its
    // mask - value)): the bracket is the total where the component is not masked; where a NEGATIVE
    // numerator masked it (:959, :858-861) it grows by that negative value -- a component confidently
    // below zero is the same exact 0 in every arithmetic and carries no risk, one within rounding of
    // zero may be a small positive number in float64 (the one value of the global grid that was still
    // off by 0.9 of itself) and keeps its budget. A NaN never flags (v_min3 passes it over).
    o.zc = o.zs = zero;
#ifndef MOD16_MIXED_NO_CANCEL       // (measurement builds: what the detection costs)
    {
        const f2 k = splat(kMixedCancel);
        o.zc = __builtin_elementwise_fma((o.total + canopy) - ev, k, s_radc * k_c);
        o.zs = __builtin_elementwise_fma((o.total + soil) - es, k, (s_t1 * rdens) * wet);
    }
#endif
    return o;
}

// Two pixels. in[k][j]: driver k of pixel j; l0 / l1: the pixels' columns of
// the float64 BPLUT table in LDS ([row][kLutCols] layout, row stride `ls`).
// vpd64: the two periods' VPD in float64 where the caller has it (raw drivers:
// it is a difference of two exponentials), else NULL = the float32 inputs widened.
// f0 / f1: the same columns of the table rounded to float32 (a second copy in LDS: the
// packed arithmetic wants float32 parameters, 15 per pixel, and converting them pixel by
// pixel was 60 vector instructions per four pixels), used with LUTF; else l0 / l1 are converted.
template <bool PET, bool LUTF = false>
__device__ __forceinline__ void et_pair_mixed_parts(const float (&in)[14][2], const double* l0,
                                                    const double* l1, int ls, const double* tb,
                                                    Parts2& day, Parts2& night,
                                                    const double (*vpd64)[2] = nullptr,
                                                    const float* f0 = nullptr, const float* f1 = nullptr,
                                                    const Humid2* hraw = nullptr, bool* cancel = nullptr) {
    const f2 zero = splat(0.f);
    auto col = [&](int k) { return f2{in[k][0], in[k][1]}; };
    auto par = [&](int row) {
        if constexpr (LUTF) return f2{lds_f32(f0 + row * ls), lds_f32(f1 + row * ls)};
        else return f2{(float)l0[row * ls], (float)l1[row * ls]};
    };
    // rbl_max - rbl_min: row kLutDrbl of the float32 copy of the table (its row of the float64 table,
    // the slope of the float64 forms, is not used by this form)
    auto par_drbl = [&]() {
        if constexpr (LUTF) return f2{lds_f32(f0 + kLutDrbl * ls), lds_f32(f1 + kLutDrbl * ls)};
        else return f2{(float)(l0[9 * ls] - l0[8 * ls]), (float)(l1[9 * ls] - l1[8 * ls])};
    };
    // ---- radiation received by the soil, :963-1119, float32
    const f2 lw_d = col(0), lw_n = col(1), sw_d = col(2), sw_n = col(3), alb = col(4);
    const f2 t_d = col(5), t_n = col(6), t_ann = col(7), fpar = col(12);
    const f2 omf = splat(1.f) - fpar;
    // A = sw (1 - albedo) + lw as (sw + lw) - sw albedo: where the two cancel (an overcast winter
    // day: A a hundredth of its terms) 1 - albedo rounded to float32 would cost A 5e-8 of sw (round 6:
    // the next-worst values of the global grid, 9e-4 off, were these)
    f2 a_d = __builtin_elementwise_fma(-sw_d, alb, sw_d + lw_d);
    const f2 a_n = lw_n;
    // x < 298.15 (float64) <=> x < RU(298.15); x >= 273.15 + tmin_close <=> x >= row 15 of
    // the table; t_d - t_n is exact in float32 for temperatures within a factor 2
    const f2 p15 = par(15), dtd = t_d - t_n;
    const bool cond0 = (t_ann.x < 298.150024f) & (t_ann.x >= p15.x) & (dtd.x >= 5.f);
    const bool cond1 = (t_ann.y < 298.150024f) & (t_ann.y >= p15.y) & (dtd.y >= 5.f);
    // t - 273.15 as (t - 273) - 0.15: both differences are exact or correctly rounded
    const f2 gd0 = sel(cond0, cond1, __builtin_elementwise_fma(splat(4.73f), (t_d - splat(273.f)) - splat(0.15f), splat(-20.87f)), zero);
    const f2 gn0 = sel(cond0, cond1, __builtin_elementwise_fma(splat(4.73f), (t_n - splat(273.f)) - splat(0.15f), splat(-20.87f)), zero);
    // (0.39 |A| = |0.39 A| exactly: one product serves the test and the cap, :1112)
    const f2 cap_d = splat(0.39f) * a_d, cap_n = splat(0.39f) * a_n;
    const f2 lim_d = __builtin_elementwise_abs(cap_d), lim_n = __builtin_elementwise_abs(cap_n);
    const f2 agd = __builtin_elementwise_abs(gd0), agn = __builtin_elementwise_abs(gn0);
    const f2 gd1 = sel(agd.x > lim_d.x, agd.y > lim_d.y, cap_d, gd0);
    const f2 gn1 = sel(agn.x > lim_n.x, agn.y > lim_n.y, cap_n, gn0);
    const f2 dd = a_d - gd1;
    const f2 gd2 = sel((dd.x < 0.f) & (a_d.x > 0.f), (dd.y < 0.f) & (a_d.y > 0.f), a_d, gd1);
    const f2 ang = a_n - gn1, half_ad = splat(0.5f) * a_d;
    const f2 dn = ang + half_ad;
    // (:1042-1046 replace G_night by A_night + A_day / 2; what the soil receives is then A_night - G =
    // -A_day / 2 -- taken directly: as a_n - (a_n + a_d / 2) the float32 difference lost five digits
    // where |A_day| << |A_night|, the largest errors left on the global grid, round 6)
    f2 rs_d = omf * (a_d - gd2);
    f2 rs_n = omf * sel((a_d.x > 0.f) & (dn.x < 0.f), (a_d.y > 0.f) & (dn.y < 0.f), -half_ad, ang);
    f2 rn_n = __builtin_elementwise_fma(-sw_n, alb, sw_n + lw_n);
    // near-ties of the computed comparisons: operands within 1e-3 W m-2 (their float32
    // errors are below 1e-4) -> this wave redoes the balance in float64 for these pixels
    const f2 tol = splat(1e-3f);
    // (the smallest of the five distances of each pixel against the tolerance: packed min,
    // one comparison per pixel)
    const f2 dist = __builtin_elementwise_min(
        __builtin_elementwise_min(__builtin_elementwise_abs(agd - lim_d), __builtin_elementwise_abs(agn - lim_n)),
        __builtin_elementwise_min(__builtin_elementwise_min(__builtin_elementwise_abs(dd), __builtin_elementwise_abs(a_d)),
                                  __builtin_elementwise_abs(dn)));
    if (__builtin_expect(__any((dist.x <= tol.x) | (dist.y <= tol.y)), 0)) {     // (out of line: the hot path falls through)
        float r[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double xlw_d = in[0][j], xlw_n = in[1][j], xsw_d = in[2][j], xsw_n = in[3][j], xalb = in[4][j];
            const double xt_d = in[5][j], xt_n = in[6][j], xt_ann = in[7][j], xfpar = in[12][j];
            const double tmin_close = (j ? l1 : l0)[0 * ls];
            const double xoma = 1.0 - xalb, xomf = 1.0 - xfpar;
            const double ad = __builtin_fma(xsw_d, xoma, xlw_d);
            const double an = xlw_n;
            const bool cnd = (xt_ann < 273.15 + 25.0) && (xt_ann >= (273.15 + tmin_close)) && ((xt_d - xt_n) >= 5.0);
            double g_d = cnd ? (4.73 * (xt_d - 273.15)) - 20.87 : 0.0;
            g_d = (__builtin_fabs(g_d) > (0.39 * __builtin_fabs(ad))) ? 0.39 * ad : g_d;
            double g_n = cnd ? (4.73 * (xt_n - 273.15)) - 20.87 : 0.0;
            g_n = (__builtin_fabs(g_n) > (0.39 * __builtin_fabs(an))) ? 0.39 * an : g_n;
            g_d = ((ad - g_d < 0.0) && (ad > 0.0)) ? ad : g_d;
            g_n = ((ad > 0.0) && ((an - g_n) < (-0.5 * ad))) ? an + (0.5 * ad) : g_n;
            r[0][j] = (float)ad;
            r[1][j] = (float)(xomf * (ad - g_d));
            r[2][j] = (float)(xomf * (an - g_n));
            r[3][j] = (float)__builtin_fma(xsw_n, xoma, xlw_n);
        }
        a_d = f2{r[0][0], r[0][1]}; rs_d = f2{r[1][0], r[1][1]};
        rs_n = f2{r[2][0], r[2][1]}; rn_n = f2{r[3][0], r[3][1]};
    }
    // ---- float64: humidity of both periods (an all-float32 form of this section was
    // built and measured in round 2 -- value + error pairs for esat, 67 packed instructions
    // per two pixels and period against 2 x 45 here: same error table, same kernel time;
    // profiles/r02_experiments_not_kept.txt)
    Humid2 hd, hn;
    if (hraw) {             // raw drivers: the humidity came with the VPD (raw_pair_mixed)
        hd = hraw[0];
        hn = hraw[1];
    } else {
        float esat_d[2], rh_d[2], fwet_d[2], omw_d[2], esat_n[2], rh_n[2], fwet_n[2], omw_n[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double vd = vpd64 ? vpd64[0][j] : (double)in[9][j], vn = vpd64 ? vpd64[1][j] : (double)in[10][j];
            humid64((double)in[5][j], vd, tb, esat_d[j], rh_d[j], fwet_d[j], omw_d[j], hd.dry[j], hd.open_w[j]);
            humid64((double)in[6][j], vn, tb, esat_n[j], rh_n[j], fwet_n[j], omw_n[j], hn.dry[j], hn.open_w[j]);
        }
        hd.esat = f2{esat_d[0], esat_d[1]}; hd.rh = f2{rh_d[0], rh_d[1]};
        hd.fwet = f2{fwet_d[0], fwet_d[1]}; hd.omw = f2{omw_d[0], omw_d[1]};
        hn.esat = f2{esat_n[0], esat_n[1]}; hn.rh = f2{rh_n[0], rh_n[1]};
        hn.fwet = f2{fwet_n[0], fwet_n[1]}; hn.omw = f2{omw_n[0], omw_n[1]};
    }

    // ---- float32, packed
    ClassPar2 p;
    p.vpd_open = par(2); p.gl_sh = par(4); p.gl_wv = par(5);
    p.g_cut = par(6); p.rbl_min = par(8);
    p.inv_dvpd = par(12); p.drbl = par_drbl(); p.inv_beta = par(14);
    const f2 pa = col(11), lai = col(13);
    Shared2 sh;
    sh.fpar = fpar;
    sh.omf = omf;
    sh.p_rel = pa * splat((float)(1.0 / 101300.0));
    sh.k_p = pa * splat((float)(1013.0 / 0.622));
    sh.p_mbar_k = pa * splat((float)(1013.0 * 0.348444 / 100.0));
    // lai <= tiny (:961; a LAI of 0 became `tiny` at :935 and is one of them) is asked in float64 of
    // the float32 value: float32(1e-7) is ABOVE 1e-7, so a LAI of exactly that float is not masked by
    // the reference (pair fuzz, round 3): x <= 1e-7  <=>  x <= 0x1.ad7f28p-24f, the largest float32 below
    constexpr float kTinyBelow = 0x1.ad7f28p-24f;
    sh.lai_tiny[0] = lai.x <= kTinyBelow;
    sh.lai_tiny[1] = lai.y <= kTinyBelow;
    sh.lai_pos[0] = lai.x > 0.f; sh.lai_pos[1] = lai.y > 0.f;
    sh.glwv_l = p.gl_wv * lai;
    sh.glsh_lai = p.gl_sh * lai;
    // csl x the Tmin ramp, :1148: clamp01((tmin - 273.15 - tmin_close) / (tmin_open - tmin_close)),
    // continuous (a float32 tie decides nothing). The clamp drops a NaN Tmin, which the reference
    // carries into the day's transpiration: tmin * 0 brings it back (an INFINITE Tmin -- 1 or 0 in the
    // reference -- is outside the form's domain since round 6: pair_out_of_domain).
    const f2 tmin = col(8);
    const f2 tm = (tmin - splat(273.f)) - splat(0.15f);
    sh.csl_m = __builtin_elementwise_fma(par(7), clamp01((tm - par(0)) * par(11)), tmin * zero);
    day = period_mixed<true, PET>(p, sh, hd, t_d, col(9), a_d, rs_d);
    night = period_mixed<false, PET>(p, sh, hn, t_n, col(10), rn_n, rs_n);
    // the cancellation class (see period_mixed): the caller marks these pixels for the pass behind the
    // pipeline's loop, which computes them in float64
    if (cancel) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float m;
            asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(day.zc[e]), "v"(day.zs[e]), "v"(night.zc[e]));
            asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(m), "v"(night.zs[e]));
            cancel[e] = m < 0.f;
        }
    }
}

// Raw drivers (calibration.py:380-423) for the mixed form: the
// pixel-function inputs of two pixels from their raw fields. VPD = svp(T) - avp
// is a difference of nearly equal numbers in humid air and feeds rh, so it is
// float64 (and handed on in float64); air pressure from elevation and the byte
// decodings are float32.
__device__ __forceinline__ void raw_pair_mixed(const float (&raw)[14][2], const unsigned (&fpar_pct)[2],
                                               const unsigned (&lai_x10)[2], const double* tb,
                                               float (&in)[14][2], Humid2 (&hum)[2]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) { in[k][0] = raw[k][0]; in[k][1] = raw[k][1]; }
    float esat[2][2], rh[2][2], fwet[2][2], omw[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const double vd = humid64_raw<false>(raw[5][j], raw[9][j], raw[11][j], tb, esat[0][j], rh[0][j], fwet[0][j],
                                             omw[0][j], hum[0].dry[j], hum[0].open_w[j]);
        const double vn = humid64_raw<true>(raw[6][j], raw[10][j], raw[12][j], tb, esat[1][j], rh[1][j], fwet[1][j],
                                            omw[1][j], hum[1].dry[j], hum[1].open_w[j]);
        in[9][j] = (float)vd;
        in[10][j] = (float)vn;
        in[12][j] = (fpar_pct[j] >= 249u) ? __builtin_nanf("") : (float)fpar_pct[j] * 0.01f;
        in[13][j] = (lai_x10[j] >= 249u) ? __builtin_nanf("") : (float)lai_x10[j] * 0.1f;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        hum[e].esat = f2{esat[e][0], esat[e][1]}; hum[e].rh = f2{rh[e][0], rh[e][1]};
        hum[e].fwet = f2{fwet[e][0], fwet[e][1]}; hum[e].omw = f2{omw[e][0], omw[e][1]};
    }
    // 101325 (1 - 0.0065 z / 288.15)^5.2559, MOD16.air_pressure :414-447
    const f2 ratio = __builtin_elementwise_fma(f2{raw[13][0], raw[13][1]}, splat((float)(-0.0065 / 288.15)), splat(1.f));
    const f2 pa = splat(101325.f) * exp2_2(splat((float)(9.80665 / (0.0065 * (8.3143 / 28.9644e-3)))) * log2_2(ratio));
    in[11][0] = pa.x; in[11][1] = pa.y;
}

// ---- domain guard of the mixed-precision form (see mod16_physics.hpp, "domain guard").
// float32 products leave their range long before float64 ones do, and the form is only
// meant for physical drivers, so its domain is drawn tightly around them (mapped with
// tests/fuzz_domain.py: outside it NaN / zero masks start to differ from the float64
// arithmetic's, or values by more than 1e-3):
//     |lw|, |sw|, |albedo|, |vpd|, |fpar|, |lai|, |tmin| < 1e5;  1e3 <= pressure < 1e7 Pa;
//     90 K < temp_day, temp_night < 1332 K                  (NaN anywhere: inside)
// Every condition is brought to the form |y| >= 1e5 -- the temperatures and the pressure by
// one packed fma for both pixels -- and the thirteen values of a pixel go through one chain of
// v_max3_f32 with |.| modifiers, which ignores NaN operands: 8.5 vector instructions per pixel.
// A flagged pixel is computed again by et_pixel_exact<double> on the widened inputs, like a
// flagged pixel of the FAST form on a float32 raster (mod16_stream.hpp).
__device__ __forceinline__ float max3_abs_f32(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
constexpr float kGuardMixed = 1e5f;
// bit e set: pixel e of the pair has a |y[k]| >= 1e5
template <int N>
__device__ __forceinline__ unsigned pair_guard(const f2 (&y)[N]) {
#ifdef MOD16_NO_GUARD
    return 0u;
#else
    static_assert(N >= 3, "at least one v_max3");
    unsigned bad = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float m = max3_abs_f32(y[0][e], y[1][e], y[2][e]);
#pragma unroll
        for (int k = 3; k + 1 < N; k += 2) m = max3_abs_f32(m, y[k][e], y[k + 1][e]);
        if ((N - 3) & 1) m = max3_abs_f32(m, y[N - 1][e], y[N - 1][e]);
        bad |= (m >= kGuardMixed) ? 1u << e : 0u;
    }
    return bad;
#endif
}
// 90 < t < 1332  <=>  |t - 711| < 621 ;  1e3 <= p < 1e7  <=>  |p - 5.0005e6| <= 4.9995e6
__device__ __forceinline__ f2 guard_scale_t(f2 t) {
    return __builtin_elementwise_fma(t, splat(kGuardMixed / 621.f), splat(-711.f * (kGuardMixed / 621.f)));
}
__device__ __forceinline__ f2 guard_scale_p(f2 p) {
    return __builtin_elementwise_fma(p, splat(kGuardMixed / 4.9995e6f), splat(-5.0005e6f * (kGuardMixed / 4.9995e6f)));
}
__device__ __forceinline__ unsigned pair_out_of_domain(const float (&in)[14][2]) {
    auto col = [&](int k) { return f2{in[k][0], in[k][1]}; };
    const f2 y[13] = {col(0), col(1), col(2), col(3), col(4), col(9), col(10), col(12), col(13),
                      guard_scale_t(col(5)), guard_scale_t(col(6)), guard_scale_p(col(11)), col(8)};
    return pair_guard<13>(y);
}
// raw drivers (raw_pair_mixed): the fields that pass through as above; specific humidity
// below 1 kg/kg, the surface pressures like the air pressure, |elevation| below 25 km (the air
// pressure computed from it then stays inside 1e3 .. 1e7 Pa: 1.1e3 Pa at 25 km, 1.1e6 at -25 km)
// ... and the raw forms' temperature interval, 190 K < T < 360 K (kRawTmin / kRawTmax, mod16_physics.hpp:
// where the shared exponential of the two saturation formulas holds)
__device__ __forceinline__ f2 guard_scale_traw(f2 t) {
    constexpr float mid = 0.5f * (float)(kRawTmin + kRawTmax), half = 0.5f * (float)(kRawTmax - kRawTmin);
    return __builtin_elementwise_fma(t, splat(kGuardMixed / half), splat(-mid * (kGuardMixed / half)));
}
__device__ __forceinline__ unsigned raw_pair_out_of_domain(const float (&raw)[14][2]) {
    auto col = [&](int k) { return f2{raw[k][0], raw[k][1]}; };
    const f2 y[13] = {col(0), col(1), col(2), col(3), col(4), col(9) * splat(kGuardMixed),
                      col(10) * splat(kGuardMixed), col(13) * splat(4.f),
                      guard_scale_traw(col(5)), guard_scale_traw(col(6)), guard_scale_p(col(11)),
                      guard_scale_p(col(12)), col(8)};
    return pair_guard<13>(y);
}

// The same two guards for ONE pixel given as (widened) float64 values -- the slow branch
// re-reads a pixel's inputs and asks again which of a thread's pixels it was. Same float32
// operations as above, so the answer is the same.
__device__ __forceinline__ bool guard_list_f32(const float* y, int n) {
    float m = 0.f;
    for (int k = 0; k < n; ++k) m = max3_abs_f32(m, y[k], y[k]);
    return m >= kGuardMixed;
}
__device__ __forceinline__ float guard_scale_t1(float t) {
    return __builtin_fmaf(t, kGuardMixed / 621.f, -711.f * (kGuardMixed / 621.f));
}
__device__ __forceinline__ float guard_scale_p1(float p) {
    return __builtin_fmaf(p, kGuardMixed / 4.9995e6f, -5.0005e6f * (kGuardMixed / 4.9995e6f));
}
__device__ __forceinline__ float guard_scale_traw1(float t) {
    constexpr float mid = 0.5f * (float)(kRawTmin + kRawTmax), half = 0.5f * (float)(kRawTmax - kRawTmin);
    return __builtin_fmaf(t, kGuardMixed / half, -mid * (kGuardMixed / half));
}
__device__ __forceinline__ bool out_of_domain_f32(const PixelIn<double>& x) {
#ifdef MOD16_NO_GUARD
    return false;
#else
    const float y[13] = {(float)x.lw_d, (float)x.lw_n, (float)x.sw_d, (float)x.sw_n, (float)x.alb,
                         (float)x.vpd_d, (float)x.vpd_n, (float)x.fpar, (float)x.lai,
                         guard_scale_t1((float)x.t_d), guard_scale_t1((float)x.t_n), guard_scale_p1((float)x.pa),
                         (float)x.tmin};
    return guard_list_f32(y, 13);
#endif
}
__device__ __forceinline__ bool raw_out_of_domain_f32(const RawIn<double>& r) {
#ifdef MOD16_NO_GUARD
    return false;
#else
    const float y[13] = {(float)r.lw_d, (float)r.lw_n, (float)r.sw_d, (float)r.sw_n, (float)r.alb,
                         (float)r.qv_d * kGuardMixed, (float)r.qv_n * kGuardMixed, (float)r.elev * 4.f,
                         guard_scale_traw1((float)r.t_d), guard_scale_traw1((float)r.t_n),
                         guard_scale_p1((float)r.ps_d), guard_scale_p1((float)r.ps_n), (float)r.tmin};
    return guard_list_f32(y, 13);
#endif
}

// totals only (mod16/__init__.py:792: (canopy + soil) + transpiration)
__device__ __forceinline__ void et_pair_mixed(const float (&in)[14][2], const double* l0,
                                              const double* l1, int ls, const double* tb,
                                              f2& day, f2& night) {
    Parts2 d, n;
    et_pair_mixed_parts<false>(in, l0, l1, ls, tb, d, n);
    day = d.total;
    night = n.total;
}

}  // namespace mod16
