/*
 * mod16_hip.h -- C ABI of libmod16hip.so, the MI355X (gfx950) engine behind the
 * MOD16 forward run.
 *
 * The reference (arthur-e/MOD16 v1.2.0) is pure Python/numpy and has no FFI of
 * its own; the boundary this library replaces is the body of
 *     MOD16.evapotranspiration()            reference mod16/__init__.py:675-793
 * and the sub-methods it calls (:795-1258, :1261-1397), plus the per-pixel
 * parameter gather `bplut[key][pft_map]` that precedes it for multi-class
 * rasters (reference mod16/utils.py:81-117 + forward-run notebook cell 32).
 * The Python class `mod16_amd.MOD16` binds these entry points with ctypes
 * (see INTEGRATION.md for the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - plain C types only; no exceptions cross the ABI; every call returns
 *     MOD16_OK (0) or a negative MOD16_ERR_* code; mod16_last_error() has text;
 *   - the library never frees or retains caller memory;
 *   - `where` says whether the data pointers are HOST (pageable or pinned
 *     memory; the call stages tiles through the GPU and is synchronous) or
 *     DEVICE (zero-copy, asynchronous on `stream`, a hipStream_t or NULL);
 *     HOST calls of at most MOD16_SMALL_PIXELS pixels (environment, default
 *     65536, 0 = never; mod16_et_*, mod16_et_raw_*, mod16_method_*,
 *     mod16_et_static_*) issue no copy commands: the kernel reads its inputs
 *     from a page-locked buffer of the ctx and writes its outputs there (one
 *     launch, one synchronisation; same kernels, same results -- with ONE
 *     exception: a MOD16_MATH_MIXED call whose pixel count is no multiple of
 *     four computes its last, incomplete vector through the pipeline here and
 *     with the float64 (FAST) arithmetic on the staged path: those up to three
 *     pixels agree to the mixed form's tolerance, not bit for bit), and a class
 *     code >= 13 is found before anything is launched;
 *   - DEVICE launches of one ctx share its diagnostics workspace: on ONE stream
 *     they are ordered anyway and nothing is added between them; the first
 *     launch a ctx makes on a second stream waits for the device once, and
 *     from then on every launch records an event the next one on another
 *     stream waits for (a stream may be destroyed by its owner at any time:
 *     the library never touches a stream it is not launching on). That one
 *     wait is hipDeviceSynchronize(): other contexts of the process stall with
 *     it, and it fails (MOD16_ERR_HIP) while another thread captures a stream
 *     in hipStreamCaptureModeGlobal -- a program that does either gives every
 *     stream its own ctx, or launches once on each stream before it starts
 *     capturing;
 *   - a launch that did not process its whole raster (kStatusIncomplete, see
 *     mod16_check_status) is found by the kernels that run BEHIND the pipeline
 *     kernel: the guard's pass (every guarded launch) or the diagnostics sum
 *     (MOD16_DOMAIN_TRUSTED launches) -- a trusted launch without diagnostics
 *     is not checked;
 *   - a ctx serialises the calls made on it (every entry point holds the ctx's
 *     mutex), so sharing one between host threads is safe; for concurrency use
 *     one ctx per host thread and GPU (each owns its staging slabs, streams,
 *     BPLUT copy and workspace).
 */
#ifndef MOD16_HIP_H
#define MOD16_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOD16_ABI_VERSION 5

#if defined(__GNUC__)
#define MOD16_API __attribute__((visibility("default")))
#else
#define MOD16_API
#endif

/* MOD16.evapotranspiration() argument order, mod16/__init__.py:675-682 */
#define MOD16_N_DRIVERS 14
enum mod16_driver {
    MOD16_LW_NET_DAY = 0, MOD16_LW_NET_NIGHT, MOD16_SW_RAD_DAY,
    MOD16_SW_RAD_NIGHT, MOD16_SW_ALBEDO, MOD16_TEMP_DAY, MOD16_TEMP_NIGHT,
    MOD16_TEMP_ANNUAL, MOD16_TMIN, MOD16_VPD_DAY, MOD16_VPD_NIGHT,
    MOD16_PRESSURE, MOD16_FPAR, MOD16_LAI
};

/* MOD16.required_parameters order, mod16/__init__.py:152-155 */
#define MOD16_N_PARAMS 11
enum mod16_param {
    MOD16_TMIN_CLOSE = 0, MOD16_TMIN_OPEN, MOD16_VPD_OPEN, MOD16_VPD_CLOSE,
    MOD16_GL_SH, MOD16_GL_WV, MOD16_G_CUTICULAR, MOD16_CSL, MOD16_RBL_MIN,
    MOD16_RBL_MAX, MOD16_BETA
};

/* restore_bplut() arrays have 13 entries indexed by PFT code (utils.py:104) */
#define MOD16_N_CLASSES 13

/* order of the optional component outputs (`separate=True`, :789-790) */
#define MOD16_N_COMPONENTS 6
enum mod16_component {
    MOD16_CANOPY_DAY = 0, MOD16_SOIL_DAY, MOD16_TRANS_DAY,
    MOD16_CANOPY_NIGHT, MOD16_SOIL_NIGHT, MOD16_TRANS_NIGHT
};

enum mod16_status {
    MOD16_OK = 0,
    MOD16_ERR_ARG = -1,          /* NULL / misaligned / inconsistent argument */
    MOD16_ERR_HIP = -2,          /* a HIP runtime call failed                 */
    MOD16_ERR_CLASS_RANGE = -3,  /* a class code >= 13 (numpy: IndexError)    */
    MOD16_ERR_NOMEM = -4,
    MOD16_ERR_NO_DEVICE = -5,    /* no usable gfx950 device                   */
    MOD16_ERR_NO_BPLUT = -6      /* class raster given before a BPLUT was set */
};

enum mod16_where { MOD16_HOST = 0, MOD16_DEVICE = 1 };

/* flags of mod16_et_*
 *
 * MOD16_MATH_FAST rearranges the smooth arithmetic (conductances instead of resistances,
 * shared per-period terms, table exp / log) and keeps every comparison of the reference:
 * within 1e-9 of the reference-order kernel, NaN and exact-zero masks identical. Its own
 * domain is finite drivers of physical sign and magnitude (NaN anywhere, zeros and the usual
 * fill values in the radiation / albedo / VPD / fPAR / LAI fields included). A pixel outside
 * it is detected in the kernel (csrc/mod16_physics.hpp, fast_out_of_domain) and computed in the
 * reference's operation order instead. The test, exactly:
 *     |A_day| = |sw_rad_day (1 - albedo) + lw_net_day|, |lw_net_night|,
 *     |sw_rad_night (1 - albedo) + lw_net_night|, |fPAR|, |LAI|, |pressure|, |VPD day|,
 *     |VPD night| at or above 1e50 (or infinite);  pressure below 1 Pa (zero, negative);
 *     a day or night temperature at or outside 36 K .. 1332 K (35.85 K is the pole of the
 *     saturation formula, from 1332.4 K the latent heat is negative).
 * NaN operands never flag a pixel. With this the DEFAULT arithmetic returned what the
 * reference returns on every input class tested: a ladder of 56 magnitudes in each driver,
 * 1.2 M random pairs and a storm of independently special drivers (tests/fuzz_domain.py),
 * signalling-NaN patterns next to an infinity (tests/test_gpu_guard.py), NaN / zero / inf
 * masks included. The raw-driver forms test the raw fields instead: |specific humidity| < 1
 * kg/kg, |surface pressure| < 1e50 Pa, an elevation between -2000 m and 12000 m (the interval on
 * which the fast form evaluates air pressure as a polynomial, 3.7e-14 relative; MIXED: |z| < 25 km),
 * the radiation terms as above, and day / night temperatures between 190 K and 360 K (-83 C ..
 * +87 C, FAST and MIXED): on that interval the two saturation formulas of the path -- MOD16.vpd's
 * 610.7 exp(17.38 tc / (239 + tc)) and svp's 610.8 exp(17.27 tc / (237.3 + tc)) -- share one
 * exponential (their exponents differ by |delta| < 0.043, exp(delta) is a short polynomial; round 5).
 * MOD16_MATH_EXACT is the reference's operation order throughout. */
#define MOD16_MATH_FAST   0u  /* strength-reduced arithmetic (default)          */
#define MOD16_MATH_EXACT  1u  /* reference operation order, IEEE divide/pow     */
#define MOD16_MATH_MIXED  2u  /* float32 rasters: float64 where it decides a mask or
                                 feeds the humidity terms, packed float32 elsewhere
                                 (dense class rasters; other shapes run FAST). Its domain
                                 is physical drivers (|lw|, |sw|, |albedo|, |vpd|, |fpar|,
                                 |lai|, |tmin| < 1e5, 1e3 <= pressure < 1e7 Pa, 90 K < T < 1332 K);
                                 pixels outside it are computed in the reference's order,
                                 float64, like FAST's. Values whose wet-canopy or bare-soil
                                 numerator cancels below 1/320 of its terms relative to the
                                 period's total (one pixel in 700 of the synthetic grid) are
                                 computed again with the FAST form's float64 arithmetic behind
                                 the pipeline kernel (round 6): against that arithmetic no value
                                 of the global grid is off by more than 2.0e-4 of itself. With
                                 MOD16_DOMAIN_TRUSTED nothing is revisited, that tail included */
/* The caller vouches that every driver lies inside the domain above (quality-controlled or
 * NaN-masked rasters): the totals form of the production pipeline (dense class raster,
 * outputs day + night; mod16_et_*, mod16_et_diag_*, mod16_et_tiled_*, their graphs) then runs
 * the instance without the domain test and without the dispatch that revisits flagged pixels
 * (-1.6 % kernel time on the float64 global grid, round 5; MIXED: -7 %, of which 5 % are the
 * cancellation budgets of round 6: profiles/r06_mixed_cancel_threshold.txt). A pixel outside the domain then
 * gets whatever the rearranged arithmetic gives. Other forms and shapes ignore the flag. */
#define MOD16_DOMAIN_TRUSTED 4u

typedef struct mod16_ctx mod16_ctx;

MOD16_API int mod16_version(void);
MOD16_API const char* mod16_strerror(int status);
/* text of the last failure on this ctx ("" if none); valid until next call */
MOD16_API const char* mod16_last_error(const mod16_ctx* ctx);

MOD16_API int mod16_device_count(int* count);
MOD16_API int mod16_create(int device, mod16_ctx** out);
MOD16_API int mod16_destroy(mod16_ctx* ctx);

/*
 * Biome-properties look-up table: `lut` is a host array [13][11] of float64,
 * row = PFT code, column = enum mod16_param; rows of invalid classes hold NaN
 * (what restore_bplut() leaves at 0 and 11, utils.py:104-116). Replaces the
 * reference's per-pixel gather `params_dict[key][pft_map]`.
 */
MOD16_API int mod16_set_bplut_f64(mod16_ctx* ctx, const double* lut);

/*
 * The forward run over n pixels: replaces MOD16.evapotranspiration()
 * (mod16/__init__.py:675-793) and everything below it.
 *
 *   cls      uint8 class raster [n], or NULL. With a class raster the
 *            parameters come from the BPLUT (set beforehand); a code >= 13
 *            makes the call (HOST) or mod16_check_status() (DEVICE) return
 *            MOD16_ERR_CLASS_RANGE and yields NaN for that pixel.
 *   drivers  the 14 driver arrays, order enum mod16_driver.
 *   dstride  element stride of each driver: 1 = dense array of n elements,
 *            0 = one value broadcast to every pixel (numpy scalar).
 *   params   used only when cls == NULL: the 11 parameters, each dense
 *            (pstride 1, per-pixel `params_dict[key][pft_map]` arrays) or a
 *            broadcast scalar (pstride 0, the single-PFT MOD16(params) case).
 *   out_day, out_night   [n] totals in kg m-2 s-1 (:792); may be NULL if
 *            only components are wanted.
 *   out_sep  NULL, or 6 arrays [n] in enum mod16_component order (any entry
 *            may be NULL) -- the `separate=True` outputs (:790).
 *   flags    MOD16_MATH_*.
 */
MOD16_API int mod16_et_f64(mod16_ctx* ctx, const uint8_t* cls,
                 const double* const* drivers, const int64_t* dstride,
                 const double* const* params, const int64_t* pstride,
                 int64_t n, double* out_day, double* out_night,
                 double* const* out_sep, unsigned flags, int where,
                 void* stream);

/*
 * The same with the reference's 2-level broadcasting: its callers pass (N,)
 * arrays -- one value per site: pressure, temp_annual, the per-site parameters
 * `params_dict[key][pft_map]` -- against (T, N) drivers (mod16/__init__.py:180-181,
 * forward-run notebook cell 17), and numpy broadcasts them. Here every input
 * carries a broadcast kind instead of a 0 / 1 stride and nothing is made dense:
 * with the (T, N) raster flattened to n = T * inner pixels (inner = N),
 *
 *     MOD16_BC_SCALAR  one value                     element 0
 *     MOD16_BC_DENSE   [n]                           element i
 *     MOD16_BC_ROW     [inner]   an (N,) array       element i % inner
 *     MOD16_BC_COL     [n/inner] a (T, 1) array      element i / inner
 *
 * `cls_kind` is the kind of the class raster (a per-site PFT vector is a ROW).
 * A call with a ROW or COL input runs the one-pixel-per-thread kernel (in HOST
 * mode the small arrays are uploaded whole, once); without one it is mod16_et_*.
 */
enum mod16_broadcast { MOD16_BC_SCALAR = 0, MOD16_BC_DENSE = 1, MOD16_BC_ROW = 2, MOD16_BC_COL = 3 };
MOD16_API int mod16_et2_f64(mod16_ctx* ctx, const uint8_t* cls, int cls_kind,
                  const double* const* drivers, const int64_t* dkind,
                  const double* const* params, const int64_t* pkind,
                  int64_t inner, int64_t n, double* out_day, double* out_night,
                  double* const* out_sep, unsigned flags, int where, void* stream);
MOD16_API int mod16_et2_f32(mod16_ctx* ctx, const uint8_t* cls, int cls_kind,
                  const float* const* drivers, const int64_t* dkind,
                  const float* const* params, const int64_t* pkind,
                  int64_t inner, int64_t n, float* out_day, float* out_night,
                  float* const* out_sep, unsigned flags, int where, void* stream);

/* float32 data: MOD16_MATH_FAST widens to float64 on load, computes in float64 and
 * rounds once on store; MOD16_MATH_EXACT keeps float32 arithmetic in the reference's
 * operation order (what numpy does for the reference on float32 inputs);
 * MOD16_MATH_MIXED (dense class rasters: totals, components, potential ET, raw drivers): float64 for
 * the humidity terms (and for the radiation balance next to its clamps), packed float32
 * for the rest, every
 * decision behind a NaN or an exact zero made as in FAST; same masks as FAST, median
 * relative difference 1e-7, absolute difference below 1e-6 of the largest value,
 * 1.5x faster */
MOD16_API int mod16_et_f32(mod16_ctx* ctx, const uint8_t* cls,
                 const float* const* drivers, const int64_t* dstride,
                 const float* const* params, const int64_t* pstride,
                 int64_t n, float* out_day, float* out_night,
                 float* const* out_sep, unsigned flags, int where,
                 void* stream);

/*
 * HOST mode with diagnostics, and the unit the multi-GPU entry points of the Python layer deal
 * out. The numpy entry points (MOD16.evapotranspiration(), mod16/__init__.py:675-793, and
 * evapotranspiration_raster) stage host arrays through the GPU in tiles of
 * mod16_host_tile_pixels() pixels. mod16_et_hdiag_* is mod16_et_* with where = MOD16_HOST and
 * both totals, plus ONE diagnostics vector per staged tile: tile_diag is a HOST array
 * [ceil(n / mod16_host_tile_pixels())][8] (layout of mod16_reduce_diag_*), row t reduced on the
 * device, in a fixed order, from the outputs of pixels [t * tile, (t + 1) * tile) while they are
 * still there. Because a tile's row depends on that tile alone, a raster cut at tile boundaries
 * and dealt over several contexts / GPUs (one host thread per ctx, each writing its rows of the
 * one array) gives the same rows -- and, folded in tile order with mod16_fold_diag_host, the same
 * diagnostics bit for bit -- as the undivided call, whatever the number of GPUs (SURVEY.md 8e:
 * "tiles of the global grid shard embarrassingly", here for the PCIe-bound HOST paths, one link
 * per GPU, no collective).
 */
MOD16_API int64_t mod16_host_tile_pixels(void);
MOD16_API int mod16_et_hdiag_f64(mod16_ctx* ctx, const uint8_t* cls,
                       const double* const* drivers, const int64_t* dstride,
                       const double* const* params, const int64_t* pstride,
                       int64_t n, double* out_day, double* out_night,
                       unsigned flags, double* tile_diag);
MOD16_API int mod16_et_hdiag_f32(mod16_ctx* ctx, const uint8_t* cls,
                       const float* const* drivers, const int64_t* dstride,
                       const float* const* params, const int64_t* pstride,
                       int64_t n, float* out_day, float* out_night,
                       unsigned flags, double* tile_diag);
/* diag[8] = parts[0] (+) parts[1] (+) ... in the order given: sums and counts [0..5] added,
 * maxima [6..7] maximised (host arrays; the rule of mod16_fold_diag). Needs no ctx. */
MOD16_API int mod16_fold_diag_host(const double* parts, int64_t count, double* diag);

/*
 * Forward run that also returns potential ET: as
 * mod16_et_* (out_day / out_night may be NULL), plus pet_day / pet_night [n]
 * in kg m-2 s-1 = wet-canopy evaporation + saturated-soil evaporation +
 * unsaturated-soil evaporation without the soil-moisture constraint +
 * Priestley-Taylor potential transpiration (reference README.md:404-424;
 * MOD16.potential_soil_evaporation :449-544, MOD16.potential_transpiration
 * :546-602 with alpha = 1.26, evaporation_wet_canopy :866-961).
 */
MOD16_API int mod16_et_pet_f64(mod16_ctx* ctx, const uint8_t* cls,
                     const double* const* drivers, const int64_t* dstride,
                     const double* const* params, const int64_t* pstride,
                     int64_t n, double* out_day, double* out_night,
                     double* pet_day, double* pet_night, unsigned flags,
                     int where, void* stream);
MOD16_API int mod16_et_pet_f32(mod16_ctx* ctx, const uint8_t* cls,
                     const float* const* drivers, const int64_t* dstride,
                     const float* const* params, const int64_t* pstride,
                     int64_t n, float* out_day, float* out_night,
                     float* pet_day, float* pet_night, unsigned flags,
                     int where, void* stream);

/*
 * Forward run on raw drivers: the pre-processing the
 * reference does in front of the forward run (mod16/calibration.py:380-423) is
 * folded into the pixel kernel --
 *   vpd_day   = MOD16.vpd(qv10m_day, ps_day, temp_day)            :604-644
 *   vpd_night = max(MOD16.vpd(qv10m_night, ps_night, temp_night), 0)   calibration.py:398-401
 *   pressure  = MOD16.air_pressure(elevation)                      :414-447
 *   fpar = fpar_pct / 100, lai = lai_x10 / 10                      calibration.py:422-423
 * with fPAR / LAI given as the MODIS uint8 encodings (codes >= 249 = fill ->
 * NaN). `raw` holds 14 arrays in enum mod16_raw_driver order (rstride 0/1 as
 * elsewhere); cls + BPLUT give the parameters. Outputs: out_day / out_night
 * [kg m-2 s-1] and/or, with `day_hours` (hours of daylight per pixel, hstride
 * 0/1), out_total8 = (day h + night (24 - h)) * 8 * 3600 [kg m-2 (8 d)-1], the
 * MOD16A2 unit (reference tests/verification/verify2.py:113-115).
 */
#define MOD16_N_RAW_DRIVERS 14
enum mod16_raw_driver {
    MOD16_RAW_LW_NET_DAY = 0, MOD16_RAW_LW_NET_NIGHT, MOD16_RAW_SW_RAD_DAY,
    MOD16_RAW_SW_RAD_NIGHT, MOD16_RAW_SW_ALBEDO, MOD16_RAW_TEMP_DAY,
    MOD16_RAW_TEMP_NIGHT, MOD16_RAW_TEMP_ANNUAL, MOD16_RAW_TMIN,
    MOD16_RAW_QV10M_DAY, MOD16_RAW_QV10M_NIGHT, MOD16_RAW_PS_DAY,
    MOD16_RAW_PS_NIGHT, MOD16_RAW_ELEVATION
};
MOD16_API int mod16_et_raw_f64(mod16_ctx* ctx, const uint8_t* cls,
                     const double* const* raw, const int64_t* rstride,
                     const uint8_t* fpar_pct, const uint8_t* lai_x10,
                     const double* day_hours, int64_t hstride, int64_t n,
                     double* out_day, double* out_night, double* out_total8,
                     unsigned flags, int where, void* stream);
MOD16_API int mod16_et_raw_f32(mod16_ctx* ctx, const uint8_t* cls,
                     const float* const* raw, const int64_t* rstride,
                     const uint8_t* fpar_pct, const uint8_t* lai_x10,
                     const float* day_hours, int64_t hstride, int64_t n,
                     float* out_day, float* out_night, float* out_total8,
                     unsigned flags, int where, void* stream);

/*
 * Forward run and diagnostics in one pass, DEVICE pointers only: as
 * mod16_et_* with a class raster (BPLUT parameters) and both totals, plus the
 * diagnostics vector of mod16_reduce_diag_* written to ddiag (device, 8
 * doubles). On the production path (dense 16-byte-aligned drivers, n a
 * multiple of the vector width) the outputs are reduced while still in
 * registers, so the extra 16 B/pixel read of a separate reduction is saved;
 * otherwise the call runs the forward run and then the reduction. The sums
 * are deterministic for a given n and device. Asynchronous on `stream`.
 */
MOD16_API int mod16_et_diag_f64(mod16_ctx* ctx, const uint8_t* cls,
                      const double* const* drivers, const int64_t* dstride,
                      int64_t n, double* out_day, double* out_night,
                      unsigned flags, double* ddiag, void* stream);
MOD16_API int mod16_et_diag_f32(mod16_ctx* ctx, const uint8_t* cls,
                      const float* const* drivers, const int64_t* dstride,
                      int64_t n, float* out_day, float* out_night,
                      unsigned flags, double* ddiag, void* stream);

/*
 * mod16_et_diag_* for a raster that is processed again and again (one call per
 * time step of a series): the launch sequence -- ticket-counter reset, pipeline
 * kernel, staged fixed-order sum of the diagnostics -- is captured once into a
 * HIP graph and replayed by mod16_graph_launch on any stream. All pointers are
 * DEVICE pointers and must stay valid and in place; the raster's contents may
 * change between launches. Replays of one graph must be ordered (one stream, or
 * events); deferred errors surface through mod16_check_status as usual.
 * A graph belongs to the context it was captured with -- its kernels read the
 * context's parameter and exp / log tables and report into its status word:
 * destroy the graph first. Once the context is gone, mod16_graph_launch and
 * mod16_time_graph return MOD16_ERR_ARG (nothing is launched);
 * mod16_graph_destroy still frees the graph, in either order.
 */
typedef struct mod16_graph mod16_graph;
MOD16_API int mod16_graph_et_diag_f64(mod16_ctx* ctx, const uint8_t* cls,
                      const double* const* drivers, const int64_t* dstride,
                      int64_t n, double* out_day, double* out_night,
                      unsigned flags, double* ddiag, mod16_graph** out);
MOD16_API int mod16_graph_et_diag_f32(mod16_ctx* ctx, const uint8_t* cls,
                      const float* const* drivers, const int64_t* dstride,
                      int64_t n, float* out_day, float* out_night,
                      unsigned flags, double* ddiag, mod16_graph** out);
MOD16_API int mod16_graph_launch(mod16_graph* graph, void* stream);
MOD16_API int mod16_graph_destroy(mod16_graph* graph);

/*
 * The sub-methods of the reference's class surface (mod16/__init__.py:384-673,
 * :795-1258, :1261-1397), reference operation order. `method` selects one;
 * `in` holds MOD16_METHOD_MAX_IN pointers in the order of the reference
 * signature (table below), NULL = optional argument not given (then computed as
 * the reference does); istride 0 = broadcast scalar, 1 = dense; `params` as in
 * mod16_et_* (may be NULL for the static / module-level methods); `out` holds 2
 * pointers (second NULL unless the method returns a pair); `alpha` is used by
 * POT_TRANSPIRATION only, `tiny` (the reference's argument of that name,
 * :869, :1157; default 1e-7) by EVAP_WET_CANOPY and TRANSPIRATION_*. `where` as
 * in mod16_et_*.
 *
 *   SVP                 temp_k                                     :1340
 *   SVP_SLOPE           temp_k, s?                                 :1370
 *   LHV                 temp_k                                     :121
 *   PSYCHROMETRIC       pressure, temp_k                           :1261
 *   RADIATION_NET       sw_rad, sw_albedo, temp_k                  :1293
 *   AIR_DENSITY         temp_k, pressure, rhumidity                :384
 *   AIR_PRESSURE        elevation_m                                :414
 *   VPD                 qv10m, pressure, tmean                     :604
 *   RHUMIDITY           temp_k, vpd                                :646
 *   POT_SOIL_EVAP       pressure, temp_k, vpd, fpar, rad_soil, r_corr?, lhv?, rh?, f_wet? -> (sat, unsat)  :449
 *   POT_TRANSPIRATION   lw_net, sw_rad, sw_albedo, pressure, temp_k, vpd, fpar, rh?, f_wet?       :546
 *   EVAP_SOIL           pressure, temp_k, vpd, fpar, rad_soil, r_corr?, lhv?, rh?, f_wet?         :795
 *   EVAP_WET_CANOPY     pressure, temp_k, vpd, lai, fpar, rad_canopy, lhv?, rh?, f_wet?           :866
 *   RADIATION_SOIL      lw_d, lw_n, sw_d, sw_n, albedo, t_d, t_n, t_annual, fpar -> (day, night)  :963
 *   SOIL_HEAT_FLUX      rad_net_day, rad_net_night, t_d, t_n, t_annual -> (day, night)            :1055
 *   SURFACE_CONDUCTANCE tmin, vpd_day                              :1121
 *   TRANSPIRATION_DAY / _NIGHT  pressure, temp_k, vpd, lai, fpar, rad_canopy, tmin, r_corr?, lhv?, rh?, f_wet?  :1152
 */
#define MOD16_METHOD_MAX_IN 13
enum mod16_method {
    MOD16_M_SVP = 0, MOD16_M_SVP_SLOPE, MOD16_M_LHV, MOD16_M_PSYCHROMETRIC,
    MOD16_M_RADIATION_NET, MOD16_M_AIR_DENSITY, MOD16_M_AIR_PRESSURE, MOD16_M_VPD,
    MOD16_M_RHUMIDITY, MOD16_M_POT_SOIL_EVAP, MOD16_M_POT_TRANSPIRATION,
    MOD16_M_EVAP_SOIL, MOD16_M_EVAP_WET_CANOPY, MOD16_M_RADIATION_SOIL,
    MOD16_M_SOIL_HEAT_FLUX, MOD16_M_SURFACE_CONDUCTANCE,
    MOD16_M_TRANSPIRATION_DAY, MOD16_M_TRANSPIRATION_NIGHT, MOD16_M_COUNT
};
MOD16_API int mod16_method_f64(mod16_ctx* ctx, int method,
                     const double* const* in, const int64_t* istride,
                     const double* const* params, const int64_t* pstride,
                     int64_t n, double* const* out, double alpha, double tiny,
                     int where, void* stream);
MOD16_API int mod16_method_f32(mod16_ctx* ctx, int method,
                     const float* const* in, const int64_t* istride,
                     const float* const* params, const int64_t* pstride,
                     int64_t n, float* const* out, float alpha, float tiny,
                     int where, void* stream);

/*
 * The vectorised calibration path MOD16._evapotranspiration (reference
 * mod16/__init__.py:195-382; MOD16._et :162-193 is out_day + out_night):
 * latent heat flux [W m-2] for day and night, reference operation order. It
 * is a different algorithm from mod16_et_* (other clamps, tmin_open in the
 * soil-heat-flux condition, transpiration switched for the whole array on
 * any(g_surf > 0)). `rcorr` = NULL or two arrays (day, night), the reference's
 * `r_corr_list`; `tiny` is the reference's argument of that name (:199,
 * default 1e-7); strides as elsewhere. HOST or DEVICE; synchronous in HOST.
 */
MOD16_API int mod16_et_static_f64(mod16_ctx* ctx, const double* const* drivers,
                        const int64_t* dstride, const double* const* params,
                        const int64_t* pstride, const double* const* rcorr,
                        const int64_t* rstride, int64_t n, double* out_day,
                        double* out_night, double tiny, int where, void* stream);
MOD16_API int mod16_et_static_f32(mod16_ctx* ctx, const float* const* drivers,
                        const int64_t* dstride, const float* const* params,
                        const int64_t* pstride, const float* const* rcorr,
                        const int64_t* rstride, int64_t n, float* out_day,
                        float* out_night, float tiny, int where, void* stream);

/*
 * The calibration path batched over parameter vectors:
 * MOD16._evapotranspiration / MOD16._et (reference mod16/__init__.py:162-382)
 * for `ndraw` parameter vectors over the same n pixels in one call -- what the
 * reference's MCMC sampler (calibration.py:907) and Sobol analysis
 * (sensitivity.py:95) evaluate draw by draw. `params` is [ndraw][11] row-major
 * in mod16_param order; outputs are [ndraw][n] row-major, each may be NULL:
 * out_day / out_night [W m-2] and out_total = day + night (MOD16._et). With
 * flags = MOD16_MATH_EXACT row d is bit-identical to mod16_et_static_* called
 * with the scalars params[d] (r_corr computed, `rcorr` = NULL); MOD16_MATH_FAST
 * uses the strength-reduced arithmetic of the forward run (float64 throughout,
 * within 1e-9 of EXACT, same NaN and zero masks, several times faster); pixels outside that
 * arithmetic's domain (the test of MOD16_MATH_FAST above, on the drivers) are left out by the
 * FAST kernels and computed in the reference's operation order behind them, so FAST returns
 * what EXACT returns for them (here also |VPD| >= 1e18 Pa: this path does not clamp the relative
 * humidity from above, mod16/__init__.py:280-281). With `observed` [n] (and optional
 * `weights` [n]) the call also reduces each draw to
 *     sse[d]   = sum_i (weights[i] * (out_total[d][i] - observed[i]))^2
 *     count[d] = number of pairs used (NaN pairs are skipped),
 * both float64 [ndraw], summed in a fixed order (deterministic); the caller
 * forms its objective (e.g. RMSD = sqrt(sse / count)) from them. In DEVICE mode
 * out_total must be given when sse is (it is the workspace). HOST: synchronous.
 */
MOD16_API int mod16_et_static_batch_f64(mod16_ctx* ctx, const double* const* drivers,
                        const int64_t* dstride, int64_t n, const double* params,
                        int64_t ndraw, double* out_day, double* out_night,
                        double* out_total, const double* observed,
                        const double* weights, double* sse, double* count,
                        unsigned flags, int where, void* stream);
MOD16_API int mod16_et_static_batch_f32(mod16_ctx* ctx, const float* const* drivers,
                        const int64_t* dstride, int64_t n, const float* params,
                        int64_t ndraw, float* out_day, float* out_night,
                        float* out_total, const float* observed,
                        const float* weights, double* sse, double* count,
                        unsigned flags, int where, void* stream);

/*
 * The same problem RESIDENT on the device: what the reference's MCMC sampler
 * (calibration.py:907-909) and Sobol analysis (sensitivity.py:94-96) actually do is evaluate
 * MOD16._et thousands of times on the SAME drivers with new parameter vectors.
 * mod16_static_batch_bind_* takes the 14 drivers (dstride 0 / 1 as above), the observations
 * and optional weights once -- where = MOD16_HOST: copied to the device; MOD16_DEVICE: the
 * caller's device arrays are used in place and must outlive the object -- marks the pixels
 * outside the FAST domain, and sizes a workspace for up to `max_draws` parameter vectors.
 *   mod16_static_batch_objective  params [ndraw][11] (HOST, the problem's data type) ->
 *       sse[ndraw], count[ndraw] (HOST float64, as defined above). One evaluation = the
 *       parameters up (ndraw x 88 bytes), ONE graph launch, 16 bytes per draw down; nothing of
 *       size [ndraw][n] exists: the residuals are reduced where they are computed, in a fixed
 *       order (lanes, waves, blocks: the result does not depend on the schedule). The
 *       reference's whole-array branch any(g_surf > 0) (mod16/__init__.py:343-348) is resolved
 *       inside the same launch sequence: every draw is evaluated with transpiration while the
 *       blocks report whether any pixel has g_surf > 0, and the (normally zero) draws without it
 *       are evaluated again without. flags = MOD16_MATH_EXACT at bind time: the kernels of the
 *       unbound call on the resident drivers (bit-identical sums to mod16_et_static_batch_*).
 *   mod16_static_batch_rows       the [ndraw][n] rows (HOST; day / night / total, any may be
 *       NULL): the unbound call's kernels on the resident drivers -- bit-identical rows.
 *   mod16_static_batch_info       n, max_draws and the number of pixels outside the FAST domain.
 *   mod16_static_batch_time       mean milliseconds of the GPU part of the last-shaped
 *       objective evaluation (graph replays bracketed by HIP events).
 * Calls on one object are serialised by its ctx's mutex. Synchronous.
 *
 * where = MOD16_DEVICE: the object works on a private stream and the interface takes none, so the
 * caller's arrays must be COMPLETE when mod16_static_batch_bind_* is called (synchronise the stream
 * that produced them first) and must not change while the object lives -- the pixels outside the
 * FAST domain are listed once, at bind time, from the values found then. Device memory that cannot
 * be had is MOD16_ERR_NOMEM (bind: resident copies, workspace for max_draws parameter vectors -- about
 * 190 bytes per draw; objective: the per-block partials, 20 bytes x draws x ceil(n / 256), sized for
 * the draws actually evaluated and grown on demand -- a problem bound for max_draws = 4096 that is
 * evaluated 256 draws at a time holds a sixteenth of what 4096 would take; EXACT problems: none).
 */
typedef struct mod16_batch mod16_batch;
MOD16_API int mod16_static_batch_bind_f64(mod16_ctx* ctx, const double* const* drivers,
                        const int64_t* dstride, int64_t n, const double* observed,
                        const double* weights, int64_t max_draws, unsigned flags, int where,
                        mod16_batch** out);
MOD16_API int mod16_static_batch_bind_f32(mod16_ctx* ctx, const float* const* drivers,
                        const int64_t* dstride, int64_t n, const float* observed,
                        const float* weights, int64_t max_draws, unsigned flags, int where,
                        mod16_batch** out);
MOD16_API int mod16_static_batch_objective(mod16_batch* problem, const void* params, int64_t ndraw,
                        double* sse, double* count);
MOD16_API int mod16_static_batch_rows(mod16_batch* problem, const void* params, int64_t ndraw,
                        void* out_day, void* out_night, void* out_total);
MOD16_API int mod16_static_batch_info(const mod16_batch* problem, int64_t* n, int64_t* max_draws,
                        int64_t* n_outside_domain);
MOD16_API int mod16_static_batch_time(mod16_batch* problem, int launches, float* ms);
MOD16_API int mod16_static_batch_destroy(mod16_batch* problem);

/*
 * Waits for the ctx's outstanding work on `stream` and reports deferred
 * errors of DEVICE-mode calls (MOD16_ERR_CLASS_RANGE, MOD16_ERR_HIP).
 */
MOD16_API int mod16_check_status(mod16_ctx* ctx, void* stream);

/*
 * Diagnostics of a (day, night) result pair [n] (device pointers): a
 * deterministic two-level tree sum in fixed order. diag (host, 8 doubles) =
 * { sum_day, sum_night, n_finite_day, n_finite_night, n_nan_day, n_nan_night,
 *   max_day, max_night }, NaN pixels skipped as np.nansum would. Synchronous.
 * ddiag (device, 8 doubles, may be NULL) receives the same vector
 * asynchronously for an RCCL all-reduce without a host round trip; pass
 * diag == NULL to stay asynchronous.
 */
MOD16_API int mod16_reduce_diag_f64(mod16_ctx* ctx, const double* day,
                          const double* night, int64_t n, double* diag,
                          double* ddiag, void* stream);
MOD16_API int mod16_reduce_diag_f32(mod16_ctx* ctx, const float* day,
                          const float* night, int64_t n, double* diag,
                          double* ddiag, void* stream);

/*
 * On-device synthetic driver fields: fills the class
 * raster and the 14 dense driver arrays (device pointers, n elements each)
 * for global pixel indices [pixel_offset, pixel_offset + n) of time step
 * `step` from a counter-based generator keyed on (seed, step, variable,
 * pixel), so any tiling of a raster over any number of GPUs sees the same
 * field. Asynchronous on `stream`.
 */
MOD16_API int mod16_synth_f64(mod16_ctx* ctx, uint64_t seed, int64_t step,
                    int64_t pixel_offset, int64_t n, uint8_t* cls,
                    double* const* drivers, void* stream);
MOD16_API int mod16_synth_f32(mod16_ctx* ctx, uint64_t seed, int64_t step,
                    int64_t pixel_offset, int64_t n, uint8_t* cls,
                    float* const* drivers, void* stream);

/*
 * Timing aid for bench.py: runs `launches` back-to-back launches of the
 * DEVICE-mode f64/f32 forward run on `stream`, bracketed by HIP events on
 * that stream, and returns the mean milliseconds per launch in *ms.
 * Arguments as mod16_et_f64 / mod16_et_f32 (is_f32 selects), where = DEVICE;
 * with ddiag != NULL the launches are those of mod16_et_diag_*.
 */
MOD16_API int mod16_time_et(mod16_ctx* ctx, int is_f32, const uint8_t* cls,
                  const void* const* drivers, const int64_t* dstride,
                  const void* const* params, const int64_t* pstride,
                  int64_t n, void* out_day, void* out_night,
                  void* const* out_sep, unsigned flags, double* ddiag,
                  int launches, void* stream, float* ms);

/*
 * Tiled rasters: the engine's own layout for rasters that stay resident on the
 * device (DEVICE pointers only). The reference passes 16 separate arrays
 * (14 drivers, day, night); streamed side by side they lie GiB apart in HBM and
 * the 14-read + 2-write mix reaches 5.7 TB/s. Cut into tiles of `tile` pixels
 * and interleaved -- [tile][field][tile pixels], every array keeps its own base
 * pointer and reads as a 2-D strided view -- the same bytes stream at 6.5 TB/s
 * (tools/probe_layout.hip, tile = 32-64 KiB per field). Pixel i of an array:
 *
 *     base[(i / tile) * row + (i % tile)]
 *
 * `tile` is a power of two (>= 8 KiB per field); `driver_row`, `out_row` are in
 * elements and apply to every driver / output array, `cls_row` in bytes to the
 * class raster; rows are multiples of the 16-byte vector width, bases 16-byte
 * aligned, n a multiple of the vector width (the storage is padded to whole
 * tiles by whoever allocates it). tile = 0 in mod16_synth_tiled_* means plain
 * arrays. mod16_et_tiled_* = mod16_et_diag_* on that layout (ddiag may be NULL;
 * MOD16_MATH_FAST or, float32, MOD16_MATH_MIXED); mod16_graph_et_tiled_* captures
 * it for replay with mod16_graph_launch. Host arrays reach the layout with 2-D
 * copies (hipMemcpy2DAsync, width = tile, destination pitch = row).
 * Limits of one pipeline launch (any layout; MOD16_ERR_ARG beyond them): 2^30
 * pieces of 64 vectors -- 1.4e11 float64 or 2.7e11 float32 pixels, three orders of
 * magnitude above what 288 GB hold -- and rows below 2^32 elements: the kernel
 * keeps its piece numbers and tile offsets in single 32-bit scalar registers.
 */
typedef struct mod16_layout {
    int64_t tile;        /* pixels per tile */
    int64_t driver_row;  /* elements between successive tiles of a driver array */
    int64_t out_row;     /* elements between successive tiles of an output array */
    int64_t cls_row;     /* bytes between successive tiles of the class raster */
} mod16_layout;
MOD16_API int mod16_et_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout,
                       const uint8_t* cls, const double* const* drivers, int64_t n,
                       double* out_day, double* out_night, unsigned flags,
                       double* ddiag, void* stream);
MOD16_API int mod16_et_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout,
                       const uint8_t* cls, const float* const* drivers, int64_t n,
                       float* out_day, float* out_night, unsigned flags,
                       double* ddiag, void* stream);
MOD16_API int mod16_graph_et_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout,
                       const uint8_t* cls, const double* const* drivers, int64_t n,
                       double* out_day, double* out_night, unsigned flags,
                       double* ddiag, mod16_graph** out);
MOD16_API int mod16_graph_et_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout,
                       const uint8_t* cls, const float* const* drivers, int64_t n,
                       float* out_day, float* out_night, unsigned flags,
                       double* ddiag, mod16_graph** out);
MOD16_API int mod16_synth_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout,
                       uint64_t seed, int64_t step, int64_t pixel_offset, int64_t n,
                       uint8_t* cls, double* const* drivers, void* stream);
MOD16_API int mod16_synth_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout,
                       uint64_t seed, int64_t step, int64_t pixel_offset, int64_t n,
                       uint8_t* cls, float* const* drivers, void* stream);

/*
 * The other forms of the forward run on a tiled raster -- what mod16_et_pet_*,
 * mod16_et_* with out_sep and mod16_et_raw_* compute on plain arrays, on the
 * layout above (device pointers, MOD16_MATH_FAST or, float32, MOD16_MATH_MIXED;
 * asynchronous on `stream`). `wide`: the form's driver arrays (driver_row applies to
 * each), `bytes`: its byte rasters (cls_row applies to each), `outs`: its outputs
 * (out_row applies to each); every array of the form is required:
 *
 *   form                          wide                      bytes                     outs
 *   MOD16_FORM_TOTALS             14 drivers                cls                       day, night
 *   MOD16_FORM_PET                14 drivers                cls                       day, night, pet day, pet night
 *   MOD16_FORM_COMPONENTS         14 drivers                cls                       canopy, soil, transpiration (day), then (night)
 *   MOD16_FORM_TOTALS_COMPONENTS  14 drivers                cls                       day, night, then the six components
 *   MOD16_FORM_RAW                14 raw (mod16_raw_driver) cls, fpar_pct, lai_x10    day, night
 *   MOD16_FORM_RAW_TOTAL8         14 raw                    cls, fpar_pct, lai_x10    day, night, total8 (day_hours: one value)
 *   MOD16_FORM_RAW_TOTAL8_HOURS   14 raw + hours of daylight cls, fpar_pct, lai_x10   day, night, total8
 *
 * mod16_form_shape() returns the three counts of a form.
 */
enum mod16_form {
    MOD16_FORM_TOTALS = 0, MOD16_FORM_PET, MOD16_FORM_COMPONENTS, MOD16_FORM_TOTALS_COMPONENTS,
    MOD16_FORM_RAW, MOD16_FORM_RAW_TOTAL8, MOD16_FORM_RAW_TOTAL8_HOURS
};
MOD16_API int mod16_form_shape(int form, int* n_wide, int* n_bytes, int* n_out);
MOD16_API int mod16_et_form_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout, int form,
                       const uint8_t* const* bytes, const double* const* wide,
                       double* const* outs, double day_hours, int64_t n,
                       unsigned flags, void* stream);
MOD16_API int mod16_et_form_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout, int form,
                       const uint8_t* const* bytes, const float* const* wide,
                       float* const* outs, double day_hours, int64_t n,
                       unsigned flags, void* stream);

/* mod16_time_et for a tiled raster: `launches` back-to-back direct (not captured)
 * launches of mod16_et_tiled_*, HIP events on `stream`, mean milliseconds per launch.
 * For steps shorter than a graph replay's fixed cost (a 1200 x 1200 tile) this is
 * the faster way to issue them, and how bench.py times configs[1]. */
MOD16_API int mod16_time_et_tiled(mod16_ctx* ctx, int is_f32, const mod16_layout* layout,
                       const uint8_t* cls, const void* const* drivers, int64_t n,
                       void* out_day, void* out_night, unsigned flags, double* ddiag,
                       int launches, void* stream, float* ms);

/* Mean milliseconds per replay of a captured step: `launches` back-to-back
 * mod16_graph_launch calls bracketed by HIP events on `stream`. Synchronous. */
MOD16_API int mod16_time_graph(mod16_graph* graph, int launches, void* stream, float* ms);

/*
 * Page-locked host memory for the arrays a HOST-mode call writes its results to.
 * The reference returns freshly allocated arrays (mod16/__init__.py:789-793); a
 * device-to-host copy into fresh pageable memory is bound by the kernel's
 * page-fault rate (13 GB/s measured), into page-locked memory by PCIe (57 GB/s).
 * The Python layer keeps a bounded pool of such blocks behind the numpy arrays
 * it returns. Not tied to a ctx; thread-safe.
 */
MOD16_API int mod16_host_alloc(int64_t bytes, void** out);
MOD16_API int mod16_host_free(void* p);

/*
 * Measurement aid for bench.py ("roofline vs measured
 * copy bandwidth"): allocates two device buffers of `bytes`, times a one-shot
 * 16-byte-per-lane copy kernel `reps` times with HIP events and returns the
 * best rate, (bytes read + bytes written) / time, in GB/s. Synchronous.
 */
MOD16_API int mod16_measure_copy(mod16_ctx* ctx, int64_t bytes, int reps, float* gbps);

/*
 * Identity of this build: a hex digest of the library's sources and compiler flags, put in
 * by mod16_amd/csrc/build.py. profiles/ records of a kernel (HBM traffic from the PMC passes)
 * carry it, and bench.py reports such a record only for the build it was measured on.
 */
MOD16_API const char* mod16_build_id(void);

/*
 * The rank-order fold behind the all-gather of the diagnostics vectors: `gathered`
 * is [world][8] float64 on the device (rank r's vector at row r), `diag` [8] receives sums and
 * counts [0..5] added in rank order (rank 0 + rank 1 + ...: the same bits on every rank and
 * from run to run) and the maxima [6..7]. One small kernel, asynchronous on `stream`.
 */
MOD16_API int mod16_fold_diag(mod16_ctx* ctx, const double* gathered, int world, double* diag,
                              void* stream);

/*
 * Parameter rasters -> class raster (round 6). The reference's multi-class idiom gathers the BPLUT
 * per pixel -- MOD16({k: bplut[k][pft_map]}), mod16/utils.py:81-117 and notebook cell 32 -- and hands
 * evapotranspiration() eleven parameter rasters that hold, pixel for pixel, one of at most 13 rows.
 * mod16_classify_* turns DEVICE rasters of that kind back into a class raster: params[11] (order of
 * MOD16.required_parameters) with pstride 1 (a raster of n values) or 0 (one value); `rows` a HOST
 * array [nrows][11] of candidate rows, 1 <= nrows <= 13; every pixel whose eleven values equal a row
 * bit for bit (a NaN row matches itself) gets that row's index in cls[n] (device). *unmatched
 * receives -1 if every pixel matched, else the smallest index of a pixel that matched no row (the
 * caller adds that pixel's row and calls again, or gives up: the rasters are not a gather of <= 13
 * rows). Waits for `stream`. What the Python layer does with it: MOD16.evapotranspiration on device
 * tensors with per-pixel parameter tensors takes the production pipeline (14 drivers + 1 byte per
 * pixel) instead of the plain kernel (25 arrays).
 */
MOD16_API int mod16_classify_f64(mod16_ctx* ctx, const double* const* params, const int64_t* pstride,
                                 int64_t n, const double* rows, int nrows, uint8_t* cls,
                                 int64_t* unmatched, void* stream);
MOD16_API int mod16_classify_f32(mod16_ctx* ctx, const float* const* params, const int64_t* pstride,
                                 int64_t n, const float* rows, int nrows, uint8_t* cls,
                                 int64_t* unmatched, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOD16_HIP_H */
