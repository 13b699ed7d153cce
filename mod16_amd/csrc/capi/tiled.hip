// libmod16hip.so -- resident tiled rasters: mod16_et_tiled_*, mod16_et_form_tiled_*, their graphs and timers
#include "internal.hpp"

// ------------------------------------------------ tiled rasters (device resident)
// The production pipeline on the engine's own raster layout: fields interleaved in
// tiles ([tile][field][tile pixels]) so that the 16 streams of a wave lie within one
// ~1 MiB block of HBM instead of 16 places GiB apart (tools/probe_layout.hip: 6.5 TB/s
// against 5.7 for the same bytes).
template <typename T>
static int tiled_entry(mod16_ctx* ctx, const mod16_layout* lay, const uint8_t* cls,
                       const T* const* drivers, int64_t n, T* out_day, T* out_night,
                       unsigned flags, double* ddiag, void* stream) {
    constexpr int V = VecOf<T>::v;
    if (!ctx) return MOD16_ERR_ARG;
    if (!lay || !cls || !drivers || !out_day || !out_night || n < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_tiled: NULL argument or n < 0");
    if (!ctx->have_lut) return fail(ctx, MOD16_ERR_NO_BPLUT, "mod16_et_tiled: mod16_set_bplut_f64 was not called");
    if (flags & MOD16_MATH_EXACT) return fail(ctx, MOD16_ERR_ARG, "mod16_et_tiled: MOD16_MATH_EXACT runs on plain arrays only");
    const int px_shift = tile_log2(lay->tile, (int64_t)64 * V * kDynRun);
    if (px_shift < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et_tiled: tile must be a power of two of at least 8 KiB per field");
    auto al16 = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    bool ok = al16(out_day) && al16(out_night) && reinterpret_cast<uintptr_t>(cls) % V == 0 &&
              lay->driver_row >= lay->tile && lay->out_row >= lay->tile && lay->cls_row >= lay->tile &&
              lay->driver_row % V == 0 && lay->out_row % V == 0 && lay->cls_row % V == 0 && n % V == 0;
    for (int k = 0; k < 14 && ok; ++k) ok = drivers[k] && al16(drivers[k]);
    if (!ok) return fail(ctx, MOD16_ERR_ARG, "mod16_et_tiled: arrays must be 16-byte aligned, rows >= tile and multiples of the vector width, n a multiple of it");
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    StreamArgs<T> s;
    memset(&s, 0, sizeof s);
    for (int k = 0; k < 14; ++k) s.wide[k] = drivers[k];
    s.bytes[0] = cls;
    s.out[0] = out_day;
    s.out[1] = out_night;
    s.n = n;
    int pv = 0;
    while ((1 << pv) < 64 * V) ++pv;
    s.tile_shift = px_shift - pv;             // pieces per tile
    s.wide_row = lay->driver_row;
    s.out_row = lay->out_row;
    s.byte_row = lay->cls_row;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = launch_totals<T>(ctx, s, st, ddiag, flags);
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    return MOD16_OK;
}

extern "C" int mod16_et_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout, const uint8_t* cls,
                                  const double* const* drivers, int64_t n, double* out_day,
                                  double* out_night, unsigned flags, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return tiled_entry<double>(ctx, layout, cls, drivers, n, out_day, out_night, flags, ddiag, stream);
}
extern "C" int mod16_et_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout, const uint8_t* cls,
                                  const float* const* drivers, int64_t n, float* out_day,
                                  float* out_night, unsigned flags, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return tiled_entry<float>(ctx, layout, cls, drivers, n, out_day, out_night, flags, ddiag, stream);
}

// The other forms of the forward run (potential ET, components, raw drivers) on the same
// layout: `wide` holds the form's 16-byte-per-lane input arrays, `bytes` its byte rasters
// (class raster first), `outs` its outputs, in the order of mod16_form_shape().
static bool form_shape(int form, int* nw, int* nb, int* no) {
    switch (form) {
    case MOD16_FORM_TOTALS: *nw = 14; *nb = 1; *no = 2; return true;
    case MOD16_FORM_PET: *nw = 14; *nb = 1; *no = 4; return true;
    case MOD16_FORM_COMPONENTS: *nw = 14; *nb = 1; *no = 6; return true;
    case MOD16_FORM_TOTALS_COMPONENTS: *nw = 14; *nb = 1; *no = 8; return true;
    case MOD16_FORM_RAW: *nw = 14; *nb = 3; *no = 2; return true;
    case MOD16_FORM_RAW_TOTAL8: *nw = 14; *nb = 3; *no = 3; return true;
    case MOD16_FORM_RAW_TOTAL8_HOURS: *nw = 15; *nb = 3; *no = 3; return true;
    }
    return false;
}

extern "C" int mod16_form_shape(int form, int* n_wide, int* n_bytes, int* n_out) {
    int nw, nb, no;
    if (!form_shape(form, &nw, &nb, &no)) return MOD16_ERR_ARG;
    if (n_wide) *n_wide = nw;
    if (n_bytes) *n_bytes = nb;
    if (n_out) *n_out = no;
    return MOD16_OK;
}

template <typename T>
static int form_tiled_entry(mod16_ctx* ctx, const mod16_layout* lay, int form,
                            const uint8_t* const* bytes, const T* const* wide, T* const* outs,
                            double day_hours, int64_t n, unsigned flags, void* stream) {
    constexpr int V = VecOf<T>::v;
    if (!ctx) return MOD16_ERR_ARG;
    int nw, nb, no;
    if (!form_shape(form, &nw, &nb, &no)) return fail(ctx, MOD16_ERR_ARG, "mod16_et_form_tiled: unknown form");
    if (!lay || !bytes || !wide || !outs || n < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_form_tiled: NULL argument or n < 0");
    if (!ctx->have_lut) return fail(ctx, MOD16_ERR_NO_BPLUT, "mod16_et_form_tiled: mod16_set_bplut_f64 was not called");
    if (flags & MOD16_MATH_EXACT) return fail(ctx, MOD16_ERR_ARG, "mod16_et_form_tiled: MOD16_MATH_EXACT runs on plain arrays only");
    const int px_shift = tile_log2(lay->tile, (int64_t)64 * V * kDynRun);
    if (px_shift < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et_form_tiled: tile must be a power of two of at least 8 KiB per field");
    auto al16 = [](const void* p) { return p && reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    bool ok = lay->driver_row >= lay->tile && lay->out_row >= lay->tile && lay->cls_row >= lay->tile &&
              lay->driver_row % V == 0 && lay->out_row % V == 0 && lay->cls_row % V == 0 && n % V == 0;
    for (int k = 0; k < nw && ok; ++k) ok = al16(wide[k]);
    for (int k = 0; k < no && ok; ++k) ok = al16(outs[k]);
    for (int k = 0; k < nb && ok; ++k) ok = bytes[k] && reinterpret_cast<uintptr_t>(bytes[k]) % V == 0;
    if (!ok) return fail(ctx, MOD16_ERR_ARG, "mod16_et_form_tiled: every array of the form is required, 16-byte aligned; rows >= tile and multiples of the vector width, n a multiple of it");
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    StreamArgs<T> s;
    memset(&s, 0, sizeof s);
    for (int k = 0; k < nw; ++k) s.wide[k] = wide[k];
    for (int k = 0; k < nb; ++k) s.bytes[k] = bytes[k];
    for (int k = 0; k < no; ++k) s.out[k] = outs[k];
    s.hours = day_hours;
    s.n = n;
    int pv = 0;
    while ((1 << pv) < 64 * V) ++pv;
    s.tile_shift = px_shift - pv;
    s.wide_row = lay->driver_row;
    s.out_row = lay->out_row;
    s.byte_row = lay->cls_row;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc = MOD16_OK;
    bool mixed = false;
    if constexpr (std::is_same<T, float>::value) {
        mixed = (flags & MOD16_MATH_MIXED) != 0;
        if (mixed) {
            switch (form) {
            case MOD16_FORM_TOTALS: rc = launch_totals<T>(ctx, s, st, nullptr, flags); break;
            case MOD16_FORM_PET: rc = launch_stream<T, kStreamPetMixed>(ctx, s, st); break;
            case MOD16_FORM_COMPONENTS: rc = launch_stream<T, kStreamSep6Mixed>(ctx, s, st); break;
            case MOD16_FORM_TOTALS_COMPONENTS: rc = launch_stream<T, kStreamSep8Mixed>(ctx, s, st); break;
            case MOD16_FORM_RAW: rc = launch_stream<T, kStreamRawMixed>(ctx, s, st); break;
            case MOD16_FORM_RAW_TOTAL8: rc = launch_stream<T, kStreamRawTotalMixed>(ctx, s, st); break;
            default: rc = launch_stream<T, kStreamRawTotalHoursMixed>(ctx, s, st); break;
            }
        }
    }
    if (!mixed) {
        switch (form) {
        case MOD16_FORM_TOTALS: rc = launch_totals<T>(ctx, s, st, nullptr, flags); break;
        case MOD16_FORM_PET: rc = launch_stream<T, kStreamPet>(ctx, s, st); break;
        case MOD16_FORM_COMPONENTS: rc = launch_stream<T, kStreamSep6>(ctx, s, st); break;
        case MOD16_FORM_TOTALS_COMPONENTS: rc = launch_stream<T, kStreamSep8>(ctx, s, st); break;
        case MOD16_FORM_RAW: rc = launch_stream<T, kStreamRaw>(ctx, s, st); break;
        case MOD16_FORM_RAW_TOTAL8: rc = launch_stream<T, kStreamRawTotal>(ctx, s, st); break;
        default: rc = launch_stream<T, kStreamRawTotalHours>(ctx, s, st); break;
        }
    }
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    return MOD16_OK;
}

extern "C" int mod16_et_form_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout, int form,
                                       const uint8_t* const* bytes, const double* const* wide,
                                       double* const* outs, double day_hours, int64_t n,
                                       unsigned flags, void* stream) {
    MOD16_LOCK(ctx);
    return form_tiled_entry<double>(ctx, layout, form, bytes, wide, outs, day_hours, n, flags, stream);
}
extern "C" int mod16_et_form_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout, int form,
                                       const uint8_t* const* bytes, const float* const* wide,
                                       float* const* outs, double day_hours, int64_t n,
                                       unsigned flags, void* stream) {
    MOD16_LOCK(ctx);
    return form_tiled_entry<float>(ctx, layout, form, bytes, wide, outs, day_hours, n, flags, stream);
}

template <typename T>
static int graph_tiled_entry(mod16_ctx* ctx, const mod16_layout* lay, const uint8_t* cls,
                             const T* const* drivers, int64_t n, T* out_day, T* out_night,
                             unsigned flags, double* ddiag, mod16_graph** out) {
    if (!ctx || !out) return MOD16_ERR_ARG;
    *out = nullptr;
    if (!lay || !ddiag) return fail(ctx, MOD16_ERR_ARG, "mod16_graph_et_tiled: layout and ddiag are required");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->streams[0]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking));
    hipStream_t st = ctx->streams[0];
    mod16_graph* g = new (std::nothrow) mod16_graph;
    if (!g) return MOD16_ERR_NOMEM;
    g->ctx = ctx;
    g->device = ctx->device;
    int rc = [&]() -> int {
        // (nothing runs here: the step is only recorded -- argument errors come back from the
        // recording call, launch errors from the instantiation -- so no wait for whatever the
        // caller's streams are still doing to the raster is needed; replays are ordered by
        // the stream they are launched on)
        HIPCHK(ctx, hipMalloc(&g->counter, 128));
        HIPCHK(ctx, hipMemset(g->counter, 0, 128));       // (not captured: the launches keep it at zero)
        ctx->force_counter = g->counter;
        int pv = 0, tsh = lay->tile > 0 ? tile_log2(lay->tile, 1) : -1;
        while ((1 << pv) < 64 * VecOf<T>::v) ++pv;
        if (tsh < pv) return fail(ctx, MOD16_ERR_ARG, "mod16_graph_et_tiled: bad tile");
        HIPCHK(ctx, ws_alloc(g->ws, std::max<int64_t>(kDiagBlocks, stream_ws_blocks(stream_geom(ctx, std::max<int64_t>(n, 0), VecOf<T>::v, tsh - pv).nruns))));
        ctx->force_ws = &g->ws;
        HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int r = tiled_entry<T>(ctx, lay, cls, drivers, n, out_day, out_night, flags, ddiag, st);
        hipError_t e = hipStreamEndCapture(st, &g->graph);
        if (r != MOD16_OK) return r;
        HIPCHK(ctx, e);
        HIPCHK(ctx, hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
        return MOD16_OK;
    }();
    ctx->force_counter = nullptr;
    ctx->force_ws = nullptr;
    if (rc != MOD16_OK) {
        mod16_graph_destroy(g);
        return rc;
    }
    graph_register(ctx, g);
    *out = g;
    return MOD16_OK;
}

extern "C" int mod16_graph_et_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout, const uint8_t* cls,
                                        const double* const* drivers, int64_t n, double* out_day,
                                        double* out_night, unsigned flags, double* ddiag,
                                        mod16_graph** out) {
    MOD16_LOCK(ctx);
    return graph_tiled_entry<double>(ctx, layout, cls, drivers, n, out_day, out_night, flags, ddiag, out);
}
extern "C" int mod16_graph_et_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout, const uint8_t* cls,
                                        const float* const* drivers, int64_t n, float* out_day,
                                        float* out_night, unsigned flags, double* ddiag,
                                        mod16_graph** out) {
    MOD16_LOCK(ctx);
    return graph_tiled_entry<float>(ctx, layout, cls, drivers, n, out_day, out_night, flags, ddiag, out);
}

extern "C" int mod16_time_et_tiled(mod16_ctx* ctx, int is_f32, const mod16_layout* layout,
                                   const uint8_t* cls, const void* const* drivers, int64_t n,
                                   void* out_day, void* out_night, unsigned flags, double* ddiag,
                                   int launches, void* stream, float* ms) {
    MOD16_LOCK(ctx);
    if (!ctx || !ms || launches <= 0) return fail(ctx, MOD16_ERR_ARG, "mod16_time_et_tiled: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    int rc = MOD16_OK;
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < launches && rc == MOD16_OK; ++i)
        rc = is_f32 ? tiled_entry<float>(ctx, layout, cls, reinterpret_cast<const float* const*>(drivers), n,
                                         static_cast<float*>(out_day), static_cast<float*>(out_night), flags, ddiag, stream)
                    : tiled_entry<double>(ctx, layout, cls, reinterpret_cast<const double* const*>(drivers), n,
                                          static_cast<double*>(out_day), static_cast<double*>(out_night), flags, ddiag, stream);
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)launches;
    return rc;
}

// mean milliseconds per replay of a captured step, HIP events on `stream`
extern "C" int mod16_time_graph(mod16_graph* g, int launches, void* stream, float* ms) {
    if (!g || !g->exec || !ms || launches <= 0 || !graph_alive(g)) return MOD16_ERR_ARG;
    if (hipSetDevice(g->device) != hipSuccess) return MOD16_ERR_HIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MOD16_ERR_HIP;
    bool ok = hipEventRecord(e0, st) == hipSuccess;
    for (int i = 0; i < launches && ok; ++i) ok = hipGraphLaunch(g->exec, st) == hipSuccess;
    ok = ok && hipEventRecord(e1, st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float t = 0.f;
    ok = ok && hipEventElapsedTime(&t, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return MOD16_ERR_HIP;
    *ms = t / (float)launches;
    return MOD16_OK;
}
