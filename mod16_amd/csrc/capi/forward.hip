// libmod16hip.so -- the forward run on numpy / device arrays: mod16_et_*, mod16_et2_*, mod16_et_hdiag_*, mod16_et_diag_*, graphs of it, potential ET
#include "internal.hpp"

template <typename T>
static int et_entry(mod16_ctx* ctx, const uint8_t* cls, const T* const* drivers,
                    const int64_t* dstride, const T* const* params, const int64_t* pstride,
                    int64_t n, T* out_day, T* out_night, T* const* out_sep, unsigned flags,
                    int where, void* stream, T* pet_day = nullptr, T* pet_night = nullptr,
                    int64_t inner = 0, int cls_mode = MOD16_BC_DENSE) {
    EtArgs<T> a;
    int rc = fill_args<T>(ctx, a, cls, drivers, dstride, params, pstride, n, out_day, out_night,
                          out_sep, pet_day, pet_night, inner, cls_mode);
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (where == MOD16_DEVICE) return launch_et<T>(ctx, a, flags, static_cast<hipStream_t>(stream));
    if (where == MOD16_HOST) return run_host<T>(ctx, a, flags);
    return fail(ctx, MOD16_ERR_ARG, "mod16_et: `where` must be MOD16_HOST or MOD16_DEVICE");
}

extern "C" int mod16_et_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* drivers,
                            const int64_t* dstride, const double* const* params,
                            const int64_t* pstride, int64_t n, double* out_day,
                            double* out_night, double* const* out_sep, unsigned flags,
                            int where, void* stream) {
    MOD16_LOCK(ctx);
    return et_entry<double>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night,
                            out_sep, flags, where, stream);
}

extern "C" int mod16_et_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* drivers,
                            const int64_t* dstride, const float* const* params,
                            const int64_t* pstride, int64_t n, float* out_day, float* out_night,
                            float* const* out_sep, unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    return et_entry<float>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night,
                           out_sep, flags, where, stream);
}

extern "C" int mod16_et2_f64(mod16_ctx* ctx, const uint8_t* cls, int cls_kind,
                             const double* const* drivers, const int64_t* dkind,
                             const double* const* params, const int64_t* pkind, int64_t inner,
                             int64_t n, double* out_day, double* out_night,
                             double* const* out_sep, unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    if (ctx && inner <= 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et2: inner must be positive");
    return et_entry<double>(ctx, cls, drivers, dkind, params, pkind, n, out_day, out_night,
                            out_sep, flags, where, stream, nullptr, nullptr, inner, cls_kind);
}
extern "C" int mod16_et2_f32(mod16_ctx* ctx, const uint8_t* cls, int cls_kind,
                             const float* const* drivers, const int64_t* dkind,
                             const float* const* params, const int64_t* pkind, int64_t inner,
                             int64_t n, float* out_day, float* out_night, float* const* out_sep,
                             unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    if (ctx && inner <= 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et2: inner must be positive");
    return et_entry<float>(ctx, cls, drivers, dkind, params, pkind, n, out_day, out_night,
                           out_sep, flags, where, stream, nullptr, nullptr, inner, cls_kind);
}

// ---- HOST mode with diagnostics (mod16_et_hdiag_*): the forward run of mod16_et_* on host arrays,
// plus one diagnostics vector PER STAGED TILE of mod16_host_tile_pixels() pixels, reduced on the
// device while the tile's outputs are there (nothing is uploaded again).
extern "C" int64_t mod16_host_tile_pixels(void) { return kTilePixels; }

template <typename T>
static int hdiag_entry(mod16_ctx* ctx, const uint8_t* cls, const T* const* drivers, const int64_t* dstride,
                       const T* const* params, const int64_t* pstride, int64_t n, T* out_day, T* out_night,
                       unsigned flags, double* tile_diag) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!out_day || !out_night || !tile_diag)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_hdiag: out_day, out_night and tile_diag are required");
    EtArgs<T> a;
    int rc = fill_args<T>(ctx, a, cls, drivers, dstride, params, pstride, n, out_day, out_night, nullptr);
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return run_host<T>(ctx, a, flags, tile_diag);
}
extern "C" int mod16_et_hdiag_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* drivers,
                                  const int64_t* dstride, const double* const* params, const int64_t* pstride,
                                  int64_t n, double* out_day, double* out_night, unsigned flags,
                                  double* tile_diag) {
    MOD16_LOCK(ctx);
    return hdiag_entry<double>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night, flags, tile_diag);
}
extern "C" int mod16_et_hdiag_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* drivers,
                                  const int64_t* dstride, const float* const* params, const int64_t* pstride,
                                  int64_t n, float* out_day, float* out_night, unsigned flags,
                                  double* tile_diag) {
    MOD16_LOCK(ctx);
    return hdiag_entry<float>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night, flags, tile_diag);
}

// The fold of `count` diagnostics vectors on the host, in the order given: sums and counts [0..5]
// added first to last, maxima [6..7] maximised -- mod16_fold_diag's rule (NaN maxima of empty
// parts are skipped the same way: `o > acc`).
extern "C" int mod16_fold_diag_host(const double* parts, int64_t count, double* diag) {
    if (!parts || !diag || count < 1) return MOD16_ERR_ARG;
    for (int k = 0; k < kDiag; ++k) {
        double acc = parts[k];
        for (int64_t r = 1; r < count; ++r) {
            const double o = parts[r * kDiag + k];
            acc = k < 6 ? acc + o : (o > acc ? o : acc);
        }
        diag[k] = acc;
    }
    return MOD16_OK;
}

template <typename T>
static int et_diag_entry(mod16_ctx* ctx, const uint8_t* cls, const T* const* drivers,
                         const int64_t* dstride, int64_t n, T* out_day, T* out_night,
                         unsigned flags, double* ddiag, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!cls || !out_day || !out_night || !ddiag)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_diag: cls, out_day, out_night and ddiag are required");
    EtArgs<T> a;
    int rc = fill_args<T>(ctx, a, cls, drivers, dstride, nullptr, nullptr, n, out_day, out_night, nullptr);
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return launch_et<T>(ctx, a, flags, static_cast<hipStream_t>(stream), ddiag);
}

extern "C" int mod16_et_diag_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* drivers,
                                 const int64_t* dstride, int64_t n, double* out_day,
                                 double* out_night, unsigned flags, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return et_diag_entry<double>(ctx, cls, drivers, dstride, n, out_day, out_night, flags, ddiag, stream);
}
extern "C" int mod16_et_diag_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* drivers,
                                 const int64_t* dstride, int64_t n, float* out_day,
                                 float* out_night, unsigned flags, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return et_diag_entry<float>(ctx, cls, drivers, dstride, n, out_day, out_night, flags, ddiag, stream);
}


extern "C" int mod16_graph_destroy(mod16_graph* g) {
    if (!g) return MOD16_OK;
    {
        std::lock_guard<std::mutex> lock(graph_registry_mu());
        if (g->ctx) {
            std::vector<mod16_graph*>& v = g->ctx->graphs;
            v.erase(std::remove(v.begin(), v.end(), g), v.end());
        }
    }
    (void)hipSetDevice(g->device);           // the context may be gone already (interpreter exit)
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->counter) (void)hipFree(g->counter);
    if (g->ws.partial) (void)hipFree(g->ws.partial);
    delete g;
    return MOD16_OK;
}

template <typename T>
static int graph_entry(mod16_ctx* ctx, const uint8_t* cls, const T* const* drivers,
                       const int64_t* dstride, int64_t n, T* out_day, T* out_night, unsigned flags,
                       double* ddiag, mod16_graph** out) {
    if (!ctx || !out) return MOD16_ERR_ARG;
    *out = nullptr;
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
        // the graph's kernel nodes keep pointing at this workspace for as long as
        // the graph lives, whatever the context's own workspace does meanwhile
        HIPCHK(ctx, ws_alloc(g->ws, std::max<int64_t>(kDiagBlocks, stream_ws_blocks(stream_geom(ctx, std::max<int64_t>(n, 0), VecOf<T>::v).nruns))));
        ctx->force_ws = &g->ws;
        HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int r = et_diag_entry<T>(ctx, cls, drivers, dstride, n, out_day, out_night, flags, ddiag, st);
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

extern "C" int mod16_graph_et_diag_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* drivers,
                                       const int64_t* dstride, int64_t n, double* out_day,
                                       double* out_night, unsigned flags, double* ddiag,
                                       mod16_graph** out) {
    MOD16_LOCK(ctx);
    return graph_entry<double>(ctx, cls, drivers, dstride, n, out_day, out_night, flags, ddiag, out);
}
extern "C" int mod16_graph_et_diag_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* drivers,
                                       const int64_t* dstride, int64_t n, float* out_day,
                                       float* out_night, unsigned flags, double* ddiag,
                                       mod16_graph** out) {
    MOD16_LOCK(ctx);
    return graph_entry<float>(ctx, cls, drivers, dstride, n, out_day, out_night, flags, ddiag, out);
}
extern "C" int mod16_graph_launch(mod16_graph* g, void* stream) {
    if (!g || !g->exec) return MOD16_ERR_ARG;
    // (a graph may outlive the context it was built with -- to be destroyed, not replayed: its
    // kernels point into the context's tables. No error text through g->ctx either way)
    if (!graph_alive(g)) {
        fprintf(stderr, "mod16_graph_launch: the context this graph was captured with has been destroyed\n");
        return MOD16_ERR_ARG;
    }
    const hipError_t e = hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) {
        fprintf(stderr, "mod16_graph_launch: %s\n", hipGetErrorString(e));
        return MOD16_ERR_HIP;
    }
    return MOD16_OK;
}

extern "C" int mod16_et_pet_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* drivers,
                                const int64_t* dstride, const double* const* params,
                                const int64_t* pstride, int64_t n, double* out_day,
                                double* out_night, double* pet_day, double* pet_night,
                                unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    if (ctx && !pet_day && !pet_night) return fail(ctx, MOD16_ERR_ARG, "mod16_et_pet: no PET output given");
    return et_entry<double>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night,
                            nullptr, flags, where, stream, pet_day, pet_night);
}
extern "C" int mod16_et_pet_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* drivers,
                                const int64_t* dstride, const float* const* params,
                                const int64_t* pstride, int64_t n, float* out_day,
                                float* out_night, float* pet_day, float* pet_night,
                                unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    if (ctx && !pet_day && !pet_night) return fail(ctx, MOD16_ERR_ARG, "mod16_et_pet: no PET output given");
    return et_entry<float>(ctx, cls, drivers, dstride, params, pstride, n, out_day, out_night,
                           nullptr, flags, where, stream, pet_day, pet_night);
}


extern "C" int mod16_time_et(mod16_ctx* ctx, int is_f32, const uint8_t* cls,
                             const void* const* drivers, const int64_t* dstride,
                             const void* const* params, const int64_t* pstride, int64_t n,
                             void* out_day, void* out_night, void* const* out_sep,
                             unsigned flags, double* ddiag, int launches, void* stream,
                             float* ms) {
    MOD16_LOCK(ctx);
    if (!ctx || !ms || launches <= 0) return fail(ctx, MOD16_ERR_ARG, "mod16_time_et: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    int rc = MOD16_OK;
    HIPCHK(ctx, hipEventRecord(e0, st));
    for (int i = 0; i < launches && rc == MOD16_OK; ++i) {
        if (ddiag && is_f32)
            rc = mod16_et_diag_f32(ctx, cls, reinterpret_cast<const float* const*>(drivers), dstride, n,
                                   static_cast<float*>(out_day), static_cast<float*>(out_night), flags,
                                   ddiag, stream);
        else if (ddiag)
            rc = mod16_et_diag_f64(ctx, cls, reinterpret_cast<const double* const*>(drivers), dstride, n,
                                   static_cast<double*>(out_day), static_cast<double*>(out_night), flags,
                                   ddiag, stream);
        else if (is_f32)
            rc = mod16_et_f32(ctx, cls, reinterpret_cast<const float* const*>(drivers), dstride,
                              reinterpret_cast<const float* const*>(params), pstride, n,
                              static_cast<float*>(out_day), static_cast<float*>(out_night),
                              reinterpret_cast<float* const*>(out_sep), flags, MOD16_DEVICE, stream);
        else
            rc = mod16_et_f64(ctx, cls, reinterpret_cast<const double* const*>(drivers), dstride,
                              reinterpret_cast<const double* const*>(params), pstride, n,
                              static_cast<double*>(out_day), static_cast<double*>(out_night),
                              reinterpret_cast<double* const*>(out_sep), flags, MOD16_DEVICE, stream);
    }
    HIPCHK(ctx, hipEventRecord(e1, st));
    HIPCHK(ctx, hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&t, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms = t / (float)launches;
    return rc;
}
