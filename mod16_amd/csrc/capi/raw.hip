// libmod16hip.so -- raw reanalysis drivers (N1): mod16_et_raw_*
#include "internal.hpp"
#include "../mod16_methods.hpp"

// ----------------------------------------------------- raw drivers (N1)
template <typename T>
static int raw_entry(mod16_ctx* ctx, const uint8_t* cls, const T* const* raw,
                     const int64_t* rstride, const uint8_t* fpar_pct, const uint8_t* lai_x10,
                     const T* day_hours, int64_t hstride, int64_t n, T* out_day, T* out_night,
                     T* out_total8, unsigned flags, int where, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!cls || !raw || !rstride || !fpar_pct || !lai_x10 || n < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_raw: NULL argument or n < 0");
    if (!out_day && !out_night && !out_total8) return fail(ctx, MOD16_ERR_ARG, "mod16_et_raw: no output array given");
    if (out_total8 && !day_hours) return fail(ctx, MOD16_ERR_ARG, "mod16_et_raw: out_total8 needs day_hours");
    if (!ctx->have_lut) return fail(ctx, MOD16_ERR_NO_BPLUT, "mod16_et_raw: mod16_set_bplut_f64 was not called");
    RawArgs<T> a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 14; ++k) {
        if (!raw[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et_raw: NULL driver array");
        a.drv[k] = raw[k];
        if (rstride[k]) a.dense_drv |= 1u << k;
    }
    a.fpar_pct = fpar_pct;
    a.lai_x10 = lai_x10;
    a.cls = cls;
    a.day_hours = out_total8 ? day_hours : nullptr;
    a.dense_hours = hstride ? 1u : 0u;
    a.out[0] = out_day;
    a.out[1] = out_night;
    a.out[2] = out_total8;
    a.n = n;
    a.lut = ctx_lut<T>(ctx);
    a.lut64 = ctx->lut64;
    a.tab = ctx->tab64;
    a.status = ctx->status;
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool fast = (flags & MOD16_MATH_EXACT) == 0;
    // d holds device pointers. Dense, 16-byte-aligned rasters run their vector
    // body on the production pipeline (et_stream_kernel); the ragged tail and
    // every other shape run the plain kernel. host_hours: the scalar hours of
    // daylight when it is known on the host (HOST mode).
    auto launch = [&](const RawArgs<T>& d, hipStream_t st, const T* host_hours) -> int {
        constexpr int V = VecOf<T>::v;
        auto al = [](const void* p, size_t to) { return reinterpret_cast<uintptr_t>(p) % to == 0; };
        bool ok = fast && ctx->use_dma && d.dense_drv == 0x3fffu && d.out[0] && d.out[1];
        for (int k = 0; k < 14 && ok; ++k) ok = al(d.drv[k], 16);
        ok = ok && al(d.fpar_pct, V) && al(d.lai_x10, V) && al(d.cls, V) && al(d.out[0], 16) && al(d.out[1], 16);
        int mode = kStreamRaw;
        if (ok && d.out[2]) {
            ok = al(d.out[2], 16);
            if (d.dense_hours) { mode = kStreamRawTotalHours; ok = ok && al(d.day_hours, 16); }
            else if (host_hours) mode = kStreamRawTotal;
            else ok = false;
        }
        const int64_t nbody = ok ? (d.n / V) * V : 0;
        if (nbody) {
            StreamArgs<T> s;
            memset(&s, 0, sizeof s);
            for (int k = 0; k < 14; ++k) s.wide[k] = d.drv[k];
            s.wide[14] = d.day_hours;
            s.bytes[0] = d.cls; s.bytes[1] = d.fpar_pct; s.bytes[2] = d.lai_x10;
            for (int k = 0; k < 3; ++k) s.out[k] = d.out[k];
            s.hours = host_hours ? (double)*host_hours : 0.0;
            s.n = nbody;
            int rc = MOD16_OK;
            bool mixed = false;
            if constexpr (std::is_same<T, float>::value) {
                mixed = (flags & MOD16_MATH_MIXED) != 0;
                if (mixed)
                    rc = mode == kStreamRaw ? launch_stream<T, kStreamRawMixed>(ctx, s, st)
                         : mode == kStreamRawTotal ? launch_stream<T, kStreamRawTotalMixed>(ctx, s, st)
                                                   : launch_stream<T, kStreamRawTotalHoursMixed>(ctx, s, st);
            }
            if (!mixed)
                rc = mode == kStreamRaw ? launch_stream<T, kStreamRaw>(ctx, s, st)
                     : mode == kStreamRawTotal ? launch_stream<T, kStreamRawTotal>(ctx, s, st)
                                               : launch_stream<T, kStreamRawTotalHours>(ctx, s, st);
            if (rc != MOD16_OK) return rc;
        }
        if (nbody < d.n) {
            RawArgs<T> t = d;
            for (int k = 0; k < 14; ++k) if ((t.dense_drv >> k) & 1u) t.drv[k] += nbody;
            t.fpar_pct += nbody; t.lai_x10 += nbody; t.cls += nbody;
            if (t.day_hours && t.dense_hours) t.day_hours += nbody;
            for (int k = 0; k < 3; ++k) if (t.out[k]) t.out[k] += nbody;
            t.n = d.n - nbody;
            const int grid = grid_for(ctx, t.n);
            if (fast) hipLaunchKernelGGL((et_raw_kernel<T, true>), dim3(grid), dim3(kBlock), 0, st, t);
            else hipLaunchKernelGGL((et_raw_kernel<T, false>), dim3(grid), dim3(kBlock), 0, st, t);
        }
        return MOD16_OK;
    };
    if (where == MOD16_DEVICE) {
        int rc = launch(a, static_cast<hipStream_t>(stream), nullptr);
        if (rc != MOD16_OK) return rc;
        HIPCHK(ctx, hipGetLastError());
        return MOD16_OK;
    }
    if (where != MOD16_HOST) return fail(ctx, MOD16_ERR_ARG, "mod16_et_raw: bad `where`");
    size_t per_arr_small = 0;
    if (n <= ctx->small_pixels && small_reserve(ctx, n, sizeof(T), 14 + 1 + 3, &per_arr_small)) {
        // small calls: no copy commands, the kernel reads and writes one page-locked buffer
        // (run_host_small; whole vectors, the pad pixels repeat the last one; classes checked here)
        for (int64_t i = 0; i < n; ++i)
            if (cls[i] >= MOD16_N_CLASSES)
                return fail(ctx, MOD16_ERR_CLASS_RANGE, "class raster holds a code >= 13 (numpy would raise IndexError)");
        const size_t per_arr = per_arr_small;
        int rc = MOD16_OK;
        hipStream_t st = ctx->streams[0];
        char* hb = static_cast<char*>(ctx->small_host);
        char* db = static_cast<char*>(ctx->small_dev);
        T* hsc = reinterpret_cast<T*>(hb);
        const T* dscal = reinterpret_cast<const T*>(db);
        constexpr int V = VecOf<T>::v;
        const int64_t npad = (n + V - 1) / V * V;
        const size_t per_b = per_arr / sizeof(T);       // the buffer's capacity in pixels
        auto arr = [&](int k) { return (size_t)256 + per_arr * k; };
        auto put = [&](size_t off, const void* src, size_t elem) {
            memcpy(hb + off, src, elem * n);
            for (int64_t i = n; i < npad; ++i) memcpy(hb + off + elem * i, static_cast<const char*>(src) + elem * (n - 1), elem);
        };
        RawArgs<T> d = a;
        d.n = npad;
        for (int k = 0; k < 14; ++k) {
            if ((a.dense_drv >> k) & 1u) {
                put(arr(k), a.drv[k], sizeof(T));
                d.drv[k] = reinterpret_cast<const T*>(db + arr(k));
            } else {
                hsc[k] = a.drv[k][0];
                d.drv[k] = dscal + k;
            }
        }
        T host_hours = T(0);
        if (a.day_hours) {
            if (a.dense_hours) {
                put(arr(14), a.day_hours, sizeof(T));
                d.day_hours = reinterpret_cast<const T*>(db + arr(14));
            } else {
                host_hours = hsc[14] = a.day_hours[0];
                d.day_hours = dscal + 14;
            }
        }
        const uint8_t* hbytes[3] = {a.fpar_pct, a.lai_x10, a.cls};
        const uint8_t** dbytes[3] = {&d.fpar_pct, &d.lai_x10, &d.cls};
        for (int k = 0; k < 3; ++k) {
            const size_t off = arr(18) + per_b * k;
            put(off, hbytes[k], 1);
            *dbytes[k] = reinterpret_cast<const uint8_t*>(db + off);
        }
        for (int k = 0; k < 3; ++k) d.out[k] = a.out[k] ? reinterpret_cast<T*>(db + arr(15 + k)) : nullptr;
        rc = launch(d, st, (a.day_hours && !a.dense_hours) ? &host_hours : nullptr);
        if (rc != MOD16_OK) return rc;
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < 3; ++k)
            if (a.out[k]) memcpy(a.out[k], hb + arr(15 + k), sizeof(T) * n);
        return MOD16_OK;
    }
    // HOST: tiles of kTilePixels staged through the context's slabs, one host thread and one stream
    // per slot, as run_host does for the processed drivers (round 5; one slab and one thread before:
    // the copies from pageable memory, which the runtime stages on the calling thread, are what bounds
    // this mode, and the light input form -- 58 bytes per pixel in float32 -- is the one worth feeding
    // at the link's rate)
    const int64_t tile = std::min<int64_t>(n, kTilePixels);
    const int64_t ntiles = (n + tile - 1) / tile;
    const int nslots = (int)std::min<int64_t>(ntiles, ctx->host_threads);
    if (nslots > 1) ctx->ws_multi = true;       // one stream per slot: the launches leave their events (ws_release)
    const size_t per_arr = (((size_t)tile * sizeof(T)) + 255) / 256 * 256 + kStagger;
    const size_t need = per_arr * (14 + 1 + 3) + 3 * ((size_t)tile + 256) + 256;
    if (ctx->slab_bytes < need) {
        for (int s = 0; s < kSlots; ++s) {
            if (ctx->slab[s]) HIPCHK(ctx, hipFree(ctx->slab[s]));
            ctx->slab[s] = nullptr;
        }
        ctx->slab_bytes = need;
    }
    for (int s = 0; s < nslots; ++s) {
        if (!ctx->slab[s]) HIPCHK(ctx, hipMalloc(&ctx->slab[s], ctx->slab_bytes));
        if (!ctx->streams[s]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[s], hipStreamNonBlocking));
    }
    T hs[16];
    for (int k = 0; k < 14; ++k) hs[k] = ((a.dense_drv >> k) & 1u) ? T(0) : a.drv[k][0];
    hs[14] = (a.day_hours && !a.dense_hours) ? a.day_hours[0] : T(0);
    HIPCHK(ctx, hipMemcpy(ctx->scalars, hs, sizeof(T) * 15, hipMemcpyHostToDevice));
    const T* dscal = static_cast<const T*>(ctx->scalars);
    const size_t per_b = ((size_t)tile + 255) / 256 * 256;
    {   // the kernels' shared workspace at its final size before any thread launches
        const int64_t npiece = (tile / VecOf<T>::v + 63) / 64;
        int rc = reserve_diag(ctx, npiece / 2 + 2048);
        if (rc != MOD16_OK) return rc;
    }
    auto stage = [&](int slot, int64_t off, int64_t m) -> int {
        hipStream_t st = ctx->streams[slot];
        char* base = static_cast<char*>(ctx->slab[slot]);
        uint8_t* bytes = reinterpret_cast<uint8_t*>(base + per_arr * 18);
        RawArgs<T> d = a;
        d.n = m;
        for (int k = 0; k < 14; ++k) {
            if ((a.dense_drv >> k) & 1u) {
                T* dp = reinterpret_cast<T*>(base + per_arr * k);
                HIPCHK(ctx, hipMemcpyAsync(dp, a.drv[k] + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
                d.drv[k] = dp;
            } else {
                d.drv[k] = dscal + k;
            }
        }
        if (a.day_hours) {
            if (a.dense_hours) {
                T* dp = reinterpret_cast<T*>(base + per_arr * 14);
                HIPCHK(ctx, hipMemcpyAsync(dp, a.day_hours + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
                d.day_hours = dp;
            } else {
                d.day_hours = dscal + 14;
            }
        }
        const uint8_t* hb[3] = {a.fpar_pct, a.lai_x10, a.cls};
        const uint8_t** db[3] = {&d.fpar_pct, &d.lai_x10, &d.cls};
        for (int k = 0; k < 3; ++k) {
            uint8_t* dp = bytes + per_b * k;
            HIPCHK(ctx, hipMemcpyAsync(dp, hb[k] + off, (size_t)m, hipMemcpyHostToDevice, st));
            *db[k] = dp;
        }
        for (int k = 0; k < 3; ++k) d.out[k] = a.out[k] ? reinterpret_cast<T*>(base + per_arr * (15 + k)) : nullptr;
        {
            std::lock_guard<std::mutex> lock(ctx->launch_mu);      // (the launches share the context's workspace)
            int rc = launch(d, st, (a.day_hours && !a.dense_hours) ? &hs[14] : nullptr);
            if (rc != MOD16_OK) return rc;
            HIPCHK(ctx, hipGetLastError());
        }
        for (int k = 0; k < 3; ++k)
            if (a.out[k]) HIPCHK(ctx, hipMemcpyAsync(a.out[k] + off, d.out[k], sizeof(T) * m, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));      // the slab of this slot is free again
        return MOD16_OK;
    };
    if (nslots == 1) {
        for (int64_t off = 0; off < n; off += tile) {
            int rc = stage(0, off, std::min(tile, n - off));
            if (rc != MOD16_OK) return rc;
        }
    } else {
        int rcs[kSlots] = {};
        std::vector<std::thread> workers;
        for (int s = 0; s < nslots; ++s)
            workers.emplace_back([&, s]() {
                if (hipSetDevice(ctx->device) != hipSuccess) { rcs[s] = MOD16_ERR_HIP; return; }
                for (int64_t t = s; t < ntiles && rcs[s] == MOD16_OK; t += nslots)
                    rcs[s] = stage(s, t * tile, std::min(tile, n - t * tile));
            });
        for (auto& w : workers) w.join();
        for (int s = 0; s < nslots; ++s)
            if (rcs[s] != MOD16_OK) return rcs[s];
    }
    return read_status(ctx, ctx->streams[0]);
}

extern "C" int mod16_et_raw_f64(mod16_ctx* ctx, const uint8_t* cls, const double* const* raw,
                                const int64_t* rstride, const uint8_t* fpar_pct,
                                const uint8_t* lai_x10, const double* day_hours, int64_t hstride,
                                int64_t n, double* out_day, double* out_night, double* out_total8,
                                unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    return raw_entry<double>(ctx, cls, raw, rstride, fpar_pct, lai_x10, day_hours, hstride, n,
                             out_day, out_night, out_total8, flags, where, stream);
}
extern "C" int mod16_et_raw_f32(mod16_ctx* ctx, const uint8_t* cls, const float* const* raw,
                                const int64_t* rstride, const uint8_t* fpar_pct,
                                const uint8_t* lai_x10, const float* day_hours, int64_t hstride,
                                int64_t n, float* out_day, float* out_night, float* out_total8,
                                unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    return raw_entry<float>(ctx, cls, raw, rstride, fpar_pct, lai_x10, day_hours, hstride, n,
                            out_day, out_night, out_total8, flags, where, stream);
}
