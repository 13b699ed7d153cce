// libmod16hip.so -- the class-surface sub-methods (mod16_method_*)
#include "internal.hpp"
#include "../mod16_methods.hpp"

// ------------------------------------------------------- class-surface methods
template <typename T>
static int method_entry(mod16_ctx* ctx, int method, const T* const* in, const int64_t* istride,
                        const T* const* params, const int64_t* pstride, int64_t n,
                        T* const* out, T alpha, T tiny, int where, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (method < 0 || method >= MOD16_M_COUNT || !in || !istride || !out || !out[0] || n < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_method: bad argument");
    MethodArgs<T> a;
    memset(&a, 0, sizeof a);
    a.method = method;
    a.alpha = alpha;
    a.tiny = tiny;
    a.n = n;
    static const T nan_param = std::numeric_limits<T>::quiet_NaN();
    for (int k = 0; k < kMethodMaxIn; ++k) {
        a.in[k] = in[k];
        if (in[k]) {
            a.present_in |= 1u << k;
            if (istride[k]) a.dense_in |= 1u << k;
        }
    }
    for (int k = 0; k < 11; ++k) {
        a.par[k] = params ? params[k] : nullptr;
        if (a.par[k] && pstride && pstride[k]) a.dense_par |= 1u << k;
    }
    a.out[0] = out[0];
    a.out[1] = out[1];
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    auto launch = [&](const MethodArgs<T>& d, hipStream_t st) {
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((d.n + kBlock - 1) / kBlock, (int64_t)ctx->cus * 8));
        hipLaunchKernelGGL((method_kernel<T>), dim3(grid), dim3(kBlock), 0, st, d);
    };
    if (where == MOD16_DEVICE) {
        // absent parameters read as NaN scalars from the ctx scratch
        T hs[11];
        for (int k = 0; k < 11; ++k) hs[k] = nan_param;
        hipStream_t st = static_cast<hipStream_t>(stream);
        bool need = false;
        for (int k = 0; k < 11; ++k) if (!a.par[k]) need = true;
        if (need) {
            HIPCHK(ctx, hipMemcpyAsync(ctx->scalars, hs, sizeof hs, hipMemcpyHostToDevice, st));
            for (int k = 0; k < 11; ++k) if (!a.par[k]) a.par[k] = static_cast<const T*>(ctx->scalars) + k;
        }
        launch(a, st);
        HIPCHK(ctx, hipGetLastError());
        return MOD16_OK;
    }
    if (where != MOD16_HOST) return fail(ctx, MOD16_ERR_ARG, "mod16_method: bad `where`");
    size_t per_arr_small = 0;
    if (n <= ctx->small_pixels && small_reserve(ctx, n, sizeof(T), kMethodMaxIn + 11 + 2, &per_arr_small)) {
        const size_t per_arr = per_arr_small;
        // small calls (what the class surface is used for: scalars, a site's series): no copy
        // commands, the kernel reads and writes one page-locked buffer (run_host_small)
        hipStream_t st = ctx->streams[0];
        char* hb = static_cast<char*>(ctx->small_host);
        char* db = static_cast<char*>(ctx->small_dev);
        T* hs = reinterpret_cast<T*>(hb);
        const T* dscal = reinterpret_cast<const T*>(db);
        static_assert(sizeof(double) * (kMethodMaxIn + 11) <= 256, "scalars of a method call fit the buffer's head");
        auto arr = [&](int k) { return (size_t)256 + per_arr * k; };
        MethodArgs<T> d = a;
        for (int k = 0; k < kMethodMaxIn; ++k) {
            if (!a.in[k]) continue;
            if ((a.dense_in >> k) & 1u) {
                memcpy(hb + arr(k), a.in[k], sizeof(T) * n);
                d.in[k] = reinterpret_cast<const T*>(db + arr(k));
            } else {
                hs[k] = a.in[k][0];
                d.in[k] = dscal + k;
            }
        }
        for (int k = 0; k < 11; ++k) {
            if (a.par[k] && ((a.dense_par >> k) & 1u)) {
                memcpy(hb + arr(kMethodMaxIn + k), a.par[k], sizeof(T) * n);
                d.par[k] = reinterpret_cast<const T*>(db + arr(kMethodMaxIn + k));
            } else {
                hs[kMethodMaxIn + k] = a.par[k] ? a.par[k][0] : nan_param;
                d.par[k] = dscal + kMethodMaxIn + k;
            }
        }
        for (int k = 0; k < 2; ++k)
            d.out[k] = a.out[k] ? reinterpret_cast<T*>(db + arr(kMethodMaxIn + 11 + k)) : nullptr;
        launch(d, st);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < 2; ++k)
            if (a.out[k]) memcpy(a.out[k], hb + arr(kMethodMaxIn + 11 + k), sizeof(T) * n);
        return MOD16_OK;
    }
    // HOST: one slab, tile by tile (no double buffering)
    const int64_t tile = std::min<int64_t>(n, kTilePixels);
    const size_t per_arr = (((size_t)tile * sizeof(T)) + 255) / 256 * 256;
    const size_t need = per_arr * (14 + 11 + 8) + (size_t)tile + 256;
    if (ctx->slab_bytes < need) {
        for (int s = 0; s < kSlots; ++s) {
            if (ctx->slab[s]) HIPCHK(ctx, hipFree(ctx->slab[s]));
            ctx->slab[s] = nullptr;
        }
        ctx->slab_bytes = need;
    }
    if (!ctx->slab[0]) HIPCHK(ctx, hipMalloc(&ctx->slab[0], ctx->slab_bytes));      // (this mode uses one slot)
    if (!ctx->streams[0]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking));
    hipStream_t st = ctx->streams[0];
    T hs[32];
    for (int k = 0; k < kMethodMaxIn; ++k) hs[k] = (a.in[k] && !((a.dense_in >> k) & 1u)) ? a.in[k][0] : T(0);
    for (int k = 0; k < 11; ++k)
        hs[kMethodMaxIn + k] = !a.par[k] ? nan_param : (((a.dense_par >> k) & 1u) ? T(0) : a.par[k][0]);
    HIPCHK(ctx, hipMemcpy(ctx->scalars, hs, sizeof(T) * (kMethodMaxIn + 11), hipMemcpyHostToDevice));
    const T* dscal = static_cast<const T*>(ctx->scalars);
    char* base = static_cast<char*>(ctx->slab[0]);
    for (int64_t off = 0; off < n; off += tile) {
        const int64_t m = std::min(tile, n - off);
        MethodArgs<T> d = a;
        d.n = m;
        for (int k = 0; k < kMethodMaxIn; ++k) {
            if (!a.in[k]) continue;
            if ((a.dense_in >> k) & 1u) {
                T* dp = reinterpret_cast<T*>(base + per_arr * k);
                HIPCHK(ctx, hipMemcpyAsync(dp, a.in[k] + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
                d.in[k] = dp;
            } else {
                d.in[k] = dscal + k;
            }
        }
        for (int k = 0; k < 11; ++k) {
            if (a.par[k] && ((a.dense_par >> k) & 1u)) {
                T* dp = reinterpret_cast<T*>(base + per_arr * (kMethodMaxIn + k));
                HIPCHK(ctx, hipMemcpyAsync(dp, a.par[k] + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
                d.par[k] = dp;
            } else {
                d.par[k] = dscal + kMethodMaxIn + k;
            }
        }
        for (int k = 0; k < 2; ++k)
            d.out[k] = a.out[k] ? reinterpret_cast<T*>(base + per_arr * (kMethodMaxIn + 11 + k)) : nullptr;
        launch(d, st);
        HIPCHK(ctx, hipGetLastError());
        for (int k = 0; k < 2; ++k)
            if (a.out[k]) HIPCHK(ctx, hipMemcpyAsync(a.out[k] + off, d.out[k], sizeof(T) * m, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
    }
    return MOD16_OK;
}

extern "C" int mod16_method_f64(mod16_ctx* ctx, int method, const double* const* in,
                                const int64_t* istride, const double* const* params,
                                const int64_t* pstride, int64_t n, double* const* out,
                                double alpha, double tiny, int where, void* stream) {
    MOD16_LOCK(ctx);
    return method_entry<double>(ctx, method, in, istride, params, pstride, n, out, alpha, tiny, where, stream);
}
extern "C" int mod16_method_f32(mod16_ctx* ctx, int method, const float* const* in,
                                const int64_t* istride, const float* const* params,
                                const int64_t* pstride, int64_t n, float* const* out, float alpha,
                                float tiny, int where, void* stream) {
    MOD16_LOCK(ctx);
    return method_entry<float>(ctx, method, in, istride, params, pstride, n, out, alpha, tiny, where, stream);
}
