// libmod16hip.so -- diagnostics, the class raster of gathered parameters, the synthetic generator
#include "internal.hpp"

extern "C" int mod16_reduce_diag_f64(mod16_ctx* ctx, const double* day, const double* night,
                                     int64_t n, double* diag, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return reduce_entry<double>(ctx, day, night, n, diag, ddiag, stream);
}
extern "C" int mod16_reduce_diag_f32(mod16_ctx* ctx, const float* day, const float* night,
                                     int64_t n, double* diag, double* ddiag, void* stream) {
    MOD16_LOCK(ctx);
    return reduce_entry<float>(ctx, day, night, n, diag, ddiag, stream);
}

// rank-order fold of the gathered diagnostics vectors (mod16_amd/dist.py)
__global__ void fold_diag_kernel(const double* gathered, int world, double* diag) {
    const int k = threadIdx.x;
    if (k >= kDiag) return;
    double acc = gathered[k];
    for (int r = 1; r < world; ++r) {          // fixed order: rank 0 + rank 1 + ...
        const double o = gathered[r * kDiag + k];
        acc = k < 6 ? acc + o : (o > acc ? o : acc);
    }
    diag[k] = acc;
}
extern "C" int mod16_fold_diag(mod16_ctx* ctx, const double* gathered, int world, double* diag, void* stream) {
    MOD16_LOCK(ctx);
    if (!ctx || !gathered || !diag || world < 1) return fail(ctx, MOD16_ERR_ARG, "mod16_fold_diag: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(fold_diag_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), gathered, world, diag);
    HIPCHK(ctx, hipGetLastError());
    return MOD16_OK;
}


// ---------------------------------------------- parameter rasters -> class raster
// (mod16_classify_*: DEVICE pointers; waits for the stream, because the caller decides on the answer)
template <typename T>
static int classify_entry(mod16_ctx* ctx, const T* const* params, const int64_t* pstride, int64_t n,
                          const T* rows, int nrows, uint8_t* cls, int64_t* unmatched, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!params || !pstride || !rows || !cls || !unmatched || n < 0 || nrows < 1 || nrows > kClassRows)
        return fail(ctx, MOD16_ERR_ARG, "mod16_classify: params, pstride, rows (1 .. 13), cls and unmatched are required");
    *unmatched = -1;
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    ClassifyArgs<T> a;
    a.dense = 0;
    for (int k = 0; k < kClassPars; ++k) {
        if (!params[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_classify: a parameter pointer is NULL");
        if (pstride[k] != 0 && pstride[k] != 1) return fail(ctx, MOD16_ERR_ARG, "mod16_classify: strides are 0 (one value) or 1 (a raster)");
        a.par[k] = params[k];
        if (pstride[k]) a.dense |= 1u << k;
    }
    // rows (host) and the answer word share one small device block
    char* block = nullptr;
    const size_t rbytes = (sizeof(T) * kClassRows * kClassPars + 15) / 16 * 16;      // (the 64-bit answer word behind them: aligned)
    HIPCHK(ctx, hipMalloc(&block, rbytes + 8));
    const unsigned long long none = ~0ull;
    hipError_t e = hipMemcpyAsync(block, rows, sizeof(T) * nrows * kClassPars, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(block + rbytes, &none, 8, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        a.rows = reinterpret_cast<const T*>(block);
        a.nrows = nrows;
        a.n = n;
        a.cls = cls;
        a.unmatched = reinterpret_cast<unsigned long long*>(block + rbytes);
        const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)ctx->cus * 16));
        hipLaunchKernelGGL(classify_kernel<T>, dim3(grid), dim3(kBlock), 0, st, a);
        e = hipGetLastError();
    }
    unsigned long long got = none;
    if (e == hipSuccess) e = hipMemcpyAsync(&got, block + rbytes, 8, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(block);
    HIPCHK(ctx, e);
    *unmatched = got == none ? -1 : (int64_t)got;
    return MOD16_OK;
}

extern "C" int mod16_classify_f64(mod16_ctx* ctx, const double* const* params, const int64_t* pstride, int64_t n,
                                  const double* rows, int nrows, uint8_t* cls, int64_t* unmatched, void* stream) {
    MOD16_LOCK(ctx);
    return classify_entry<double>(ctx, params, pstride, n, rows, nrows, cls, unmatched, stream);
}
extern "C" int mod16_classify_f32(mod16_ctx* ctx, const float* const* params, const int64_t* pstride, int64_t n,
                                  const float* rows, int nrows, uint8_t* cls, int64_t* unmatched, void* stream) {
    MOD16_LOCK(ctx);
    return classify_entry<float>(ctx, params, pstride, n, rows, nrows, cls, unmatched, stream);
}


template <typename T>
static int synth_entry(mod16_ctx* ctx, uint64_t seed, int64_t step, int64_t pixel_offset,
                       int64_t n, uint8_t* cls, T* const* drivers, void* stream,
                       const mod16_layout* lay = nullptr) {
    if (!ctx || !drivers || n < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_synth: bad argument");
    SynthArgs<T> a;
    a.tile_shift = 62;
    a.drv_row = a.cls_row = 0;
    if (lay && lay->tile > 0) {
        a.tile_shift = tile_log2(lay->tile, 1);
        if (a.tile_shift < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_synth_tiled: tile must be a power of two");
        a.drv_row = lay->driver_row;
        a.cls_row = lay->cls_row;
    }
    a.cls = cls;
    for (int k = 0; k < 14; ++k) {
        if (!drivers[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_synth: NULL driver array");
        a.drv[k] = drivers[k];
    }
    a.seed = seed;
    a.step = step;
    a.offset = pixel_offset;
    a.n = n;
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int grid = (int)std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)ctx->cus * 16);
    hipLaunchKernelGGL((synth_kernel<T>), dim3(grid), dim3(kBlock), 0, static_cast<hipStream_t>(stream), a);
    HIPCHK(ctx, hipGetLastError());
    return MOD16_OK;
}

extern "C" int mod16_synth_f64(mod16_ctx* ctx, uint64_t seed, int64_t step, int64_t pixel_offset,
                               int64_t n, uint8_t* cls, double* const* drivers, void* stream) {
    MOD16_LOCK(ctx);
    return synth_entry<double>(ctx, seed, step, pixel_offset, n, cls, drivers, stream);
}
extern "C" int mod16_synth_f32(mod16_ctx* ctx, uint64_t seed, int64_t step, int64_t pixel_offset,
                               int64_t n, uint8_t* cls, float* const* drivers, void* stream) {
    MOD16_LOCK(ctx);
    return synth_entry<float>(ctx, seed, step, pixel_offset, n, cls, drivers, stream);
}

extern "C" int mod16_synth_tiled_f64(mod16_ctx* ctx, const mod16_layout* layout, uint64_t seed,
                                     int64_t step, int64_t pixel_offset, int64_t n, uint8_t* cls,
                                     double* const* drivers, void* stream) {
    MOD16_LOCK(ctx);
    return synth_entry<double>(ctx, seed, step, pixel_offset, n, cls, drivers, stream, layout);
}
extern "C" int mod16_synth_tiled_f32(mod16_ctx* ctx, const mod16_layout* layout, uint64_t seed,
                                     int64_t step, int64_t pixel_offset, int64_t n, uint8_t* cls,
                                     float* const* drivers, void* stream) {
    MOD16_LOCK(ctx);
    return synth_entry<float>(ctx, seed, step, pixel_offset, n, cls, drivers, stream, layout);
}
