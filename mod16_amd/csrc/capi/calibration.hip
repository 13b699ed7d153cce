// libmod16hip.so -- the vectorised calibration path (N2): mod16_et_static_*, mod16_et_static_batch_*, mod16_static_batch_*
#include "internal.hpp"
#include "../mod16_methods.hpp"

// ------------------------------------------- vectorised calibration path (N2)
// HOST-mode workspace of the calibration entry points, kept in the context between calls (a
// calibration loop repeats the same shape thousands of times): it only grows; above kBatchKeepBytes
// it is given back after the call.
constexpr size_t kBatchKeepBytes = size_t(8) << 30;
static int batch_reserve(mod16_ctx* ctx, size_t total) {
    if (ctx->batch_bytes >= total) return MOD16_OK;
    if (ctx->batch_buf) HIPCHK(ctx, hipFree(ctx->batch_buf));
    ctx->batch_buf = nullptr;
    ctx->batch_bytes = 0;
    if (hipMalloc(&ctx->batch_buf, total) != hipSuccess) {
        (void)hipGetLastError();
        ctx->batch_buf = nullptr;
        return fail(ctx, MOD16_ERR_NOMEM, "mod16_et_static*: device memory for the calibration workspace");
    }
    ctx->batch_bytes = total;
    return MOD16_OK;
}
static void batch_trim(mod16_ctx* ctx) {
    if (ctx->batch_bytes <= kBatchKeepBytes) return;
    (void)hipFree(ctx->batch_buf);
    ctx->batch_buf = nullptr;
    ctx->batch_bytes = 0;
}

template <typename T>
static int static_entry(mod16_ctx* ctx, const T* const* drivers, const int64_t* dstride,
                        const T* const* params, const int64_t* pstride, const T* const* rcorr,
                        const int64_t* rstride, int64_t n, T* out_day, T* out_night, T tiny,
                        int where, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!drivers || !dstride || !params || !pstride || !out_day || !out_night || n < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_static: bad argument");
    StaticArgs<T> a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 14; ++k) {
        if (!drivers[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static: NULL driver");
        a.drv[k] = drivers[k];
        if (dstride[k]) a.dense_drv |= 1u << k;
    }
    for (int k = 0; k < 11; ++k) {
        if (!params[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static: NULL parameter");
        a.par[k] = params[k];
        if (pstride[k]) a.dense_par |= 1u << k;
    }
    if (rcorr) {
        if (!rcorr[0] || !rcorr[1] || !rstride) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static: r_corr_list needs two arrays");
        for (int k = 0; k < 2; ++k) {
            a.rc[k] = rcorr[k];
            if (rstride[k]) a.dense_rc |= 1u << k;
        }
    }
    a.out[0] = out_day;
    a.out[1] = out_night;
    a.n = n;
    a.tiny = tiny;
    if (n == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->static_flag) HIPCHK(ctx, hipMalloc(&ctx->static_flag, sizeof(unsigned)));
    a.flag = ctx->static_flag;
    auto grid_of = [&](int64_t m) {
        return (int)std::max<int64_t>(1, std::min<int64_t>((m + kBlock - 1) / kBlock, (int64_t)ctx->cus * 8));
    };
    if (where == MOD16_DEVICE) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        HIPCHK(ctx, hipMemsetAsync(a.flag, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL((static_flag_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, a);
        hipLaunchKernelGGL((static_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, a);
        HIPCHK(ctx, hipGetLastError());
        return MOD16_OK;
    }
    if (where != MOD16_HOST) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static: bad `where`");
    constexpr int kArr = 14 + 11 + 2 + 2;
    size_t per_arr_small = 0;
    if (n <= ctx->small_pixels && small_reserve(ctx, n, sizeof(T), kArr, &per_arr_small)) {
        // what a sampler calls once per draw (a few sites x a year): no allocation, no copy commands --
        // the two kernels read the page-locked buffer and write their outputs there (run_host_small)
        const size_t per_arr = per_arr_small;
        hipStream_t st = ctx->streams[0];
        char* hb = static_cast<char*>(ctx->small_host);
        char* db = static_cast<char*>(ctx->small_dev);
        StaticArgs<T> d = a;
        int slot = 0;
        auto put = [&](const T* src, bool dense) -> const T* {
            const size_t off = 256 + per_arr * slot++;
            memcpy(hb + off, src, sizeof(T) * (dense ? n : 1));
            return reinterpret_cast<const T*>(db + off);
        };
        for (int k = 0; k < 14; ++k) d.drv[k] = put(a.drv[k], (a.dense_drv >> k) & 1u);
        for (int k = 0; k < 11; ++k) d.par[k] = put(a.par[k], (a.dense_par >> k) & 1u);
        for (int k = 0; k < 2; ++k) d.rc[k] = a.rc[k] ? put(a.rc[k], (a.dense_rc >> k) & 1u) : nullptr;
        const size_t o0 = 256 + per_arr * 27, o1 = 256 + per_arr * 28;
        d.out[0] = reinterpret_cast<T*>(db + o0);
        d.out[1] = reinterpret_cast<T*>(db + o1);
        HIPCHK(ctx, hipMemsetAsync(d.flag, 0, sizeof(unsigned), st));
        hipLaunchKernelGGL((static_flag_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, d);
        hipLaunchKernelGGL((static_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, d);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(st));
        memcpy(out_day, hb + o0, sizeof(T) * n);
        memcpy(out_night, hb + o1, sizeof(T) * n);
        return MOD16_OK;
    }
    // HOST: the whole-array branch needs every pixel before any output, so the
    // inputs are made resident once (calibration-sized arrays, not rasters)
    const size_t per_arr = (((size_t)n * sizeof(T)) + 255) / 256 * 256;
    // the workspace the context keeps between calibration calls (mod16_et_static_batch_* shares it;
    // until round 5 this entry point allocated and freed its own on every call)
    int rcw = batch_reserve(ctx, per_arr * kArr);
    if (rcw != MOD16_OK) return rcw;
    if (!ctx->streams[0]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking));
    hipStream_t st = ctx->streams[0];
    char* base = static_cast<char*>(ctx->batch_buf);
    StaticArgs<T> d = a;
    int slot = 0;
    int rc_status = MOD16_OK;
    auto up = [&](const T* src, bool dense) -> const T* {
        T* dp = reinterpret_cast<T*>(base + per_arr * slot++);
        hipError_t e = hipMemcpyAsync(dp, src, sizeof(T) * (dense ? n : 1), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) rc_status = MOD16_ERR_HIP;
        return dp;
    };
    for (int k = 0; k < 14; ++k) d.drv[k] = up(a.drv[k], (a.dense_drv >> k) & 1u);
    for (int k = 0; k < 11; ++k) d.par[k] = up(a.par[k], (a.dense_par >> k) & 1u);
    for (int k = 0; k < 2; ++k) d.rc[k] = a.rc[k] ? up(a.rc[k], (a.dense_rc >> k) & 1u) : nullptr;
    slot = 27;
    d.out[0] = reinterpret_cast<T*>(base + per_arr * slot++);
    d.out[1] = reinterpret_cast<T*>(base + per_arr * slot++);
    if (rc_status == MOD16_OK) {
        (void)hipMemsetAsync(d.flag, 0, sizeof(unsigned), st);
        hipLaunchKernelGGL((static_flag_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, d);
        hipLaunchKernelGGL((static_kernel<T>), dim3(grid_of(n)), dim3(kBlock), 0, st, d);
        if (hipGetLastError() != hipSuccess) rc_status = MOD16_ERR_HIP;
        if (hipMemcpyAsync(out_day, d.out[0], sizeof(T) * n, hipMemcpyDeviceToHost, st) != hipSuccess) rc_status = MOD16_ERR_HIP;
        if (hipMemcpyAsync(out_night, d.out[1], sizeof(T) * n, hipMemcpyDeviceToHost, st) != hipSuccess) rc_status = MOD16_ERR_HIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) rc_status = MOD16_ERR_HIP;
    batch_trim(ctx);
    if (rc_status != MOD16_OK) ctx->err = "mod16_et_static: HIP call failed";
    return rc_status;
}

extern "C" int mod16_et_static_f64(mod16_ctx* ctx, const double* const* drivers,
                                   const int64_t* dstride, const double* const* params,
                                   const int64_t* pstride, const double* const* rcorr,
                                   const int64_t* rstride, int64_t n, double* out_day,
                                   double* out_night, double tiny, int where, void* stream) {
    MOD16_LOCK(ctx);
    return static_entry<double>(ctx, drivers, dstride, params, pstride, rcorr, rstride, n, out_day, out_night, tiny, where, stream);
}
extern "C" int mod16_et_static_f32(mod16_ctx* ctx, const float* const* drivers,
                                   const int64_t* dstride, const float* const* params,
                                   const int64_t* pstride, const float* const* rcorr,
                                   const int64_t* rstride, int64_t n, float* out_day,
                                   float* out_night, float tiny, int where, void* stream) {
    MOD16_LOCK(ctx);
    return static_entry<float>(ctx, drivers, dstride, params, pstride, rcorr, rstride, n, out_day, out_night, tiny, where, stream);
}

// ---------------------- calibration path batched over parameter vectors (N2)
__global__ void zero_u32_kernel(unsigned* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) p[i] = 0u;
}

// The device-side pass of the batched calibration path over device pointers: d.drv / d.params /
// d.out are set; rows [ndraw][n] into d.out, and with dsse the objective from d.out[2]. FAST: the
// pixels outside the domain of the strength-reduced arithmetic (dskip, [n] bytes of workspace) are
// left out by the FAST kernels and computed in the reference's operation order behind them.
template <typename T>
static int static_batch_rows(mod16_ctx* ctx, StaticBatchArgs<T> d, int64_t ndraw, const T* dobs, const T* dw,
                             double* dsse, double* dcnt, unsigned* dflags, uint8_t* dskip, unsigned flags,
                             hipStream_t st, bool skip_ready = false) {
    const int64_t n = d.n;
    d.flags = dflags;
    d.tab = ctx->tab64;
    d.ndraw = ndraw;
    const bool fast = (flags & MOD16_MATH_EXACT) == 0;
    d.skip = fast ? dskip : nullptr;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((unsigned)((ndraw + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, dflags, ndraw);
    const int gx = (int)std::max<int64_t>(1, std::min<int64_t>((n + kBlock - 1) / kBlock, (int64_t)ctx->cus * 8));
    if (fast && !skip_ready)
        hipLaunchKernelGGL((static_domain_kernel<T>), dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d, dskip);
    for (int64_t d0 = 0; d0 < ndraw; d0 += 32768 * (int64_t)kBatchDraws) {
        const unsigned gy = (unsigned)((std::min<int64_t>(32768 * (int64_t)kBatchDraws, ndraw - d0) + kBatchDraws - 1) / kBatchDraws);
        d.draw0 = d0;
        if (fast) {
            hipLaunchKernelGGL((static_batch_flag_fast_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
            hipLaunchKernelGGL((static_batch_flag_skipped_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
            hipLaunchKernelGGL((static_batch_fast_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
            hipLaunchKernelGGL((static_batch_redo_rows_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
        } else {
            hipLaunchKernelGGL((static_batch_flag_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
            hipLaunchKernelGGL((static_batch_kernel<T>), dim3(gx, gy), dim3(kBlock), 0, st, d);
        }
    }
    if (dsse)
        hipLaunchKernelGGL((static_batch_sse_kernel<T>), dim3((unsigned)ndraw), dim3(kBlock), 0, st,
                           d.out[2], dobs, dw, n, dsse, dcnt);
    HIPCHK(ctx, hipGetLastError());
    return MOD16_OK;
}

template <typename T>
static int static_batch_entry(mod16_ctx* ctx, const T* const* drivers, const int64_t* dstride,
                              int64_t n, const T* params, int64_t ndraw, T* out_day, T* out_night,
                              T* out_total, const T* observed, const T* weights, double* sse,
                              double* count, unsigned flags, int where, void* stream) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!drivers || !dstride || !params || n < 0 || ndraw < 0)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: bad argument");
    if (!out_day && !out_night && !out_total && !sse)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: no output requested");
    if ((sse != nullptr) != (count != nullptr) || (sse && !observed))
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: sse needs count and observed");
    if (where == MOD16_DEVICE && sse && !out_total)
        return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: sse on device pointers needs out_total as workspace");
    StaticBatchArgs<T> a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 14; ++k) {
        if (!drivers[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: NULL driver");
        if (dstride[k] != 0 && dstride[k] != 1) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: driver stride must be 0 or 1");
        a.drv[k] = drivers[k];
        if (dstride[k]) a.dense_drv |= 1u << k;
    }
    a.n = n;
    if (n == 0 || ndraw == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ndraw > 0x7fffffff) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: too many draws");
    if (where == MOD16_DEVICE) {
        hipStream_t st = static_cast<hipStream_t>(stream);
        unsigned* dflags = nullptr;
        uint8_t* dskip = nullptr;
        // flags and the domain mask: per-call allocations freed on the stream (asynchronous)
        HIPCHK(ctx, hipMallocAsync(reinterpret_cast<void**>(&dflags), sizeof(unsigned) * ndraw, st));
        HIPCHK(ctx, hipMallocAsync(reinterpret_cast<void**>(&dskip), (size_t)n, st));
        a.params = params;
        a.out[0] = out_day; a.out[1] = out_night; a.out[2] = out_total;
        int rc = static_batch_rows<T>(ctx, a, ndraw, observed, weights, sse, count, dflags, dskip, flags, st);
        (void)hipFreeAsync(dflags, st);
        (void)hipFreeAsync(dskip, st);
        return rc;
    }
    if (where != MOD16_HOST) return fail(ctx, MOD16_ERR_ARG, "mod16_et_static_batch: bad `where`");
    // HOST: drivers / parameters resident once, outputs [ndraw][n] come back
    if (!ctx->streams[0]) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking));
    hipStream_t st = ctx->streams[0];
    const size_t per_arr = (((size_t)n * sizeof(T)) + 255) / 256 * 256;
    const size_t per_out = (((size_t)n * (size_t)ndraw * sizeof(T)) + 255) / 256 * 256;
    const bool want[3] = {out_day != nullptr, out_night != nullptr, out_total != nullptr || sse != nullptr};
    T* const host_out[3] = {out_day, out_night, out_total};
    const size_t par_b = (((size_t)ndraw * 11 * sizeof(T)) + 255) / 256 * 256;
    const size_t red_b = (((size_t)ndraw * sizeof(double)) + 255) / 256 * 256;
    const size_t flag_b = (((size_t)ndraw * sizeof(unsigned)) + 255) / 256 * 256;
    const size_t skip_b = ((size_t)n + 255) / 256 * 256;
    const size_t total = per_arr * 16 + par_b + 2 * red_b + flag_b + skip_b +
                         per_out * ((int)want[0] + (int)want[1] + (int)want[2]);
    // workspace kept in the context between calls (a calibration loop repeats the same
    // shape thousands of times -- better still: mod16_static_batch_bind_*); it only grows, up to
    // kBatchKeepBytes it is kept
    {
        int rcw = batch_reserve(ctx, total);
        if (rcw != MOD16_OK) return rcw;
    }
    char* base = static_cast<char*>(ctx->batch_buf);
    int rc = MOD16_OK;
    auto chk = [&](hipError_t e) { if (e != hipSuccess && rc == MOD16_OK) { rc = MOD16_ERR_HIP; ctx->err = hipGetErrorString(e); } };
    char* cur = base;
    auto take = [&](size_t b) { char* p = cur; cur += b; return p; };
    StaticBatchArgs<T> d = a;
    for (int k = 0; k < 14; ++k) {
        T* dp = reinterpret_cast<T*>(take(per_arr));
        chk(hipMemcpyAsync(dp, a.drv[k], sizeof(T) * (((a.dense_drv >> k) & 1u) ? n : 1), hipMemcpyHostToDevice, st));
        d.drv[k] = dp;
    }
    T* dobs = reinterpret_cast<T*>(take(per_arr));
    T* dw = reinterpret_cast<T*>(take(per_arr));
    if (sse) chk(hipMemcpyAsync(dobs, observed, sizeof(T) * n, hipMemcpyHostToDevice, st));
    if (sse && weights) chk(hipMemcpyAsync(dw, weights, sizeof(T) * n, hipMemcpyHostToDevice, st));
    T* dpar = reinterpret_cast<T*>(take(par_b));
    chk(hipMemcpyAsync(dpar, params, sizeof(T) * ndraw * 11, hipMemcpyHostToDevice, st));
    d.params = dpar;
    double* dsse = reinterpret_cast<double*>(take(red_b));
    double* dcnt = reinterpret_cast<double*>(take(red_b));
    unsigned* dflags = reinterpret_cast<unsigned*>(take(flag_b));
    uint8_t* dskip = reinterpret_cast<uint8_t*>(take(skip_b));
    for (int k = 0; k < 3; ++k) d.out[k] = want[k] ? reinterpret_cast<T*>(take(per_out)) : nullptr;
    if (rc == MOD16_OK)
        rc = static_batch_rows<T>(ctx, d, ndraw, dobs, (sse && weights) ? dw : nullptr, sse ? dsse : nullptr, dcnt,
                                  dflags, dskip, flags, st);
    if (rc == MOD16_OK) {
        for (int k = 0; k < 3; ++k)
            if (host_out[k]) chk(hipMemcpyAsync(host_out[k], d.out[k], sizeof(T) * n * ndraw, hipMemcpyDeviceToHost, st));
        if (sse) {
            chk(hipMemcpyAsync(sse, dsse, sizeof(double) * ndraw, hipMemcpyDeviceToHost, st));
            chk(hipMemcpyAsync(count, dcnt, sizeof(double) * ndraw, hipMemcpyDeviceToHost, st));
        }
    }
    chk(hipStreamSynchronize(st));
    batch_trim(ctx);
    return rc;
}

extern "C" int mod16_et_static_batch_f64(mod16_ctx* ctx, const double* const* drivers,
                                         const int64_t* dstride, int64_t n, const double* params,
                                         int64_t ndraw, double* out_day, double* out_night,
                                         double* out_total, const double* observed,
                                         const double* weights, double* sse, double* count,
                                         unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    return static_batch_entry<double>(ctx, drivers, dstride, n, params, ndraw, out_day, out_night,
                                      out_total, observed, weights, sse, count, flags, where, stream);
}
extern "C" int mod16_et_static_batch_f32(mod16_ctx* ctx, const float* const* drivers,
                                         const int64_t* dstride, int64_t n, const float* params,
                                         int64_t ndraw, float* out_day, float* out_night,
                                         float* out_total, const float* observed,
                                         const float* weights, double* sse, double* count,
                                         unsigned flags, int where, void* stream) {
    MOD16_LOCK(ctx);
    return static_batch_entry<float>(ctx, drivers, dstride, n, params, ndraw, out_day, out_night,
                                     out_total, observed, weights, sse, count, flags, where, stream);
}

// ---- the calibration problem RESIDENT on the device (mod16_static_batch_bind_*): drivers,
// observations and weights go up once; an evaluation is parameters up, one graph launch (kernels
// only), (sse, count) down.
struct mod16_batch {
    mod16_ctx* ctx = nullptr;
    int device = 0;
    bool f32 = false;
    unsigned flags = 0;
    int64_t n = 0, max_draws = 0;
    int gx = 0;
    void* owned = nullptr;              // the resident copies (HOST bind); NULL when the caller's device arrays are used
    const void* drv[14] = {};
    uint32_t dense_drv = 0;
    const void* obs = nullptr;
    const void* wts = nullptr;
    uint8_t* skip = nullptr;            // [n]: 1 = outside the FAST domain
    int64_t* list = nullptr;            // those pixels, ascending
    int64_t nlist = 0;
    void* ws = nullptr;                 // evaluation workspace (one allocation)
    void* dparams = nullptr;            // [max_draws][11] of the data type
    double *par16 = nullptr, *partial = nullptr, *redo = nullptr, *dsse = nullptr, *dcnt = nullptr;
    unsigned *any_gs = nullptr, *any_draw = nullptr, *dflags = nullptr;
    void* eval_ws = nullptr;            // partial + any_gs of the FAST objective: sized for the draws actually evaluated
    int64_t eval_draws = 0;             //   (grown on demand; max_draws x blocks x 20 bytes would be GBs for large n)
    void* rows = nullptr;               // [ndraw][n] x up to 3: rows workspace, allocated when first asked for
    size_t rows_bytes = 0;
    void* hparams = nullptr;            // pinned staging
    double* hout = nullptr;             // pinned [2][max_draws]
    hipStream_t st = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int64_t graph_ndraw = -1;
};

extern "C" int mod16_static_batch_destroy(mod16_batch* b) {
    if (!b) return MOD16_OK;
    (void)hipSetDevice(b->device);
    if (b->st) (void)hipStreamSynchronize(b->st);
    if (b->exec) (void)hipGraphExecDestroy(b->exec);
    if (b->graph) (void)hipGraphDestroy(b->graph);
    if (b->owned) (void)hipFree(b->owned);
    if (b->skip) (void)hipFree(b->skip);
    if (b->list) (void)hipFree(b->list);
    if (b->ws) (void)hipFree(b->ws);
    if (b->eval_ws) (void)hipFree(b->eval_ws);
    if (b->rows) (void)hipFree(b->rows);
    if (b->hparams) (void)hipHostFree(b->hparams);
    if (b->hout) (void)hipHostFree(b->hout);
    if (b->st) (void)hipStreamDestroy(b->st);
    delete b;
    return MOD16_OK;
}

template <typename T>
static StaticBatchArgs<T> batch_args(const mod16_batch* b) {
    StaticBatchArgs<T> a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 14; ++k) a.drv[k] = static_cast<const T*>(b->drv[k]);
    a.dense_drv = b->dense_drv;
    a.n = b->n;
    a.params = static_cast<const T*>(b->dparams);
    return a;
}

template <typename T>
static int batch_bind(mod16_ctx* ctx, const T* const* drivers, const int64_t* dstride, int64_t n,
                      const T* observed, const T* weights, int64_t max_draws, unsigned flags, int where,
                      mod16_batch** out) {
    if (!ctx || !out) return MOD16_ERR_ARG;
    *out = nullptr;
    // (a launch evaluates 32 draws per block row: 65535 rows at most)
    if (!drivers || !dstride || n <= 0 || max_draws <= 0 || max_draws > (int64_t)65535 * kObjDraws)
        return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_bind: NULL drivers, n <= 0 or max_draws outside 1 .. 2097120");
    if (weights && !observed) return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_bind: weights need observed");
    if (where != MOD16_HOST && where != MOD16_DEVICE) return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_bind: bad `where`");
    for (int k = 0; k < 14; ++k) {
        if (!drivers[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_bind: NULL driver");
        if (dstride[k] != 0 && dstride[k] != 1) return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_bind: driver stride must be 0 or 1");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    mod16_batch* b = new (std::nothrow) mod16_batch;
    if (!b) return MOD16_ERR_NOMEM;
    b->ctx = ctx;
    b->device = ctx->device;
    b->f32 = std::is_same<T, float>::value;
    b->flags = flags;
    b->n = n;
    b->max_draws = max_draws;
    b->gx = (int)((n + kBlock - 1) / kBlock);
    int rc = [&]() -> int {
        HIPCHK(ctx, hipStreamCreateWithFlags(&b->st, hipStreamNonBlocking));
        const size_t per_arr = (((size_t)n * sizeof(T)) + 255) / 256 * 256;
        for (int k = 0; k < 14; ++k) if (dstride[k]) b->dense_drv |= 1u << k;
        // device memory that cannot be had is MOD16_ERR_NOMEM, not a HIP error
        auto dmalloc = [&](void** p, size_t bytes, const char* what) -> int {
            if (hipMalloc(p, bytes) == hipSuccess) return MOD16_OK;
            (void)hipGetLastError();
            *p = nullptr;
            ctx->err = std::string("mod16_static_batch_bind: device memory for ") + what;
            return MOD16_ERR_NOMEM;
        };
#define MOD16_DMALLOC(p, bytes, what) do { int r_ = dmalloc(reinterpret_cast<void**>(p), bytes, what); if (r_ != MOD16_OK) return r_; } while (0)
        if (where == MOD16_HOST) {
            MOD16_DMALLOC(&b->owned, per_arr * 16, "the resident drivers");
            char* base = static_cast<char*>(b->owned);
            for (int k = 0; k < 14; ++k) {
                HIPCHK(ctx, hipMemcpyAsync(base + per_arr * k, drivers[k], sizeof(T) * (dstride[k] ? n : 1), hipMemcpyHostToDevice, b->st));
                b->drv[k] = base + per_arr * k;
            }
            if (observed) {
                HIPCHK(ctx, hipMemcpyAsync(base + per_arr * 14, observed, sizeof(T) * n, hipMemcpyHostToDevice, b->st));
                b->obs = base + per_arr * 14;
            }
            if (weights) {
                HIPCHK(ctx, hipMemcpyAsync(base + per_arr * 15, weights, sizeof(T) * n, hipMemcpyHostToDevice, b->st));
                b->wts = base + per_arr * 15;
            }
        } else {
            for (int k = 0; k < 14; ++k) b->drv[k] = drivers[k];
            b->obs = observed;
            b->wts = weights;
        }
        // evaluation workspace
        const int64_t D = max_draws;
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        // (the per-block partials of the FAST objective -- draws x blocks x 20 bytes, 3.2 GB at 4096
        // draws x 10 M pixels -- are NOT part of this: batch_eval_ws sizes them for the draws an
        // evaluation actually brings; an EXACT problem never has them)
        const size_t sz_par = al((size_t)D * 11 * sizeof(T)), sz_p16 = al((size_t)D * kPar16 * 8),
                     sz_d = al((size_t)D * 8), sz_redo = al((size_t)D * 40), sz_u = al((size_t)D * 4);
        MOD16_DMALLOC(&b->ws, sz_par + sz_p16 + 2 * sz_d + sz_redo + 2 * sz_u, "the evaluation workspace");
        char* cur = static_cast<char*>(b->ws);
        auto take = [&](size_t x) { char* p = cur; cur += x; return p; };
        b->dparams = take(sz_par);
        b->par16 = reinterpret_cast<double*>(take(sz_p16));
        b->dsse = reinterpret_cast<double*>(take(sz_d));
        b->dcnt = reinterpret_cast<double*>(take(sz_d));
        b->redo = reinterpret_cast<double*>(take(sz_redo));
        b->any_draw = reinterpret_cast<unsigned*>(take(sz_u));
        b->dflags = reinterpret_cast<unsigned*>(take(sz_u));
        HIPCHK(ctx, hipHostMalloc(&b->hparams, (size_t)D * 11 * sizeof(T)));
        HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&b->hout), (size_t)D * 16));
        // the pixels outside the domain of the FAST arithmetic: marked once, listed in ascending order
        MOD16_DMALLOC(&b->skip, (size_t)n, "the domain mask");
        StaticBatchArgs<T> a = batch_args<T>(b);
        hipLaunchKernelGGL((static_domain_kernel<T>), dim3((unsigned)b->gx), dim3(kBlock), 0, b->st, a, b->skip);
        HIPCHK(ctx, hipGetLastError());
        std::vector<uint8_t> mask((size_t)n);
        HIPCHK(ctx, hipMemcpyAsync(mask.data(), b->skip, (size_t)n, hipMemcpyDeviceToHost, b->st));
        HIPCHK(ctx, hipStreamSynchronize(b->st));
        std::vector<int64_t> list;
        for (int64_t i = 0; i < n; ++i) if (mask[(size_t)i]) list.push_back(i);
        b->nlist = (int64_t)list.size();
        if (b->nlist) {
            MOD16_DMALLOC(&b->list, sizeof(int64_t) * list.size(), "the list of pixels outside the domain");
#undef MOD16_DMALLOC
            HIPCHK(ctx, hipMemcpy(b->list, list.data(), sizeof(int64_t) * list.size(), hipMemcpyHostToDevice));
        }
        return MOD16_OK;
    }();
    if (rc != MOD16_OK) {
        mod16_static_batch_destroy(b);
        return rc;
    }
    *out = b;
    return MOD16_OK;
}

extern "C" int mod16_static_batch_bind_f64(mod16_ctx* ctx, const double* const* drivers, const int64_t* dstride,
                                           int64_t n, const double* observed, const double* weights,
                                           int64_t max_draws, unsigned flags, int where, mod16_batch** out) {
    MOD16_LOCK(ctx);
    return batch_bind<double>(ctx, drivers, dstride, n, observed, weights, max_draws, flags, where, out);
}
extern "C" int mod16_static_batch_bind_f32(mod16_ctx* ctx, const float* const* drivers, const int64_t* dstride,
                                           int64_t n, const float* observed, const float* weights,
                                           int64_t max_draws, unsigned flags, int where, mod16_batch** out) {
    MOD16_LOCK(ctx);
    return batch_bind<float>(ctx, drivers, dstride, n, observed, weights, max_draws, flags, where, out);
}

extern "C" int mod16_static_batch_info(const mod16_batch* b, int64_t* n, int64_t* max_draws, int64_t* n_outside_domain) {
    if (!b) return MOD16_ERR_ARG;
    if (n) *n = b->n;
    if (max_draws) *max_draws = b->max_draws;
    if (n_outside_domain) *n_outside_domain = b->nlist;
    return MOD16_OK;
}

// the kernels of one objective evaluation (FAST arithmetic), enqueued on b->st
// The per-block partials and flags of the FAST objective for `ndraw` draws (grown to the next power
// of two, at most max_draws; a captured graph holds the old addresses: dropped with them).
static int batch_eval_ws(mod16_batch* b, int64_t ndraw) {
    if (ndraw <= b->eval_draws) return MOD16_OK;
    mod16_ctx* ctx = b->ctx;
    int64_t want = 64;
    while (want < ndraw) want *= 2;
    want = std::min(want, b->max_draws);
    if (b->exec) (void)hipGraphExecDestroy(b->exec);
    if (b->graph) (void)hipGraphDestroy(b->graph);
    b->exec = nullptr;
    b->graph = nullptr;
    b->graph_ndraw = -1;
    HIPCHK(ctx, hipStreamSynchronize(b->st));
    if (b->eval_ws) HIPCHK(ctx, hipFree(b->eval_ws));
    b->eval_ws = nullptr;
    b->eval_draws = 0;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t sz_part = al((size_t)want * b->gx * 16), sz_any = al((size_t)want * b->gx * 4);
    if (hipMalloc(&b->eval_ws, sz_part + sz_any) != hipSuccess) {
        (void)hipGetLastError();
        b->eval_ws = nullptr;
        return fail(ctx, MOD16_ERR_NOMEM, "mod16_static_batch_objective: device memory for the per-block partials of this many draws");
    }
    b->partial = reinterpret_cast<double*>(b->eval_ws);
    b->any_gs = reinterpret_cast<unsigned*>(static_cast<char*>(b->eval_ws) + sz_part);
    b->eval_draws = want;
    return MOD16_OK;
}

template <typename T>
static void batch_objective_launches(mod16_batch* b, int64_t ndraw) {
    hipStream_t st = b->st;
    const unsigned gd = (unsigned)((ndraw + kBlock - 1) / kBlock);
    hipLaunchKernelGGL((static_obj_params_kernel<T>), dim3(gd), dim3(kBlock), 0, st, static_cast<const T*>(b->dparams), ndraw, b->par16);
    StaticObjArgs<T> a;
    memset(&a, 0, sizeof a);
    for (int k = 0; k < 14; ++k) a.drv[k] = static_cast<const T*>(b->drv[k]);
    a.dense_drv = b->dense_drv;
    a.n = b->n;
    a.observed = static_cast<const T*>(b->obs);
    a.weights = static_cast<const T*>(b->wts);
    a.skip = b->nlist ? b->skip : nullptr;
    a.par16 = b->par16;
    a.tab = b->ctx->tab64;
    a.ndraw = ndraw;
    a.any_draw = b->any_draw;
    a.partial = b->partial;
    a.any_gs = b->any_gs;
    const dim3 grid((unsigned)b->gx, (unsigned)((ndraw + kObjDraws - 1) / kObjDraws));
    hipLaunchKernelGGL((static_obj_kernel<T, true>), grid, dim3(kBlock), 0, st, a);
    if (b->nlist) {
        StaticObjRedoArgs<T> r;
        memset(&r, 0, sizeof r);
        for (int k = 0; k < 14; ++k) r.drv[k] = static_cast<const T*>(b->drv[k]);
        r.dense_drv = b->dense_drv;
        r.params = static_cast<const T*>(b->dparams);
        r.observed = a.observed;
        r.weights = a.weights;
        r.list = b->list;
        r.nlist = b->nlist;
        r.redo = b->redo;
        hipLaunchKernelGGL((static_obj_redo_kernel<T>), dim3((unsigned)ndraw), dim3(kBlock), 0, st, r);
    }
    const double* redo = b->nlist ? b->redo : nullptr;
    const unsigned gr = (unsigned)((ndraw + kObjPerBlock - 1) / kObjPerBlock);
    hipLaunchKernelGGL(static_obj_any_kernel, dim3(gr), dim3(kBlock), 0, st, b->any_gs, redo, ndraw, b->gx, b->any_draw);
    hipLaunchKernelGGL((static_obj_kernel<T, false>), grid, dim3(kBlock), 0, st, a);
    hipLaunchKernelGGL(static_obj_final_kernel, dim3(gr), dim3(kBlock), 0, st, b->partial, redo, b->any_draw, ndraw, b->gx,
                       b->dsse, b->dcnt);
}

template <typename T>
static int batch_objective(mod16_batch* b, const T* params, int64_t ndraw, double* sse, double* count) {
    mod16_ctx* ctx = b->ctx;
    if (!params || !sse || !count || ndraw < 0 || ndraw > b->max_draws)
        return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_objective: NULL argument or more draws than the problem was bound for");
    if (!b->obs) return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_objective: the problem was bound without observations");
    if (ndraw == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(b->device));
    memcpy(b->hparams, params, sizeof(T) * (size_t)ndraw * 11);
    HIPCHK(ctx, hipMemcpyAsync(b->dparams, b->hparams, sizeof(T) * (size_t)ndraw * 11, hipMemcpyHostToDevice, b->st));
    if (b->flags & MOD16_MATH_EXACT) {
        // reference order: rows into a workspace, then the residuals' sums (the kernels of the unbound call)
        const size_t need = sizeof(T) * (size_t)ndraw * (size_t)b->n;
        if (b->rows_bytes < need) {
            if (b->rows) HIPCHK(ctx, hipFree(b->rows));
            b->rows = nullptr;
            b->rows_bytes = 0;
            if (hipMalloc(&b->rows, need) != hipSuccess) {
                (void)hipGetLastError();
                return fail(ctx, MOD16_ERR_NOMEM, "mod16_static_batch_objective: device memory for the [ndraw][n] rows");
            }
            b->rows_bytes = need;
        }
        StaticBatchArgs<T> a = batch_args<T>(b);
        a.out[2] = static_cast<T*>(b->rows);
        int rc = static_batch_rows<T>(ctx, a, ndraw, static_cast<const T*>(b->obs), static_cast<const T*>(b->wts), b->dsse, b->dcnt,
                                      b->dflags, b->skip, b->flags, b->st, true);
        if (rc != MOD16_OK) return rc;
    } else {
        int rc = batch_eval_ws(b, ndraw);
        if (rc != MOD16_OK) return rc;
        if (b->graph_ndraw != ndraw) {          // (re)capture: the kernels' arguments hold the number of draws
            if (b->exec) (void)hipGraphExecDestroy(b->exec);
            if (b->graph) (void)hipGraphDestroy(b->graph);
            b->exec = nullptr;
            b->graph = nullptr;
            b->graph_ndraw = -1;
            HIPCHK(ctx, hipStreamBeginCapture(b->st, hipStreamCaptureModeThreadLocal));
            batch_objective_launches<T>(b, ndraw);
            hipError_t e = hipStreamEndCapture(b->st, &b->graph);
            HIPCHK(ctx, e);
            HIPCHK(ctx, hipGraphInstantiate(&b->exec, b->graph, nullptr, nullptr, 0));
            b->graph_ndraw = ndraw;
        }
        HIPCHK(ctx, hipGraphLaunch(b->exec, b->st));
    }
    HIPCHK(ctx, hipMemcpyAsync(b->hout, b->dsse, sizeof(double) * (size_t)ndraw, hipMemcpyDeviceToHost, b->st));
    HIPCHK(ctx, hipMemcpyAsync(b->hout + b->max_draws, b->dcnt, sizeof(double) * (size_t)ndraw, hipMemcpyDeviceToHost, b->st));
    HIPCHK(ctx, hipStreamSynchronize(b->st));
    memcpy(sse, b->hout, sizeof(double) * (size_t)ndraw);
    memcpy(count, b->hout + b->max_draws, sizeof(double) * (size_t)ndraw);
    return MOD16_OK;
}

extern "C" int mod16_static_batch_objective(mod16_batch* b, const void* params, int64_t ndraw, double* sse, double* count) {
    if (!b) return MOD16_ERR_ARG;
    MOD16_LOCK(b->ctx);
    return b->f32 ? batch_objective<float>(b, static_cast<const float*>(params), ndraw, sse, count)
                  : batch_objective<double>(b, static_cast<const double*>(params), ndraw, sse, count);
}

// rows [ndraw][n] (host) of the bound problem: the kernels of the unbound call on the resident drivers
template <typename T>
static int batch_rows(mod16_batch* b, const T* params, int64_t ndraw, T* out_day, T* out_night, T* out_total) {
    mod16_ctx* ctx = b->ctx;
    if (!params || ndraw < 0 || ndraw > b->max_draws || (!out_day && !out_night && !out_total))
        return fail(ctx, MOD16_ERR_ARG, "mod16_static_batch_rows: NULL argument, no output or more draws than the problem was bound for");
    if (ndraw == 0) return MOD16_OK;
    HIPCHK(ctx, hipSetDevice(b->device));
    T* const host_out[3] = {out_day, out_night, out_total};
    const size_t per_out = (sizeof(T) * (size_t)ndraw * (size_t)b->n + 255) / 256 * 256;
    const size_t need = per_out * ((out_day != nullptr) + (out_night != nullptr) + (out_total != nullptr));
    if (b->rows_bytes < need) {
        if (b->rows) HIPCHK(ctx, hipFree(b->rows));
        b->rows = nullptr;
        b->rows_bytes = 0;
        if (hipMalloc(&b->rows, need) != hipSuccess) {
            (void)hipGetLastError();
            return fail(ctx, MOD16_ERR_NOMEM, "mod16_static_batch_rows: device memory for the [ndraw][n] rows");
        }
        b->rows_bytes = need;
    }
    memcpy(b->hparams, params, sizeof(T) * (size_t)ndraw * 11);
    HIPCHK(ctx, hipMemcpyAsync(b->dparams, b->hparams, sizeof(T) * (size_t)ndraw * 11, hipMemcpyHostToDevice, b->st));
    StaticBatchArgs<T> a = batch_args<T>(b);
    char* cur = static_cast<char*>(b->rows);
    for (int k = 0; k < 3; ++k)
        if (host_out[k]) { a.out[k] = reinterpret_cast<T*>(cur); cur += per_out; }
    int rc = static_batch_rows<T>(ctx, a, ndraw, nullptr, nullptr, nullptr, nullptr, b->dflags, b->skip, b->flags, b->st, true);
    if (rc != MOD16_OK) return rc;
    for (int k = 0; k < 3; ++k)
        if (host_out[k]) HIPCHK(ctx, hipMemcpyAsync(host_out[k], a.out[k], sizeof(T) * (size_t)ndraw * (size_t)b->n, hipMemcpyDeviceToHost, b->st));
    HIPCHK(ctx, hipStreamSynchronize(b->st));
    return MOD16_OK;
}

extern "C" int mod16_static_batch_rows(mod16_batch* b, const void* params, int64_t ndraw, void* out_day, void* out_night,
                                       void* out_total) {
    if (!b) return MOD16_ERR_ARG;
    MOD16_LOCK(b->ctx);
    return b->f32 ? batch_rows<float>(b, static_cast<const float*>(params), ndraw, static_cast<float*>(out_day),
                                      static_cast<float*>(out_night), static_cast<float*>(out_total))
                  : batch_rows<double>(b, static_cast<const double*>(params), ndraw, static_cast<double*>(out_day),
                                       static_cast<double*>(out_night), static_cast<double*>(out_total));
}

// mean milliseconds of the GPU part of an objective evaluation (graph replays on the problem's
// stream, HIP events): what bench.py puts next to the wall-clock rate of the call
extern "C" int mod16_static_batch_time(mod16_batch* b, int launches, float* ms) {
    if (!b || !ms || launches <= 0 || !b->exec) return MOD16_ERR_ARG;
    MOD16_LOCK(b->ctx);
    if (hipSetDevice(b->device) != hipSuccess) return MOD16_ERR_HIP;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MOD16_ERR_HIP;
    bool ok = hipEventRecord(e0, b->st) == hipSuccess;
    for (int i = 0; i < launches && ok; ++i) ok = hipGraphLaunch(b->exec, b->st) == hipSuccess;
    ok = ok && hipEventRecord(e1, b->st) == hipSuccess && hipEventSynchronize(e1) == hipSuccess;
    float t = 0.f;
    ok = ok && hipEventElapsedTime(&t, e0, e1) == hipSuccess;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return MOD16_ERR_HIP;
    *ms = t / (float)launches;
    return MOD16_OK;
}
