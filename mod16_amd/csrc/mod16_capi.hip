// libmod16hip.so -- host side of the C ABI declared in include/mod16_hip.h.
// Owns the device context (BPLUT, status word, staging tiles, reduction
// workspace) and launches the gfx950 kernels of mod16_kernels.hpp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mod16_hip.h"
#include "mod16_kernels.hpp"
#include "mod16_stream.hpp"
#include "mod16_methods.hpp"

using namespace mod16;

namespace {
constexpr int64_t kTilePixels = int64_t(1) << 21;   // HOST mode: pixels per staged tile
constexpr int kDiagBlocks = 1024;
constexpr int kSlots = 12;                          // staging slots = host threads of the HOST mode
constexpr size_t kStagger = 33 * 1024;              // see RasterEngine.STAGGER_BYTES
constexpr int kSmallPixels = 65536;                 // HOST mode: calls up to this size take the copy-free path
constexpr int kSmallPixelsMax = 1 << 18;            // ... and what MOD16_SMALL_PIXELS may raise it to
constexpr int kSmallUnavailable = 1;                // run_host_small: no page-locked buffer -- the caller stages the call
}  // namespace

// Workspace of the per-run diagnostics partials of et_stream_kernel (and of the
// stand-alone reduction). The context owns one, sized on demand; a captured
// graph owns its own, so growing the context's never pulls memory from under
// a graph that is replayed later.
struct DiagWs {
    double* partial = nullptr;   // device [capacity][kDiag], then 128 bytes: the "blocks done" counter
    int64_t capacity = 0;        // in partials
    unsigned* done() const { return reinterpret_cast<unsigned*>(partial + capacity * 8); }
};
// hipMalloc of a workspace for `blocks` partials + the (zeroed) counter behind them
static hipError_t ws_alloc(DiagWs& ws, int64_t blocks) {
    hipError_t e = hipMalloc(&ws.partial, sizeof(double) * (blocks * 8 + 16));
    if (e != hipSuccess) return e;
    ws.capacity = blocks;
    return hipMemset(ws.done(), 0, 128);
}

struct mod16_ctx {
    std::recursive_mutex api_mu;     // every entry point holds it: a ctx may be shared by threads
    int device = 0;
    int cus = 256;
    int grid_mult = 64;              // blocks per CU in the grid-stride launches of the plain kernels
    int host_threads = 8;            // MOD16_HOST_THREADS: staging threads of the HOST mode (1..kSlots)
    // launch geometry; fixed in the shipped library, overridable in -DMOD16_EXPERIMENTS builds only
    bool use_dma = true;             // production pipeline (mod16_stream.hpp); off: plain kernels only
    int run_shift = -1;              // force 2^k pieces per run
    int stream_blocks = 2;           // blocks of the pipeline kernel per CU (1 = one wave per SIMD)
    int static_below = 8;            // runs per wave below which runs are dealt out statically (0: never)
    int use_pitch = 1;               // scalar base + pitch addressing for slab layouts
    int poison_byte = -1;            // ... MOD16_POISON_BYTE=b: the byte every byte of that ticket is set to (default: the ticket becomes 2^40)
    int poison_ticket = 0;           // experiments build, MOD16_POISON_TICKET=k: the k-th dynamically scheduled launch finds
                                     // its ticket counter in use (what an abandoned launch leaves behind): the test of kStatusIncomplete
    unsigned long long* dyn_counters = nullptr;   // ring of ticket counters, 128 B apart
    int dyn_next = 0;
    bool have_lut = false;
    double* lut64 = nullptr;     // device [MOD16_LUT_ROWS][kLutCols]
    float* lut32 = nullptr;
    double* tab64 = nullptr;         // exp/log tables of FastMath<double>
    unsigned* status = nullptr;      // device status word
    unsigned* status_host = nullptr; // pinned mirror
    unsigned* static_flag = nullptr; // device word of mod16_et_static_*
    DiagWs ws;                       // diagnostics partials of launches outside a graph
    std::vector<void*> retired;      // outgrown workspaces (freed with the context)
    DiagWs* force_ws = nullptr;      // workspace to use instead (graph capture)
    hipEvent_t ws_event = nullptr;   // recorded behind the last launch that produced diagnostics in `ws`
    hipStream_t ws_stream = nullptr; // ... and the stream it ran on
    bool ws_pending = false;
    bool ws_recorded = false;        // ... and whether ws_event was recorded behind it
    bool ws_multi = false;           // the context has launched on more than one stream (or runs HOST tiles on
                                     // its slots): every launch records ws_event from now on
    double* diag_dev = nullptr;      // device [kDiag]
    double* diag_host = nullptr;     // pinned [kDiag]
    double* hdiag_dev = nullptr;     // device [kSlots][kDiag]: per-tile diagnostics of the HOST mode (mod16_et_hdiag_*)
    // HOST-mode staging: per slot one device slab + one stream
    void* slab[kSlots] = {};
    size_t slab_bytes = 0;
    hipStream_t streams[kSlots] = {};
    std::mutex launch_mu;            // HOST mode: kernel launches of the staging threads
    void* scalars = nullptr;         // device copies of broadcast scalars
    // HOST mode, small calls (a flux-tower site, a year of one pixel): one page-locked buffer the
    // kernel reads its inputs from and writes its outputs to over the link -- no copy commands at all
    int small_pixels = kSmallPixels; // MOD16_SMALL_PIXELS: calls of at most this many pixels go that way (0: none)
    void* small_host = nullptr;      // hipHostMalloc'ed
    void* small_dev = nullptr;       // ... as the device addresses it
    size_t small_bytes = 0;
    unsigned long long* force_counter = nullptr;   // ticket counter to use instead of the ring (graph capture)
    void* bc_buf = nullptr;          // HOST mode: device copies of (N,) / (T, 1) inputs (mod16_et2_*)
    size_t bc_bytes = 0;
    void* batch_buf = nullptr;       // HOST-mode workspace of mod16_et_static_batch_*
    size_t batch_bytes = 0;
    std::string err;
};

#define HIPCHK(ctx, call)                                                          \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                    \
            char b_[512];                                                          \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call,                \
                     hipGetErrorString(e_), __FILE__, __LINE__);                   \
            (ctx)->err = b_;                                                       \
            return MOD16_ERR_HIP;                                                  \
        }                                                                          \
    } while (0)

// every entry point that takes a ctx holds its mutex for the duration of the call
#define MOD16_LOCK(ctx) std::unique_lock<std::recursive_mutex> api_lock_; \
    if (ctx) api_lock_ = std::unique_lock<std::recursive_mutex>((ctx)->api_mu)

static int fail(mod16_ctx* ctx, int code, const char* msg) {
    if (ctx) ctx->err = msg;
    return code;
}

extern "C" int mod16_version(void) { return MOD16_ABI_VERSION; }

#ifndef MOD16_BUILD_ID
#define MOD16_BUILD_ID "unknown"
#endif
// (behind a marker that build.py finds in the file: a library whose id is not the digest of the
// sources next to it is rebuilt, whatever the files' dates say)
static const char kBuildIdMarker[] = "mod16-build-id=" MOD16_BUILD_ID;
extern "C" const char* mod16_build_id(void) { return kBuildIdMarker + sizeof("mod16-build-id=") - 1; }

extern "C" const char* mod16_strerror(int status) {
    switch (status) {
        case MOD16_OK: return "ok";
        case MOD16_ERR_ARG: return "invalid argument";
        case MOD16_ERR_HIP: return "HIP runtime error";
        case MOD16_ERR_CLASS_RANGE: return "class code out of range (>= 13)";
        case MOD16_ERR_NOMEM: return "out of memory";
        case MOD16_ERR_NO_DEVICE: return "no usable gfx950 device";
        case MOD16_ERR_NO_BPLUT: return "class raster given but no BPLUT set";
        default: return "unknown status";
    }
}

extern "C" const char* mod16_last_error(const mod16_ctx* ctx) {
    return ctx ? ctx->err.c_str() : "";
}

extern "C" int mod16_device_count(int* count) {
    if (!count) return MOD16_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return MOD16_OK;
}

extern "C" int mod16_destroy(mod16_ctx* ctx) {
    if (!ctx) return MOD16_OK;
    (void)hipSetDevice(ctx->device);
    for (int s = 0; s < kSlots; ++s) {
        if (ctx->slab[s]) (void)hipFree(ctx->slab[s]);
        if (ctx->streams[s]) (void)hipStreamDestroy(ctx->streams[s]);
    }
    if (ctx->scalars) (void)hipFree(ctx->scalars);
    if (ctx->small_host) (void)hipHostFree(ctx->small_host);
    if (ctx->batch_buf) (void)hipFree(ctx->batch_buf);
    if (ctx->bc_buf) (void)hipFree(ctx->bc_buf);
    if (ctx->lut64) (void)hipFree(ctx->lut64);
    if (ctx->lut32) (void)hipFree(ctx->lut32);
    if (ctx->tab64) (void)hipFree(ctx->tab64);
    if (ctx->dyn_counters) (void)hipFree(ctx->dyn_counters);
    if (ctx->status) (void)hipFree(ctx->status);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->static_flag) (void)hipFree(ctx->static_flag);
    if (ctx->ws.partial) (void)hipFree(ctx->ws.partial);
    for (void* p : ctx->retired) (void)hipFree(p);
    if (ctx->ws_event) (void)hipEventDestroy(ctx->ws_event);
    if (ctx->diag_dev) (void)hipFree(ctx->diag_dev);
    if (ctx->diag_host) (void)hipHostFree(ctx->diag_host);
    if (ctx->hdiag_dev) (void)hipFree(ctx->hdiag_dev);
    delete ctx;
    return MOD16_OK;
}

extern "C" int mod16_create(int device, mod16_ctx** out) {
    if (!out) return MOD16_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n)
        return MOD16_ERR_NO_DEVICE;
    mod16_ctx* ctx = new (std::nothrow) mod16_ctx;
    if (!ctx) return MOD16_ERR_NOMEM;
    ctx->device = device;
    int rc = [&]() -> int {
        HIPCHK(ctx, hipSetDevice(device));
        hipDeviceProp_t prop;
        HIPCHK(ctx, hipGetDeviceProperties(&prop, device));
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            ctx->err = std::string("device is ") + prop.gcnArchName + ", this library is gfx950 only";
            return MOD16_ERR_NO_DEVICE;
        }
        ctx->cus = prop.multiProcessorCount;
        // the one documented tuning knob of the shipped library: staging threads of the HOST mode
        if (const char* g = getenv("MOD16_HOST_THREADS")) ctx->host_threads = std::max(1, std::min(kSlots, atoi(g)));
        // ... and where the HOST mode's copy-free path for small calls ends (0: every call is staged)
        if (const char* g = getenv("MOD16_SMALL_PIXELS")) ctx->small_pixels = std::max(0, std::min(kSmallPixelsMax, atoi(g))) / 4 * 4;
#ifdef MOD16_EXPERIMENTS
        // launch-geometry overrides of the experiments build (libmod16hip_exp.so: tools/, and the
        // tests that put the flag record through the other schedules); never in the shipped library
        if (const char* g = getenv("MOD16_GRID_MULT")) ctx->grid_mult = std::max(1, atoi(g));
        if (const char* g = getenv("MOD16_NO_DMA")) ctx->use_dma = atoi(g) == 0;
        if (const char* g = getenv("MOD16_PITCH")) ctx->use_pitch = atoi(g);
        if (const char* g = getenv("MOD16_RUN_SHIFT")) ctx->run_shift = std::max(1, std::min(6, atoi(g)));
        if (const char* g = getenv("MOD16_STATIC_BELOW")) ctx->static_below = std::max(0, std::min(64, atoi(g)));
        if (const char* g = getenv("MOD16_STREAM_BLOCKS")) ctx->stream_blocks = std::max(1, std::min(2, atoi(g)));
        if (const char* g = getenv("MOD16_POISON_TICKET")) ctx->poison_ticket = std::max(0, atoi(g));
        if (const char* g = getenv("MOD16_POISON_BYTE")) ctx->poison_byte = std::max(-1, std::min(255, atoi(g)));
#endif
        HIPCHK(ctx, hipMalloc(&ctx->dyn_counters, 64 * 128));
        {   // ticket = 0, blocks done = 0, and a serial number (word [3]) that starts somewhere else in
            // every slot of the ring: successive launches take successive slots and share the
            // diagnostics workspace, so their markers (kSerialField) must differ -- launch j carries
            // (j % 64) * 1021 + j / 64
            unsigned init[64 * 32] = {};
            for (unsigned i = 0; i < 64; ++i) init[i * 32 + 3] = i * 1021u;
            HIPCHK(ctx, hipMemcpy(ctx->dyn_counters, init, sizeof init, hipMemcpyHostToDevice));
        }
        const size_t nlut = MOD16_LUT_ROWS * kLutCols;
        HIPCHK(ctx, hipMalloc(&ctx->lut64, nlut * sizeof(double)));
        HIPCHK(ctx, hipMalloc(&ctx->lut32, nlut * sizeof(float)));
        {   // exp/log tables of FastMath<double> (mod16_math.hpp)
            constexpr int n = FastMath<double>::kTabDoubles;
            double t[n];
            for (int j = 0; j < 64; ++j) t[j] = (double)exp2l((long double)j / 64.0L);
            for (int j = 0; j < 128; ++j) {
                const double inv = 1.0 / (1.0 + (j + 0.5) / 128.0);
                t[64 + 2 * j] = inv;
                t[64 + 2 * j + 1] = (double)(-logl((long double)inv));
            }
            // entry 0 serves x = 1 (m = 1): make log_tab(1) cancel to exactly 0
            t[64 + 1] = -FastMath<double>::log1p_poly(std::fma(1.0, t[64], -1.0));
            // air pressure [Pa] from elevation, MOD16.air_pressure (mod16/__init__.py:414-447):
            // 101325 (1 - 0.0065 z / 288.15)^5.2559 interpolated at the 10 Chebyshev nodes of
            // -2000 m .. 12000 m, in powers of u = (z - 5000) / 7000: 3.7e-14 relative on that
            // interval (numpy fit, tools/fit_air_pressure.py); outside it the domain guard hands
            // the pixel to the reference-order arithmetic
            static const double kPressurePoly[10] = {
                0x1.a607a9266ab84p+15, -0x1.8ac7a6364460ap+15, 0x1.2b06fedbccfd8p+14, -0x1.ce13d340b53c7p+11,
                0x1.730bd6d1a9cc9p+8, -0x1.096571085f8f7p+4, 0x1.01d9b2280ab84p-3, 0x1.3835059a0bfaap-9,
                0x1.8732949feb6b7p-14, 0x1.555f18e36b65ap-18};
            for (int j = 0; j < 16; ++j) t[FastMath<double>::kTabRaw + j] = j < 10 ? kPressurePoly[j] : 0.0;
            HIPCHK(ctx, hipMalloc(&ctx->tab64, sizeof t));
            HIPCHK(ctx, hipMemcpy(ctx->tab64, t, sizeof t, hipMemcpyHostToDevice));
        }
        HIPCHK(ctx, hipMalloc(&ctx->status, sizeof(unsigned)));
        HIPCHK(ctx, hipMemset(ctx->status, 0, sizeof(unsigned)));
        HIPCHK(ctx, hipHostMalloc(&ctx->status_host, sizeof(unsigned)));
        HIPCHK(ctx, ws_alloc(ctx->ws, kDiagBlocks));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ws_event, hipEventDisableTiming));
        HIPCHK(ctx, hipMalloc(&ctx->diag_dev, sizeof(double) * kDiag));
        HIPCHK(ctx, hipHostMalloc(&ctx->diag_host, sizeof(double) * kDiag));
        HIPCHK(ctx, hipMalloc(&ctx->hdiag_dev, sizeof(double) * kDiag * kSlots));
        HIPCHK(ctx, hipMalloc(&ctx->scalars, 32 * sizeof(double)));
        return MOD16_OK;
    }();
    if (rc != MOD16_OK) {
        // keep the message for the caller? the ctx is gone: print it once
        fprintf(stderr, "mod16_create: %s\n", ctx->err.c_str());
        mod16_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return MOD16_OK;
}

extern "C" int mod16_set_bplut_f64(mod16_ctx* ctx, const double* lut) {
    MOD16_LOCK(ctx);
    if (!ctx || !lut) return fail(ctx, MOD16_ERR_ARG, "mod16_set_bplut_f64: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double nan = std::numeric_limits<double>::quiet_NaN();
    double h64[MOD16_LUT_ROWS * kLutCols];
    float h32[MOD16_LUT_ROWS * kLutCols];
    for (int c = 0; c < kLutCols; ++c) {
        double row[MOD16_LUT_ROWS];
        for (int k = 0; k < MOD16_LUT_ROWS; ++k) row[k] = nan;
        if (c < MOD16_N_CLASSES) {
            const double* p = lut + (size_t)c * MOD16_N_PARAMS;
            for (int k = 0; k < MOD16_N_PARAMS; ++k) row[k] = p[k];
            row[11] = 1.0 / (p[MOD16_TMIN_OPEN] - p[MOD16_TMIN_CLOSE]);
            row[12] = 1.0 / (p[MOD16_VPD_CLOSE] - p[MOD16_VPD_OPEN]);
            row[13] = (p[MOD16_RBL_MAX] - p[MOD16_RBL_MIN]) / (p[MOD16_VPD_CLOSE] - p[MOD16_VPD_OPEN]);
            row[14] = 1.0 / p[MOD16_BETA];
            // smallest float32 >= 273.15 + tmin_close: for a float32 x,
            // x >= 273.15 + tmin_close (in float64) <=> x >= this (mixed-precision form)
            const double thr = 273.15 + p[MOD16_TMIN_CLOSE];
            float tf = (float)thr;
            if ((double)tf < thr) tf = std::nextafterf(tf, std::numeric_limits<float>::infinity());
            row[15] = (double)tf;
        }
        for (int k = 0; k < MOD16_LUT_ROWS; ++k) {
            h64[k * kLutCols + c] = row[k];
            h32[k * kLutCols + c] = (float)row[k];
        }
    }
    HIPCHK(ctx, hipMemcpy(ctx->lut64, h64, sizeof h64, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->lut32, h32, sizeof h32, hipMemcpyHostToDevice));
    ctx->have_lut = true;
    return MOD16_OK;
}

// ------------------------------------------------------------------ launch
template <typename T> static const T* ctx_lut(const mod16_ctx* ctx);
template <> const double* ctx_lut<double>(const mod16_ctx* ctx) { return ctx->lut64; }
template <> const float* ctx_lut<float>(const mod16_ctx* ctx) { return ctx->lut32; }


template <typename T> struct VecOf;
template <> struct VecOf<double> { static constexpr int v = 2; };
template <> struct VecOf<float> { static constexpr int v = 4; };

// Instantiated variants: the production (FAST, vectorised) kernel gets the
// SEP / DENSE specialisations; the scalar-tail and EXACT kernels are generic.
template <typename T, int V>
static void launch_variant(const EtArgs<T>& a, bool lut, bool fast, bool sep, bool dense,
                           int grid, hipStream_t st) {
    // FAST on float32 with 4 pixels per thread is never built (see launch_et): EXACT only
#ifdef MOD16_REPRO_V4   // reproduction builds of DESIGN.md 5.2 only (tools/repro_v4.py)
    constexpr bool kFastOk = true;
#else
    constexpr bool kFastOk = !(std::is_same<T, float>::value && V == 4);
#endif
    if (a.out[8] || a.out[9]) {   // potential ET wanted: the generic all-outputs form
#define MOD16_LAUNCH_PET(LUT, FAST) \
    hipLaunchKernelGGL((et_kernel<T, V, LUT, FAST, true, false, true>), dim3(grid), dim3(kBlock), 0, st, a)
        if constexpr (kFastOk) {
            if (fast) {
                if (lut) {
                    MOD16_LAUNCH_PET(true, true);
                } else {
                    // per-pixel parameter arrays + potential ET: one pixel per thread (with two,
                    // the 25 inputs, 10 outputs and the guard's slow branch do not fit the
                    // register budget of two waves per SIMD without spilling)
                    const int g1 = (int)std::min<int64_t>((a.n + kBlock - 1) / kBlock, (int64_t)grid * V);
                    hipLaunchKernelGGL((et_kernel<T, 1, false, true, true, false, true>), dim3(std::max(1, g1)),
                                       dim3(kBlock), 0, st, a);
                }
                return;
            }
        }
        if (lut) MOD16_LAUNCH_PET(true, false); else MOD16_LAUNCH_PET(false, false);
#undef MOD16_LAUNCH_PET
        return;
    }
#define MOD16_LAUNCH(LUT, FAST, SEP, DENSE) \
    hipLaunchKernelGGL((et_kernel<T, V, LUT, FAST, SEP, DENSE>), dim3(grid), dim3(kBlock), 0, st, a)
    if constexpr (kFastOk && V > 1) {
        if (fast) {
            if (lut) {
                if (sep) { if (dense) MOD16_LAUNCH(true, true, true, true); else MOD16_LAUNCH(true, true, true, false); }
                else     { if (dense) MOD16_LAUNCH(true, true, false, true); else MOD16_LAUNCH(true, true, false, false); }
            } else if (sep) {
                // per-pixel parameter arrays + the six components: one pixel per thread (with two,
                // 25 inputs, 8 outputs and the guard's slow branch spill two registers)
                const int g1 = (int)std::min<int64_t>((a.n + kBlock - 1) / kBlock, (int64_t)grid * V);
                hipLaunchKernelGGL((et_kernel<T, 1, false, true, true, false>), dim3(std::max(1, g1)),
                                   dim3(kBlock), 0, st, a);
            } else {
                if (dense) MOD16_LAUNCH(false, true, false, true); else MOD16_LAUNCH(false, true, false, false);
            }
            return;
        }
    } else if constexpr (kFastOk) {
        if (fast) {
            if (lut) MOD16_LAUNCH(true, true, true, false); else MOD16_LAUNCH(false, true, true, false);
            return;
        }
    }
    if (lut) MOD16_LAUNCH(true, false, true, false); else MOD16_LAUNCH(false, false, true, false);
#undef MOD16_LAUNCH
}

static int grid_for(const mod16_ctx* ctx, int64_t nvec) {
    int64_t need = (nvec + kBlock - 1) / kBlock;
    int64_t cap = (int64_t)ctx->cus * ctx->grid_mult;
    return (int)std::max<int64_t>(1, std::min(need, cap));
}

// All pointers are device pointers here.
// -> the workspace for `blocks` partials: the forced one (graph capture; it was
// sized by its owner) or the context's, grown if need be. Captured graphs never
// point into the context's workspace, so it can be replaced once the device is idle.
static int reserve_diag(mod16_ctx* ctx, int64_t blocks, DiagWs** out = nullptr) {
    if (ctx->force_ws) {
        if (blocks > ctx->force_ws->capacity)
            return fail(ctx, MOD16_ERR_ARG, "internal: graph workspace smaller than its launch");
        if (out) *out = ctx->force_ws;
        return MOD16_OK;
    }
    if (out) *out = &ctx->ws;
    if (blocks <= ctx->ws.capacity) return MOD16_OK;
    // Growing: earlier launches may still use the old block. No device-wide wait (other
    // contexts of the process -- the workers of mod16_amd.io -- would stall with this one) and
    // no hipFree (which synchronises the device): the old block is retired and freed with the
    // context; sizes at least double, so the retired blocks add up to less than the live one.
    // Launches that follow use the new block and are ordered behind the old one's by the
    // workspace event as before.
    ctx->retired.push_back(ctx->ws.partial);
    ctx->ws.partial = nullptr;
    const int64_t want = std::max<int64_t>(blocks, 2 * ctx->ws.capacity);
    ctx->ws.capacity = 0;
    HIPCHK(ctx, ws_alloc(ctx->ws, want));
    return MOD16_OK;
}

// The context's workspace is shared by its launches (every pipeline launch
// writes per-run partials, wanted or not). Launches on ONE stream are ordered
// anyway; a launch on another stream than the previous one waits for it, so the
// previous launch's final sum has read its partials before they are overwritten.
// Inside a graph capture the graph's own workspace is used instead.
static int ws_acquire(mod16_ctx* ctx, hipStream_t st) {
    if (ctx->force_ws) return MOD16_OK;
    // A caller capturing its own stream into a graph would bake the context's workspace (which
    // may be replaced later) and this event bookkeeping into it: refused -- mod16_graph_* builds
    // graphs that own their workspace.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (st && hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
        return fail(ctx, MOD16_ERR_ARG, "the stream is being captured into a graph: use mod16_graph_et_diag_* / "
                                        "mod16_graph_et_tiled_* (they capture the step with a workspace of its own)");
    // Ordering across streams costs a marker packet behind EVERY launch (the event) -- part of the
    // 5.5 us that lie between two 1200 x 1200 launches on one stream -- so a context pays for it
    // only once it has seen a second stream (round 5): until then nothing is recorded; the first
    // launch that arrives on another stream waits for the DEVICE (once per context: the earlier
    // stream may be gone by now -- its owner may destroy it, and an event cannot be recorded on a
    // destroyed stream), and from then on every launch leaves its event behind (ws_release).
    if (ctx->ws_pending && st != ctx->ws_stream) {
        if (ctx->ws_recorded) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ws_event, 0));
        else HIPCHK(ctx, hipDeviceSynchronize());
        ctx->ws_multi = true;
        ctx->ws_pending = false;
    }
    return MOD16_OK;
}
static int ws_release(mod16_ctx* ctx, hipStream_t st) {
    if (ctx->force_ws) return MOD16_OK;
    ctx->ws_recorded = false;
    if (ctx->ws_multi) {
        HIPCHK(ctx, hipEventRecord(ctx->ws_event, st));
        ctx->ws_recorded = true;
    }
    ctx->ws_stream = st;
    ctx->ws_pending = true;
    return MOD16_OK;
}

template <typename T>
static int reduce_entry(mod16_ctx* ctx, const T* day, const T* night, int64_t n, double* diag,
                        double* ddiag, void* stream);

// Launch geometry of the production pipeline for n pixels, V per 16-byte vector.
struct StreamGeom { int run_shift; int64_t npiece, nruns; int grid; int static_sched; };
static StreamGeom stream_geom(const mod16_ctx* ctx, int64_t n, int V, int tile_shift = kNoTile) {
    StreamGeom g;
    g.npiece = (n / V + 63) / 64;
    // run length: kDynRun pieces, halved for small rasters until every wave the
    // chip holds (2 blocks of 4 per CU) gets at least one run
    const int64_t chip_waves = (int64_t)ctx->cus * 2 * (kBlock / 64);
    int run_shift = 0;
    while ((1 << run_shift) < kDynRun) ++run_shift;
    // tiled rasters: runs of 2 kDynRun pieces (16 KiB per field, half a default tile):
    // -0.8 % on the global grid in two same-box A/Bs, where on plain arrays runs of 16
    // measured +0.4-1.4 % (round 1); fewer claims and partials, the tail stays < 0.5 %
    if (tile_shift != kNoTile) ++run_shift;
    while (run_shift > 1 && (g.npiece >> run_shift) < chip_waves) --run_shift;   // >= 2 pieces: the claim of a run is consumed in its second iteration
    // A small raster (fewer than kStaticBelow runs per wave) is latency-bound and ends
    // with its slowest SIMD: single pieces dealt out round-robin -- all waves start together,
    // no claim's round trip sits on a path that is only a few iterations long, and the load is
    // balanced to within one PIECE per wave. (Rounds 2-3 dealt runs of 2 pieces: a 1200 x 1200
    // raster, 11250 pieces over 2048 waves, then gave the waves 6 or 4 pieces, and since a CU holds
    // blocks c and c + 256, the first 127 CUs got 12 pieces per SIMD against an average of 11;
    // piece by piece it is 6 or 5 per wave and at most 11 per SIMD.)
    g.static_sched = 0;
    if (ctx->static_below > 0 && (g.npiece >> run_shift) < (int64_t)ctx->static_below * chip_waves) {
        g.static_sched = 1;
        run_shift = 0;
    }
    if (ctx->run_shift > 0) run_shift = ctx->run_shift;
    run_shift = std::min(run_shift, tile_shift);     // a run never straddles two tiles
    g.run_shift = run_shift;
    g.nruns = (g.npiece + (int64_t(1) << run_shift) - 1) >> run_shift;
    // persistent waves: 2 blocks per CU is what the LDS slots allow
    g.grid = (int)std::max<int64_t>(1, std::min<int64_t>(
        (g.nruns + (kBlock / 64) - 1) / (kBlock / 64), (int64_t)ctx->cus * ctx->stream_blocks));
    return g;
}
constexpr int kStage = 1024;     // slices of the two-level sum of the per-run partials
// workspace of a pipeline launch, in partials (64 bytes): one per run, the stage of the two-level sum,
// and one more per run -- the run's cancellation list (mixed-precision forms, mod16_stream.hpp)
static_assert(kCancelCap * sizeof(uint16_t) == kDiag * sizeof(double), "a cancellation list is the size of a partial");
static int64_t stream_ws_blocks(int64_t nruns) { return 2 * nruns + kStage; }
constexpr int64_t kFuseFinalBelow = 16384;   // partials up to which the pipeline kernel sums them itself

// The production pipeline for dense class rasters (mod16_stream.hpp). s.n must
// be a multiple of the vector width. ddiag != NULL: also the fixed-order sum
// of the per-run diagnostics partials -> ddiag (8 doubles on the device), over
// n_valid_total pixels.
// GUARD = false: MOD16_DOMAIN_TRUSTED (the instance without the domain test; totals forms).
template <typename T, int MODE, bool GUARD = true>
static int launch_stream(mod16_ctx* ctx, StreamArgs<T> s, hipStream_t st, double* ddiag = nullptr) {
    constexpr int V = VecOf<T>::v;
    s.lut64 = ctx->lut64;
    s.tab = ctx->tab64;
    s.status = ctx->status;
    if (s.tile_shift <= 0) {       // plain arrays: one "tile"
        s.tile_shift = kNoTile;
        s.wide_row = s.out_row = s.byte_row = 0;
    }
    const StreamGeom g = stream_geom(ctx, s.n, V, s.tile_shift);
    unsigned long long* ctr = ctx->force_counter ? ctx->force_counter
                                                 : ctx->dyn_counters + 16 * (ctx->dyn_next++ % 64);
    // the ticket counter of the dynamic schedule (a statically scheduled raster never reads it):
    // zero when it was allocated, and every launch leaves it at zero again (the kernel's last
    // block resets it) -- no memset in front of the kernel, see et_stream_kernel
    s.dyn_counter = ctr;
#ifdef MOD16_EXPERIMENTS
    if (ctx->poison_ticket > 0 && !g.static_sched && --ctx->poison_ticket == 0) {
        // (MOD16_POISON_TICKET=k: the k-th dynamically scheduled launch of the context) the ticket
        // -- zero between launches -- becomes 2^40: "every run has been claimed", as a counter left
        // behind by an abandoned launch says: the waves process their first, statically assigned
        // runs and find nothing to claim
        // MOD16_POISON_BYTE=b (round 6): all eight bytes of the ticket become b instead -- 0x3f is the
        // pattern whose (nwaves + ticket) << run_shift overflowed into a negative base in round 5 (a
        // wild read and a wild store); 0xff is -1. The kernel clamps the ticket before it forms a base.
        if (ctx->poison_byte >= 0) HIPCHK(ctx, hipMemsetAsync(ctr, ctx->poison_byte, 8, st));
        else HIPCHK(ctx, hipMemsetAsync(reinterpret_cast<char*>(ctr) + 5, 1, 1, st));
    }
#endif
    s.run_shift = g.run_shift;
    s.static_sched = g.static_sched;
    const int grid = g.grid;
    // partials: one per run, or (static schedule) one per block
    const int64_t nruns = g.static_sched ? grid : g.nruns;
    DiagWs* ws = nullptr;
    int rc = reserve_diag(ctx, stream_ws_blocks(nruns), &ws);
    if (rc != MOD16_OK) return rc;
    rc = ws_acquire(ctx, st);
    if (rc != MOD16_OK) return rc;
    s.diag_partial = ws->partial;
    s.cancel_list = reinterpret_cast<uint16_t*>(ws->partial + (nruns + kStage) * kDiag);
    // few partials: the kernel's last block adds them up itself (two dispatches less)
    // (only under the static schedule: a dynamically scheduled raster's flagged pieces are
    // revisited by the kernel BEHIND this one, which corrects the partials before they are summed)
    const bool fused_final = ddiag && g.static_sched && nruns <= kFuseFinalBelow;
    s.diag_out = fused_final ? ddiag : nullptr;
    s.done_counter = ws->done();
    s.nruns = nruns;
    // equally spaced wide arrays (one slab): scalar base + k * pitch
    constexpr int NW = StreamSpec<MODE>::NW;
    const ptrdiff_t pitch_b = reinterpret_cast<const char*>(s.wide[1]) - reinterpret_cast<const char*>(s.wide[0]);
    bool pitched = ctx->use_pitch && pitch_b % (ptrdiff_t)sizeof(T) == 0;
    for (int k = 2; k < NW && pitched; ++k)
        pitched = reinterpret_cast<const char*>(s.wide[k]) - reinterpret_cast<const char*>(s.wide[0]) == k * pitch_b;
    s.wide_pitch = pitched ? pitch_b / (ptrdiff_t)sizeof(T) : 0;
    if (pitched) hipLaunchKernelGGL((et_stream_kernel<T, MODE, true, GUARD>), dim3(grid), dim3(kBlock), 0, st, s);
    else hipLaunchKernelGGL((et_stream_kernel<T, MODE, false, GUARD>), dim3(grid), dim3(kBlock), 0, st, s);
    // pixels outside the domain of the production arithmetic (mod16_physics.hpp, "domain
    // guard"): a statically scheduled (small) raster has revisited them inside the kernel; a
    // large one left one flag per piece in its runs' partials for this kernel
#ifndef MOD16_NO_REDO_LAUNCH
    if constexpr (GUARD) if (!g.static_sched) {
        const int64_t groups = (nruns + 63) / 64;
        // mixed-precision forms: first the runs' cancellation lists (mod16_mixed.hpp, period_mixed)
        if constexpr (stream_is_mixed(MODE)) {
            const int cgrid = (int)std::max<int64_t>(1, std::min<int64_t>((groups + kBlock / 64 - 1) / (kBlock / 64),
                                                                          (int64_t)ctx->cus * 8));
            hipLaunchKernelGGL((et_stream_cancel_kernel<T, MODE>), dim3(cgrid), dim3(kBlock), 0, st, s);
        }
        const int rgrid = (int)std::max<int64_t>(1, std::min<int64_t>((groups + kBlock / 64 - 1) / (kBlock / 64),
                                                                      (int64_t)ctx->cus * 4));
        hipLaunchKernelGGL((et_stream_redo_kernel<T, MODE>), dim3(rgrid), dim3(kBlock), 0, st, s);
    }
#endif
    if (ddiag && !fused_final) {
        const double* fin = ws->partial;
        int64_t count = nruns;
        // (a trusted launch has no kernel behind it that looks at every run: the kernel that reads
        // the runs' own partials compares every run's marker)
        bool check = !GUARD && !g.static_sched;
        const unsigned* serial_word = reinterpret_cast<const unsigned*>(ctr) + 3;
        if (count > 4 * kStage) {   // two-level: 1024 fixed slices, then one block
            double* stage = ws->partial + nruns * kDiag;
            const int64_t per = (count + kStage - 1) / kStage;
            hipLaunchKernelGGL(diag_stage_kernel, dim3(kStage), dim3(kBlock), 0, st, fin, count, per, stage,
                               check ? serial_word : (const unsigned*)nullptr,
                               check ? ctx->status : (unsigned*)nullptr);
            fin = stage;
            count = (count + per - 1) / per;
            check = false;
        }
        hipLaunchKernelGGL(diag_final_fused_kernel, dim3(1), dim3(kFinalBlock), 0, st,
                           fin, (int)count, s.n, ddiag,
                           check ? serial_word : (const unsigned*)nullptr,
                           check ? ctx->status : (unsigned*)nullptr);
    }
    return ws_release(ctx, st);
}

// The totals form (the production step): FAST or, float32, MIXED arithmetic; with
// MOD16_DOMAIN_TRUSTED the instance without the domain test.
template <typename T>
static int launch_totals(mod16_ctx* ctx, const StreamArgs<T>& s, hipStream_t st, double* ddiag, unsigned flags) {
    const bool trusted = (flags & MOD16_DOMAIN_TRUSTED) != 0;
    if constexpr (std::is_same<T, float>::value) {
        if (flags & MOD16_MATH_MIXED)
            return trusted ? launch_stream<T, kStreamTotalsMixed, false>(ctx, s, st, ddiag)
                           : launch_stream<T, kStreamTotalsMixed>(ctx, s, st, ddiag);
    }
    return trusted ? launch_stream<T, kStreamTotals, false>(ctx, s, st, ddiag)
                   : launch_stream<T, kStreamTotals>(ctx, s, st, ddiag);
}

template <typename T> static bool has_rows_or_cols(const EtArgs<T>& a) {
    return (a.row_drv | a.col_drv | a.row_par | a.col_par) != 0u || (a.cls && a.cls_mode != MOD16_BC_DENSE);
}

// ddiag != NULL: also produce the diagnostics vector (device, 8 doubles).
template <typename T>
static int launch_et(mod16_ctx* ctx, EtArgs<T> a, unsigned flags, hipStream_t st,
                     double* ddiag = nullptr) {
    constexpr int V = VecOf<T>::v;
    const bool lut = a.cls != nullptr;
    const bool fast = (flags & MOD16_MATH_EXACT) == 0;
    if (a.n <= 0) return MOD16_OK;
    a.lut = ctx_lut<T>(ctx);
    a.lut64 = ctx->lut64;
    a.tab = ctx->tab64;
    a.status = ctx->status;
    // 16-byte vector path needs every dense pointer 16-byte aligned
    bool aligned = true;
    auto chk = [&](const void* p, size_t al) {
        if (p && (reinterpret_cast<uintptr_t>(p) % al)) aligned = false;
    };
    for (int k = 0; k < 14; ++k) if ((a.dense_drv >> k) & 1u) chk(a.drv[k], 16);
    if (lut) chk(a.cls, V);
    else for (int k = 0; k < 11; ++k) if ((a.dense_par >> k) & 1u) chk(a.par[k], 16);
    bool sep = false;
    for (int k = 0; k < 10; ++k) {
        chk(a.out[k], 16);
        if (k >= 2 && a.out[k]) sep = true;
    }
    const bool dense = a.dense_drv == 0x3fffu;
    // (N,) rows or (T, 1) columns among the inputs: the one-pixel-per-thread kernel
    // indexes them; the vector kernels see plain dense arrays and scalars only
    const int64_t nbody = (aligned && !has_rows_or_cols(a)) ? (a.n / V) * V : 0;
    bool fused_diag = false;
    // dense class rasters with one of the supported output sets take the
    // production pipeline (mod16_stream.hpp), everything else the plain kernel
    int smode = -1;
    if (ctx->use_dma && lut && fast && dense) {
        bool all6 = true, none6 = true;
        for (int k = 2; k < 8; ++k) { all6 = all6 && a.out[k]; none6 = none6 && !a.out[k]; }
        const bool tot = a.out[0] && a.out[1], notot = !a.out[0] && !a.out[1];
        const bool pet = a.out[8] && a.out[9], nopet = !a.out[8] && !a.out[9];
        if (tot && none6 && nopet) smode = kStreamTotals;
        else if (ddiag) smode = -1;      // the fused diagnostics belong to the totals form
        else if (tot && none6 && pet) smode = kStreamPet;
        else if (tot && all6 && nopet) smode = kStreamSep8;
        else if (notot && all6 && nopet) smode = kStreamSep6;
    }
    if (nbody && smode >= 0) {
        StreamArgs<T> s;
        memset(&s, 0, sizeof s);
        for (int k = 0; k < 14; ++k) s.wide[k] = a.drv[k];
        s.bytes[0] = a.cls;
        s.n = nbody;
        int rc;
        if (smode == kStreamTotals) {
            s.out[0] = a.out[0]; s.out[1] = a.out[1];
            fused_diag = ddiag && nbody == a.n;
            rc = launch_totals<T>(ctx, s, st, fused_diag ? ddiag : nullptr, flags);
        } else {
            // float32 rasters: MOD16_MATH_MIXED selects the mixed-precision pixel function
            bool mixed = false;
            if constexpr (std::is_same<T, float>::value) mixed = (flags & MOD16_MATH_MIXED) != 0;
            if (smode == kStreamPet) {
                s.out[0] = a.out[0]; s.out[1] = a.out[1]; s.out[2] = a.out[8]; s.out[3] = a.out[9];
            } else if (smode == kStreamSep8) {
                for (int k = 0; k < 8; ++k) s.out[k] = a.out[k];
            } else {
                for (int k = 0; k < 6; ++k) s.out[k] = a.out[k + 2];
            }
            rc = MOD16_OK;
            if constexpr (std::is_same<T, float>::value) {
                if (mixed) {
                    rc = smode == kStreamPet ? launch_stream<T, kStreamPetMixed>(ctx, s, st)
                         : smode == kStreamSep8 ? launch_stream<T, kStreamSep8Mixed>(ctx, s, st)
                                                : launch_stream<T, kStreamSep6Mixed>(ctx, s, st);
                }
            }
            if (!mixed)
                rc = smode == kStreamPet ? launch_stream<T, kStreamPet>(ctx, s, st)
                     : smode == kStreamSep8 ? launch_stream<T, kStreamSep8>(ctx, s, st)
                                            : launch_stream<T, kStreamSep6>(ctx, s, st);
        }
        if (rc != MOD16_OK) return rc;
    } else if (nbody) {
        EtArgs<T> b = a;
        b.n = nbody;
        // float32 rasters with the FAST (float64) arithmetic: 2 pixels per thread. The
        // 4-pixel instances need ~400 registers; built from the round-1 sources at -O2 / -O3
        // they computed wrong values (which instance goes wrong moves with the
        // scheduler's settings, -O1 is right, today's sources are right) -- they stay unbuilt.
#ifndef MOD16_REPRO_V4
        if (fast && std::is_same<T, float>::value)
            launch_variant<T, 2>(b, lut, fast, sep, dense, grid_for(ctx, nbody / 2), st);
        else
#endif
            launch_variant<T, V>(b, lut, fast, sep, dense, grid_for(ctx, nbody / V), st);
    }
    if (nbody < a.n) {   // ragged tail (or unaligned input): scalar variant
        EtArgs<T> t = a;
        const int64_t off = nbody;
        for (int k = 0; k < 14; ++k) if ((t.dense_drv >> k) & 1u) t.drv[k] += off;
        if (lut) { if (t.cls_mode == MOD16_BC_DENSE) t.cls += off; }
        else for (int k = 0; k < 11; ++k) if ((t.dense_par >> k) & 1u) t.par[k] += off;
        for (int k = 0; k < 10; ++k) if (t.out[k]) t.out[k] += off;
        t.n = a.n - off;
        t.base = a.base + off;
        launch_variant<T, 1>(t, lut, fast, sep, dense, grid_for(ctx, t.n), st);
    }
    HIPCHK(ctx, hipGetLastError());
    if (ddiag && !fused_diag) {
        if (!a.out[0] || !a.out[1]) return fail(ctx, MOD16_ERR_ARG, "diagnostics need both out_day and out_night");
        return reduce_entry<T>(ctx, a.out[0], a.out[1], a.n, nullptr, ddiag, st);
    }
    return MOD16_OK;
}

// dstride / pstride hold a broadcast kind per array: MOD16_BC_SCALAR (0), MOD16_BC_DENSE (1)
// and, with inner > 0 (mod16_et2_*), MOD16_BC_ROW (2) / MOD16_BC_COL (3).
template <typename T>
static int fill_args(mod16_ctx* ctx, EtArgs<T>& a, const uint8_t* cls, const T* const* drivers,
                     const int64_t* dstride, const T* const* params, const int64_t* pstride,
                     int64_t n, T* out_day, T* out_night, T* const* out_sep,
                     T* pet_day = nullptr, T* pet_night = nullptr, int64_t inner = 0,
                     int cls_mode = MOD16_BC_DENSE) {
    if (!ctx) return MOD16_ERR_ARG;
    if (!drivers || !dstride || n < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et: NULL drivers/strides or n < 0");
    memset(&a, 0, sizeof a);
    const int64_t max_kind = inner > 0 ? MOD16_BC_COL : MOD16_BC_DENSE;
    if (inner > 0 && n % inner != 0) return fail(ctx, MOD16_ERR_ARG, "mod16_et2: n must be a multiple of inner");
    a.inner = inner > 0 ? inner : 1;
    a.base = 0;
    for (int k = 0; k < 14; ++k) {
        if (!drivers[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et: NULL driver array");
        if (dstride[k] < 0 || dstride[k] > max_kind) return fail(ctx, MOD16_ERR_ARG, "mod16_et: driver stride must be 0 or 1 (mod16_et2: a MOD16_BC_* kind)");
        a.drv[k] = drivers[k];
        if (dstride[k] == MOD16_BC_DENSE) a.dense_drv |= 1u << k;
        if (dstride[k] == MOD16_BC_ROW) a.row_drv |= 1u << k;
        if (dstride[k] == MOD16_BC_COL) a.col_drv |= 1u << k;
    }
    a.cls = cls;
    a.cls_mode = (uint32_t)cls_mode;
    if (cls) {
        if (!ctx->have_lut) return fail(ctx, MOD16_ERR_NO_BPLUT, "mod16_et: class raster given but mod16_set_bplut_f64 was not called");
        if (cls_mode < 0 || cls_mode > max_kind) return fail(ctx, MOD16_ERR_ARG, "mod16_et2: bad broadcast kind of the class raster");
    } else {
        if (!params || !pstride) return fail(ctx, MOD16_ERR_ARG, "mod16_et: neither a class raster nor parameter arrays given");
        for (int k = 0; k < 11; ++k) {
            if (!params[k]) return fail(ctx, MOD16_ERR_ARG, "mod16_et: NULL parameter array");
            if (pstride[k] < 0 || pstride[k] > max_kind) return fail(ctx, MOD16_ERR_ARG, "mod16_et: parameter stride must be 0 or 1 (mod16_et2: a MOD16_BC_* kind)");
            a.par[k] = params[k];
            if (pstride[k] == MOD16_BC_DENSE) a.dense_par |= 1u << k;
            if (pstride[k] == MOD16_BC_ROW) a.row_par |= 1u << k;
            if (pstride[k] == MOD16_BC_COL) a.col_par |= 1u << k;
        }
    }
    a.out[0] = out_day;
    a.out[1] = out_night;
    bool any = out_day || out_night;
    if (out_sep)
        for (int k = 0; k < 6; ++k) {
            a.out[2 + k] = out_sep[k];
            any = any || out_sep[k];
        }
    a.out[8] = pet_day;
    a.out[9] = pet_night;
    any = any || pet_day || pet_night;
    if (!any) return fail(ctx, MOD16_ERR_ARG, "mod16_et: no output array given");
    a.n = n;
    return MOD16_OK;
}

static int read_status(mod16_ctx* ctx, hipStream_t st) {
    HIPCHK(ctx, hipMemcpyAsync(ctx->status_host, ctx->status, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipMemsetAsync(ctx->status, 0, sizeof(unsigned), st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    if (*ctx->status_host & kStatusIncomplete)
        return fail(ctx, MOD16_ERR_HIP, "a launch processed only part of its raster: it found its ticket counter in use "
                                        "(an earlier launch on this context ended abnormally, or more launches were in "
                                        "flight than the context has counters) -- the outputs of that step are not valid");
    if (*ctx->status_host & kStatusClassRange)
        return fail(ctx, MOD16_ERR_CLASS_RANGE, "class raster holds a code >= 13 (numpy would raise IndexError)");
    return MOD16_OK;
}

// HOST mode: tiles of kTilePixels staged through kSlots device slabs, one host
// thread and one stream per slot. The copies from and to pageable numpy memory
// are what bounds this mode (the HIP runtime stages them through its own pinned
// buffers on the calling thread), so the slots run them concurrently; kernel
// launches are serialised (they share the context's workspace).
// device copies of the inputs that are neither dense nor scalars: (N,) rows and
// (T, 1) columns, uploaded whole once per call
template <typename T> struct BcTable {
    const T* drv[14] = {};
    const T* par[11] = {};
    const uint8_t* cls = nullptr;
};

template <typename T>
static int stage_tile(mod16_ctx* ctx, const EtArgs<T>& h, unsigned flags, const T* dscal,
                      size_t per_arr, int slot, int64_t off, int64_t m, const BcTable<T>& bc,
                      double* tile_diag = nullptr) {
    hipStream_t st = ctx->streams[slot];
    char* base = static_cast<char*>(ctx->slab[slot]);
    EtArgs<T> d = h;
    d.n = m;
    d.base = off;
    for (int k = 0; k < 14; ++k) {
        if ((h.dense_drv >> k) & 1u) {
            T* dp = reinterpret_cast<T*>(base + per_arr * k);
            HIPCHK(ctx, hipMemcpyAsync(dp, h.drv[k] + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
            d.drv[k] = dp;
        } else if (bc.drv[k]) {
            d.drv[k] = bc.drv[k];
        } else {
            d.drv[k] = dscal + k;
        }
    }
    if (h.cls) {
        if (h.cls_mode == MOD16_BC_DENSE) {
            uint8_t* dc = reinterpret_cast<uint8_t*>(base + per_arr * 35);
            HIPCHK(ctx, hipMemcpyAsync(dc, h.cls + off, (size_t)m, hipMemcpyHostToDevice, st));
            d.cls = dc;
        } else {
            d.cls = bc.cls;
        }
    } else {
        for (int k = 0; k < 11; ++k) {
            if ((h.dense_par >> k) & 1u) {
                T* dp = reinterpret_cast<T*>(base + per_arr * (14 + k));
                HIPCHK(ctx, hipMemcpyAsync(dp, h.par[k] + off, sizeof(T) * m, hipMemcpyHostToDevice, st));
                d.par[k] = dp;
            } else if (bc.par[k]) {
                d.par[k] = bc.par[k];
            } else {
                d.par[k] = dscal + 14 + k;
            }
        }
    }
    for (int k = 0; k < 10; ++k)
        d.out[k] = h.out[k] ? reinterpret_cast<T*>(base + per_arr * (25 + k)) : nullptr;
    // tile_diag: the diagnostics vector of THIS tile (host, 8 doubles), reduced on the device
    // while the tile's outputs are there
    double* dd = tile_diag ? ctx->hdiag_dev + (size_t)slot * kDiag : nullptr;
    {
        std::lock_guard<std::mutex> lock(ctx->launch_mu);
        int rc = launch_et<T>(ctx, d, flags, st, dd);
        if (rc != MOD16_OK) return rc;
    }
    for (int k = 0; k < 10; ++k)
        if (h.out[k]) HIPCHK(ctx, hipMemcpyAsync(h.out[k] + off, d.out[k], sizeof(T) * m, hipMemcpyDeviceToHost, st));
    if (dd) HIPCHK(ctx, hipMemcpyAsync(tile_diag, dd, sizeof(double) * kDiag, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));      // the slab of this slot is free again
    return MOD16_OK;
}

// The page-locked buffer of the small calls: 256 bytes of scalars, `arrays` arrays of `elem`-byte
// values and up to three of bytes behind them, for n pixels. It grows with the largest call seen
// (powers of two from 1024 pixels: a caller of scalars pins 0.3 MB, one of 256 x 256 windows 18 MB).
// Also makes sure of streams[0]. -> false: no page-locked memory to be had (the context stops
// asking: its calls are staged from now on).
static bool small_reserve(mod16_ctx* ctx, int64_t n, size_t elem, int arrays, size_t* per_arr) {
    int64_t cap = 1024;
    while (cap < n) cap *= 2;
    *per_arr = (size_t)cap * elem;
    const size_t need = 256 + *per_arr * arrays + 3 * (size_t)cap + 256;
    bool ok = true;
    if (ctx->small_bytes < need) {
        if (ctx->small_host) (void)hipHostFree(ctx->small_host);
        ctx->small_host = ctx->small_dev = nullptr;
        ctx->small_bytes = 0;
        ok = hipHostMalloc(&ctx->small_host, need, hipHostMallocDefault) == hipSuccess &&
             hipHostGetDevicePointer(&ctx->small_dev, ctx->small_host, 0) == hipSuccess;
        if (ok) ctx->small_bytes = need;
    }
    if (ok && !ctx->streams[0]) ok = hipStreamCreateWithFlags(&ctx->streams[0], hipStreamNonBlocking) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (ctx->small_host) (void)hipHostFree(ctx->small_host);
        ctx->small_host = ctx->small_dev = nullptr;
        ctx->small_bytes = 0;
        ctx->small_pixels = 0;
    }
    return ok;
}

// HOST mode, small calls. The staged path costs a dozen copy commands whatever the size (each
// dense input its own, pageable memory: the runtime stages and waits), a status read-back and
// three synchronisations -- 64 us for ONE pixel, where the reference's numpy takes 86 us for its
// whole forward run (BASELINE.json configs[0]: a flux-tower site), ~290 us up to 16 k pixels.
// Measured against it (tools/smallcall.py, profiles/r05_small_calls.jsonl): 18 us against 64 for one
// pixel, 102 against 273 at 100 x 100, 371 against 420 at 256 x 256, even at ~90 k pixels, slower
// beyond (the CPU's copies into the buffer grow faster than the runtime's DMA): kSmallPixels.
// Here the CPU copies the inputs into one page-locked buffer, the kernel reads them from there and
// writes its outputs there (host memory is in the device's address space: a few KB over the link),
// and the CPU copies the outputs on: one launch, one synchronisation, the same kernels on the same
// values -- the same bits as the staged path gives. Class codes are checked here instead of by the
// kernel (the staged path reads the kernel's status word back).
template <typename T>
static int run_host_small(mod16_ctx* ctx, const EtArgs<T>& h, unsigned flags) {
    const int64_t n = h.n;
    size_t per_arr = 0;
    if (!small_reserve(ctx, n, sizeof(T), 14 + 11 + 10, &per_arr)) return kSmallUnavailable;
    hipStream_t st = ctx->streams[0];
    if (h.cls) {       // (dense: a broadcast class raster is has_rows_or_cols' business)
        for (int64_t i = 0; i < n; ++i)
            if (h.cls[i] >= MOD16_N_CLASSES)
                return fail(ctx, MOD16_ERR_CLASS_RANGE, "class raster holds a code >= 13 (numpy would raise IndexError)");
    }
    char* hb = static_cast<char*>(ctx->small_host);
    char* db = static_cast<char*>(ctx->small_dev);
    T* hs = reinterpret_cast<T*>(hb);              // 25 broadcast scalars in the first 256 bytes
    const T* dscal = reinterpret_cast<const T*>(db);
    EtArgs<T> d = h;
    d.base = 0;
    // whole 16-byte vectors: a ragged end would cost a second launch (the one-pixel-per-thread
    // kernel behind the vector kernel) -- the buffer has the room, the pad pixels repeat the last
    // pixel (so they are no new case for the domain guard), and their outputs stay in the buffer
    constexpr int V = VecOf<T>::v;
    const int64_t npad = (n + V - 1) / V * V;
    d.n = npad;
    auto arr = [&](int k) { return (size_t)256 + per_arr * k; };
    auto put = [&](size_t off, const void* src, size_t elem) {
        memcpy(hb + off, src, elem * n);
        for (int64_t i = n; i < npad; ++i) memcpy(hb + off + elem * i, static_cast<const char*>(src) + elem * (n - 1), elem);
    };
    for (int k = 0; k < 14; ++k) {
        if ((h.dense_drv >> k) & 1u) {
            put(arr(k), h.drv[k], sizeof(T));
            d.drv[k] = reinterpret_cast<const T*>(db + arr(k));
        } else {
            hs[k] = h.drv[k][0];
            d.drv[k] = dscal + k;
        }
    }
    if (h.cls) {
        const size_t off = arr(35);
        put(off, h.cls, 1);
        d.cls = reinterpret_cast<const uint8_t*>(db + off);
    } else {
        for (int k = 0; k < 11; ++k) {
            if ((h.dense_par >> k) & 1u) {
                put(arr(14 + k), h.par[k], sizeof(T));
                d.par[k] = reinterpret_cast<const T*>(db + arr(14 + k));
            } else {
                hs[14 + k] = h.par[k][0];
                d.par[k] = dscal + 14 + k;
            }
        }
    }
    for (int k = 0; k < 10; ++k)
        d.out[k] = h.out[k] ? reinterpret_cast<T*>(db + arr(25 + k)) : nullptr;
    int rc = launch_et<T>(ctx, d, flags, st);
    if (rc != MOD16_OK) return rc;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < 10; ++k)
        if (h.out[k]) memcpy(h.out[k], hb + arr(25 + k), sizeof(T) * n);
    return MOD16_OK;
}

template <typename T>
static int run_host(mod16_ctx* ctx, const EtArgs<T>& h, unsigned flags, double* tile_diag = nullptr) {
    const int64_t n = h.n;
    if (n == 0) return MOD16_OK;
    if (n <= ctx->small_pixels && !tile_diag && !has_rows_or_cols(h)) {
        const int rc = run_host_small<T>(ctx, h, flags);
        if (rc != kSmallUnavailable) return rc;
    }
    const int64_t tile = std::min<int64_t>(n, kTilePixels);
    const int64_t ntiles = (n + tile - 1) / tile;
    const int nslots = (int)std::min<int64_t>(ntiles, ctx->host_threads);
    if (nslots > 1) ctx->ws_multi = true;       // one stream per slot: the launches leave their events (ws_release)
    // slab layout per slot: 14 drivers | 11 params | 10 outputs (T each) | class bytes
    // successive staged arrays are kStagger bytes apart on top of their size
    const size_t per_arr = (((size_t)tile * sizeof(T)) + 255) / 256 * 256 + kStagger;
    const size_t need = per_arr * (14 + 11 + 10) + (size_t)tile + 256;
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
    // broadcast scalars live in one small device array
    T hs[32];
    for (int k = 0; k < 14; ++k) hs[k] = ((h.dense_drv >> k) & 1u) ? T(0) : h.drv[k][0];
    for (int k = 0; k < 11; ++k) hs[14 + k] = (!h.cls && !((h.dense_par >> k) & 1u)) ? h.par[k][0] : T(0);
    HIPCHK(ctx, hipMemcpy(ctx->scalars, hs, sizeof(T) * 25, hipMemcpyHostToDevice));
    const T* dscal = static_cast<const T*>(ctx->scalars);
    // (N,) rows and (T, 1) columns: whole, once, next to the tiles
    BcTable<T> bc;
    if (has_rows_or_cols(h)) {
        const int64_t nrow = h.inner, ncol = n / h.inner;
        auto len_of = [&](bool row) { return (size_t)(row ? nrow : ncol); };
        size_t need_bc = 256;
        for (int k = 0; k < 14; ++k)
            if (((h.row_drv | h.col_drv) >> k) & 1u) need_bc += (len_of((h.row_drv >> k) & 1u) * sizeof(T) + 255) / 256 * 256;
        for (int k = 0; k < 11 && !h.cls; ++k)
            if (((h.row_par | h.col_par) >> k) & 1u) need_bc += (len_of((h.row_par >> k) & 1u) * sizeof(T) + 255) / 256 * 256;
        if (h.cls && h.cls_mode != MOD16_BC_DENSE)
            need_bc += (h.cls_mode == MOD16_BC_SCALAR ? 1 : len_of(h.cls_mode == MOD16_BC_ROW)) + 256;
        if (ctx->bc_bytes < need_bc) {
            if (ctx->bc_buf) HIPCHK(ctx, hipFree(ctx->bc_buf));
            ctx->bc_buf = nullptr;
            ctx->bc_bytes = 0;
            HIPCHK(ctx, hipMalloc(&ctx->bc_buf, need_bc));
            ctx->bc_bytes = need_bc;
        }
        char* cur = static_cast<char*>(ctx->bc_buf);
        auto up = [&](const void* src, size_t bytes) -> const void* {
            char* p = cur;
            if (hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
            cur += (bytes + 255) / 256 * 256;
            return p;
        };
        for (int k = 0; k < 14; ++k)
            if (((h.row_drv | h.col_drv) >> k) & 1u) {
                bc.drv[k] = static_cast<const T*>(up(h.drv[k], len_of((h.row_drv >> k) & 1u) * sizeof(T)));
                if (!bc.drv[k]) return fail(ctx, MOD16_ERR_HIP, "mod16_et2: upload of a broadcast input failed");
            }
        for (int k = 0; k < 11 && !h.cls; ++k)
            if (((h.row_par | h.col_par) >> k) & 1u) {
                bc.par[k] = static_cast<const T*>(up(h.par[k], len_of((h.row_par >> k) & 1u) * sizeof(T)));
                if (!bc.par[k]) return fail(ctx, MOD16_ERR_HIP, "mod16_et2: upload of a broadcast input failed");
            }
        if (h.cls && h.cls_mode != MOD16_BC_DENSE) {
            bc.cls = static_cast<const uint8_t*>(up(h.cls, h.cls_mode == MOD16_BC_SCALAR ? 1 : len_of(h.cls_mode == MOD16_BC_ROW)));
            if (!bc.cls) return fail(ctx, MOD16_ERR_HIP, "mod16_et2: upload of the class raster failed");
        }
    }
    // the kernels' shared workspace at its final size before any thread launches
    {
        const int64_t npiece = (tile / VecOf<T>::v + 63) / 64;
        int rc = reserve_diag(ctx, npiece / 2 + 2048);
        if (rc != MOD16_OK) return rc;
    }
    if (nslots == 1) {
        for (int64_t off = 0; off < n; off += tile) {
            int rc = stage_tile<T>(ctx, h, flags, dscal, per_arr, 0, off, std::min(tile, n - off), bc,
                                   tile_diag ? tile_diag + (off / tile) * kDiag : nullptr);
            if (rc != MOD16_OK) return rc;
        }
    } else {
        int rcs[kSlots] = {};
        std::vector<std::thread> workers;
        for (int s = 0; s < nslots; ++s)
            workers.emplace_back([&, s]() {
                if (hipSetDevice(ctx->device) != hipSuccess) { rcs[s] = MOD16_ERR_HIP; return; }
                for (int64_t t = s; t < ntiles && rcs[s] == MOD16_OK; t += nslots)
                    rcs[s] = stage_tile<T>(ctx, h, flags, dscal, per_arr, s, t * tile, std::min(tile, n - t * tile), bc,
                                           tile_diag ? tile_diag + t * kDiag : nullptr);
            });
        for (auto& w : workers) w.join();
        for (int s = 0; s < nslots; ++s)
            if (rcs[s] != MOD16_OK) return rcs[s];
    }
    for (int s = 0; s < nslots; ++s) HIPCHK(ctx, hipStreamSynchronize(ctx->streams[s]));
    return read_status(ctx, ctx->streams[0]);
}

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

// ---- the forward run + diagnostics of one raster as a HIP graph: the launch
// sequence of mod16_et_diag_* (counter reset, pipeline kernel, staged fixed-order
// sum) captured once and replayed with one call per time step.
struct mod16_graph {
    mod16_ctx* ctx = nullptr;                // for error text at launch; not touched by destroy
    int device = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long* counter = nullptr;   // its own ticket counter: replays never meet the ring
    DiagWs ws;                               // its own diagnostics workspace (freed with the graph)
};

extern "C" int mod16_graph_destroy(mod16_graph* g) {
    if (!g) return MOD16_OK;
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
    // (a graph may outlive the context it was built with: no error text through g->ctx)
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

extern "C" int mod16_check_status(mod16_ctx* ctx, void* stream) {
    MOD16_LOCK(ctx);
    if (!ctx) return MOD16_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return read_status(ctx, static_cast<hipStream_t>(stream));
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

// Page-locked host memory for result arrays (mod16_amd/_lib.py keeps a small pool): a
// device-to-host copy into fresh pageable memory runs at the kernel's page-fault rate
// (13 GB/s measured, tools/probe_pcie.hip), into pinned memory at the PCIe rate (57 GB/s).
extern "C" int mod16_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes <= 0) return MOD16_ERR_ARG;
    *out = nullptr;
    void* p = nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return MOD16_ERR_NOMEM;
    }
    *out = p;
    return MOD16_OK;
}
extern "C" int mod16_host_free(void* p) {
    if (!p) return MOD16_OK;
    return hipHostFree(p) == hipSuccess ? MOD16_OK : MOD16_ERR_HIP;
}

extern "C" int mod16_measure_copy(mod16_ctx* ctx, int64_t bytes, int reps, float* gbps) {
    MOD16_LOCK(ctx);
    if (!ctx || !gbps || bytes < 16 || reps <= 0) return fail(ctx, MOD16_ERR_ARG, "mod16_measure_copy: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t nvec = bytes / 16;
    void *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, nvec * 16) != hipSuccess || hipMalloc(&b, nvec * 16) != hipSuccess) {
        (void)hipGetLastError();
        if (a) (void)hipFree(a);
        return fail(ctx, MOD16_ERR_NOMEM, "mod16_measure_copy: device memory for the two buffers");
    }
    int rc = [&]() -> int {
        HIPCHK(ctx, hipMemset(a, 1, nvec * 16));
        HIPCHK(ctx, hipMemset(b, 0, nvec * 16));
        hipEvent_t e0, e1;
        HIPCHK(ctx, hipEventCreate(&e0));
        HIPCHK(ctx, hipEventCreate(&e1));
        const unsigned grid = (unsigned)((nvec + kBlock - 1) / kBlock);
        float best = 1e30f;
        for (int r = 0; r <= reps; ++r) {      // the first launch is a warm-up
            HIPCHK(ctx, hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(kBlock), 0, nullptr,
                               static_cast<const copy_vec_t*>(a), static_cast<copy_vec_t*>(b), nvec);
            HIPCHK(ctx, hipEventRecord(e1, nullptr));
            HIPCHK(ctx, hipEventSynchronize(e1));
            float ms = 0.f;
            HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms < best) best = ms;
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        *gbps = (float)(2.0 * (double)nvec * 16.0 / (best * 1e-3) / 1e9);
        return MOD16_OK;
    }();
    (void)hipFree(a);
    (void)hipFree(b);
    return rc;
}

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

// ------------------------------------------------------------- diagnostics
template <typename T>
static int reduce_entry(mod16_ctx* ctx, const T* day, const T* night, int64_t n, double* diag,
                        double* ddiag, void* stream) {
    if (!ctx || !day || !night || n < 0) return fail(ctx, MOD16_ERR_ARG, "mod16_reduce_diag: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(kDiagBlocks, (n + kBlock - 1) / kBlock));
    DiagWs* ws = nullptr;
    int rc = reserve_diag(ctx, blocks, &ws);
    if (rc != MOD16_OK) return rc;
    rc = ws_acquire(ctx, st);
    if (rc != MOD16_OK) return rc;
    hipLaunchKernelGGL((diag_partial_kernel<T>), dim3(blocks), dim3(kBlock), 0, st, day, night, n, ws->partial);
    double* dst = ddiag ? ddiag : ctx->diag_dev;
    hipLaunchKernelGGL(diag_final_kernel, dim3(1), dim3(kBlock), 0, st, ws->partial, blocks, dst);
    HIPCHK(ctx, hipGetLastError());
    rc = ws_release(ctx, st);
    if (rc != MOD16_OK) return rc;
    if (diag) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->diag_host, dst, sizeof(double) * kDiag, hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        memcpy(diag, ctx->diag_host, sizeof(double) * kDiag);
    }
    return MOD16_OK;
}

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

// ---------------------------------------------------------------- generator
// tile (pixels) -> log2, or -1 if it is not a power of two >= lo
static int tile_log2(int64_t tile, int64_t lo) {
    if (tile < lo || (tile & (tile - 1)) != 0) return -1;
    int sh = 0;
    while ((int64_t(1) << sh) < tile) ++sh;
    return sh;
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
    if (!g || !g->exec || !ms || launches <= 0) return MOD16_ERR_ARG;
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
