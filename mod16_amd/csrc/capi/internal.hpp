// libmod16hip.so, host side: what the entry-point families (capi/*.hip) share -- the device
// context, the launch machinery of the pipeline and of the plain kernels, the HOST-mode tiler.
// Internal linkage throughout (every translation unit holds its own copy of what it uses);
// mod16_capi.hip includes all families into ONE unit for single-command builds (variants,
// tests/host_asan, listings), mod16_amd/csrc/build.py compiles them side by side.
#pragma once
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

#include "../../../include/mod16_hip.h"
#include "../mod16_kernels.hpp"
#include "../mod16_stream.hpp"

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

// A captured graph's kernel nodes point into its context (parameter table, exp / log tables, status
// word): the context keeps a list of its live graphs, and mod16_destroy marks them dead -- a replay
// of one returns MOD16_ERR_ARG instead of reading freed memory; mod16_graph_destroy still frees it.
// One process-wide lock (an inline function: the same object in every translation unit of the library).
inline std::mutex& graph_registry_mu() {
    static std::mutex mu;
    return mu;
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
    std::vector<mod16_graph*> graphs; // graphs captured with this context and still alive (graph_registry_mu)
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
    // (the pipeline kernel's piece numbers and tile rows are 32-bit scalars: 2^30 pieces are 137 G
    // float64 pixels -- a raster of 1.7 TB -- and a row is 15 tiles' worth of elements)
    if (g.npiece > kMaxPieces || (uint64_t)s.wide_row >> 32 || (uint64_t)s.out_row >> 32 || (uint64_t)s.byte_row >> 32)
        return fail(ctx, MOD16_ERR_ARG, "pipeline launch: more than 2^30 pieces, or a tile row of 2^32 elements or more");
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
    s.dma_step[0] = (int64_t)pitch_b - 1024;
    s.dma_step[1] = (int64_t)pitch_b + 3072;
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


// ---- the forward run + diagnostics of one raster as a HIP graph: the launch
// sequence of mod16_et_diag_* (counter reset, pipeline kernel, staged fixed-order
// sum) captured once and replayed with one call per time step.
struct mod16_graph {
    mod16_ctx* ctx = nullptr;                // NULL once the context has been destroyed (graph_registry_mu)
    int device = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    unsigned long long* counter = nullptr;   // its own ticket counter: replays never meet the ring
    DiagWs ws;                               // its own diagnostics workspace (freed with the graph)
};
static void graph_register(mod16_ctx* ctx, mod16_graph* g) {
    std::lock_guard<std::mutex> lock(graph_registry_mu());
    ctx->graphs.push_back(g);
}
// false: the context the graph was captured with is gone (its tables with it)
static bool graph_alive(const mod16_graph* g) {
    std::lock_guard<std::mutex> lock(graph_registry_mu());
    return g->ctx != nullptr;
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


// ---------------------------------------------------------------- generator
// tile (pixels) -> log2, or -1 if it is not a power of two >= lo
static int tile_log2(int64_t tile, int64_t lo) {
    if (tile < lo || (tile & (tile - 1)) != 0) return -1;
    int sh = 0;
    while ((int64_t(1) << sh) < tile) ++sh;
    return sh;
}


