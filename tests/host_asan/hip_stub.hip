// A stand-in for the HIP runtime under the HOST half of libmod16hip (tests/host_asan): SURVEY.md
// section 5 asks for a sanitizer build of the host shim, and GPU AddressSanitizer is not available on
// the pool, so the library's host code -- mod16_amd/csrc/mod16_capi.hip compiled with
// `hipcc --cuda-host-only -fsanitize=address,undefined` -- is linked against THIS instead of
// libamdhip64: "device" memory is host heap (so every staged copy, memset and workspace is under
// AddressSanitizer), streams / events / graphs are inert handles, and a kernel launch runs the
// kernel's SHADOW: the launch's address arithmetic -- tile / row / pitch offsets of every piece,
// workspace sizes, ticket-ring slots, grid and block shapes -- replayed on the host against the table
// of live allocations, touching nothing. An address range outside every allocation, a misaligned
// vector access, an impossible launch shape or a use of freed memory ends the run with a message.
//
// Allocations above 64 MiB are address-space reservations without memory (PROT_NONE): the shadows
// only compute addresses, so rasters of 43200 x 21600 or 2^31 + 12345 pixels run through the real
// launch geometry in a container without a GPU. Copies and memsets that touch such a range are range-checked
// and skipped.
//
// Compiled as HIP (host only) so that it sees the kernels' own argument structures
// (StreamArgs, EtArgs, SynthArgs: mod16_amd/csrc/*.hpp) instead of copies of them.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mod16_hip.h"
// (this translation unit's copy of the kernels' declarations lives in a namespace of its own: the
// non-template kernels of the headers would otherwise be defined twice in the linked program)
#define mod16 mod16_shadow
#include "../../mod16_amd/csrc/mod16_kernels.hpp"
#include "../../mod16_amd/csrc/mod16_stream.hpp"

using namespace mod16;

namespace stub {

constexpr size_t kRealBelow = size_t(64) << 20;

struct Block { size_t size; bool fake; };
// (constructed on first use: the library's module constructor registers its kernels before this
// translation unit's globals would be initialised)
static std::map<uintptr_t, Block>& blocks() { static auto* m = new std::map<uintptr_t, Block>; return *m; }   // base -> block
static std::mutex& mu() { static auto* m = new std::mutex; return *m; }
static std::map<const void*, std::string>& kernels() { static auto* m = new std::map<const void*, std::string>; return *m; }   // host stub -> mangled device name
static std::map<std::string, long>& launches() { static auto* m = new std::map<std::string, long>; return *m; }
static std::map<std::string, long>& unchecked() { static auto* m = new std::map<std::string, long>; return *m; }
#define g_blocks blocks()
#define g_mu mu()
#define g_kernels kernels()
#define g_launches launches()
#define g_unchecked unchecked()
static long g_checked_ranges = 0;
static int g_cus = 256;

[[noreturn]] static void die(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "hip_stub: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
    abort();
}

// [p, p + bytes) must lie inside ONE live allocation
static bool inside(const void* p, size_t bytes, bool* fake = nullptr) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(p);
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_blocks.upper_bound(a);
    if (it == g_blocks.begin()) return false;
    --it;
    if (a + bytes > it->first + it->second.size) return false;
    if (fake) *fake = it->second.fake;
    return true;
}
static void need(const void* p, size_t bytes, size_t align, const char* what, const char* kernel) {
    ++g_checked_ranges;
    if (bytes == 0) return;
    if (!p) die("%s: %s is NULL", kernel, what);
    if (reinterpret_cast<uintptr_t>(p) % align) die("%s: %s at %p is not %zu-byte aligned", kernel, what, p, align);
    if (!inside(p, bytes)) die("%s: %s [%p, +%zu) lies outside every live device allocation", kernel, what, p, bytes);
}

static void* alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    void* p;
    bool fake = bytes >= kRealBelow;
    if (fake) {
        p = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return nullptr;
    } else {
        p = aligned_alloc(256, (bytes + 255) / 256 * 256);      // hipMalloc hands out 256-byte aligned blocks
        if (!p) return nullptr;
        memset(p, 0xA5, bytes);
    }
    std::lock_guard<std::mutex> lock(g_mu);
    g_blocks[reinterpret_cast<uintptr_t>(p)] = Block{bytes, fake};
    return p;
}
static hipError_t release(void* p) {
    if (!p) return hipSuccess;
    Block b;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        auto it = g_blocks.find(reinterpret_cast<uintptr_t>(p));
        if (it == g_blocks.end()) die("free of %p, which is not the base of a live allocation", p);
        b = it->second;
        g_blocks.erase(it);
    }
    if (b.fake) munmap(p, b.size);
    else free(p);
    return hipSuccess;
}

// ------------------------------------------------------------------ shadows
static int stream_counts(int mode, int* nw, int* nb, int* nout) {
    switch (mode) {
#define C(M) case M: *nw = StreamSpec<M>::NW; *nb = StreamSpec<M>::NB; *nout = StreamSpec<M>::NOUT; return 1;
        C(kStreamPet) C(kStreamSep8) C(kStreamSep6) C(kStreamRaw) C(kStreamRawTotal) C(kStreamRawTotalHours)
        C(kStreamTotals) C(kStreamTotalsMixed) C(kStreamPetMixed) C(kStreamSep8Mixed) C(kStreamSep6Mixed)
        C(kStreamRawMixed) C(kStreamRawTotalMixed) C(kStreamRawTotalHoursMixed)
#undef C
    }
    return 0;
}

// et_stream_kernel<T, MODE, PITCHED, GUARD>: every piece's reads and writes, the workspace, the ticket
template <typename T>
static void shadow_stream(const StreamArgs<T>& a, int mode, bool pitched, dim3 grid, dim3 block, const char* name) {
    constexpr int V = 16 / (int)sizeof(T);
    int NW, NB, NOUT;
    if (!stream_counts(mode, &NW, &NB, &NOUT)) die("%s: unknown mode %d", name, mode);
    if (block.x != (unsigned)kBlock || block.y != 1 || block.z != 1) die("%s: block %u x %u x %u", name, block.x, block.y, block.z);
    if (grid.x < 1 || grid.x > (unsigned)(g_cus * 2) || grid.y != 1) die("%s: grid %u (persistent waves: 1 .. 2 blocks per CU)", name, grid.x);
    if (a.n <= 0 || a.n % V) die("%s: n = %lld is not a positive multiple of the vector width", name, (long long)a.n);
    if (a.run_shift < 0 || a.run_shift > 6) die("%s: run_shift %d", name, a.run_shift);
    if (a.run_shift > a.tile_shift) die("%s: a run of 2^%d pieces straddles tiles of 2^%d", name, a.run_shift, a.tile_shift);
    const int64_t nvec = a.n / V, npiece = (nvec + 63) / 64;
    const int64_t nruns_geom = (npiece + (int64_t(1) << a.run_shift) - 1) >> a.run_shift;
    if (a.static_sched) {
        if (a.nruns != (int64_t)grid.x) die("%s: static schedule with %lld partials for %u blocks", name, (long long)a.nruns, grid.x);
    } else {
        if (a.nruns != nruns_geom) die("%s: %lld partials for %lld runs", name, (long long)a.nruns, (long long)nruns_geom);
        need(a.dyn_counter, 16, 8, "ticket counter", name);
    }
    need(a.lut64, sizeof(double) * MOD16_LUT_ROWS * kLutCols, 8, "BPLUT", name);
    need(a.tab, sizeof(double) * FastMath<double>::kTabDoubles, 16, "exp / log tables", name);
    need(a.status, 4, 4, "status word", name);
    need(a.diag_partial, sizeof(double) * kDiag * (size_t)a.nruns, 8, "diagnostics partials", name);
    // the runs' cancellation lists (mixed-precision forms under the dynamic schedule write them)
    if (!a.static_sched && stream_is_mixed(mode))
        need(a.cancel_list, sizeof(uint16_t) * kCancelCap * (size_t)a.nruns, 2, "cancellation lists", name);
    if (a.diag_out) {
        need(a.diag_out, sizeof(double) * kDiag, 8, "diagnostics vector", name);
        need(a.done_counter, 4, 4, "blocks-done counter", name);
    }
    if (pitched) {
        for (int k = 1; k < NW; ++k)
            if (a.wide[k] != a.wide[0] + k * a.wide_pitch) die("%s: wide[%d] is not wide[0] + %d * pitch", name, k, k);
        // the two steps between successive DMA bases (four LDS slots per value of M0, et_stream_kernel's issue())
        const int64_t pitch_b = a.wide_pitch * (int64_t)sizeof(T);
        if (a.dma_step[0] != pitch_b - 1024 || a.dma_step[1] != pitch_b + 3072)
            die("%s: DMA steps %lld / %lld for a pitch of %lld bytes", name, (long long)a.dma_step[0], (long long)a.dma_step[1], (long long)pitch_b);
    }
    // 32-bit piece numbers and tile rows in the kernel
    if (npiece > kMaxPieces || (uint64_t)a.wide_row >> 32 || (uint64_t)a.out_row >> 32 || (uint64_t)a.byte_row >> 32)
        die("%s: %lld pieces / rows %lld %lld %lld do not fit the kernel's 32-bit scalars", name, (long long)npiece,
            (long long)a.wide_row, (long long)a.out_row, (long long)a.byte_row);
    const bool tiled = a.tile_shift != kNoTile;
    // plain arrays: one check per array; tiled: per tile (pieces of a tile are contiguous)
    const int64_t per_tile = tiled ? (int64_t(1) << a.tile_shift) : npiece;
    for (int64_t p0 = 0; p0 < npiece; p0 += per_tile) {
        const int64_t p1 = std::min(npiece, p0 + per_tile);
        const int64_t tile = tiled ? (p0 >> a.tile_shift) : 0;
        const int64_t first = (p0 - (tiled ? (tile << a.tile_shift) : 0)) * (int64_t)(64 * V);
        const int64_t elems = std::min<int64_t>(a.n - p0 * 64 * V, (p1 - p0) * 64 * V);
        const int64_t ow = tile * a.wide_row + first, oo = tile * a.out_row + first, ob = tile * a.byte_row + first;
        for (int k = 0; k < NW; ++k) need(a.wide[k] + ow, sizeof(T) * elems, 16, "driver array", name);
        for (int k = 0; k < NB; ++k) need(a.bytes[k] + ob, (size_t)elems, V, "byte raster", name);
        for (int k = 0; k < NOUT; ++k) need(a.out[k] + oo, sizeof(T) * elems, 16, "output array", name);
    }
}

template <typename T>
static void shadow_et(const EtArgs<T>& a, dim3 grid, dim3 block, const char* name) {
    if (block.x != (unsigned)kBlock) die("%s: block %u", name, block.x);
    if (grid.x < 1) die("%s: empty grid", name);
    if (a.n <= 0) die("%s: n = %lld", name, (long long)a.n);
    const int64_t inner = a.inner > 0 ? a.inner : 1;
    auto len = [&](bool dense, bool row, bool col) -> int64_t {
        if (dense) return a.n;
        if (row) return inner;
        if (col) return (a.base + a.n + inner - 1) / inner;
        return 1;
    };
    for (int k = 0; k < 14; ++k)
        need(a.drv[k], sizeof(T) * len((a.dense_drv >> k) & 1u, (a.row_drv >> k) & 1u, (a.col_drv >> k) & 1u), sizeof(T), "driver", name);
    if (a.cls) {
        need(a.cls, (size_t)len(a.cls_mode == MOD16_BC_DENSE, a.cls_mode == MOD16_BC_ROW, a.cls_mode == MOD16_BC_COL), 1, "class raster", name);
        need(a.lut64, sizeof(double) * MOD16_LUT_ROWS * kLutCols, 8, "BPLUT", name);
    } else {
        for (int k = 0; k < 11; ++k)
            need(a.par[k], sizeof(T) * len((a.dense_par >> k) & 1u, (a.row_par >> k) & 1u, (a.col_par >> k) & 1u), sizeof(T), "parameter", name);
    }
    for (int k = 0; k < 10; ++k)
        if (a.out[k]) need(a.out[k], sizeof(T) * a.n, sizeof(T), "output", name);
    need(a.status, 4, 4, "status word", name);
}

template <typename T>
static void shadow_synth(const SynthArgs<T>& a, const char* name) {
    if (a.n <= 0) die("%s: n = %lld", name, (long long)a.n);
    const int64_t tile = a.tile_shift >= 62 ? a.n : (int64_t(1) << a.tile_shift);
    for (int64_t i = 0; i < a.n; i += tile) {
        const int64_t m = std::min(tile, a.n - i);
        const int64_t e = a.tile_shift >= 62 ? 0 : (i >> a.tile_shift) * a.drv_row;
        const int64_t c = a.tile_shift >= 62 ? 0 : (i >> a.tile_shift) * a.cls_row;
        for (int k = 0; k < 14; ++k) need(a.drv[k] + e, sizeof(T) * m, sizeof(T), "driver array", name);
        need(a.cls + c, (size_t)m, 1, "class raster", name);
    }
}

static bool has(const std::string& s, const char* sub) { return s.find(sub) != std::string::npos; }

static void run_shadow(const std::string& name, dim3 grid, dim3 block, void** args) {
    {
        std::lock_guard<std::mutex> lock(g_mu);
        ++g_launches[name.substr(0, name.find('I') == std::string::npos ? name.size() : name.find('I'))];
    }
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.x * block.y * block.z > 1024)
        die("%s: launch shape %u x %u x %u blocks of %u x %u x %u", name.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z);
    const char* n = name.c_str();
    if (has(name, "16et_stream_kernelI")) {
        // ...et_stream_kernelI{d|f}Li<MODE>ELb<PITCHED>ELb<GUARD>EE...
        const size_t at = name.find("16et_stream_kernelI") + 19;
        const bool f64 = name[at] == 'd';
        const int mode = atoi(name.c_str() + at + 3);
        const size_t lb = name.find("ELb", at);
        const bool pitched = name[lb + 3] == '1';
        if (f64) shadow_stream(*static_cast<const StreamArgs<double>*>(args[0]), mode, pitched, grid, block, n);
        else shadow_stream(*static_cast<const StreamArgs<float>*>(args[0]), mode, pitched, grid, block, n);
    } else if (has(name, "21et_stream_redo_kernelI")) {
        const size_t at = name.find("21et_stream_redo_kernelI") + 24;
        if (name[at] == 'd') {
            const auto& a = *static_cast<const StreamArgs<double>*>(args[0]);
            need(a.diag_partial, sizeof(double) * kDiag * (size_t)a.nruns, 8, "diagnostics partials", n);
        } else {
            const auto& a = *static_cast<const StreamArgs<float>*>(args[0]);
            need(a.diag_partial, sizeof(double) * kDiag * (size_t)a.nruns, 8, "diagnostics partials", n);
            need(a.cancel_list, sizeof(uint16_t) * kCancelCap * (size_t)a.nruns, 2, "cancellation lists", n);
        }
    } else if (has(name, "23et_stream_cancel_kernelI")) {
        const auto& a = *static_cast<const StreamArgs<float>*>(args[0]);      // (float32 rasters only)
        need(a.diag_partial, sizeof(double) * kDiag * (size_t)a.nruns, 8, "diagnostics partials", n);
        need(a.cancel_list, sizeof(uint16_t) * kCancelCap * (size_t)a.nruns, 2, "cancellation lists", n);
    } else if (has(name, "9et_kernelI")) {
        const size_t at = name.find("9et_kernelI") + 11;
        if (name[at] == 'd') shadow_et(*static_cast<const EtArgs<double>*>(args[0]), grid, block, n);
        else shadow_et(*static_cast<const EtArgs<float>*>(args[0]), grid, block, n);
    } else if (has(name, "12synth_kernelI")) {
        const size_t at = name.find("12synth_kernelI") + 15;
        if (name[at] == 'd') shadow_synth(*static_cast<const SynthArgs<double>*>(args[0]), n);
        else shadow_synth(*static_cast<const SynthArgs<float>*>(args[0]), n);
    } else if (has(name, "17diag_stage_kernel")) {
        const double* partial = *static_cast<const double* const*>(args[0]);
        const int64_t count = *static_cast<const int64_t*>(args[1]), per = *static_cast<const int64_t*>(args[2]);
        double* out = *static_cast<double* const*>(args[3]);
        if (per < 1 || (int64_t)grid.x * per < count) die("%s: %u slices of %lld do not cover %lld partials", n, grid.x, (long long)per, (long long)count);
        need(partial, sizeof(double) * kDiag * (size_t)count, 8, "partials", n);
        need(out, sizeof(double) * kDiag * grid.x, 8, "stage output", n);
    } else if (has(name, "23diag_final_fused_kernel") || has(name, "17diag_final_kernel")) {
        const double* partial = *static_cast<const double* const*>(args[0]);
        const int nb = *static_cast<const int*>(args[1]);
        double* out = *static_cast<double* const*>(args[has(name, "fused") ? 3 : 2]);
        if (nb < 1) die("%s: %d partials", n, nb);
        need(partial, sizeof(double) * kDiag * (size_t)nb, 8, "partials", n);
        need(out, sizeof(double) * kDiag, 8, "diagnostics vector", n);
    } else if (has(name, "19diag_partial_kernelI")) {
        const int64_t cnt = *static_cast<const int64_t*>(args[2]);
        const bool f64 = name[name.find("19diag_partial_kernelI") + 22] == 'd';
        need(*static_cast<const void* const*>(args[0]), (f64 ? 8 : 4) * (size_t)cnt, f64 ? 8 : 4, "day", n);
        need(*static_cast<const void* const*>(args[1]), (f64 ? 8 : 4) * (size_t)cnt, f64 ? 8 : 4, "night", n);
        need(*static_cast<double* const*>(args[3]), sizeof(double) * kDiag * grid.x, 8, "partials", n);
    } else if (has(name, "16fold_diag_kernel")) {
        const int world = *static_cast<const int*>(args[1]);
        need(*static_cast<const double* const*>(args[0]), sizeof(double) * kDiag * (size_t)world, 8, "gathered vectors", n);
        need(*static_cast<double* const*>(args[2]), sizeof(double) * kDiag, 8, "diagnostics vector", n);
    } else if (has(name, "11copy_kernel")) {
        const int64_t nvec = *static_cast<const int64_t*>(args[2]);
        need(*static_cast<const void* const*>(args[0]), 16 * (size_t)nvec, 16, "source", n);
        need(*static_cast<void* const*>(args[1]), 16 * (size_t)nvec, 16, "destination", n);
    } else {
        std::lock_guard<std::mutex> lock(g_mu);
        ++g_unchecked[name.substr(0, 48)];
    }
}

}  // namespace stub

extern "C" void mod16_stub_report(FILE* f) {
    fprintf(f, "hip_stub: %ld address ranges checked; live allocations %zu\n", stub::g_checked_ranges, stub::g_blocks.size());
    for (auto& kv : stub::g_launches) fprintf(f, "  launches  %-44s %ld\n", kv.first.c_str(), kv.second);
    for (auto& kv : stub::g_unchecked) fprintf(f, "  launch shape only (no shadow)  %-48s %ld\n", kv.first.c_str(), kv.second);
}
extern "C" size_t mod16_stub_live_allocations(void) { return stub::g_blocks.size(); }

// ------------------------------------------------------------------ the runtime's entry points
extern "C" {
void** __hipRegisterFatBinary(const void*) { static void* handle[1]; return handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* host, char*, const char* device_name, unsigned, void*, void*, void*, void*, int*) {
    stub::g_kernels[host] = device_name;
}
void __hipRegisterVar(void**, void*, char*, char*, int, size_t, int, int) {}

static thread_local struct { dim3 grid, block; size_t shmem; hipStream_t stream; } t_cfg;
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg = {grid, block, shmem, stream};
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
    *grid = t_cfg.grid; *block = t_cfg.block; *shmem = t_cfg.shmem; *stream = t_cfg.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void* func, dim3 grid, dim3 block, void** args, size_t, hipStream_t) {
    auto it = stub::g_kernels.find(func);
    if (it == stub::g_kernels.end()) stub::die("launch of an unregistered kernel %p", func);
    stub::run_shadow(it->second, grid, block, args);
    return hipSuccess;
}

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_t* p, int) {
    memset(p, 0, sizeof *p);
    strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = stub::g_cus;
    return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }

hipError_t hipMalloc(void** p, size_t bytes) { *p = stub::alloc(bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { return stub::release(p); }
hipError_t hipMallocAsync(void** p, size_t bytes, hipStream_t) { return hipMalloc(p, bytes); }
hipError_t hipFreeAsync(void* p, hipStream_t) { return stub::release(p); }
// page-locked host memory is in the device's address space too (the HOST mode's small calls hand it
// to their kernels): a tracked block like any other, always real memory
static int g_fail_host_malloc = 0;      // mod16_stub_fail_host_malloc(k): the next k hipHostMalloc calls fail
extern "C" void mod16_stub_fail_host_malloc(int k) { g_fail_host_malloc = k; }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
    if (g_fail_host_malloc > 0) { --g_fail_host_malloc; *p = nullptr; return hipErrorOutOfMemory; }
    if (bytes >= stub::kRealBelow) stub::die("hipHostMalloc of %zu bytes: the library pins small buffers only", bytes);
    *p = stub::alloc(bytes);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void* p) { return stub::release(p); }
hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) { *dev = host; return hipSuccess; }

static hipError_t copy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
    // the device side of a copy must be a device allocation; host sides are plain memory under ASan
    bool fake = false;
    if (kind == hipMemcpyHostToDevice || kind == hipMemcpyDeviceToDevice)
        if (!stub::inside(dst, bytes, &fake)) stub::die("copy of %zu bytes to device %p: outside every allocation", bytes, dst);
    if (kind == hipMemcpyDeviceToHost || kind == hipMemcpyDeviceToDevice)
        if (!stub::inside(src, bytes, &fake)) stub::die("copy of %zu bytes from device %p: outside every allocation", bytes, src);
    if (!fake) memmove(dst, src, bytes);
    return hipSuccess;
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) { return copy(dst, src, bytes, kind); }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t) { return copy(dst, src, bytes, kind); }
static hipError_t fill(void* dst, int v, size_t bytes) {
    bool fake = false;
    if (!stub::inside(dst, bytes, &fake)) stub::die("memset of %zu bytes at device %p: outside every allocation", bytes, dst);
    if (!fake) memset(dst, v, bytes);
    return hipSuccess;
}
hipError_t hipMemset(void* dst, int v, size_t bytes) { return fill(dst, v, bytes); }
hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t) { return fill(dst, v, bytes); }

// streams, events and graphs: inert handles (heap objects, so leaks and double destroys show)
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(new int(1)); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t* s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t s) { delete reinterpret_cast<int*>(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static thread_local bool t_capturing = false;
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* st) {
    *st = t_capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { t_capturing = true; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { t_capturing = false; *g = reinterpret_cast<hipGraph_t>(new int(2)); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = reinterpret_cast<hipGraphExec_t>(new int(3)); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete reinterpret_cast<int*>(e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { delete reinterpret_cast<int*>(g); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = reinterpret_cast<hipEvent_t>(new int(4)); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<int*>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
}
