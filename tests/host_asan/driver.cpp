// Drives the HOST half of libmod16hip -- built with AddressSanitizer + UndefinedBehaviorSanitizer
// against tests/host_asan/hip_stub.hip -- through the C ABI (include/mod16_hip.h): ragged sizes,
// every form, every raster layout, bad layouts, more than 2^31 pixels, graphs, the HOST-mode tiler
// with its staging threads, the resident calibration problem. "Device" memory is host heap or (large
// rasters) an address-space reservation; kernel launches run their shadows (address arithmetic only).
// Exit code 0 and "host_asan: ok" = clean.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/mod16_hip.h"

extern "C" void mod16_stub_report(FILE* f);
extern "C" size_t mod16_stub_live_allocations(void);
extern "C" void mod16_stub_fail_host_malloc(int k);

static int g_checks = 0;
#define EXPECT(cond)                                                                       \
    do {                                                                                   \
        ++g_checks;                                                                        \
        if (!(cond)) { fprintf(stderr, "host_asan: %s:%d: %s failed\n", __FILE__, __LINE__, #cond); exit(1); } \
    } while (0)
#define OK(call)                                                                           \
    do {                                                                                   \
        ++g_checks;                                                                        \
        int rc_ = (call);                                                                  \
        if (rc_ != MOD16_OK) { fprintf(stderr, "host_asan: %s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, mod16_last_error(ctx)); exit(1); } \
    } while (0)

static void* dmalloc(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess || !p) { fprintf(stderr, "host_asan: no memory for %zu bytes\n", bytes); exit(1); }
    return p;
}

template <typename T> struct TypeOps;
template <> struct TypeOps<double> {
    static constexpr int V = 2;
    static int et(mod16_ctx* c, const uint8_t* cls, const double* const* d, const int64_t* ds, const double* const* p, const int64_t* ps,
                  int64_t n, double* a, double* b, double* const* sep, unsigned f, int w) { return mod16_et_f64(c, cls, d, ds, p, ps, n, a, b, sep, f, w, nullptr); }
    static int tiled(mod16_ctx* c, const mod16_layout* l, const uint8_t* cls, const double* const* d, int64_t n, double* a, double* b, unsigned f, double* dd) { return mod16_et_tiled_f64(c, l, cls, d, n, a, b, f, dd, nullptr); }
    static int form(mod16_ctx* c, const mod16_layout* l, int form, const uint8_t* const* by, const double* const* w, double* const* o, int64_t n, unsigned f) { return mod16_et_form_tiled_f64(c, l, form, by, w, o, 12.0, n, f, nullptr); }
    static int graph(mod16_ctx* c, const mod16_layout* l, const uint8_t* cls, const double* const* d, int64_t n, double* a, double* b, unsigned f, double* dd, mod16_graph** g) { return mod16_graph_et_tiled_f64(c, l, cls, d, n, a, b, f, dd, g); }
    static int synth(mod16_ctx* c, const mod16_layout* l, int64_t n, uint8_t* cls, double* const* d) { return mod16_synth_tiled_f64(c, l, 16, 0, 0, n, cls, d, nullptr); }
    static int diag(mod16_ctx* c, const uint8_t* cls, const double* const* d, const int64_t* ds, int64_t n, double* a, double* b, unsigned f, double* dd) { return mod16_et_diag_f64(c, cls, d, ds, n, a, b, f, dd, nullptr); }
};
template <> struct TypeOps<float> {
    static constexpr int V = 4;
    static int et(mod16_ctx* c, const uint8_t* cls, const float* const* d, const int64_t* ds, const float* const* p, const int64_t* ps,
                  int64_t n, float* a, float* b, float* const* sep, unsigned f, int w) { return mod16_et_f32(c, cls, d, ds, p, ps, n, a, b, sep, f, w, nullptr); }
    static int tiled(mod16_ctx* c, const mod16_layout* l, const uint8_t* cls, const float* const* d, int64_t n, float* a, float* b, unsigned f, double* dd) { return mod16_et_tiled_f32(c, l, cls, d, n, a, b, f, dd, nullptr); }
    static int form(mod16_ctx* c, const mod16_layout* l, int form, const uint8_t* const* by, const float* const* w, float* const* o, int64_t n, unsigned f) { return mod16_et_form_tiled_f32(c, l, form, by, w, o, 12.0, n, f, nullptr); }
    static int graph(mod16_ctx* c, const mod16_layout* l, const uint8_t* cls, const float* const* d, int64_t n, float* a, float* b, unsigned f, double* dd, mod16_graph** g) { return mod16_graph_et_tiled_f32(c, l, cls, d, n, a, b, f, dd, g); }
    static int synth(mod16_ctx* c, const mod16_layout* l, int64_t n, uint8_t* cls, float* const* d) { return mod16_synth_tiled_f32(c, l, 16, 0, 0, n, cls, d, nullptr); }
    static int diag(mod16_ctx* c, const uint8_t* cls, const float* const* d, const int64_t* ds, int64_t n, float* a, float* b, unsigned f, double* dd) { return mod16_et_diag_f32(c, cls, d, ds, n, a, b, f, dd, nullptr); }
};

// A tiled raster of one form: [tile][field][tile pixels] as mod16_amd/raster.py lays it out
template <typename T> struct Raster {
    mod16_layout lay;
    int64_t n, ntiles;
    char* slab;
    std::vector<const T*> wide;
    std::vector<T*> outs;
    std::vector<const uint8_t*> bytes;
    Raster(int64_t n_, int64_t tile, int nw, int nb, int no, int64_t extra_row = 0) : n(n_) {
        ntiles = (n + tile - 1) / tile;
        if (ntiles < 1) ntiles = 1;
        const int64_t wrow = nw * tile + extra_row, orow = no * tile + extra_row, brow = nb * tile + extra_row;
        const size_t wb = (size_t)ntiles * wrow * sizeof(T), ob = (size_t)ntiles * orow * sizeof(T), bb = (size_t)ntiles * brow;
        slab = static_cast<char*>(dmalloc(wb + ob + bb + 4096));
        lay = mod16_layout{tile, wrow, orow, brow};
        for (int k = 0; k < nw; ++k) wide.push_back(reinterpret_cast<const T*>(slab) + k * tile);
        for (int k = 0; k < no; ++k) outs.push_back(reinterpret_cast<T*>(slab + wb) + k * tile);
        for (int k = 0; k < nb; ++k) bytes.push_back(reinterpret_cast<const uint8_t*>(slab + wb + ob) + k * tile);
    }
    ~Raster() { (void)hipFree(slab); }
};

template <typename T>
static void tiled_cases(mod16_ctx* ctx, const char* what) {
    constexpr int V = TypeOps<T>::V;
    double* ddiag = static_cast<double*>(dmalloc(64));
    const int64_t tile = 32768 / (int64_t)sizeof(T);
    // sizes: one piece, ragged ends, a 1200 x 1200 tile (static schedule), the global grid and more than 2^31 pixels
    const int64_t sizes[] = {V, 64 * V, 64 * V + V, tile - V, tile, tile + V, 1200 * 1200, 7 * tile + 5 * V,
                             (int64_t)43200 * 21600, ((int64_t)1 << 31) + 12344};
    for (int64_t n : sizes) {
        Raster<T> r(n, tile, 14, 1, 2);
        OK(TypeOps<T>::synth(ctx, &r.lay, n, const_cast<uint8_t*>(r.bytes[0]), reinterpret_cast<T* const*>(const_cast<T**>(const_cast<const T**>(r.wide.data())))));
        OK(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), n, r.outs[0], r.outs[1], MOD16_MATH_FAST, ddiag));
        OK(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), n, r.outs[0], r.outs[1], MOD16_MATH_FAST | MOD16_DOMAIN_TRUSTED, nullptr));
        if (sizeof(T) == 4) OK(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), n, r.outs[0], r.outs[1], MOD16_MATH_MIXED, ddiag));
        mod16_graph* g = nullptr;
        OK(TypeOps<T>::graph(ctx, &r.lay, r.bytes[0], r.wide.data(), n, r.outs[0], r.outs[1], MOD16_MATH_FAST, ddiag, &g));
        EXPECT(mod16_graph_launch(g, nullptr) == MOD16_OK);
        float ms = 0;
        EXPECT(mod16_time_graph(g, 2, nullptr, &ms) == MOD16_OK);
        EXPECT(mod16_graph_destroy(g) == MOD16_OK);
        OK(mod16_check_status(ctx, nullptr));
    }
    // rows wider than the tile (a padded pitch), other tile sizes
    for (int64_t tl : {tile / 4, tile * 2, tile * 16}) {
        Raster<T> r(5 * tl + 3 * V, tl, 14, 1, 2, 64);
        OK(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), r.n, r.outs[0], r.outs[1], MOD16_MATH_FAST, ddiag));
    }
    // every form on the tiled layout
    for (int form = MOD16_FORM_TOTALS; form <= MOD16_FORM_RAW_TOTAL8_HOURS; ++form) {
        int nw, nb, no;
        EXPECT(mod16_form_shape(form, &nw, &nb, &no) == MOD16_OK);
        for (int64_t n : {(int64_t)(3 * tile + 7 * V), (int64_t)10800 * 43200}) {
            Raster<T> r(n, tile, nw, nb, no);
            OK(TypeOps<T>::form(ctx, &r.lay, form, r.bytes.data(), r.wide.data(), r.outs.data(), n, MOD16_MATH_FAST));
            if (sizeof(T) == 4) OK(TypeOps<T>::form(ctx, &r.lay, form, r.bytes.data(), r.wide.data(), r.outs.data(), n, MOD16_MATH_MIXED));
        }
    }
    int dummy;
    EXPECT(mod16_form_shape(99, &dummy, &dummy, &dummy) == MOD16_ERR_ARG);
    // layouts that must be refused: tile not a power of two / too small, rows narrower than the tile or
    // not a multiple of the vector width, misaligned bases, n not a multiple of the vector width
    {
        Raster<T> r(4 * tile, tile, 14, 1, 2);
        auto bad = [&](mod16_layout lay, const uint8_t* cls, const T* const* w, int64_t n, T* day) {
            return TypeOps<T>::tiled(ctx, &lay, cls, w, n, day, r.outs[1], MOD16_MATH_FAST, ddiag);
        };
        mod16_layout l = r.lay;
        l.tile = tile - 64; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.tile = 64; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.driver_row = tile - V; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.out_row += 1; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.cls_row = 0; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        // what the pipeline kernel's 32-bit scalars cannot hold (round 6): more than 2^30 pieces of 64 vectors,
        // a row between tiles of 2^32 elements -- refused before anything is launched
        EXPECT(bad(r.lay, r.bytes[0], r.wide.data(), (((int64_t)1 << 30) + 1) * 64 * V, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.driver_row = (int64_t)1 << 32; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        l = r.lay; l.out_row = (int64_t)1 << 32; EXPECT(bad(l, r.bytes[0], r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        EXPECT(bad(r.lay, r.bytes[0] + 1, r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        EXPECT(bad(r.lay, r.bytes[0], r.wide.data(), r.n - 1, r.outs[0]) == MOD16_ERR_ARG);
        EXPECT(bad(r.lay, r.bytes[0], r.wide.data(), r.n, r.outs[0] + 1) == MOD16_ERR_ARG);
        std::vector<const T*> w = r.wide;
        w[3] += 1; EXPECT(bad(r.lay, r.bytes[0], w.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        w[3] = nullptr; EXPECT(bad(r.lay, r.bytes[0], w.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        EXPECT(bad(r.lay, nullptr, r.wide.data(), r.n, r.outs[0]) == MOD16_ERR_ARG);
        EXPECT(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), r.n, r.outs[0], r.outs[1], MOD16_MATH_EXACT, ddiag) == MOD16_ERR_ARG);
        EXPECT(TypeOps<T>::tiled(ctx, &r.lay, r.bytes[0], r.wide.data(), 0, r.outs[0], r.outs[1], MOD16_MATH_FAST, ddiag) == MOD16_OK);
    }
    (void)hipFree(ddiag);
    printf("host_asan: tiled rasters, %s: done\n", what);
}

// plain device arrays: aligned slab (pitched), scattered, misaligned (scalar kernels), scalars, ragged n
template <typename T>
static void plain_device_cases(mod16_ctx* ctx, const char* what) {
    constexpr int V = TypeOps<T>::V;
    double* ddiag = static_cast<double*>(dmalloc(64));
    for (int64_t n : {(int64_t)1, (int64_t)V - 1, (int64_t)64 * V + 1, (int64_t)1200 * 1200, (int64_t)1200 * 1200 + 3, (int64_t)2700 * 43200}) {
        const size_t per = ((size_t)n * sizeof(T) + 4095) / 4096 * 4096 + 33 * 1024;
        char* slab = static_cast<char*>(dmalloc(16 * per + n + 4096));
        const T* drv[14];
        int64_t ds[14];
        for (int k = 0; k < 14; ++k) { drv[k] = reinterpret_cast<const T*>(slab + k * per); ds[k] = 1; }
        T* day = reinterpret_cast<T*>(slab + 14 * per);
        T* night = reinterpret_cast<T*>(slab + 15 * per);
        const uint8_t* cls = reinterpret_cast<const uint8_t*>(slab + 16 * per);
        OK(TypeOps<T>::diag(ctx, cls, drv, ds, n, day, night, MOD16_MATH_FAST, ddiag));
        OK(TypeOps<T>::et(ctx, cls, drv, ds, nullptr, nullptr, n, day, night, nullptr, MOD16_MATH_EXACT, MOD16_DEVICE));
        // a broadcast scalar among the drivers, and a misaligned one: the plain kernels
        T* scalar = static_cast<T*>(dmalloc(sizeof(T)));
        const T* d2[14];
        int64_t s2[14];
        for (int k = 0; k < 14; ++k) { d2[k] = drv[k]; s2[k] = 1; }
        d2[7] = scalar; s2[7] = 0;
        OK(TypeOps<T>::diag(ctx, cls, d2, s2, n, day, night, MOD16_MATH_FAST, ddiag));
        if (n > 8) {
            d2[7] = drv[7] + 1; s2[7] = 1;
            OK(TypeOps<T>::et(ctx, cls, d2, s2, nullptr, nullptr, n - 1, day, night, nullptr, MOD16_MATH_FAST, MOD16_DEVICE));
        }
        (void)hipFree(scalar);
        // components and per-pixel parameter arrays
        std::vector<T*> sep(6);
        std::vector<const T*> par(11);
        std::vector<int64_t> ps(11, 1);
        char* extra = static_cast<char*>(dmalloc(17 * per));
        for (int k = 0; k < 6; ++k) sep[k] = reinterpret_cast<T*>(extra + k * per);
        for (int k = 0; k < 11; ++k) par[k] = reinterpret_cast<const T*>(extra + (6 + k) * per);
        OK(TypeOps<T>::et(ctx, cls, drv, ds, nullptr, nullptr, n, day, night, sep.data(), MOD16_MATH_FAST, MOD16_DEVICE));
        OK(TypeOps<T>::et(ctx, cls, drv, ds, nullptr, nullptr, n, nullptr, nullptr, sep.data(), MOD16_MATH_FAST, MOD16_DEVICE));
        OK(TypeOps<T>::et(ctx, nullptr, drv, ds, par.data(), ps.data(), n, day, night, sep.data(), MOD16_MATH_FAST, MOD16_DEVICE));
        OK(mod16_check_status(ctx, nullptr));
        (void)hipFree(extra);
        (void)hipFree(slab);
    }
    EXPECT(TypeOps<T>::et(ctx, nullptr, nullptr, nullptr, nullptr, nullptr, 4, nullptr, nullptr, nullptr, 0, MOD16_DEVICE) == MOD16_ERR_ARG);
    (void)hipFree(ddiag);
    printf("host_asan: plain device arrays, %s: done\n", what);
}

// HOST mode: host arrays staged through the slabs by the library's own threads
template <typename T>
static void host_cases(mod16_ctx* ctx, const char* what) {
    const int64_t tile = mod16_host_tile_pixels();
    for (int64_t n : {(int64_t)5, (int64_t)tile + 12345, (int64_t)3 * tile}) {
        std::vector<std::vector<T>> drv(14, std::vector<T>(n, T(1)));
        std::vector<uint8_t> cls(n, 1);
        std::vector<T> day(n), night(n);
        const T* dp[14];
        int64_t ds[14];
        for (int k = 0; k < 14; ++k) { dp[k] = drv[k].data(); ds[k] = 1; }
        OK(TypeOps<T>::et(ctx, cls.data(), dp, ds, nullptr, nullptr, n, day.data(), night.data(), nullptr, MOD16_MATH_FAST, MOD16_HOST));
        T one = T(300);
        dp[5] = &one; ds[5] = 0;                 // a broadcast scalar
        std::vector<double> tile_diag(8 * ((n + tile - 1) / tile), -1.0);
        if (sizeof(T) == 8)
            OK(mod16_et_hdiag_f64(ctx, cls.data(), reinterpret_cast<const double* const*>(dp), ds, nullptr, nullptr, n,
                                  reinterpret_cast<double*>(day.data()), reinterpret_cast<double*>(night.data()), MOD16_MATH_FAST, tile_diag.data()));
        else
            OK(mod16_et_hdiag_f32(ctx, cls.data(), reinterpret_cast<const float* const*>(dp), ds, nullptr, nullptr, n,
                                  reinterpret_cast<float*>(day.data()), reinterpret_cast<float*>(night.data()), MOD16_MATH_FAST, tile_diag.data()));
        double folded[8];
        EXPECT(mod16_fold_diag_host(tile_diag.data(), (int64_t)tile_diag.size() / 8, folded) == MOD16_OK);
    }
    // small calls: the copy-free path (page-locked buffer read and written by the kernel) up to its
    // limit, the staged path one pixel above it; parameters as arrays / scalars, every output set,
    // a class code the reference would refuse
    for (int64_t n : {(int64_t)1, (int64_t)365, (int64_t)1025, (int64_t)65535, (int64_t)65536, (int64_t)65537}) {
        std::vector<std::vector<T>> drv(14, std::vector<T>(n, T(1))), par(11, std::vector<T>(n, T(2)));
        std::vector<std::vector<T>> outs(6, std::vector<T>(n));
        std::vector<uint8_t> cls(n, 1);
        std::vector<T> day(n), night(n);
        const T *dp[14], *pp[11];
        int64_t ds[14], ps[11];
        T one = T(300);
        for (int k = 0; k < 14; ++k) { dp[k] = k % 3 ? drv[k].data() : &one; ds[k] = k % 3 ? 1 : 0; }
        for (int k = 0; k < 11; ++k) { pp[k] = k % 2 ? par[k].data() : &one; ps[k] = k % 2 ? 1 : 0; }
        T* sep[6];
        for (int k = 0; k < 6; ++k) sep[k] = outs[k].data();
        OK(TypeOps<T>::et(ctx, nullptr, dp, ds, pp, ps, n, day.data(), night.data(), nullptr, MOD16_MATH_FAST, MOD16_HOST));
        OK(TypeOps<T>::et(ctx, nullptr, dp, ds, pp, ps, n, nullptr, nullptr, sep, MOD16_MATH_EXACT, MOD16_HOST));
        OK(TypeOps<T>::et(ctx, cls.data(), dp, ds, nullptr, nullptr, n, day.data(), night.data(), sep, MOD16_MATH_FAST, MOD16_HOST));
        if (n <= 65536) {    // (above it the kernel reports the code; the stand-in's kernels report nothing)
            cls[n - 1] = 13;
            EXPECT(TypeOps<T>::et(ctx, cls.data(), dp, ds, nullptr, nullptr, n, day.data(), night.data(), nullptr, MOD16_MATH_FAST, MOD16_HOST) == MOD16_ERR_CLASS_RANGE);
        }
    }
    // raw drivers through the same staging (threads and slots since round 5), hours dense / scalar / absent
    for (int64_t n : {(int64_t)7, (int64_t)2 * tile + 4321}) {
        std::vector<std::vector<T>> raw(14, std::vector<T>(n, T(280)));
        std::vector<uint8_t> cls(n, 1), fpar(n, 50), lai(n, 20);
        std::vector<T> day(n), night(n), total(n), hours(n, T(12));
        const T* rp[14];
        int64_t rs[14];
        for (int k = 0; k < 14; ++k) { rp[k] = raw[k].data(); rs[k] = 1; }
        T elev = T(350);
        rp[13] = &elev; rs[13] = 0;
        if (sizeof(T) == 8) {
            auto R = reinterpret_cast<const double* const*>(rp);
            OK(mod16_et_raw_f64(ctx, cls.data(), R, rs, fpar.data(), lai.data(), reinterpret_cast<const double*>(hours.data()), 1, n,
                                reinterpret_cast<double*>(day.data()), reinterpret_cast<double*>(night.data()), reinterpret_cast<double*>(total.data()), MOD16_MATH_FAST, MOD16_HOST, nullptr));
            OK(mod16_et_raw_f64(ctx, cls.data(), R, rs, fpar.data(), lai.data(), reinterpret_cast<const double*>(hours.data()), 0, n,
                                reinterpret_cast<double*>(day.data()), reinterpret_cast<double*>(night.data()), reinterpret_cast<double*>(total.data()), MOD16_MATH_FAST, MOD16_HOST, nullptr));
            OK(mod16_et_raw_f64(ctx, cls.data(), R, rs, fpar.data(), lai.data(), nullptr, 0, n,
                                reinterpret_cast<double*>(day.data()), reinterpret_cast<double*>(night.data()), nullptr, MOD16_MATH_EXACT, MOD16_HOST, nullptr));
        } else {
            auto R = reinterpret_cast<const float* const*>(rp);
            OK(mod16_et_raw_f32(ctx, cls.data(), R, rs, fpar.data(), lai.data(), reinterpret_cast<const float*>(hours.data()), 1, n,
                                reinterpret_cast<float*>(day.data()), reinterpret_cast<float*>(night.data()), reinterpret_cast<float*>(total.data()), MOD16_MATH_MIXED, MOD16_HOST, nullptr));
            OK(mod16_et_raw_f32(ctx, cls.data(), R, rs, fpar.data(), lai.data(), nullptr, 0, n,
                                reinterpret_cast<float*>(day.data()), reinterpret_cast<float*>(night.data()), nullptr, MOD16_MATH_FAST, MOD16_HOST, nullptr));
        }
    }
    double x[8];
    EXPECT(mod16_fold_diag_host(nullptr, 1, x) == MOD16_ERR_ARG);
    EXPECT(mod16_fold_diag_host(x, 0, x) == MOD16_ERR_ARG);
    printf("host_asan: HOST mode, %s: done\n", what);
}

int main(int argc, char** argv) {
    mod16_ctx* ctx = nullptr;
    EXPECT(mod16_create(3, &ctx) == MOD16_ERR_NO_DEVICE);
    EXPECT(mod16_create(0, &ctx) == MOD16_OK && ctx);
    if (argc > 1 && !strcmp(argv[1], "--fault")) {
        // the harness must SEE a fault: a raster whose storage is one tile short of what the call says
        // (the library cannot know; on a GPU this launch would read and write past the allocation)
        double lut[13 * 11] = {};
        OK(mod16_set_bplut_f64(ctx, lut));
        const int64_t tile = 4096;
        Raster<double> r(3 * tile, tile, 14, 1, 2);
        fprintf(stderr, "host_asan: launching 4 tiles on a raster of 3\n");
        (void)mod16_et_tiled_f64(ctx, &r.lay, r.bytes[0], r.wide.data(), 4 * tile, r.outs[0], r.outs[1], MOD16_MATH_FAST, nullptr, nullptr);
        fprintf(stderr, "host_asan: the fault went unnoticed\n");
        return 0;
    }
    {   // a class raster before a BPLUT: refused
        mod16_layout lay{4096, 14 * 4096, 2 * 4096, 4096};
        const double* w[14] = {};
        double o[2];
        uint8_t c[2] = {};
        EXPECT(mod16_et_tiled_f64(ctx, &lay, c, w, 4096, o, o, 0, nullptr, nullptr) == MOD16_ERR_NO_BPLUT);
    }
    double lut[13 * 11];
    for (int i = 0; i < 13 * 11; ++i) lut[i] = 1.0 + i;
    OK(mod16_set_bplut_f64(ctx, lut));
    tiled_cases<double>(ctx, "float64");
    tiled_cases<float>(ctx, "float32");
    plain_device_cases<double>(ctx, "float64");
    plain_device_cases<float>(ctx, "float32");
    host_cases<double>(ctx, "float64");
    host_cases<float>(ctx, "float32");
    {   // a graph and its context, destroyed in either order: once the context is gone a replay is
        // refused (its kernels would read the freed tables), the graph itself can still be freed
        mod16_ctx* c2 = nullptr;
        EXPECT(mod16_create(0, &c2) == MOD16_OK && c2);
        OK(mod16_set_bplut_f64(c2, lut));
        const int64_t tile = 4096;
        Raster<double> r(3 * tile, tile, 14, 1, 2);
        double* dd = static_cast<double*>(dmalloc(64));
        mod16_graph *g1 = nullptr, *g2 = nullptr;
        EXPECT(mod16_graph_et_tiled_f64(c2, &r.lay, r.bytes[0], r.wide.data(), r.n, r.outs[0], r.outs[1], MOD16_MATH_FAST, dd, &g1) == MOD16_OK);
        EXPECT(mod16_graph_et_tiled_f64(c2, &r.lay, r.bytes[0], r.wide.data(), r.n, r.outs[0], r.outs[1], MOD16_MATH_FAST, dd, &g2) == MOD16_OK);
        EXPECT(mod16_graph_launch(g1, nullptr) == MOD16_OK);
        EXPECT(mod16_graph_destroy(g1) == MOD16_OK);          // graph first: leaves the context's list
        EXPECT(mod16_destroy(c2) == MOD16_OK);                // context first: g2 is dead ...
        float ms = 0;
        EXPECT(mod16_graph_launch(g2, nullptr) == MOD16_ERR_ARG);
        EXPECT(mod16_time_graph(g2, 1, nullptr, &ms) == MOD16_ERR_ARG);
        EXPECT(mod16_graph_destroy(g2) == MOD16_OK);          // ... and still freed
        (void)hipFree(dd);
        printf("host_asan: graph lifetime: done\n");
    }
    {   // the stand-alone reduction, the rank-order fold, the copy probe
        const int64_t n = 1200 * 1200 + 1;
        double* day = static_cast<double*>(dmalloc(8 * n));
        double* gathered = static_cast<double*>(dmalloc(8 * 64));
        double host[8];
        OK(mod16_reduce_diag_f64(ctx, day, day, n, host, nullptr, nullptr));
        OK(mod16_fold_diag(ctx, gathered, 8, gathered, nullptr));
        EXPECT(mod16_fold_diag(ctx, gathered, 0, gathered, nullptr) == MOD16_ERR_ARG);
        float gbps = 0;
        OK(mod16_measure_copy(ctx, 1 << 20, 1, &gbps));
        (void)hipFree(day);
        (void)hipFree(gathered);
    }
    {   // the resident calibration problem: bind, objective, rows, destroy
        const int64_t n = 1000, ndraw = 7;
        std::vector<std::vector<double>> drv(14, std::vector<double>(n, 280.0));
        std::vector<double> obs(n, 10.0), par(ndraw * 11, 1.0), sse(ndraw), cnt(ndraw), rows(ndraw * n);
        const double* dp[14];
        int64_t ds[14];
        for (int k = 0; k < 14; ++k) { dp[k] = drv[k].data(); ds[k] = 1; }
        ds[7] = 0;
        mod16_batch* b = nullptr;
        OK(mod16_static_batch_bind_f64(ctx, dp, ds, n, obs.data(), nullptr, 16, MOD16_MATH_FAST, MOD16_HOST, &b));
        EXPECT(mod16_static_batch_objective(b, par.data(), ndraw, sse.data(), cnt.data()) == MOD16_OK);
        EXPECT(mod16_static_batch_rows(b, par.data(), ndraw, nullptr, nullptr, rows.data()) == MOD16_OK);
        EXPECT(mod16_static_batch_objective(b, par.data(), 17, sse.data(), cnt.data()) != MOD16_OK);   // more than max_draws
        EXPECT(mod16_static_batch_destroy(b) == MOD16_OK);
        OK(mod16_et_static_batch_f64(ctx, dp, ds, n, par.data(), ndraw, nullptr, nullptr, nullptr, obs.data(), nullptr, sse.data(),
                                     cnt.data(), MOD16_MATH_EXACT, MOD16_HOST, nullptr));
    }
    EXPECT(mod16_destroy(ctx) == MOD16_OK);
    {   // no page-locked memory for the small calls' buffer: the call is staged instead (and the
        // context stops asking), every entry point that has the small path
        mod16_ctx* c2 = nullptr;
        EXPECT(mod16_create(0, &c2) == MOD16_OK && c2);
        ctx = c2;
        OK(mod16_set_bplut_f64(ctx, lut));
        const int64_t n = 9;
        std::vector<std::vector<double>> drv(14, std::vector<double>(n, 280.0)), par(11, std::vector<double>(n, 2.0));
        std::vector<uint8_t> cls(n, 1), fpar(n, 50), lai(n, 20);
        std::vector<double> day(n), night(n);
        const double *dp[14], *pp[11];
        int64_t ds[14], ps[11];
        for (int k = 0; k < 14; ++k) { dp[k] = drv[k].data(); ds[k] = 1; }
        for (int k = 0; k < 11; ++k) { pp[k] = par[k].data(); ps[k] = 1; }
        mod16_stub_fail_host_malloc(1000);
        OK(mod16_et_f64(ctx, cls.data(), dp, ds, nullptr, nullptr, n, day.data(), night.data(), nullptr, MOD16_MATH_FAST, MOD16_HOST, nullptr));
        OK(mod16_et_static_f64(ctx, dp, ds, pp, ps, nullptr, nullptr, n, day.data(), night.data(), 1e-7, MOD16_HOST, nullptr));
        OK(mod16_et_raw_f64(ctx, cls.data(), dp, ds, fpar.data(), lai.data(), nullptr, 0, n, day.data(), night.data(), nullptr, MOD16_MATH_FAST, MOD16_HOST, nullptr));
        const double* in[13] = {dp[5], dp[9]};
        int64_t is[13] = {1, 1};
        double* outs[2] = {day.data(), nullptr};
        OK(mod16_method_f64(ctx, MOD16_M_RHUMIDITY, in, is, nullptr, nullptr, n, outs, 1.26, 1e-7, MOD16_HOST, nullptr));
        mod16_stub_fail_host_malloc(0);
        EXPECT(mod16_destroy(ctx) == MOD16_OK);
    }
    mod16_stub_report(stdout);
    // everything the library allocated is gone with its context
    EXPECT(mod16_stub_live_allocations() == 0);
    printf("host_asan: ok (%d checks)\n", g_checks);
    return 0;
}
