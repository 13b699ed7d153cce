// libmod16hip.so -- the context: version, errors, create / destroy, the parameter table, status, page-locked memory
#include "internal.hpp"

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
    {   // graphs captured with this context can be destroyed, but not replayed any more
        std::lock_guard<std::mutex> lock(graph_registry_mu());
        for (mod16_graph* g : ctx->graphs) g->ctx = nullptr;
        ctx->graphs.clear();
    }
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


extern "C" int mod16_check_status(mod16_ctx* ctx, void* stream) {
    MOD16_LOCK(ctx);
    if (!ctx) return MOD16_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return read_status(ctx, static_cast<hipStream_t>(stream));
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
