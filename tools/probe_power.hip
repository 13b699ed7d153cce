// Power and shader clock of the device under single-instruction loads (gfx950): is a kernel
// that keeps the vector pipe busy bound by the pipe or by the package power cap, and what does
// an instruction of each kind cost in energy? Every kernel runs ~1 s on the whole chip at the
// production kernel's occupancy (2 waves per SIMD); a host thread samples the hwmon files of the
// card (power1_input, freq1_input -- read-only) meanwhile.
// hipcc --offload-arch=gfx950 -O3 tools/probe_power.hip -o tools/bin/probe_power -lpthread
#include <hip/hip_runtime.h>
#include <glob.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cctype>
#include <climits>
#include <string>
#include <thread>
#include <vector>

#define REP8(x) x x x x x x x x
// four independent chains per statement, operands that change from instruction to instruction
#define KERNEL(NAME, ASM, ...)                                                                 \
    __global__ void __launch_bounds__(256) k_##NAME(double* out, int iters) {                  \
        double r0 = 1.0 + threadIdx.x * 1e-3, r1 = 1.5 + threadIdx.x * 1e-4, r2 = 0.75 + threadIdx.x * 1e-5,   \
               r3 = 1.25 - threadIdx.x * 1e-4;                                                 \
        double b = 1.0000001 + threadIdx.x * 1e-9, c = 1e-7 * (threadIdx.x + 1);               \
        unsigned i0 = threadIdx.x * 2654435761u, i1 = i0 ^ 0x9e3779b9u, i2 = i0 * 3u, i3 = ~i0; \
        unsigned k = 0x85ebca6bu + threadIdx.x;                                                \
        for (int i = 0; i < iters; ++i) {                                                      \
            REP8(asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(i0), "+v"(i1), \
                                    "+v"(i2), "+v"(i3)                                          \
                              : "v"(b), "v"(c), "v"(k) : __VA_ARGS__);)                         \
        }                                                                                      \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + (double)(i0 + i1 + i2 + i3);  \
    }
// %0-%3 doubles, %4-%7 unsigned, %8 b, %9 c, %10 k
KERNEL(fma_f64, "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9", "memory")
KERNEL(mul_f64, "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8", "memory")
KERNEL(add_f64, "v_add_f64 %0, %0, %9\n v_add_f64 %1, %1, %9\n v_add_f64 %2, %2, %9\n v_add_f64 %3, %3, %9", "memory")
KERNEL(max_f64, "v_max_f64 %0, |%1|, |%8|\n v_max_f64 %1, |%2|, |%9|\n v_max_f64 %2, |%3|, |%8|\n v_max_f64 %3, |%0|, |%9|", "memory")
KERNEL(rcp_f64, "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3", "memory")
KERNEL(cmp_f64, "v_cmp_gt_f64 vcc, %0, %1\n v_cmp_gt_f64 vcc, %1, %2\n v_cmp_gt_f64 vcc, %2, %3\n v_cmp_gt_f64 vcc, %3, %0", "vcc", "memory")
KERNEL(pk_fma_f32, "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9", "memory")
KERNEL(fma_f32, "v_fma_f32 %4, %4, %10, %5\n v_fma_f32 %5, %5, %10, %6\n v_fma_f32 %6, %6, %10, %7\n v_fma_f32 %7, %7, %10, %4", "memory")
KERNEL(cndmask, "v_cndmask_b32 %4, %5, %10, vcc\n v_cndmask_b32 %5, %6, %10, vcc\n v_cndmask_b32 %6, %7, %10, vcc\n v_cndmask_b32 %7, %4, %10, vcc", "memory")
KERNEL(mov_b32, "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %10", "memory")
KERNEL(add_u32, "v_add_u32 %4, %4, %10\n v_add_u32 %5, %5, %10\n v_add_u32 %6, %6, %10\n v_add_u32 %7, %7, %10", "memory")
KERNEL(min3_u32, "v_min3_u32 %4, %5, %6, %10\n v_min3_u32 %5, %6, %7, %10\n v_min3_u32 %6, %7, %4, %10\n v_min3_u32 %7, %4, %5, %10", "memory")
KERNEL(lshl_add_u32, "v_lshl_add_u32 %4, %5, 1, %10\n v_lshl_add_u32 %5, %6, 1, %10\n v_lshl_add_u32 %6, %7, 1, %10\n v_lshl_add_u32 %7, %4, 1, %10", "memory")
KERNEL(max3_f32, "v_max3_f32 %4, |%5|, |%6|, |%10|\n v_max3_f32 %5, |%6|, |%7|, |%10|\n v_max3_f32 %6, |%7|, |%4|, |%10|\n v_max3_f32 %7, |%4|, |%5|, |%10|", "memory")
KERNEL(readlane, "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %5, 5\n v_readlane_b32 s22, %6, 7\n v_readlane_b32 s23, %7, 9", "s20", "s21", "s22", "s23", "memory")
KERNEL(s_nop, "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0", "memory")
KERNEL(s_mov, "s_mov_b32 s20, s2\n s_mov_b32 s21, s3\n s_mov_b32 s22, s2\n s_mov_b32 s23, s3", "s20", "s21", "s22", "s23", "memory")

// the production mix in miniature: 2 fma + 2.4 mul + 0.6 add per ... (approximate): 4 fma, 5 mul, 1 add, 2 cndmask, 1 cmp
KERNEL(mix_f64, "v_fma_f64 %0, %0, %8, %9\n v_mul_f64 %1, %1, %8\n v_fma_f64 %2, %2, %8, %9\n v_mul_f64 %3, %3, %8\n"
                "v_cndmask_b32 %4, %5, %10, vcc\n v_mul_f64 %1, %1, %8\n v_add_f64 %0, %0, %9\n v_cmp_gt_f64 vcc, %2, %3", "vcc", "memory")

typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_copy(const f4* __restrict__ src, f4* __restrict__ dst, long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

// variants of the stream: cache policy, read only / write only, and the production kernel's
// path (LDS-DMA into the wave's slot, ds_read_b128, store) -- what does a byte cost on each?
__global__ void __launch_bounds__(256) k_copy_plain(const f4* __restrict__ src, f4* __restrict__ dst, long n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_read_nt(const f4* __restrict__ src, f4* __restrict__ dst, long n, int reps) {
    f4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
            acc += __builtin_nontemporal_load(src + i);
    if (acc.x == 1.2345f) dst[threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_write_nt(const f4* __restrict__ src, f4* __restrict__ dst, long n, int reps) {
    const f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256)
            __builtin_nontemporal_store(v, dst + i);
}
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <bool DMA>
__global__ void __launch_bounds__(256) k_copy_8(const f4* __restrict__ src, f4* __restrict__ dst, long n, int reps) {
    // a wave moves 8 KB per turn: 8 x (64 lanes x 16 B), like the 14 + 2 of the production kernel
    __shared__ __attribute__((aligned(16))) char stage[4 * 8192];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* ws = stage + wave * 8192;
    const long nchunk = n / 512;                      // chunks of 8 KB
    const long nw = (long)gridDim.x * 4;
    for (int r = 0; r < reps; ++r)
        for (long c = blockIdx.x * 4L + wave; c < nchunk; c += nw) {
            const f4* s0 = src + c * 512 + lane;
            f4 v[8];
            if constexpr (DMA) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    __builtin_amdgcn_global_load_lds((gptr_t)(s0 + k * 64), (lptr_t)(ws + k * 1024), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const f4*>(ws + k * 1024 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(s0 + k * 64);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(v[k], dst + c * 512 + k * 64 + lane);
        }
}

static double g_seconds = 1.2;
struct Sampler {
    std::string power, freq;
    std::atomic<bool> run{false};
    std::vector<double> pw, fq;
    std::thread th;
    static double read(const std::string& p) {
        FILE* f = fopen(p.c_str(), "r");
        if (!f) return -1;
        double v = -1;
        if (fscanf(f, "%lf", &v) != 1) v = -1;
        fclose(f);
        return v;
    }
    Sampler() {
        // the card of THIS process's device, by PCI address (a box shows the hwmon of all 8)
        char bus[64] = "";
        if (hipDeviceGetPCIBusId(bus, sizeof bus, 0) != hipSuccess) return;
        for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
        glob_t g;
        if (glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input", 0, nullptr, &g) == 0) {
            for (size_t i = 0; i < g.gl_pathc; ++i) {
                std::string p = g.gl_pathv[i];
                std::string dev = p.substr(0, p.find("/hwmon"));
                char real[4096];
                if (!realpath(dev.c_str(), real)) continue;
                if (std::string(real).find(bus) == std::string::npos) continue;
                power = p;
                freq = p.substr(0, p.rfind('/')) + "/freq1_input";
            }
        }
        globfree(&g);
    }
    void start() {
        pw.clear(); fq.clear(); run = true;
        th = std::thread([this] {
            std::this_thread::sleep_for(std::chrono::milliseconds((int)(g_seconds * 600)));   // the hwmon power is a slow average: sample the last 40 % only
            while (run) {
                pw.push_back(read(power) * 1e-6);
                fq.push_back(read(freq) * 1e-6);
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        });
    }
    void stop() { run = false; th.join(); }
    static double mean(const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return v.empty() ? -1 : s / v.size(); }
};

template <typename K> void bench(const char* name, K k, double* out, Sampler& smp, int per_stmt) {
    const int blocks = 512;          // 2 blocks of 4 waves per CU: 2 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 1000); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 20000); hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    int iters = (int)(20000 * (g_seconds * 1000.0) / ms);
    smp.start();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(e1);
    hipEventSynchronize(e1);
    smp.stop();
    hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)iters * 8 * per_stmt * blocks * 4;        // wave-instructions
    const double per_simd_s = instr / (256.0 * 4) / (ms * 1e-3);
    const double mhz = Sampler::mean(smp.fq), w = Sampler::mean(smp.pw);
    printf("{\"kernel\": \"%s\", \"ms\": %.1f, \"sclk_mhz\": %.0f, \"power_w\": %.0f, \"cycles_per_instr\": %.2f, "
           "\"wave_instr_per_s\": %.4g, \"samples\": %zu}\n",
           name, ms, mhz, w, mhz * 1e6 / per_simd_s, instr / (ms * 1e-3), smp.fq.size());
    fflush(stdout);
}

int main(int argc, char** argv) {
    if (argc > 1) g_seconds = atof(argv[1]);
    const char* only = argc > 2 ? argv[2] : nullptr;
    double* out; hipMalloc(&out, 512 * 256 * sizeof(double));
    Sampler smp;
    if (smp.power.empty()) { fprintf(stderr, "no hwmon power1_input\n"); return 1; }
    fprintf(stderr, "sampling %s\n", smp.power.c_str());
#define RUN(NAME) if (!only || strstr(only, #NAME)) bench(#NAME, k_##NAME, out, smp, 4)
    RUN(s_nop); RUN(s_mov); RUN(mov_b32); RUN(add_u32); RUN(cndmask); RUN(min3_u32); RUN(lshl_add_u32); RUN(max3_f32);
    RUN(readlane); RUN(fma_f32); RUN(pk_fma_f32); RUN(cmp_f64); RUN(max_f64); RUN(add_f64); RUN(mul_f64); RUN(fma_f64);
    RUN(rcp_f64); if (!only || strstr(only, "mix_f64")) bench("mix_f64", k_mix_f64, out, smp, 8);
    {   // streaming variants, 2 x 4 GiB
        const long n = (4L << 30) / 16;
        f4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
        hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto stream = [&](const char* name, auto k, int blocks, double bytes_per_rep) {
            if (only && !strstr(only, name)) return;
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, b, n, 2); hipDeviceSynchronize();
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, b, n, 20); hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int reps = (int)(20 * g_seconds * 1000.0 / ms) + 1;
            smp.start();
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, a, b, n, reps); hipEventRecord(e1);
            hipEventSynchronize(e1);
            smp.stop();
            hipEventElapsedTime(&ms, e0, e1);
            printf("{\"kernel\": \"%s\", \"ms\": %.1f, \"sclk_mhz\": %.0f, \"power_w\": %.0f, \"GBps\": %.0f, \"samples\": %zu}\n",
                   name, ms, Sampler::mean(smp.fq), Sampler::mean(smp.pw), reps * bytes_per_rep / (ms * 1e-3) * 1e-9, smp.fq.size());
            fflush(stdout);
        };
        stream("copy_nt", k_copy, 4096, 2.0 * n * 16);
        stream("copy_plain", k_copy_plain, 4096, 2.0 * n * 16);
        stream("read_nt", k_read_nt, 4096, 1.0 * n * 16);
        stream("write_nt", k_write_nt, 4096, 1.0 * n * 16);
        stream("copy8_direct", k_copy_8<false>, 512, 2.0 * n * 16);
        stream("copy8_ldsdma", k_copy_8<true>, 512, 2.0 * n * 16);
    }
    return 0;
}
