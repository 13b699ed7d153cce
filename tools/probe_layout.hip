// HBM bandwidth of the fused ET kernel's traffic (14 x 16 B/lane reads + 2 x 16 B/lane
// writes per vector of 2 float64 pixels) as a function of how the 16 fields are LAID OUT
// in HBM -- the question behind DESIGN.md section 6: the 14-read + 2-write mix over 16
// separate arrays reaches 5.7 TB/s where 14 reads alone reach 6.5.
//
//   soa            16 separate arrays (what the reference passes), one-shot launch
//   blocked T      fields interleaved in tiles of T vectors: [tile][field][T x 16 B];
//                  T = 64 makes every wave read ONE contiguous 14 KiB block per step
//   out-inside     the two outputs live in the same tile as the inputs ([tile][16 fields])
//   out-separate   the outputs in their own [tile][2][T] buffer
//
// hipcc --offload-arch=gfx950 -O3 tools/probe_layout.hip -o tools/bin/probe_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double d2 __attribute__((ext_vector_type(2)));
struct Args { const d2* in; d2* out; long nvec; long pitch; };   // pitch: SoA distance between arrays (vectors)

// MODE 0: SoA. MODE 1: blocked, outputs inside the tile (16 fields). MODE 2: blocked,
// inputs [tile][14][T], outputs [tile][2][T] in `out`.
template <int MODE, int T, int K, int W>
__global__ void __launch_bounds__(256) oneshot(Args a) {
    const long v = (long)blockIdx.x * 256 + threadIdx.x;
    if (v >= a.nvec) return;
    d2 s = {0.0, 0.0};
    const long tile = v / T, r = v % T;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        long idx;
        if (MODE == 0) idx = k * a.pitch + v;
        else if (MODE == 1) idx = (tile * 16 + k) * T + r;
        else idx = (tile * K + k) * T + r;
        s += __builtin_nontemporal_load(&a.in[idx]);
    }
#pragma unroll
    for (int w = 0; w < W; ++w) {
        long idx;
        d2* base = a.out;
        if (MODE == 0) idx = w * a.pitch + v;
        else if (MODE == 1) { idx = (tile * 16 + 14 + w) * T + r; base = const_cast<d2*>(a.in); }
        else idx = (tile * W + w) * T + r;
        __builtin_nontemporal_store(s + (double)w, &base[idx]);
    }
    if (W == 0 && s[0] == 1.2345e300) a.out[v] = s;
}

// persistent form: every wave walks runs of R consecutive 64-vector pieces claimed from a
// ticket counter (the production kernel's schedule), plain register loads
template <int MODE, int T, int K, int W, int R>
__global__ void __launch_bounds__(256) persistent(Args a, unsigned long long* counter) {
    const int lane = threadIdx.x & 63;
    const long npiece = a.nvec / 64;
    const long nwaves = (long)gridDim.x * 4;
    long run = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    while (run * R < npiece) {
        unsigned long long t = 0;
        if (lane == 0) t = atomicAdd(counter, 1ull);
        for (int i = 0; i < R; ++i) {
            const long v = (run * R + i) * 64 + lane;
            if (v >= a.nvec) break;
            d2 s = {0.0, 0.0};
            const long tile = v / T, r = v % T;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                long idx;
                if (MODE == 0) idx = k * a.pitch + v;
                else if (MODE == 1) idx = (tile * 16 + k) * T + r;
                else idx = (tile * K + k) * T + r;
                s += __builtin_nontemporal_load(&a.in[idx]);
            }
#pragma unroll
            for (int w = 0; w < W; ++w) {
                long idx;
                d2* base = a.out;
                if (MODE == 0) idx = w * a.pitch + v;
                else if (MODE == 1) { idx = (tile * 16 + 14 + w) * T + r; base = const_cast<d2*>(a.in); }
                else idx = (tile * W + w) * T + r;
                __builtin_nontemporal_store(s + (double)w, &base[idx]);
            }
        }
        t = __shfl(t, 0, 64);
        run = nwaves + (long)t;
    }
}

static hipEvent_t e0, e1;
template <typename F> static float best_of(F launch, int reps = 4) {
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best;
}
static void report(const char* name, float ms, long nvec, int K, int W) {
    const double bytes = (double)nvec * 16.0 * (K + W);
    printf("%-44s %8.3f ms  %8.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
    fflush(stdout);
}

template <int MODE, int T, int K, int W> static void run_oneshot(Args a, const char* name) {
    const unsigned grid = (unsigned)((a.nvec + 255) / 256);
    report(name, best_of([&] { oneshot<MODE, T, K, W><<<grid, 256>>>(a); }), a.nvec, K, W);
}
template <int MODE, int T, int K, int W, int R> static void run_persistent(Args a, unsigned long long* ctr, const char* name) {
    report(name, best_of([&] { hipMemsetAsync(ctr, 0, 8); persistent<MODE, T, K, W, R><<<512, 256>>>(a, ctr); }), a.nvec, K, W);
}

int main(int argc, char** argv) {
    // vectors of 16 B per field; default 233,280,000 = the global grid's 466.56 M float64 pairs / 2
    const long nvec = (argc > 1 ? atol(argv[1]) : 233280000L) / 1048576 * 1048576;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const long pitch = nvec + (33 * 1024 + 512L * 1024 * 1024) / 16;   // SoA: +0.5 GiB + 33 KiB between arrays
    d2 *in, *out;
    unsigned long long* ctr;
    hipMalloc((void**)&in, (size_t)pitch * 16 * 16);
    hipMalloc((void**)&out, (size_t)pitch * 2 * 16);
    hipMalloc((void**)&ctr, 128);
    hipMemset(in, 0, (size_t)pitch * 16 * 16);
    Args a = {in, out, nvec, pitch};
    const bool second = argc > 2 && atoi(argv[2]) == 2;
    for (int rep = 0; rep < 2 && !second; ++rep) {
        run_oneshot<0, 64, 14, 0>(a, "soa 14R one-shot");
        run_oneshot<0, 64, 14, 2>(a, "soa 14R+2W one-shot");
        run_oneshot<0, 64, 16, 0>(a, "soa 16R one-shot");
        run_oneshot<1, 64, 14, 2>(a, "blocked T=64 (1 KiB) out-inside one-shot");
        run_oneshot<2, 64, 14, 2>(a, "blocked T=64 (1 KiB) out-separate one-shot");
        run_oneshot<1, 256, 14, 2>(a, "blocked T=256 (4 KiB) out-inside one-shot");
        run_oneshot<2, 256, 14, 2>(a, "blocked T=256 (4 KiB) out-separate one-shot");
        run_oneshot<2, 512, 14, 2>(a, "blocked T=512 (8 KiB) out-separate one-shot");
        run_oneshot<2, 4096, 14, 2>(a, "blocked T=4096 (64 KiB) out-separate one-shot");
        run_oneshot<1, 4096, 14, 2>(a, "blocked T=4096 (64 KiB) out-inside one-shot");
        run_oneshot<2, 64, 14, 0>(a, "blocked T=64 14R only one-shot");
        run_persistent<0, 64, 14, 2, 8>(a, ctr, "soa 14R+2W persistent runs of 8");
        run_persistent<2, 64, 14, 2, 8>(a, ctr, "blocked T=64 out-separate persistent R=8");
        run_persistent<1, 64, 14, 2, 8>(a, ctr, "blocked T=64 out-inside persistent R=8");
        run_persistent<2, 512, 14, 2, 8>(a, ctr, "blocked T=512 out-separate persistent R=8");
    }
    // second series: how large should a tile be, and does the persistent schedule keep the gain
    for (int rep = 0; rep < 2 && second; ++rep) {
        run_oneshot<0, 64, 14, 2>(a, "soa 14R+2W one-shot");
        run_oneshot<2, 2048, 14, 2>(a, "blocked T=2048 (32 KiB) out-separate one-shot");
        run_oneshot<2, 4096, 14, 2>(a, "blocked T=4096 (64 KiB) out-separate one-shot");
        run_oneshot<2, 8192, 14, 2>(a, "blocked T=8192 (128 KiB) out-separate one-shot");
        run_oneshot<2, 16384, 14, 2>(a, "blocked T=16384 (256 KiB) out-separate one-shot");
        run_oneshot<2, 65536, 14, 2>(a, "blocked T=65536 (1 MiB) out-separate one-shot");
        run_oneshot<2, 262144, 14, 2>(a, "blocked T=262144 (4 MiB) out-separate one-shot");
        run_oneshot<2, 1048576, 14, 2>(a, "blocked T=1048576 (16 MiB) out-separate one-shot");
        run_oneshot<1, 16384, 14, 2>(a, "blocked T=16384 (256 KiB) out-inside one-shot");
        run_oneshot<1, 65536, 14, 2>(a, "blocked T=65536 (1 MiB) out-inside one-shot");
        run_persistent<0, 64, 14, 2, 8>(a, ctr, "soa 14R+2W persistent runs of 8");
        run_persistent<2, 4096, 14, 2, 8>(a, ctr, "blocked T=4096 out-separate persistent R=8");
        run_persistent<2, 16384, 14, 2, 8>(a, ctr, "blocked T=16384 out-separate persistent R=8");
        run_persistent<2, 65536, 14, 2, 8>(a, ctr, "blocked T=65536 out-separate persistent R=8");
        run_persistent<2, 4096, 14, 2, 16>(a, ctr, "blocked T=4096 out-separate persistent R=16");
        run_persistent<2, 4096, 14, 2, 64>(a, ctr, "blocked T=4096 out-separate persistent R=64");
    }
    return 0;
}
