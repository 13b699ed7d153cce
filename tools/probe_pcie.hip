// Host <-> device copy rates on this box: pinned and pageable memory, one and several
// threads, and what pinning user memory in place (hipHostRegister) costs -- the ceiling of
// the numpy-in / numpy-out path (MOD16_HOST mode).
// hipcc --offload-arch=gfx950 -O3 tools/probe_pcie.hip -o tools/bin/probe_pcie -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t B = size_t(1) << 30;   // 1 GiB per buffer
    void *dev, *dev2, *pin, *pin2;
    hipMalloc(&dev, B); hipMalloc(&dev2, B);
    hipHostMalloc(&pin, B); hipHostMalloc(&pin2, B);
    memset(pin, 1, B); memset(pin2, 2, B);
    char* page = (char*)aligned_alloc(4096, B); memset(page, 3, B);
    char* page2 = (char*)aligned_alloc(4096, B); memset(page2, 4, B);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    auto rate = [&](const char* name, auto f, double bytes) {
        f(); hipDeviceSynchronize();
        double best = 1e30;
        for (int r = 0; r < 3; ++r) { double t0 = now(); f(); hipDeviceSynchronize(); best = std::min(best, now() - t0); }
        printf("%-58s %7.2f GB/s\n", name, bytes / best / 1e9); fflush(stdout);
    };
    rate("H2D pinned, 1 GiB, one stream", [&] { hipMemcpyAsync(dev, pin, B, hipMemcpyHostToDevice, s1); }, B);
    rate("D2H pinned, 1 GiB, one stream", [&] { hipMemcpyAsync(pin, dev, B, hipMemcpyDeviceToHost, s1); }, B);
    rate("H2D + D2H pinned concurrently (sum)", [&] { hipMemcpyAsync(dev, pin, B, hipMemcpyHostToDevice, s1); hipMemcpyAsync(pin2, dev2, B, hipMemcpyDeviceToHost, s2); }, 2.0 * B);
    rate("H2D pinned, 64 x 16 MiB chunks, one stream", [&] { for (int i = 0; i < 64; ++i) hipMemcpyAsync((char*)dev + i * (B / 64), (char*)pin + i * (B / 64), B / 64, hipMemcpyHostToDevice, s1); }, B);
    rate("H2D pageable, 1 GiB, one thread", [&] { hipMemcpy(dev, page, B, hipMemcpyHostToDevice); }, B);
    rate("D2H pageable, 1 GiB, one thread", [&] { hipMemcpy(page, dev, B, hipMemcpyDeviceToHost); }, B);
    for (int nt : {2, 4, 8}) {
        char name[96]; snprintf(name, sizeof name, "H2D pageable, %d threads x %zu MiB", nt, B / nt >> 20);
        rate(name, [&] {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { hipSetDevice(0); hipMemcpy((char*)dev + t * (B / nt), page + t * (B / nt), B / nt, hipMemcpyHostToDevice); });
            for (auto& x : th) x.join(); }, B);
    }
    for (int nt : {1, 2, 4, 8, 16}) {
        char name[96]; snprintf(name, sizeof name, "host memcpy pageable -> pinned, %d threads", nt);
        rate(name, [&] {
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { memcpy((char*)pin + t * (B / nt), page + t * (B / nt), B / nt); });
            for (auto& x : th) x.join(); }, B);
    }
    {   // pinning user memory in place
        double t0 = now(); hipError_t e = hipHostRegister(page2, B, hipHostRegisterDefault); double t1 = now();
        printf("hipHostRegister 1 GiB: %s, %.1f ms (%.2f GB/s)\n", hipGetErrorString(e), (t1 - t0) * 1e3, B / (t1 - t0) / 1e9);
        if (e == hipSuccess) {
            rate("H2D from registered user memory, 1 GiB", [&] { hipMemcpyAsync(dev, page2, B, hipMemcpyHostToDevice, s1); }, B);
            t0 = now(); hipHostUnregister(page2); t1 = now();
            printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
        }
        // fresh (never touched) memory: registration also has to fault the pages in
        char* fresh = (char*)aligned_alloc(4096, B);
        t0 = now(); e = hipHostRegister(fresh, B, hipHostRegisterDefault); t1 = now();
        printf("hipHostRegister 1 GiB of untouched memory: %s, %.1f ms\n", hipGetErrorString(e), (t1 - t0) * 1e3);
        if (e == hipSuccess) {
            rate("D2H into registered fresh memory, 1 GiB", [&] { hipMemcpyAsync(fresh, dev, B, hipMemcpyDeviceToHost, s1); }, B);
            hipHostUnregister(fresh);
        }
        char* fresh2 = (char*)aligned_alloc(4096, B);
        t0 = now(); hipMemcpy(fresh2, dev, B, hipMemcpyDeviceToHost); t1 = now();
        printf("D2H into untouched pageable memory (first touch), 1 GiB: %.2f GB/s\n", B / (t1 - t0) / 1e9);
    }
    return 0;
}
