// What does a 1R+1W copy reach on this part, by kernel shape? (reference point
// for the mixed read/write roof of the ET kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) copy_stride(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    long step = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) b[i] = a[i];
}
__global__ void __launch_bounds__(256) copy_oneshot(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <int U> __global__ void __launch_bounds__(256) copy_unroll(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    long step = (long)gridDim.x * 256 * U;
    for (long i = (long)blockIdx.x * 256 * U + threadIdx.x; i < n; i += step) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (i + u * 256 < n) ? a[i + u * 256] : f4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * 256 < n) b[i + u * 256] = v[u];
    }
}
template <int T> __global__ void __launch_bounds__(T) copy_block(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    long step = (long)gridDim.x * T;
    for (long i = (long)blockIdx.x * T + threadIdx.x; i < n; i += step) b[i] = a[i];
}
__global__ void __launch_bounds__(256) read_only(const f4* __restrict__ a, f4* __restrict__ b, long n) {
    long step = (long)gridDim.x * 256; f4 s = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += step) s += a[i];
    if (s[0] == 1.234e30f) b[0] = s;
}

template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); float best = 1e30f;
    for (int r = 0; r < 3; ++r) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    return best;
}
int main(int argc, char** argv) {
    for (long mb : {256L, 1024L, 4096L, 16384L}) {
        long n = mb * 1024 * 1024 / 16; f4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMemset(a, 1, n * 16); hipMemset(b, 0, n * 16);
        double gb = 2.0 * n * 16 / 1e9;
        printf("--- %ld MiB per buffer\n", mb);
        for (int g : {256 * 4, 256 * 16, 256 * 64}) printf("grid-stride 256thr grid %6d : %7.1f GB/s\n", g, gb / timeit([&] { copy_stride<<<g, 256>>>(a, b, n); }) * 1e3);
        printf("one-shot 256thr            : %7.1f GB/s\n", gb / timeit([&] { copy_oneshot<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }) * 1e3);
        printf("unroll4 grid 4096          : %7.1f GB/s\n", gb / timeit([&] { copy_unroll<4><<<4096, 256>>>(a, b, n); }) * 1e3);
        printf("unroll8 grid 2048          : %7.1f GB/s\n", gb / timeit([&] { copy_unroll<8><<<2048, 256>>>(a, b, n); }) * 1e3);
        printf("512thr grid 2048           : %7.1f GB/s\n", gb / timeit([&] { copy_block<512><<<2048, 512>>>(a, b, n); }) * 1e3);
        printf("1024thr grid 1024          : %7.1f GB/s\n", gb / timeit([&] { copy_block<1024><<<1024, 1024>>>(a, b, n); }) * 1e3);
        printf("hipMemcpyDtoD              : %7.1f GB/s\n", gb / timeit([&] { hipMemcpyAsync(b, a, n * 16, hipMemcpyDeviceToDevice, 0); }) * 1e3);
        printf("read only (grid 16384)     : %7.1f GB/s\n", gb / 2 / timeit([&] { read_only<<<16384, 256>>>(a, b, n); }) * 1e3);
        hipFree(a); hipFree(b);
    }
    return 0;
}
