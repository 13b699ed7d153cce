// Probe of gfx950 float64 primitives: accuracy of v_rcp/v_rsq/v_sqrt_f64 seeds
// and issue cost of the instruction kinds the ET kernel is made of.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probe_f64.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void acc_kernel(const double* x, double* r, double* q, double* s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        r[i] = __builtin_amdgcn_rcp(x[i]);
        q[i] = __builtin_amdgcn_rsq(x[i]);
        s[i] = __builtin_amdgcn_sqrt(x[i]);
    }
}

template <int OP> __global__ void __launch_bounds__(256) thr_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x * 1e-9, b = 1.000000001, c = 0.5, d = a * 0.3;
    double e = a + 1.0, f = a + 2.0, g = a + 3.0, h = a + 4.0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c);
                           g = __builtin_fma(g, b, c); h = __builtin_fma(h, b, c); }
            if (OP == 1) { a = a * b; d = d * b; e = e * b; f = f * b; g = g * b; h = h * b; }
            if (OP == 2) { a = a + b; d = d + b; e = e + b; f = f + b; g = g + b; h = h + b; }
            if (OP == 3) { a = __builtin_amdgcn_rcp(a); d = __builtin_amdgcn_rcp(d); e = __builtin_amdgcn_rcp(e); f = __builtin_amdgcn_rcp(f);
                           g = __builtin_amdgcn_rcp(g); h = __builtin_amdgcn_rcp(h); }
            if (OP == 4) { a = (a > d) ? e : f; d = (d > e) ? f : g; e = (e > f) ? g : h; f = (f > g) ? h : a; g = (g > h) ? a : d; h = (h > a) ? d : e; }
            if (OP == 5) { a = __builtin_amdgcn_ldexp(a, 1); d = __builtin_amdgcn_ldexp(d, 1); e = __builtin_amdgcn_ldexp(e, -1); f = __builtin_amdgcn_ldexp(f, 1);
                           g = __builtin_amdgcn_ldexp(g, -1); h = __builtin_amdgcn_ldexp(h, -1); }
            if (OP == 6) { a = __builtin_amdgcn_rsq(a); d = __builtin_amdgcn_rsq(d); e = __builtin_amdgcn_rsq(e); f = __builtin_amdgcn_rsq(f);
                           g = __builtin_amdgcn_rsq(g); h = __builtin_amdgcn_rsq(h); }
            if (OP == 7) { a = __builtin_rint(a); d = __builtin_rint(d); e = __builtin_rint(e); f = __builtin_rint(f); g = __builtin_rint(g); h = __builtin_rint(h); }
            if (OP == 8) { a = __builtin_fmax(a, d); d = __builtin_fmax(d, e); e = __builtin_fmax(e, f); f = __builtin_fmax(f, g); g = __builtin_fmax(g, h); h = __builtin_fmax(h, a); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
}

template <int OP> __global__ void __launch_bounds__(256) thr32_kernel(float* out, int iters, float seed) {
    float a = seed + threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = a * 0.3f, e = a + 1.f, f = a + 2.f, g = a + 3.f, h = a + 4.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a = __builtin_fmaf(a, b, c); d = __builtin_fmaf(d, b, c); e = __builtin_fmaf(e, b, c); f = __builtin_fmaf(f, b, c);
                           g = __builtin_fmaf(g, b, c); h = __builtin_fmaf(h, b, c); }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + g + h;
}

template <typename K> double time_it(K launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> hx(n), hr(n), hq(n), hs(n);
    for (int i = 0; i < n; ++i) hx[i] = 0.5 + 3.5 * (i + 0.5) / n;
    double *x, *r, *q, *s;
    hipMalloc(&x, n * 8); hipMalloc(&r, n * 8); hipMalloc(&q, n * 8); hipMalloc(&s, n * 8);
    hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
    acc_kernel<<<n / 256, 256>>>(x, r, q, s, n);
    hipMemcpy(hr.data(), r, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hq.data(), q, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hs.data(), s, n * 8, hipMemcpyDeviceToHost);
    double er = 0, eq = 0, es = 0;
    for (int i = 0; i < n; ++i) {
        er = fmax(er, fabs(hr[i] * hx[i] - 1.0));
        eq = fmax(eq, fabs(hq[i] * sqrt(hx[i]) - 1.0));
        es = fmax(es, fabs(hs[i] / sqrt(hx[i]) - 1.0));
    }
    printf("max rel err: v_rcp_f64 %.3e (2^%.1f)  v_rsq_f64 %.3e (2^%.1f)  v_sqrt_f64 %.3e (2^%.1f)\n",
           er, log2(er), eq, log2(eq), es, log2(es > 0 ? es : 1e-300));
    double* out; hipMalloc(&out, 1024 * 8 * 256 * 8);
    const int iters = 2000, blocks = 256 * 8;   // 8 blocks/CU -> 8 waves/SIMD
    const double ops = (double)blocks * 256 * iters * 8 * 6;
    const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "cmp+cndmask f64", "v_ldexp_f64", "v_rsq_f64", "v_rndne_f64", "v_max_f64"};
    double t[9];
    t[0] = time_it([&] { thr_kernel<0><<<blocks, 256>>>(out, iters, 1.0); });
    t[1] = time_it([&] { thr_kernel<1><<<blocks, 256>>>(out, iters, 1.0); });
    t[2] = time_it([&] { thr_kernel<2><<<blocks, 256>>>(out, iters, 1.0); });
    t[3] = time_it([&] { thr_kernel<3><<<blocks, 256>>>(out, iters, 1.0); });
    t[4] = time_it([&] { thr_kernel<4><<<blocks, 256>>>(out, iters, 1.0); });
    t[5] = time_it([&] { thr_kernel<5><<<blocks, 256>>>(out, iters, 1.0); });
    t[6] = time_it([&] { thr_kernel<6><<<blocks, 256>>>(out, iters, 1.0); });
    t[7] = time_it([&] { thr_kernel<7><<<blocks, 256>>>(out, iters, 1.0); });
    t[8] = time_it([&] { thr_kernel<8><<<blocks, 256>>>(out, iters, 1.0); });
    for (int k = 0; k < 9; ++k)
        printf("%-18s %8.3f ms  %7.2f Tops/s  (%.2fx fma time)\n", names[k], t[k], ops / t[k] / 1e9, t[k] / t[0]);
    double t32 = time_it([&] { thr32_kernel<0><<<blocks, 256>>>((float*)out, iters, 1.0f); });
    printf("%-18s %8.3f ms  %7.2f Tops/s  (%.2fx f64 fma time)\n", "v_fma_f32", t32, ops / t32 / 1e9, t32 / t[0]);
    return 0;
}
