// Issue cost of single instructions on gfx950, 8 waves/SIMD, independent
// instructions (inline asm so the compiler cannot fold them).
// hipcc --offload-arch=gfx950 -O3 tools/probe_issue.hip -o tools/bin/probe_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define BODY(NAME, ASM, ...)                                                          \
    __global__ void __launch_bounds__(256) k_##NAME(double* out, int iters) {         \
        double a = 1.0 + threadIdx.x * 1e-6, b = 2.0, c = 0.5;                        \
        double r0 = a, r1 = a, r2 = a, r3 = a;                                        \
        int i0 = (threadIdx.x & 63) * 8, i1 = 3;                                                 \
        for (int i = 0; i < iters; ++i) {                                             \
            REP8(asm volatile(ASM : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(i0)  \
                              : "v"(a), "v"(b), "v"(c), "v"(i1) : __VA_ARGS__);)             \
        }                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + i0;                 \
    }

// each ASM string holds 4 independent instructions
BODY(fma64, "v_fma_f64 %0, %5, %6, %7\n v_fma_f64 %1, %5, %6, %7\n v_fma_f64 %2, %5, %6, %7\n v_fma_f64 %3, %5, %6, %7", "memory")
BODY(add64, "v_add_f64 %0, %5, %6\n v_add_f64 %1, %5, %6\n v_add_f64 %2, %5, %6\n v_add_f64 %3, %5, %6", "memory")
BODY(mul64, "v_mul_f64 %0, %5, %6\n v_mul_f64 %1, %5, %6\n v_mul_f64 %2, %5, %6\n v_mul_f64 %3, %5, %6", "memory")
BODY(min64, "v_min_f64 %0, %5, %6\n v_min_f64 %1, %5, %6\n v_min_f64 %2, %5, %6\n v_min_f64 %3, %5, %6", "memory")
BODY(cmp64_vcc, "v_cmp_gt_f64 vcc, %5, %6\n v_cmp_gt_f64 vcc, %6, %7\n v_cmp_gt_f64 vcc, %5, %7\n v_cmp_gt_f64 vcc, %7, %5", "vcc")
BODY(cmp64_sgpr, "v_cmp_gt_f64 s[20:21], %5, %6\n v_cmp_gt_f64 s[22:23], %6, %7\n v_cmp_gt_f64 s[24:25], %5, %7\n v_cmp_gt_f64 s[26:27], %7, %5", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
BODY(cmp32_vcc, "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %8, %4\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %8, %4", "vcc")
BODY(cmpclass64, "v_cmp_class_f64 vcc, %5, %8\n v_cmp_class_f64 vcc, %6, %8\n v_cmp_class_f64 vcc, %7, %8\n v_cmp_class_f64 vcc, %5, %8", "vcc")
BODY(cndmask, "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %4, %8, %4, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %4, %8, %4, vcc", "memory")
BODY(mov32, "v_mov_b32 %4, %8\n v_mov_b32 %4, %8\n v_mov_b32 %4, %8\n v_mov_b32 %4, %8", "memory")
BODY(mov64, "v_mov_b64 %0, %5\n v_mov_b64 %1, %6\n v_mov_b64 %2, %7\n v_mov_b64 %3, %5", "memory")
BODY(readlane, "v_readlane_b32 s20, %4, 3\n v_readlane_b32 s21, %4, 5\n v_readlane_b32 s22, %4, 7\n v_readlane_b32 s23, %4, 9", "s20", "s21", "s22", "s23")
BODY(writelane, "v_writelane_b32 %4, s2, 3\n v_writelane_b32 %4, s3, 5\n v_writelane_b32 %4, s2, 7\n v_writelane_b32 %4, s3, 9", "memory")
BODY(rcp64, "v_rcp_f64 %0, %5\n v_rcp_f64 %1, %6\n v_rcp_f64 %2, %7\n v_rcp_f64 %3, %5", "memory")
BODY(ldexp64, "v_ldexp_f64 %0, %5, %8\n v_ldexp_f64 %1, %6, %8\n v_ldexp_f64 %2, %7, %8\n v_ldexp_f64 %3, %5, %8", "memory")
BODY(rndne64, "v_rndne_f64 %0, %5\n v_rndne_f64 %1, %6\n v_rndne_f64 %2, %7\n v_rndne_f64 %3, %5", "memory")
BODY(cvti32f64, "v_cvt_i32_f64 %4, %5\n v_cvt_i32_f64 %4, %6\n v_cvt_i32_f64 %4, %7\n v_cvt_i32_f64 %4, %5", "memory")
BODY(cvtf64i32, "v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %8\n v_cvt_f64_i32 %2, %8\n v_cvt_f64_i32 %3, %8", "memory")
BODY(frexpm64, "v_frexp_mant_f64 %0, %5\n v_frexp_mant_f64 %1, %6\n v_frexp_mant_f64 %2, %7\n v_frexp_mant_f64 %3, %5", "memory")
BODY(frexpe64, "v_frexp_exp_i32_f64 %4, %5\n v_frexp_exp_i32_f64 %4, %6\n v_frexp_exp_i32_f64 %4, %7\n v_frexp_exp_i32_f64 %4, %5", "memory")
BODY(fma32, "v_fma_f32 %4, %8, %8, %8\n v_fma_f32 %4, %8, %8, %8\n v_fma_f32 %4, %8, %8, %8\n v_fma_f32 %4, %8, %8, %8", "memory")
BODY(pkfma32, "v_pk_fma_f32 %0, %5, %6, %7\n v_pk_fma_f32 %1, %5, %6, %7\n v_pk_fma_f32 %2, %5, %6, %7\n v_pk_fma_f32 %3, %5, %6, %7", "memory")
BODY(and32, "v_and_b32 %4, %8, %8\n v_and_b32 %4, %8, %8\n v_and_b32 %4, %8, %8\n v_and_b32 %4, %8, %8", "memory")
BODY(lshl_add64, "v_lshl_add_u64 %0, %5, 3, %6\n v_lshl_add_u64 %1, %5, 3, %6\n v_lshl_add_u64 %2, %5, 3, %6\n v_lshl_add_u64 %3, %5, 3, %6", "memory")
BODY(smov, "s_mov_b32 s20, s2\n s_mov_b32 s21, s3\n s_mov_b32 s22, s2\n s_mov_b32 s23, s3", "s20", "s21", "s22", "s23")


#define BODY2(NAME, ASM, ...)                                                         \
    __global__ void __launch_bounds__(256) k_##NAME(double* out, int iters) {         \
        double a = 1.0 + threadIdx.x * 1e-6, b = 2.0, c = 0.5;                        \
        int i0 = (threadIdx.x & 63) * 8, i1 = 3, i2 = 5, i3 = 7, k = 9;               \
        for (int i = 0; i < iters; ++i) {                                             \
            REP8(asm volatile(ASM : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(a)   \
                              : "v"(k), "v"(b), "v"(c) : __VA_ARGS__);)               \
        }                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = a + i0 + i1 + i2 + i3;                  \
    }
BODY2(cnd_indep, "v_cndmask_b32 %0, %5, %5, vcc\n v_cndmask_b32 %1, %5, %5, vcc\n v_cndmask_b32 %2, %5, %5, vcc\n v_cndmask_b32 %3, %5, %5, vcc", "memory")
BODY2(cnd_dep, "v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %0, %0, %5, vcc\n v_cndmask_b32 %0, %0, %5, vcc", "memory")
BODY2(cnd_e64, "v_cndmask_b32 %0, %5, %5, s[20:21]\n v_cndmask_b32 %1, %5, %5, s[20:21]\n v_cndmask_b32 %2, %5, %5, s[20:21]\n v_cndmask_b32 %3, %5, %5, s[20:21]", "memory")
BODY2(fma64_dep, "v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %4, %4, %6, %7", "memory")
BODY2(mov32_dep, "v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0\n v_mov_b32 %0, %0", "memory")
BODY2(mov32_ind, "v_mov_b32 %0, %5\n v_mov_b32 %1, %5\n v_mov_b32 %2, %5\n v_mov_b32 %3, %5", "memory")
BODY2(add32_dep, "v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5\n v_add_u32 %0, %0, %5", "memory")
BODY2(add32_ind, "v_add_u32 %0, %5, %5\n v_add_u32 %1, %5, %5\n v_add_u32 %2, %5, %5\n v_add_u32 %3, %5, %5", "memory")
BODY2(fma32_ind, "v_fma_f32 %0, %5, %5, %5\n v_fma_f32 %1, %5, %5, %5\n v_fma_f32 %2, %5, %5, %5\n v_fma_f32 %3, %5, %5, %5", "memory")
BODY2(sel64, "v_cmp_gt_f64 vcc, %6, %7\n v_cndmask_b32 %0, %5, %5, vcc\n v_cmp_gt_f64 vcc, %7, %6\n v_cndmask_b32 %1, %5, %5, vcc", "vcc")
BODY2(sel64_sgpr, "v_cmp_gt_f64 s[20:21], %6, %7\n v_cndmask_b32 %0, %5, %5, s[20:21]\n v_cmp_gt_f64 s[22:23], %7, %6\n v_cndmask_b32 %1, %5, %5, s[22:23]", "s20", "s21", "s22", "s23")
BODY2(selpair_vcc, "v_cmp_gt_f64 vcc, %6, %7\n v_cndmask_b32 %0, %5, %5, vcc\n v_cndmask_b32 %1, %5, %5, vcc\n v_fma_f64 %4, %4, %6, %7", "vcc")
BODY2(selpair_sgpr, "v_cmp_gt_f64 s[20:21], %6, %7\n v_cndmask_b32 %0, %5, %5, s[20:21]\n v_cndmask_b32 %1, %5, %5, s[20:21]\n v_fma_f64 %4, %4, %6, %7", "s20", "s21")
BODY2(selpair_vcc3, "v_cmp_gt_f64 vcc, %6, %7\n v_cndmask_b32_e64 %0, %5, %5, vcc\n v_cndmask_b32_e64 %1, %5, %5, vcc\n v_fma_f64 %4, %4, %6, %7", "vcc")
BODY2(fma64x3_cnd, "v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %4, %4, %6, %7\n v_fma_f64 %4, %4, %6, %7\n v_cndmask_b32 %1, %5, %5, vcc", "memory")
BODY2(dsread, "ds_read_b64 %4, %0\n ds_read_b64 %4, %0\n ds_read_b64 %4, %0\n ds_read_b64 %4, %0\n s_waitcnt lgkmcnt(0)", "memory")
BODY2(readlane_ind, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9", "s20", "s21", "s22", "s23")

template <typename K> float time_it(K k, double* out, int iters, int blocks) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 8;   // 256-thread blocks per CU = waves per SIMD
    const int iters = 1000, blocks = 256 * per_cu;
    printf("waves per SIMD: %d\n", per_cu);
    double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
    // wave-instructions issued per SIMD: blocks*4 waves / 1024 SIMDs * iters * 32
    const double wi_per_simd = (double)blocks * 4 / 1024 * iters * 32;
    float base = 0;
#define RUN(NAME) { float ms = time_it(k_##NAME, out, iters, blocks); if (!base) base = ms; \
        printf("%-12s %8.3f ms  %6.2f ns/wave-instr/SIMD  (%.2fx v_fma_f64)\n", #NAME, ms, ms * 1e6 / wi_per_simd, ms / base); }
    RUN(fma64) RUN(add64) RUN(mul64) RUN(min64) RUN(cmp64_vcc) RUN(cmp64_sgpr) RUN(cmp32_vcc) RUN(cmpclass64)
    RUN(cndmask) RUN(mov32) RUN(mov64) RUN(readlane) RUN(writelane) RUN(rcp64) RUN(ldexp64) RUN(rndne64)
    RUN(cvti32f64) RUN(cvtf64i32) RUN(frexpm64) RUN(frexpe64) RUN(fma32) RUN(pkfma32) RUN(and32) RUN(lshl_add64) RUN(smov)
    RUN(cnd_indep) RUN(cnd_dep) RUN(cnd_e64) RUN(fma64_dep) RUN(mov32_dep) RUN(mov32_ind) RUN(add32_dep) RUN(add32_ind) RUN(fma32_ind) RUN(sel64) RUN(sel64_sgpr) RUN(selpair_vcc) RUN(selpair_sgpr) RUN(selpair_vcc3) RUN(fma64x3_cnd) RUN(dsread) RUN(readlane_ind)
    return 0;
}
