// gfx950 kernels of libmod16hip: the fused ET pixel kernel, the synthetic
// driver generator and the diagnostics reduction. Launch code is in
// mod16_capi.hip.
//
// Fused ET kernel -- data movement per pixel (float64): 14 x 8 B driver
// loads + 1 B class + 2 x 8 B stores = 129 B, each driver array read exactly
// once with 16-byte (global_load_dwordx4) coalesced accesses; no MFMA (an
// element-wise map has no contraction). The BPLUT and its derived
// reciprocals sit in LDS as [row][16] so that lanes of different classes hit
// different banks and lanes of one class broadcast.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "mod16_physics.hpp"

namespace mod16 {

constexpr int kBlock = 256;
constexpr int kLutCols = 16;   // 13 class codes + NaN padding (col 13..15)
constexpr unsigned kStatusClassRange = 1u;

template <typename T> struct EtArgs {
    const T* drv[14];
    const T* par[11];
    const uint8_t* cls;
    const T* lut;          // device [MOD16_LUT_ROWS][kLutCols], EXACT arithmetic (type T)
    const double* lut64;   // the same table in float64, FAST arithmetic
    const double* tab;     // device exp/log tables, FastMath<double>::kTabDoubles values
    T* out[10];            // day, night, the 6 components (mod16_component), PET day, night
    int64_t n;
    unsigned* status;
    uint32_t dense_drv;    // bit k set: driver k is a dense array, else a broadcast scalar
    uint32_t dense_par;
    double* diag_partial;  // et_kernel_dma: [gridDim][8] per-block diagnostics
    unsigned long long* dyn_counter;   // et_kernel_dyn: run tickets, zero at launch
    int64_t drv_pitch;     // et_kernel_dyn<.., PITCHED>: drv[k] = drv[0] + k * drv_pitch (elements)
};

static_assert(__builtin_offsetof(EtArgs<double>, drv) == 0 && __builtin_offsetof(EtArgs<float>, drv) == 0,
              "et_kernel_dyn reads drv[] from offset 0 of the kernel-argument segment");

template <typename T, int V> struct Vec;
template <> struct Vec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<double, 1> { typedef double type; };
template <> struct Vec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<float, 1> { typedef float type; };

template <typename T, int V, bool DENSE>
__device__ __forceinline__ void load_vec(const T* __restrict__ p, bool dense, int64_t i,
                                         T (&dst)[V]) {
    if (DENSE || dense) {   // `dense` is wave-uniform (a kernel-argument bit)
        typedef typename Vec<T, V>::type VT;
        VT v = *reinterpret_cast<const VT*>(p + i);
        if constexpr (V == 1) {
            dst[0] = v;
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) dst[j] = v[j];
        }
    } else {
        T s = p[0];
#pragma unroll
        for (int j = 0; j < V; ++j) dst[j] = s;
    }
}

template <typename T, int V>
__device__ __forceinline__ void store_vec(T* __restrict__ p, int64_t i, const T (&src)[V]) {
    typedef typename Vec<T, V>::type VT;
    if constexpr (V == 1) {
        p[i] = src[0];
    } else {
        VT v;
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = src[j];
        *reinterpret_cast<VT*>(p + i) = v;
    }
}

// Template switches: V pixels per thread (16-byte accesses when V > 1), LUT =
// parameters from the BPLUT in LDS by class code (else per-pixel / scalar
// parameter inputs), FAST = strength-reduced arithmetic, SEP = also store the
// six components, DENSE = every driver is a dense array (no broadcast checks).
template <typename T, int V, bool LUT, bool FAST, bool SEP, bool DENSE, bool PET = false>
__global__ void __launch_bounds__(kBlock) et_kernel(const EtArgs<T> a) {
    // FAST always computes in float64 (float32 data are widened on load and
    // the result rounded once on store); EXACT computes in the data type, as
    // numpy does for the reference code.
    typedef typename std::conditional<FAST, double, T>::type C;
    constexpr int kTab = FAST ? FastMath<double>::kTabDoubles : 1;
    __shared__ C lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    if (LUT)
        for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) {
            if constexpr (FAST) lut[i] = a.lut64[i];
            else lut[i] = a.lut[i];
        }
    if (FAST)
        for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int64_t nvec = a.n / V;
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += step) {
        const int64_t i = v * V;
        T in[14][V];
#pragma unroll
        for (int k = 0; k < 14; ++k)
            load_vec<T, V, DENSE>(a.drv[k], (a.dense_drv >> k) & 1u, i, in[k]);
        unsigned cbits = 0;
        T pin[11][V];
        if (LUT) {
            if constexpr (V == 1) cbits = a.cls[i];
            else if constexpr (V == 2) cbits = *reinterpret_cast<const uint16_t*>(a.cls + i);
            else cbits = *reinterpret_cast<const uint32_t*>(a.cls + i);
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k)
                load_vec<T, V, false>(a.par[k], (a.dense_par >> k) & 1u, i, pin[k]);
        }
        T res[10][V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            PixelIn<C> x = {(C)in[0][j], (C)in[1][j], (C)in[2][j], (C)in[3][j], (C)in[4][j],
                            (C)in[5][j], (C)in[6][j], (C)in[7][j], (C)in[8][j], (C)in[9][j],
                            (C)in[10][j], (C)in[11][j], (C)in[12][j], (C)in[13][j]};
            ClassPar<C> p;
            if (LUT) {
                unsigned c = (cbits >> (8 * j)) & 0xffu;
                if (c >= 13u) {   // numpy would raise IndexError: flag it, give NaN
                    atomicOr(a.status, kStatusClassRange);
                    c = 13u;
                }
                const C* l = lut + c;
                p.tmin_close = l[0 * kLutCols];
                p.tmin_open = l[1 * kLutCols];
                p.vpd_open = l[2 * kLutCols];
                p.vpd_close = l[3 * kLutCols];
                p.gl_sh = l[4 * kLutCols];
                p.gl_wv = l[5 * kLutCols];
                p.g_cut = l[6 * kLutCols];
                p.csl = l[7 * kLutCols];
                p.rbl_min = l[8 * kLutCols];
                p.rbl_max = l[9 * kLutCols];
                p.beta = l[10 * kLutCols];
                if (FAST) {
                    p.inv_dtmin = l[11 * kLutCols];
                    p.inv_dvpd = l[12 * kLutCols];
                    p.rbl_slope = l[13 * kLutCols];
                    p.inv_beta = l[14 * kLutCols];
                }
            } else {
                p.tmin_close = (C)pin[0][j];
                p.tmin_open = (C)pin[1][j];
                p.vpd_open = (C)pin[2][j];
                p.vpd_close = (C)pin[3][j];
                p.gl_sh = (C)pin[4][j];
                p.gl_wv = (C)pin[5][j];
                p.g_cut = (C)pin[6][j];
                p.csl = (C)pin[7][j];
                p.rbl_min = (C)pin[8][j];
                p.rbl_max = (C)pin[9][j];
                p.beta = (C)pin[10][j];
                if (FAST) p.derive();
            }
            PixelOut<C> o;
            if constexpr (FAST) o = et_pixel_fast<double, PET>(x, p, tab);
            else o = et_pixel_exact<T, PET>(x, p);
            // mod16/__init__.py:792: (canopy + soil) + transpiration
            res[0][j] = (T)((o.canopy_d + o.soil_d) + o.trans_d);
            res[1][j] = (T)((o.canopy_n + o.soil_n) + o.trans_n);
            if (SEP) {
                res[2][j] = (T)o.canopy_d;
                res[3][j] = (T)o.soil_d;
                res[4][j] = (T)o.trans_d;
                res[5][j] = (T)o.canopy_n;
                res[6][j] = (T)o.soil_n;
                res[7][j] = (T)o.trans_n;
            }
            if (PET) {
                res[8][j] = (T)o.pet_d;
                res[9][j] = (T)o.pet_n;
            }
        }
#pragma unroll
        for (int k = 0; k < (PET ? 10 : (SEP ? 8 : 2)); ++k)
            if (a.out[k]) store_vec<T, V>(a.out[k], i, res[k]);
    }
}

// ------------------------------------------------------------- diagnostics
// Deterministic two-level sum: every block walks a fixed slice pattern, the
// in-block tree is fixed, and the final pass adds the per-block partials in
// index order -- so the result depends only on (n, grid), never on timing.
constexpr int kDiag = 8;

__device__ __forceinline__ void diag_merge(double (&a)[kDiag], const double (&b)[kDiag]) {
#pragma unroll
    for (int k = 0; k < 6; ++k) a[k] += b[k];
    a[6] = (b[6] > a[6]) ? b[6] : a[6];
    a[7] = (b[7] > a[7]) ? b[7] : a[7];
}

template <int BLOCK = kBlock>
__device__ __forceinline__ void diag_block_reduce(double (&acc)[kDiag], double* out) {
    __shared__ double sm[BLOCK / 64][kDiag];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = __shfl_down(acc[k], off, 64);
        diag_merge(acc, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < kDiag; ++k) sm[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK / 64; ++w) {
            double o[kDiag];
            for (int k = 0; k < kDiag; ++k) o[k] = sm[w][k];
            diag_merge(acc, o);
        }
        for (int k = 0; k < kDiag; ++k) out[k] = acc[k];
    }
}

template <typename T>
__global__ void __launch_bounds__(kBlock) diag_partial_kernel(const T* __restrict__ day,
                                                              const T* __restrict__ night,
                                                              int64_t n, double* partial) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    const int64_t step = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += step) {
        double d = (double)day[i], g = (double)night[i];
        bool dn = d != d, gn = g != g;
        acc[0] += dn ? 0.0 : d;
        acc[1] += gn ? 0.0 : g;
        acc[2] += dn ? 0.0 : 1.0;
        acc[3] += gn ? 0.0 : 1.0;
        acc[4] += dn ? 1.0 : 0.0;
        acc[5] += gn ? 1.0 : 0.0;
        acc[6] = (!dn && d > acc[6]) ? d : acc[6];
        acc[7] = (!gn && g > acc[7]) ? g : acc[7];
    }
    diag_block_reduce(acc, partial + (int64_t)blockIdx.x * kDiag);
}

__global__ void __launch_bounds__(kBlock) diag_final_kernel(const double* partial, int nblocks,
                                                            double* out) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        double o[kDiag];
        for (int k = 0; k < kDiag; ++k) o[k] = partial[(int64_t)b * kDiag + k];
        diag_merge(acc, o);
    }
    diag_block_reduce(acc, out);
}

// ---------------------------------------------------------------- LDS-DMA form
// Production kernel for dense multi-class rasters (class raster + BPLUT, totals
// only). Same arithmetic as et_kernel<.., LUT, FAST, !SEP, DENSE>; what differs
// is how the drivers reach the registers. At ~170-200 VGPRs only two waves fit a
// SIMD, too few to hide HBM latency behind other waves, and there is no room
// for a second register set to prefetch into. So each wave owns a 14.25 KiB LDS
// slot and streams the NEXT iteration's 14 driver vectors (+ class bytes) into
// it with global_load_lds (LDS-DMA, no VGPR destination) while it computes the
// current one: counted vmcnt -> ds_read_b128 x 14 -> issue next -> compute ->
// store. The slot is private to the wave that fills it, so no barrier is
// involved: the wave's own counted s_waitcnt vmcnt orders its ds_reads behind
// its DMA. Addresses are SGPR chunk base + 32-bit lane offset (no per-array
// 64-bit VALU add). Every byte is touched once, so both directions use the
// non-temporal policy: +3-4 % on the 14-read + 2-write stream mix
// (profiles/r01_probe_streams_hbm_roof.txt).
//
// The kernel also reduces its outputs into per-block diagnostics partials
// (kDiag doubles per block, the fields of diag_partial_kernel) while they are
// still in registers (DIAG), which saves the separate 16 B/pixel reduction
// pass. Only DIAG = true is instantiated: without the accumulation hipcc's
// schedule needs 50-60 more VGPRs and the kernel is slower. (Timing note:
// identical binaries differ by ~3 % from process to process on one device,
// bimodally -- compare variants over several processes, not one.)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int kDmaBlock = 256;       // threads per block (128/192/320 measured slower)
constexpr int kDmaNt = 2;            // cache-policy bits of the LDS-DMA loads: nt
#ifndef MOD16_CHUNK_RUN
#define MOD16_CHUNK_RUN 1
#endif
constexpr int kChunkRun = MOD16_CHUNK_RUN;   // consecutive chunks per block before striding

template <typename T, bool FAST, bool DIAG>
__global__ void __launch_bounds__(kBlock) et_kernel_dma(const EtArgs<T> a) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int kSlot = 15 * 1024;   // 14 x (64 lanes x 16 B) + class bytes
    // arithmetic is float64 for both data types (float32 is widened on load,
    // rounded once on store)
    static_assert(FAST, "the LDS-DMA kernel is the FAST production kernel");
    constexpr int kTab = FastMath<double>::kTabDoubles;
    __shared__ double lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    __shared__ __attribute__((aligned(16))) char stage[(kBlock / 64) * kSlot];
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) lut[i] = a.lut64[i];
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const ws = stage + wave * kSlot;
    const int64_t nvec = a.n / V;
    // chunk c covers vectors [c * 256, (c + 1) * 256); block b takes chunks
    // (b * RUN + r) + it * gridDim * RUN, r = 0..RUN-1
    const int64_t nchunk = (nvec + kBlock - 1) / kBlock;
    const int64_t cstride = (int64_t)gridDim.x * kChunkRun;
    int64_t cbase = (int64_t)blockIdx.x * kChunkRun;
    int run = 0;
    auto vec_of = [&](int64_t cb, int r) { return (cb + r) * kBlock + threadIdx.x; };
    // first element of a chunk: wave-uniform, so base + it stays in SGPRs and
    // each access is SGPR base + 32-bit lane offset (no per-array 64-bit VALU add)
    auto first_of = [&](int64_t cb, int r) { return (cb + r) * (int64_t)(kBlock * V); };
    const unsigned lane_elem = threadIdx.x * (unsigned)V;
    auto advance = [&](int64_t& cb, int& r) {
        if (++r == kChunkRun) { r = 0; cb += cstride; }
    };
    int64_t v = vec_of(cbase, run);
    double dsum_d = 0, dsum_n = 0, dmax_d = -__builtin_huge_val(), dmax_n = -__builtin_huge_val();
    unsigned nan_d = 0, nan_n = 0;   // wave-uniform NaN counts (ballot + popcount)

    auto issue = [&](int64_t first) {
#pragma unroll
        for (int k = 0; k < 14; ++k)
            __builtin_amdgcn_global_load_lds((gptr_t)((a.drv[k] + first) + lane_elem),
                                             (lptr_t)(ws + k * 1024), 16, 0, kDmaNt);
        // sub-dword LDS-DMA lands one dword per lane (measured): read back at lane * 4
        if constexpr (V == 2)
            __builtin_amdgcn_global_load_lds((gptr_t)((a.cls + first) + lane_elem),
                                             (lptr_t)(ws + 14 * 1024), 2, 0, kDmaNt);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)((a.cls + first) + lane_elem),
                                             (lptr_t)(ws + 14 * 1024), 4, 0, kDmaNt);
    };
    if (v < nvec) issue(first_of(cbase, run));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first fill: nothing to overlap with
#pragma nounroll
    for (; cbase + run < nchunk; ) {
        // this iteration's DMA was issued before the previous iteration's two
        // stores: all but the two youngest vector-memory operations must be done
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        // The slot is read with ds_read_b128 in one asm statement that also
        // waits for the data (lgkmcnt(0)): as ordinary LDS loads hipcc would
        // put a full s_waitcnt vmcnt(0) in front of them (it pairs them with the
        // LDS-DMA), which would also wait for the two stores just issued. The
        // returned reads are what allows the refill below (WAR on the slot).
        typedef typename Vec<T, V>::type VT;
        VT in[14];
        unsigned cbits;
        {
            const unsigned base = (unsigned)(uintptr_t)(lptr_t)ws + lane * 16u;
            const unsigned caddr = (unsigned)(uintptr_t)(lptr_t)ws + 14u * 1024u + lane * 4u;   // sub-dword LDS-DMA lands one dword per lane
            if constexpr (V == 2) {
                asm volatile(
                    "ds_read_b128 %0, %15\n\tds_read_b128 %1, %15 offset:1024\n\t"
                    "ds_read_b128 %2, %15 offset:2048\n\tds_read_b128 %3, %15 offset:3072\n\t"
                    "ds_read_b128 %4, %15 offset:4096\n\tds_read_b128 %5, %15 offset:5120\n\t"
                    "ds_read_b128 %6, %15 offset:6144\n\tds_read_b128 %7, %15 offset:7168\n\t"
                    "ds_read_b128 %8, %15 offset:8192\n\tds_read_b128 %9, %15 offset:9216\n\t"
                    "ds_read_b128 %10, %15 offset:10240\n\tds_read_b128 %11, %15 offset:11264\n\t"
                    "ds_read_b128 %12, %15 offset:12288\n\tds_read_b128 %13, %15 offset:13312\n\t"
                    "ds_read_u16 %14, %16\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(in[0]), "=&v"(in[1]), "=&v"(in[2]), "=&v"(in[3]), "=&v"(in[4]),
                      "=&v"(in[5]), "=&v"(in[6]), "=&v"(in[7]), "=&v"(in[8]), "=&v"(in[9]),
                      "=&v"(in[10]), "=&v"(in[11]), "=&v"(in[12]), "=&v"(in[13]), "=&v"(cbits)
                    : "v"(base), "v"(caddr)
                    : "memory");
            } else {
                asm volatile(
                    "ds_read_b128 %0, %15\n\tds_read_b128 %1, %15 offset:1024\n\t"
                    "ds_read_b128 %2, %15 offset:2048\n\tds_read_b128 %3, %15 offset:3072\n\t"
                    "ds_read_b128 %4, %15 offset:4096\n\tds_read_b128 %5, %15 offset:5120\n\t"
                    "ds_read_b128 %6, %15 offset:6144\n\tds_read_b128 %7, %15 offset:7168\n\t"
                    "ds_read_b128 %8, %15 offset:8192\n\tds_read_b128 %9, %15 offset:9216\n\t"
                    "ds_read_b128 %10, %15 offset:10240\n\tds_read_b128 %11, %15 offset:11264\n\t"
                    "ds_read_b128 %12, %15 offset:12288\n\tds_read_b128 %13, %15 offset:13312\n\t"
                    "ds_read_b32 %14, %16\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(in[0]), "=&v"(in[1]), "=&v"(in[2]), "=&v"(in[3]), "=&v"(in[4]),
                      "=&v"(in[5]), "=&v"(in[6]), "=&v"(in[7]), "=&v"(in[8]), "=&v"(in[9]),
                      "=&v"(in[10]), "=&v"(in[11]), "=&v"(in[12]), "=&v"(in[13]), "=&v"(cbits)
                    : "v"(base), "v"(caddr)
                    : "memory");
            }
        }
        int64_t cb_n = cbase;
        int run_n = run;
        advance(cb_n, run_n);
        const int64_t vn = vec_of(cb_n, run_n);
        if (vn < nvec) issue(first_of(cb_n, run_n));
        asm volatile("" ::: "memory");

        if (v < nvec) {   // only the last chunk is ragged
            VT day, night;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                PixelIn<double> x = {(double)in[0][j], (double)in[1][j], (double)in[2][j],
                                     (double)in[3][j], (double)in[4][j], (double)in[5][j],
                                     (double)in[6][j], (double)in[7][j], (double)in[8][j],
                                     (double)in[9][j], (double)in[10][j], (double)in[11][j],
                                     (double)in[12][j], (double)in[13][j]};
                unsigned c = (cbits >> (8 * j)) & 0xffu;
                if (c >= 13u) {
                    atomicOr(a.status, kStatusClassRange);
                    c = 13u;
                }
                const double* l = lut + c;
                ClassPar<double> p;
                p.tmin_close = l[0 * kLutCols];
                p.tmin_open = l[1 * kLutCols];
                p.vpd_open = l[2 * kLutCols];
                p.vpd_close = l[3 * kLutCols];
                p.gl_sh = l[4 * kLutCols];
                p.gl_wv = l[5 * kLutCols];
                p.g_cut = l[6 * kLutCols];
                p.csl = l[7 * kLutCols];
                p.rbl_min = l[8 * kLutCols];
                p.rbl_max = l[9 * kLutCols];
                p.beta = l[10 * kLutCols];
                p.inv_dtmin = l[11 * kLutCols];
                p.inv_dvpd = l[12 * kLutCols];
                p.rbl_slope = l[13 * kLutCols];
                p.inv_beta = l[14 * kLutCols];
                PixelOut<double> o = et_pixel_fast<double>(x, p, tab);
                day[j] = (T)((o.canopy_d + o.soil_d) + o.trans_d);
                night[j] = (T)((o.canopy_n + o.soil_n) + o.trans_n);
                if (DIAG) {
                    const double d = (double)day[j], g = (double)night[j];
                    const bool dn = d != d, gn = g != g;
                    nan_d += (unsigned)__builtin_popcountll(__ballot(dn));
                    nan_n += (unsigned)__builtin_popcountll(__ballot(gn));
                    dsum_d += dn ? 0.0 : d;
                    dsum_n += gn ? 0.0 : g;
                    dmax_d = __builtin_fmax(dmax_d, d);    // maxNum: skips NaN
                    dmax_n = __builtin_fmax(dmax_n, g);
                }
            }
            const int64_t first = first_of(cbase, run);
            __builtin_nontemporal_store(day, reinterpret_cast<VT*>((a.out[0] + first) + lane_elem));
            __builtin_nontemporal_store(night, reinterpret_cast<VT*>((a.out[1] + first) + lane_elem));
        }
        cbase = cb_n;
        run = run_n;
        v = vn;
    }
    if (DIAG) {
        // counts: every lane of a wave holds the wave's total; let lane 0 carry it
        const bool lead = lane == 0;
        double acc[kDiag] = {dsum_d, dsum_n, 0.0, 0.0, lead ? (double)nan_d : 0.0,
                             lead ? (double)nan_n : 0.0, dmax_d, dmax_n};
        diag_block_reduce(acc, a.diag_partial + (int64_t)blockIdx.x * kDiag);
    }
}

// ---- dynamic work distribution (experiment, -> DESIGN.md section 6) -----------
// Same kernel body; what changes is who takes which piece. Waves are persistent
// (the grid is what fits the chip) and every WAVE claims runs of kDynRun
// consecutive 64-vector pieces (kDynRun KiB per array) from a global counter,
// one run ahead, so pieces are handed out in address order to whichever wave is
// ready -- the order a one-shot launch gives (measured 4-5 % faster than a
// static grid-stride for this 14-read + 2-write mix) without giving up the
// LDS-DMA pipeline. The claim is an asm atomic issued by lane 0 in the first
// iteration of a run, in front of that iteration's DMA; the loop's counted
// vmcnt(2) of the next iteration retires it, no extra wait exists.
#ifndef MOD16_DYN_RUN
#define MOD16_DYN_RUN 16
#endif
constexpr int kDynRun = MOD16_DYN_RUN;
// PITCHED: the 14 driver arrays are equally spaced (one slab, as
// RasterEngine.alloc_raster lays them out), so array k's address is
// drv[0] + k * pitch in scalar registers instead of 14 pointers.
template <typename T, bool FAST, bool DIAG, bool PITCHED = false>
__global__ void __launch_bounds__(kBlock) et_kernel_dyn(const EtArgs<T> a) {
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int kSlot = 15 * 1024;   // 14 x (64 lanes x 16 B) + class bytes
    // arithmetic is float64 for both data types (float32 is widened on load,
    // rounded once on store)
    static_assert(FAST, "the LDS-DMA kernel is the FAST production kernel");
    constexpr int kTab = FastMath<double>::kTabDoubles;
    __shared__ double lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    __shared__ __attribute__((aligned(16))) char stage[(kBlock / 64) * kSlot];
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) lut[i] = a.lut64[i];
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const ws = stage + wave * kSlot;
    const int64_t nvec = a.n / V;
    // piece p covers vectors [p * 64, (p + 1) * 64) -- one wave-instruction per
    // array; a run is kDynRun consecutive pieces. Wave g starts with run g; run
    // (nwaves + ticket) is claimed from the global counter while a run is worked.
    const int64_t npiece = (nvec + 63) / 64;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    int64_t cbase = ((int64_t)blockIdx.x * (kBlock / 64) + wave) * kDynRun;   // first piece of the run
    int64_t next_base = npiece;                                               // claimed run (none yet)
    unsigned long long ticket = 0;
    int run = 0;
    auto vec_of = [&](int64_t cb, int r) { return (cb + r) * 64 + lane; };
    // first element of a piece; readfirstlane pins it to SGPRs (the claimed run
    // index arrives through a VGPR), so every access is SGPR base + lane offset
    auto first_of = [&](int64_t cb, int r) {
        const int64_t f = (cb + r) * (int64_t)(64 * V);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)f);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)f >> 32));
        return (int64_t)(((unsigned long long)hi << 32) | lo);
    };
    const unsigned lane_elem = (unsigned)lane * (unsigned)V;
    auto advance = [&](int64_t& cb, int& r) {
        if (++r == kDynRun) { r = 0; cb = next_base; }
    };
    int64_t v = vec_of(cbase, run);
    double dsum_d = 0, dsum_n = 0, dmax_d = -__builtin_huge_val(), dmax_n = -__builtin_huge_val();
    unsigned nan_d = 0, nan_n = 0;   // wave-uniform NaN counts (ballot + popcount)

    // scalar loads of drv[0..13] from the kernel-argument segment (EtArgs::drv
    // is at offset 0); called ahead of the wait for the DMA so that their
    // latency is hidden
    auto load_ptrs = [&](const char* (&dptr)[14]) {
        if constexpr (!PITCHED) {
            typedef const __attribute__((address_space(4))) char* kptr_t;
            kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(ka));
#pragma unroll
            for (int k = 0; k < 14; ++k)
                dptr[k] = *reinterpret_cast<const char* const __attribute__((address_space(4)))*>(ka + 8 * k);
        }
    };
    auto issue = [&](int64_t first, const char* (&dptr)[14]) {
        if constexpr (PITCHED) {
            // Opaque copies of the loop invariants keep hipcc from hoisting 14
            // per-lane 64-bit addresses (28 VGPRs) and 14 LDS offsets out of the
            // loop: each access is scalar (running array base) + 32-bit lane
            // byte offset, M0 = slot + k KiB by one s_add.
            unsigned lb = lane_elem * (unsigned)sizeof(T);
            unsigned wl = (unsigned)(uintptr_t)(lptr_t)ws;
            int64_t pitch_b = a.drv_pitch * (int64_t)sizeof(T);
            asm volatile("" : "+v"(lb));
            asm volatile("" : "+s"(wl));
            asm volatile("" : "+s"(pitch_b));
            const char* pk = reinterpret_cast<const char*>(a.drv[0] + first);
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                __builtin_amdgcn_global_load_lds((gptr_t)(pk + lb), (lptr_t)(uintptr_t)(wl + k * 1024),
                                                 16, 0, kDmaNt);
                pk += pitch_b;
            }
            const char* pc = reinterpret_cast<const char*>(a.cls + first);
            // sub-dword LDS-DMA lands one dword per lane (measured): read back at lane * 4
            if constexpr (V == 2)
                __builtin_amdgcn_global_load_lds((gptr_t)(pc + (lb >> 3)), (lptr_t)(uintptr_t)(wl + 14 * 1024),
                                                 2, 0, kDmaNt);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)(pc + (lb >> 2)), (lptr_t)(uintptr_t)(wl + 14 * 1024),
                                                 4, 0, kDmaNt);
        } else {
            // 14 independent pointers, re-read from the kernel-argument segment
            // by the caller (dptr) so that no register holds them across the
            // arithmetic
            unsigned lb = lane_elem * (unsigned)sizeof(T);
            unsigned wl = (unsigned)(uintptr_t)(lptr_t)ws;
            asm volatile("" : "+v"(lb));
            asm volatile("" : "+s"(wl));
            const int64_t first_b = first * (int64_t)sizeof(T);
#pragma unroll
            for (int k = 0; k < 14; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)((dptr[k] + first_b) + lb),
                                                 (lptr_t)(uintptr_t)(wl + k * 1024), 16, 0, kDmaNt);
            const char* pc = reinterpret_cast<const char*>(a.cls + first);
            if constexpr (V == 2)
                __builtin_amdgcn_global_load_lds((gptr_t)(pc + (lb >> 3)), (lptr_t)(uintptr_t)(wl + 14 * 1024),
                                                 2, 0, kDmaNt);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)(pc + (lb >> 2)), (lptr_t)(uintptr_t)(wl + 14 * 1024),
                                                 4, 0, kDmaNt);
        }
    };
    {
        const char* dptr[14];
        load_ptrs(dptr);
        if (v < nvec) issue(first_of(cbase, run), dptr);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first fill: nothing to overlap with
    bool flushed = false;
#pragma nounroll
    for (; cbase + run < npiece; ) {
        // this iteration's DMA (and, one iteration after a claim, the claim's
        // atomic) was issued before the previous iteration's two stores: all but
        // the two youngest vector-memory operations must be done
        // (after a diagnostics flush there is one more store behind them)
        const char* dptr[14];
        load_ptrs(dptr);
        if (flushed) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if (run == 1) {   // the claim issued in the previous iteration has returned
            asm volatile("" : "+v"(ticket));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ticket);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ticket >> 32));
            next_base = (nwaves + (int64_t)(((unsigned long long)hi << 32) | lo)) * kDynRun;
        }
        // The slot is read with ds_read_b128 in one asm statement that also
        // waits for the data (lgkmcnt(0)): as ordinary LDS loads hipcc would
        // put a full s_waitcnt vmcnt(0) in front of them (it pairs them with the
        // LDS-DMA), which would also wait for the two stores just issued. The
        // returned reads are what allows the refill below (WAR on the slot).
        typedef typename Vec<T, V>::type VT;
        VT in[14];
        unsigned cbits;
        {
            const unsigned base = (unsigned)(uintptr_t)(lptr_t)ws + lane * 16u;
            const unsigned caddr = (unsigned)(uintptr_t)(lptr_t)ws + 14u * 1024u + lane * 4u;   // sub-dword LDS-DMA lands one dword per lane
            if constexpr (V == 2) {
                asm volatile(
                    "ds_read_b128 %0, %15\n\tds_read_b128 %1, %15 offset:1024\n\t"
                    "ds_read_b128 %2, %15 offset:2048\n\tds_read_b128 %3, %15 offset:3072\n\t"
                    "ds_read_b128 %4, %15 offset:4096\n\tds_read_b128 %5, %15 offset:5120\n\t"
                    "ds_read_b128 %6, %15 offset:6144\n\tds_read_b128 %7, %15 offset:7168\n\t"
                    "ds_read_b128 %8, %15 offset:8192\n\tds_read_b128 %9, %15 offset:9216\n\t"
                    "ds_read_b128 %10, %15 offset:10240\n\tds_read_b128 %11, %15 offset:11264\n\t"
                    "ds_read_b128 %12, %15 offset:12288\n\tds_read_b128 %13, %15 offset:13312\n\t"
                    "ds_read_u16 %14, %16\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(in[0]), "=&v"(in[1]), "=&v"(in[2]), "=&v"(in[3]), "=&v"(in[4]),
                      "=&v"(in[5]), "=&v"(in[6]), "=&v"(in[7]), "=&v"(in[8]), "=&v"(in[9]),
                      "=&v"(in[10]), "=&v"(in[11]), "=&v"(in[12]), "=&v"(in[13]), "=&v"(cbits)
                    : "v"(base), "v"(caddr)
                    : "memory");
            } else {
                asm volatile(
                    "ds_read_b128 %0, %15\n\tds_read_b128 %1, %15 offset:1024\n\t"
                    "ds_read_b128 %2, %15 offset:2048\n\tds_read_b128 %3, %15 offset:3072\n\t"
                    "ds_read_b128 %4, %15 offset:4096\n\tds_read_b128 %5, %15 offset:5120\n\t"
                    "ds_read_b128 %6, %15 offset:6144\n\tds_read_b128 %7, %15 offset:7168\n\t"
                    "ds_read_b128 %8, %15 offset:8192\n\tds_read_b128 %9, %15 offset:9216\n\t"
                    "ds_read_b128 %10, %15 offset:10240\n\tds_read_b128 %11, %15 offset:11264\n\t"
                    "ds_read_b128 %12, %15 offset:12288\n\tds_read_b128 %13, %15 offset:13312\n\t"
                    "ds_read_b32 %14, %16\n\ts_waitcnt lgkmcnt(0)"
                    : "=&v"(in[0]), "=&v"(in[1]), "=&v"(in[2]), "=&v"(in[3]), "=&v"(in[4]),
                      "=&v"(in[5]), "=&v"(in[6]), "=&v"(in[7]), "=&v"(in[8]), "=&v"(in[9]),
                      "=&v"(in[10]), "=&v"(in[11]), "=&v"(in[12]), "=&v"(in[13]), "=&v"(cbits)
                    : "v"(base), "v"(caddr)
                    : "memory");
            }
        }
        if (run == 0 && lane == 0) {   // claim the next run, ahead of this iteration's DMA
            const unsigned long long one = 1;
            asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0"
                         : "=v"(ticket) : "v"(a.dyn_counter), "v"(one) : "memory");
        }
        int64_t cb_n = cbase;
        int run_n = run;
        advance(cb_n, run_n);
        const int64_t vn = vec_of(cb_n, run_n);
        if (vn < nvec) issue(first_of(cb_n, run_n), dptr);
        asm volatile("" ::: "memory");

        if (v < nvec) {   // only the last chunk is ragged
            VT day, night;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                PixelIn<double> x = {(double)in[0][j], (double)in[1][j], (double)in[2][j],
                                     (double)in[3][j], (double)in[4][j], (double)in[5][j],
                                     (double)in[6][j], (double)in[7][j], (double)in[8][j],
                                     (double)in[9][j], (double)in[10][j], (double)in[11][j],
                                     (double)in[12][j], (double)in[13][j]};
                unsigned c = (cbits >> (8 * j)) & 0xffu;
                if (c >= 13u) {
                    atomicOr(a.status, kStatusClassRange);
                    c = 13u;
                }
                const double* l = lut + c;
                ClassPar<double> p;
                p.tmin_close = l[0 * kLutCols];
                p.tmin_open = l[1 * kLutCols];
                p.vpd_open = l[2 * kLutCols];
                p.vpd_close = l[3 * kLutCols];
                p.gl_sh = l[4 * kLutCols];
                p.gl_wv = l[5 * kLutCols];
                p.g_cut = l[6 * kLutCols];
                p.csl = l[7 * kLutCols];
                p.rbl_min = l[8 * kLutCols];
                p.rbl_max = l[9 * kLutCols];
                p.beta = l[10 * kLutCols];
                p.inv_dtmin = l[11 * kLutCols];
                p.inv_dvpd = l[12 * kLutCols];
                p.rbl_slope = l[13 * kLutCols];
                p.inv_beta = l[14 * kLutCols];
                PixelOut<double> o = et_pixel_fast<double>(x, p, tab);
                day[j] = (T)((o.canopy_d + o.soil_d) + o.trans_d);
                night[j] = (T)((o.canopy_n + o.soil_n) + o.trans_n);
                if (DIAG) {
                    const double d = (double)day[j], g = (double)night[j];
                    const bool dn = d != d, gn = g != g;
                    nan_d += (unsigned)__builtin_popcountll(__ballot(dn));
                    nan_n += (unsigned)__builtin_popcountll(__ballot(gn));
                    dsum_d += dn ? 0.0 : d;
                    dsum_n += gn ? 0.0 : g;
                    dmax_d = __builtin_fmax(dmax_d, d);    // maxNum: skips NaN
                    dmax_n = __builtin_fmax(dmax_n, g);
                }
            }
            const int64_t first = first_of(cbase, run);
            __builtin_nontemporal_store(day, reinterpret_cast<VT*>((a.out[0] + first) + lane_elem));
            __builtin_nontemporal_store(night, reinterpret_cast<VT*>((a.out[1] + first) + lane_elem));
        }
        // Diagnostics are flushed per RUN, not per wave: a run is always the same
        // pixels in the same order whichever wave claimed it, so the partials --
        // and the fixed-order sums over them -- do not depend on the dynamic
        // schedule. Butterfly reduction (every lane ends with the same bits),
        // then lanes 0..7 store the 8 fields with one instruction.
        flushed = DIAG && (run_n == 0 || cb_n + run_n >= npiece);
        if (flushed) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                dsum_d += __shfl_xor(dsum_d, off, 64);
                dsum_n += __shfl_xor(dsum_n, off, 64);
                dmax_d = __builtin_fmax(dmax_d, __shfl_xor(dmax_d, off, 64));
                dmax_n = __builtin_fmax(dmax_n, __shfl_xor(dmax_n, off, 64));
            }
            // the counts are incremented under the ragged-piece mask: lane 0, always
            // active there, holds the wave's totals
            const double cnt_d = (double)__builtin_amdgcn_readfirstlane(nan_d);
            const double cnt_n = (double)__builtin_amdgcn_readfirstlane(nan_n);
            const double f = lane == 0 ? dsum_d : lane == 1 ? dsum_n : lane == 4 ? cnt_d
                           : lane == 5 ? cnt_n : lane == 6 ? dmax_d : lane == 7 ? dmax_n : 0.0;
            if (lane < kDiag) a.diag_partial[(cbase / kDynRun) * kDiag + lane] = f;
            dsum_d = dsum_n = 0.0;
            dmax_d = dmax_n = -__builtin_huge_val();
            nan_d = nan_n = 0;
        }
        cbase = cb_n;
        run = run_n;
        v = vn;
    }
}

// One stage of the fixed-order reduction of per-run partials: block b reduces
// partials [b * per, (b + 1) * per) to out[b].
__global__ void __launch_bounds__(kBlock) diag_stage_kernel(const double* partial, int64_t count,
                                                            int64_t per, double* out) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = (lo + per < count) ? lo + per : count;
    for (int64_t b = lo + threadIdx.x; b < hi; b += kBlock) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = partial[b * kDiag + k];
        diag_merge(acc, o);
    }
    diag_block_reduce(acc, out + (int64_t)blockIdx.x * kDiag);
}

// Sum of the per-block partials of et_kernel_dma, fixed order (thread
// t adds partials t, t + 1024, ...; then a fixed wave / block tree);
// n_valid = n - n_nan. 1024 threads keep the dependent-load chain short.
constexpr int kFinalBlock = 1024;
__global__ void __launch_bounds__(kFinalBlock) diag_final_fused_kernel(const double* partial,
                                                                       int nblocks, int64_t n,
                                                                       double* out) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    for (int b = threadIdx.x; b < nblocks; b += kFinalBlock) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = partial[(int64_t)b * kDiag + k];
        diag_merge(acc, o);
    }
    __shared__ double sm[kFinalBlock / 64][kDiag];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = __shfl_down(acc[k], off, 64);
        diag_merge(acc, o);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
        for (int k = 0; k < kDiag; ++k) sm[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kFinalBlock / 64; ++w) {
            double o[kDiag];
            for (int k = 0; k < kDiag; ++k) o[k] = sm[w][k];
            diag_merge(acc, o);
        }
        out[0] = acc[0];
        out[1] = acc[1];
        out[2] = (double)n - acc[4];
        out[3] = (double)n - acc[5];
        out[4] = acc[4];
        out[5] = acc[5];
        out[6] = acc[6];
        out[7] = acc[7];
    }
}

// --------------------------------------------------------- synthetic fields
// Counter-based generator: value = f(seed, step,
// variable, global pixel index) through a splitmix64 finaliser, so a raster
// tiled over any number of GPUs is the same field.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double u01(uint64_t seed, uint64_t step, unsigned var, uint64_t pix) {
    uint64_t h = mix64(seed + 0x9e3779b97f4a7c15ull * (step * 64ull + var + 1ull));
    h = mix64(h ^ (pix * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull));
    return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T> struct SynthArgs {
    uint8_t* cls;
    T* drv[14];
    uint64_t seed;
    int64_t step, offset, n;
};

template <typename T>
__global__ void __launch_bounds__(kBlock) synth_kernel(const SynthArgs<T> a) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += stride) {
        const uint64_t pix = (uint64_t)(a.offset + i);
        const uint64_t st = (uint64_t)a.step;
        auto U = [&](unsigned var, double lo, double hi) {
            return lo + (hi - lo) * u01(a.seed, st, var, pix);
        };
        double t_d = U(0, 255.0, 305.0);
        double t_n = t_d - U(1, 0.0, 12.0);
        double tmin = t_n - U(2, 0.0, 3.0);
        double t_ann = 265.0 + 35.0 * u01(a.seed, 0, 3, pix);   // climatology is static
        double rh_d = U(4, 0.05, 1.0), rh_n = U(5, 0.05, 1.0);
        double vpd_d = svp_exact<double>(t_d) * (1.0 - rh_d);
        double vpd_n = svp_exact<double>(t_n) * (1.0 - rh_n);
        double fpar = U(11, 0.02, 0.89), lai = U(12, 0.13, 5.34);
        double r = u01(a.seed, st, 13, pix);       // exact 0 / 1 / NaN specials
        fpar = (r < 0.01) ? 0.0 : ((r < 0.015) ? 1.0 : ((r < 0.02) ? __builtin_nan("") : fpar));
        r = u01(a.seed, st, 14, pix);
        lai = (r < 0.01) ? 0.0 : ((r < 0.015) ? __builtin_nan("") : lai);
        double rc = u01(a.seed, 0, 15, pix);       // land cover is static
        unsigned c = (unsigned)(u01(a.seed, 0, 16, pix) * 11.0) % 11u + 1u;   // 1..11
        c = (c == 11u) ? 12u : c;                  // PFT_VALID = 1..10, 12
        c = (rc < 0.01) ? 0u : ((rc < 0.02) ? 11u : c);
        if (a.cls) a.cls[i] = (uint8_t)c;
        a.drv[0][i] = (T)U(6, -100.0, 0.0);        // lw_net_day
        a.drv[1][i] = (T)U(7, -50.0, 0.0);         // lw_net_night
        a.drv[2][i] = (T)U(8, 0.0, 360.0);         // sw_rad_day
        a.drv[3][i] = (T)0;                        // sw_rad_night
        a.drv[4][i] = (T)U(9, 0.1, 0.22);          // sw_albedo
        a.drv[5][i] = (T)t_d;
        a.drv[6][i] = (T)t_n;
        a.drv[7][i] = (T)t_ann;
        a.drv[8][i] = (T)tmin;
        a.drv[9][i] = (T)vpd_d;
        a.drv[10][i] = (T)vpd_n;
        a.drv[11][i] = (T)U(10, 70000.0, 101340.0);  // pressure
        a.drv[12][i] = (T)fpar;
        a.drv[13][i] = (T)lai;
    }
}

}  // namespace mod16
