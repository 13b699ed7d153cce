// gfx950 kernels of libmod16hip: the plain fused ET pixel kernel (every input
// shape, EXACT arithmetic), the diagnostics reduction and the synthetic driver
// generator. The production pipeline for dense class rasters is
// mod16_stream.hpp; launch code is in mod16_capi.hip.
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
// A dynamically scheduled launch of the pipeline kernel (mod16_stream.hpp) did not process every
// run of its raster: it found a ticket counter that was not at zero (an earlier launch on that
// counter ended abnormally, or two launches in flight shared it). Such a launch keeps the previous
// step's outputs and partials for the runs it did not reach -- plausible numbers -- so it is
// detected: every run's partial carries the launch's serial number (field kSerialField) and what
// runs behind the pipeline kernel counts the runs that carry it.
constexpr unsigned kStatusIncomplete = 2u;
constexpr int kSerialField = 3;      // the field of a run's diagnostics partial that carries the marker
__device__ __forceinline__ double launch_marker(unsigned serial) { return (double)((serial & 0xffffu) + 1u); }

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
    // 2-level broadcasting of the reference's (N,) against (T, N) inputs
    // (mod16/__init__.py:180-181): pixel g = base + i of the flattened (T, N) raster
    // reads element g % inner of a ROW input ((N,), one value per site) and g / inner
    // of a COL input ((T, 1), one value per time step). Only the one-pixel-per-thread
    // kernels look at these (a ROW / COL input sends the whole call there).
    uint32_t row_drv, col_drv, row_par, col_par;
    uint32_t cls_mode;     // MOD16_BC_* of the class raster
    int64_t inner, base;
};

template <typename T, int V> struct Vec;
template <> struct Vec<double, 2> { typedef double type __attribute__((ext_vector_type(2))); };
template <> struct Vec<double, 1> { typedef double type; };
template <> struct Vec<float, 4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct Vec<float, 2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec<float, 1> { typedef float type; };

// index of pixel (i local, g global) in an input of the given broadcast kind
struct BcIndex {
    int64_t i, r, c;       // local dense index, g / inner, g % inner
    __device__ __forceinline__ int64_t of(bool dense, bool row, bool col) const {
        return dense ? i : (row ? c : (col ? r : 0));
    }
};

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
__global__ void __launch_bounds__(kBlock, FAST ? 2 : 1) et_kernel(const EtArgs<T> a) {
    // FAST always computes in float64 (float32 data are widened on load and
    // the result rounded once on store); EXACT computes in the data type, as
    // numpy does for the reference code.
    typedef typename std::conditional<FAST, double, T>::type C;
    constexpr int kTab = FAST ? FastMath<double>::kTabDoubles : 1;
    __shared__ C lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    if constexpr (FAST) ignore_signalling_nans();       // the domain guard's NaN-ignoring chain
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
        unsigned cbits = 0;
        T pin[11][V];
        if constexpr (V == 1 && !DENSE) {
            // every broadcast kind (scalar, dense, (N,) rows, (T, 1) columns)
            BcIndex ix = {i, 0, 0};
            if ((a.row_drv | a.col_drv | a.row_par | a.col_par) != 0u || a.cls_mode >= 2u) {
                const int64_t g = a.base + i;
                ix.r = g / a.inner;
                ix.c = g - ix.r * a.inner;
            }
#pragma unroll
            for (int k = 0; k < 14; ++k)
                in[k][0] = a.drv[k][ix.of((a.dense_drv >> k) & 1u, (a.row_drv >> k) & 1u, (a.col_drv >> k) & 1u)];
            if (LUT) {
                cbits = a.cls[ix.of(a.cls_mode == 1u, a.cls_mode == 2u, a.cls_mode == 3u)];
            } else {
#pragma unroll
                for (int k = 0; k < 11; ++k)
                    pin[k][0] = a.par[k][ix.of((a.dense_par >> k) & 1u, (a.row_par >> k) & 1u, (a.col_par >> k) & 1u)];
            }
        } else {
#pragma unroll
        for (int k = 0; k < 14; ++k)
            load_vec<T, V, DENSE>(a.drv[k], (a.dense_drv >> k) & 1u, i, in[k]);
        if (LUT) {
            if constexpr (V == 1) cbits = a.cls[i];
            else if constexpr (V == 2) cbits = *reinterpret_cast<const uint16_t*>(a.cls + i);
            else cbits = *reinterpret_cast<const uint32_t*>(a.cls + i);
        } else {
#pragma unroll
            for (int k = 0; k < 11; ++k)
                load_vec<T, V, false>(a.par[k], (a.dense_par >> k) & 1u, i, pin[k]);
        }
        }
        T res[10][V];
        unsigned bad = 0;      // FAST: bit j set = pixel j is outside the domain of that arithmetic
        auto emit = [&](int j, const PixelOut<C>& o) {
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
        };
        // the class parameters of a pixel: its column of the table in LDS (class code c) or
        // the 11 per-pixel / scalar parameter inputs
        auto params = [&](unsigned c, auto&& par) {
            ClassPar<C> p;
            if (LUT) {
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
                p.tmin_close = (C)par(0);
                p.tmin_open = (C)par(1);
                p.vpd_open = (C)par(2);
                p.vpd_close = (C)par(3);
                p.gl_sh = (C)par(4);
                p.gl_wv = (C)par(5);
                p.g_cut = (C)par(6);
                p.csl = (C)par(7);
                p.rbl_min = (C)par(8);
                p.rbl_max = (C)par(9);
                p.beta = (C)par(10);
                if (FAST) p.derive();
            }
            return p;
        };
#pragma unroll
        for (int j = 0; j < V; ++j) {
            PixelIn<C> x = {(C)in[0][j], (C)in[1][j], (C)in[2][j], (C)in[3][j], (C)in[4][j],
                            (C)in[5][j], (C)in[6][j], (C)in[7][j], (C)in[8][j], (C)in[9][j],
                            (C)in[10][j], (C)in[11][j], (C)in[12][j], (C)in[13][j]};
            unsigned c = 0;
            if (LUT) {
                c = (cbits >> (8 * j)) & 0xffu;
                if (c >= 13u) {   // numpy would raise IndexError: flag it, give NaN
                    atomicOr(a.status, kStatusClassRange);
                    c = 13u;
                }
            }
            const ClassPar<C> p = params(c, [&](int k) { return pin[k][j]; });
            PixelOut<C> o;
            if constexpr (FAST) {
                o = et_pixel_fast<double, PET>(x, p, tab);
                bad |= fast_out_of_domain(x) ? 1u << j : 0u;
            } else {
                o = et_pixel_exact<T, PET>(x, p);
            }
            emit(j, o);
        }
        if constexpr (FAST) {
            // pixels outside the domain of the strength-reduced arithmetic: again, in the
            // reference's operation order (mod16_physics.hpp, "domain guard"); a wave enters only
            // if one of its lanes has such a pixel, every lane takes its own one at a time
            if (__builtin_expect(__any(bad != 0u), 0)) {
                unsigned pend = bad;
#pragma nounroll
                while (__any(pend != 0u)) {
                    if (pend != 0u) {
                        const int j = __builtin_ctz(pend);
                        pend &= pend - 1u;
                        // V > 1: the pixel's inputs again from memory (dense array or broadcast
                        // scalar) -- keeping the V pixels' 14 + 11 inputs alive across this branch
                        // would push the kernel over its register budget
                        auto drv = [&](int k) -> double {
                            if constexpr (V == 1) return (double)in[k][0];
                            else return (double)((DENSE || ((a.dense_drv >> k) & 1u)) ? a.drv[k][i + j] : a.drv[k][0]);
                        };
                        auto par = [&](int k) -> double {
                            if constexpr (V == 1) return (double)pin[k][0];
                            else return (double)(((a.dense_par >> k) & 1u) ? a.par[k][i + j] : a.par[k][0]);
                        };
                        const PixelIn<double> x = {drv(0), drv(1), drv(2), drv(3), drv(4), drv(5), drv(6),
                                                   drv(7), drv(8), drv(9), drv(10), drv(11), drv(12), drv(13)};
                        unsigned c = (cbits >> (8 * j)) & 0xffu;
                        c = c >= 13u ? 13u : c;
                        const ClassPar<double> p = params(c, par);
                        const PixelOut<double> o = et_pixel_exact<double, PET, true>(x, p);
#pragma unroll
                        for (int jj = 0; jj < V; ++jj)
                            if (j == jj) emit(jj, o);
                    }
                }
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

static __global__ void __launch_bounds__(kBlock) diag_final_kernel(const double* partial, int nblocks,
                                                            double* out) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    for (int b = threadIdx.x; b < nblocks; b += kBlock) {
        double o[kDiag];
        for (int k = 0; k < kDiag; ++k) o[k] = partial[(int64_t)b * kDiag + k];
        diag_merge(acc, o);
    }
    diag_block_reduce(acc, out);
}

// One stage of the fixed-order reduction of per-run partials: block b reduces
// partials [b * per, (b + 1) * per) to out[b].
// serial_word / status (trusted launches): see diag_final_fused_kernel -- the kernel that reads the
// runs' own partials compares every run's marker.
static __global__ void __launch_bounds__(kBlock) diag_stage_kernel(const double* partial, int64_t count,
                                                            int64_t per, double* out,
                                                            const unsigned* serial_word = nullptr,
                                                            unsigned* status = nullptr) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    const int64_t lo = (int64_t)blockIdx.x * per;
    const int64_t hi = (lo + per < count) ? lo + per : count;
    double marker = 0.0;
    if (serial_word)
        marker = launch_marker(__hip_atomic_load(serial_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u);
    bool stale = false;
    for (int64_t b = lo + threadIdx.x; b < hi; b += kBlock) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = partial[b * kDiag + k];
        stale |= o[kSerialField] != marker;
        diag_merge(acc, o);
    }
    if (serial_word && stale) atomicOr(status, kStatusIncomplete);
    diag_block_reduce(acc, out + (int64_t)blockIdx.x * kDiag);
}

// Sum of the per-run partials of et_stream_kernel, fixed order (thread
// t adds partials t, t + 1024, ...; then a fixed wave / block tree);
// n_valid = n - n_nan. 1024 threads keep the dependent-load chain short.
constexpr int kFinalBlock = 1024;
// serial_word / status (trusted launches, which have no kernel behind them that revisits the runs;
// given to the kernel that reads the runs' OWN partials -- this one, or diag_stage_kernel for a
// two-level sum): the launch's serial number AFTER the pipeline kernel incremented it and the status
// word -- field kSerialField of every partial must hold this launch's marker.
static __global__ void __launch_bounds__(kFinalBlock) diag_final_fused_kernel(const double* partial,
                                                                       int nblocks, int64_t n,
                                                                       double* out,
                                                                       const unsigned* serial_word = nullptr,
                                                                       unsigned* status = nullptr) {
    double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
    // EVERY run's marker is compared (round 6; a sum of the markers against nruns * marker passed
    // stale partials whose markers lay m - 1 and m + 1 in equal number)
    double marker = 0.0;
    if (serial_word)
        marker = launch_marker(__hip_atomic_load(serial_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u);
    bool stale = false;
    for (int b = threadIdx.x; b < nblocks; b += kFinalBlock) {
        double o[kDiag];
#pragma unroll
        for (int k = 0; k < kDiag; ++k) o[k] = partial[(int64_t)b * kDiag + k];
        stale |= o[kSerialField] != marker;
        diag_merge(acc, o);
    }
    if (serial_word && stale) atomicOr(status, kStatusIncomplete);
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

// ------------------------------------------------ parameter rasters -> class raster
// The reference's multi-class idiom gathers the BPLUT per pixel (mod16/utils.py:81-117, notebook
// cell 32: MOD16({k: bplut[k][pft_map]})): eleven parameter rasters that hold, pixel for pixel, one
// of at most 13 rows. This kernel turns them back into a class raster (mod16_classify_*): every
// pixel's eleven values are compared BIT FOR BIT (a NaN row -- an invalid class -- matches itself)
// with `nrows` candidate rows and the pixel gets the index of the first row that matches; a pixel that
// matches none reports its index (the smallest such index of the launch, so that the caller can take
// that pixel's row in and try again) -- the production pipeline then reads 14 drivers + 1 byte per
// pixel where the plain kernel read 25 arrays. One pass over the 11 rasters, HBM-bound (89 B/pixel
// in float64): a hash of the pixel's values picks the candidate row.
template <typename T> struct ParamBits;
template <> struct ParamBits<double> { typedef unsigned long long type; };
template <> struct ParamBits<float> { typedef unsigned type; };
constexpr int kClassRows = 13, kClassPars = 11;

template <typename T> struct ClassifyArgs {
    const T* par[kClassPars];
    uint32_t dense;            // bit k set: parameter k is a dense raster, else ONE value (par[k][0])
    const T* rows;             // device [nrows][11]: the candidate rows
    int nrows;
    int64_t n;
    uint8_t* cls;
    unsigned long long* unmatched;     // device word, UINT64_MAX before the launch: atomicMin of the index of a pixel that matches no row
};

template <typename U> __device__ __forceinline__ unsigned mix_param(unsigned h, U bits) {
    const unsigned lo = (unsigned)bits, hi = sizeof(U) == 8 ? (unsigned)((unsigned long long)bits >> 32) : 0u;
    return (h ^ lo ^ (hi * 0x85ebca6bu)) * 0x9e3779b1u;
}

template <typename T>
__global__ void __launch_bounds__(kBlock) classify_kernel(const ClassifyArgs<T> a) {
    typedef typename ParamBits<T>::type U;
    __shared__ U rows[kClassRows][kClassPars];
    __shared__ unsigned row_hash[kClassRows];
    for (int i = threadIdx.x; i < a.nrows * kClassPars; i += kBlock)
        rows[i / kClassPars][i % kClassPars] = reinterpret_cast<const U*>(a.rows)[i];
    __syncthreads();
    if ((int)threadIdx.x < a.nrows) {
        unsigned h = 0x1234567u;
        for (int k = 0; k < kClassPars; ++k) h = mix_param<U>(h, rows[threadIdx.x][k]);
        row_hash[threadIdx.x] = h;
    }
    __syncthreads();
    U one[kClassPars];                 // the values of the parameters that are not rasters
#pragma unroll
    for (int k = 0; k < kClassPars; ++k) one[k] = ((a.dense >> k) & 1u) ? U(0) : *reinterpret_cast<const U*>(a.par[k]);
    const int64_t step = (int64_t)gridDim.x * kBlock;
    unsigned long long missing = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < a.n; i += step) {
        U v[kClassPars];
        unsigned h = 0x1234567u;
#pragma unroll
        for (int k = 0; k < kClassPars; ++k) {
            v[k] = ((a.dense >> k) & 1u) ? reinterpret_cast<const U*>(a.par[k])[i] : one[k];
            h = mix_param<U>(h, v[k]);
        }
        int found = -1;
        for (int r = 0; r < a.nrows && found < 0; ++r) {
            if (row_hash[r] != h) continue;
            bool same = true;
#pragma unroll
            for (int k = 0; k < kClassPars; ++k) same &= rows[r][k] == v[k];
            found = same ? r : -1;
        }
        a.cls[i] = (uint8_t)(found < 0 ? 0 : found);
        if (found < 0 && (unsigned long long)i < missing) missing = (unsigned long long)i;
    }
    if (missing != ~0ull) atomicMin(a.unmatched, missing);
}

// ------------------------------------------------------- copy (measurement aid)
// One-shot 16-byte-per-lane copy, the reference point "measured copy bandwidth"
// of SURVEY.md section 8d next to the 8 TB/s nominal peak (mod16_measure_copy).
typedef float copy_vec_t __attribute__((ext_vector_type(4)));
static __global__ void __launch_bounds__(kBlock) copy_kernel(const copy_vec_t* __restrict__ src,
                                                      copy_vec_t* __restrict__ dst, int64_t nvec) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < nvec) dst[i] = src[i];
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
    // tiled rasters: element index of pixel i = (i >> tile_shift) * row + (i & (2^tile_shift - 1));
    // plain arrays: tile_shift = 62 (one tile)
    int tile_shift;
    int64_t drv_row, cls_row;
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
        const int64_t tile = i >> a.tile_shift, within = i - (tile << a.tile_shift);
        const int64_t e = tile * a.drv_row + within;
        if (a.cls) a.cls[tile * a.cls_row + within] = (uint8_t)c;
        a.drv[0][e] = (T)U(6, -100.0, 0.0);        // lw_net_day
        a.drv[1][e] = (T)U(7, -50.0, 0.0);         // lw_net_night
        a.drv[2][e] = (T)U(8, 0.0, 360.0);         // sw_rad_day
        a.drv[3][e] = (T)0;                        // sw_rad_night
        a.drv[4][e] = (T)U(9, 0.1, 0.22);          // sw_albedo
        a.drv[5][e] = (T)t_d;
        a.drv[6][e] = (T)t_n;
        a.drv[7][e] = (T)t_ann;
        a.drv[8][e] = (T)tmin;
        a.drv[9][e] = (T)vpd_d;
        a.drv[10][e] = (T)vpd_n;
        a.drv[11][e] = (T)U(10, 70000.0, 101340.0);  // pressure
        a.drv[12][e] = (T)fpar;
        a.drv[13][e] = (T)lai;
    }
}

}  // namespace mod16
