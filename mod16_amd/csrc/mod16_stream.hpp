// The production pipeline of the forward run for dense class rasters
// (16-byte-aligned arrays, BPLUT parameters, FAST arithmetic): one kernel
// template over the set of arrays and the pixel function.
//
//   kStreamTotals         14 drivers + class -> day, night          (129 B/pixel in float64)
//                         MOD16.evapotranspiration(), mod16/__init__.py:675-793
//   kStreamPet            ... -> day, night, PET day, PET night      (145)
//                         (reference README.md:404-424, :546-602)
//   kStreamSep8 / Sep6    ... -> [day, night,] the six components    (177 / 161)
//                         (separate=True, mod16/__init__.py:789-793)
//   kStreamRaw*           14 raw drivers + class + uint8 fPAR/LAI [+ hours of
//                         daylight] -> day, night [, 8-day total]    (131 / 139 / 147)
//                         (calibration.py:380-423, verify2.py:113-115)
//
// How the bytes move. At ~175-200 VGPRs only two waves fit a SIMD, too few to
// hide HBM latency behind other waves, and there is no room for a second
// register set to prefetch into. So each wave owns one LDS slot (NW KiB + NB x
// 256 B) and streams the NEXT iteration's NW driver vectors and NB byte
// vectors into it with global_load_lds (LDS-DMA, no VGPR destination) while it
// computes the current one:
//     counted s_waitcnt vmcnt(NOUT) -> ds_read_b128 x NW -> issue next DMA ->
//     compute -> NOUT non-temporal 16-byte stores.
// The slot is private to the wave that fills it, so no barrier is involved: the
// wave's own counted vmcnt orders its ds_reads behind its DMA. Every byte is
// touched once, so both directions use the non-temporal policy (+3-4 % on this
// read/write mix, profiles/r01_probe_streams_hbm_roof.txt).
//
// Who takes which pixels. Waves are persistent (the grid is what fits the
// chip: 2 blocks per CU) and every WAVE claims runs of kDynRun consecutive
// 64-vector pieces (kDynRun KiB per array) from a global ticket counter, one
// run ahead, so pieces are handed out in address order to whichever wave is
// ready -- the order a one-shot launch gives (4-5 % faster than a static
// grid-stride for a 14-read + 2-write mix) without giving up the LDS-DMA
// pipeline. The claim is an asm atomic issued by lane 0 in the first
// iteration of a run, in front of that iteration's DMA; the loop's counted
// vmcnt of the next iteration retires it, no extra wait exists.
//
// Diagnostics (sums, NaN counts, maxima of day / night) are accumulated while
// the outputs are in registers and flushed per RUN: a run is always the same
// pixels in the same order whichever wave claimed it, so the partials -- and
// the fixed-order sums over them -- do not depend on the dynamic schedule.
// (The variant without the accumulation compiles to 50-60 more VGPRs and runs
// slower, so there is only this one.)
#pragma once
#include "mod16_kernels.hpp"
#include "mod16_mixed.hpp"

namespace mod16 {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int kDmaNt = 2;            // cache-policy bits of the LDS-DMA loads: nt
#ifndef MOD16_DYN_RUN
#define MOD16_DYN_RUN 8              // pieces per run: 4 saturates the ticket counter (88 atomics/us: +19 %), 16 leaves a longer tail (+0.4-1.4 %)
#endif
constexpr int kDynRun = MOD16_DYN_RUN;
constexpr int kPinLevel = 1;         // constants of the pixel function held in vector registers by the float64 totals instances (mod16_physics.hpp, KPin)

enum StreamMode {
    kStreamPet = 0, kStreamSep8, kStreamSep6, kStreamRaw, kStreamRawTotal, kStreamRawTotalHours,
    kStreamTotals,
    // float32 rasters only: the mixed-precision pixel function (mod16_mixed.hpp)
    kStreamTotalsMixed, kStreamPetMixed, kStreamSep8Mixed, kStreamSep6Mixed,
    kStreamRawMixed, kStreamRawTotalMixed, kStreamRawTotalHoursMixed
};
constexpr bool stream_is_mixed(int mode) { return mode >= kStreamTotalsMixed; }

// NW 16-byte-per-lane arrays, NB byte arrays (class raster first), NOUT outputs
template <int MODE> struct StreamSpec;
template <> struct StreamSpec<kStreamTotals> { static constexpr int NW = 14, NB = 1, NOUT = 2; };
template <> struct StreamSpec<kStreamTotalsMixed> { static constexpr int NW = 14, NB = 1, NOUT = 2; };
template <> struct StreamSpec<kStreamPetMixed> { static constexpr int NW = 14, NB = 1, NOUT = 4; };
template <> struct StreamSpec<kStreamSep8Mixed> { static constexpr int NW = 14, NB = 1, NOUT = 8; };
template <> struct StreamSpec<kStreamSep6Mixed> { static constexpr int NW = 14, NB = 1, NOUT = 6; };
template <> struct StreamSpec<kStreamRawMixed> { static constexpr int NW = 14, NB = 3, NOUT = 2; };
template <> struct StreamSpec<kStreamRawTotalMixed> { static constexpr int NW = 14, NB = 3, NOUT = 3; };
template <> struct StreamSpec<kStreamRawTotalHoursMixed> { static constexpr int NW = 15, NB = 3, NOUT = 3; };
template <> struct StreamSpec<kStreamPet> { static constexpr int NW = 14, NB = 1, NOUT = 4; };
template <> struct StreamSpec<kStreamSep8> { static constexpr int NW = 14, NB = 1, NOUT = 8; };
template <> struct StreamSpec<kStreamSep6> { static constexpr int NW = 14, NB = 1, NOUT = 6; };
template <> struct StreamSpec<kStreamRaw> { static constexpr int NW = 14, NB = 3, NOUT = 2; };
template <> struct StreamSpec<kStreamRawTotal> { static constexpr int NW = 14, NB = 3, NOUT = 3; };
template <> struct StreamSpec<kStreamRawTotalHours> { static constexpr int NW = 15, NB = 3, NOUT = 3; };

// f(integral_constant<int, K>) for K = A ... B - 1: a loop whose index is a constant expression
// (the immediate operands of the LDS-DMA builtin)
template <int A, int B, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (A < B) {
        f(std::integral_constant<int, A>{});
        static_for<A + 1, B>(f);
    }
}

template <typename T> struct StreamArgs {
    const T* wide[16];         // offset 0 of the kernel-argument segment (re-read in the loop)
    const uint8_t* bytes[4];   // offset 128: class raster, then fpar_pct, lai_x10
    T* out[8];
    const double* lut64;
    const double* tab;
    int64_t n;
    unsigned* status;
    double* diag_partial;      // [runs][8]
    unsigned long long* dyn_counter;   // ticket counter of the dynamic schedule; word [2] (unsigned) behind it: blocks finished
    double hours;              // kStreamRawTotal: the (scalar) hours of daylight
    int64_t wide_pitch;        // PITCHED: wide[k] = wide[0] + k * wide_pitch (elements)
    int64_t dma_step[2];       // PITCHED: wide_pitch in bytes - 1024, + 3072 (et_stream_kernel's issue())
    int run_shift;             // a run is 2^run_shift pieces (kDynRun for large rasters, less for
                               // small ones so that every wave of the chip gets work)
    // Tiled rasters (2-level layout): pixel i of an array lives at
    // base[(i >> log2 tile) * row + (i & (tile - 1))]. tile_shift = log2 of the PIECES
    // per tile (a piece = 64 vectors of 16 B); plain arrays: tile_shift = kNoTile, rows unused.
    int tile_shift;
    int static_sched;          // small rasters: runs dealt out round-robin (wave w: runs w, w + nwaves, ...)
                               // instead of claimed from the ticket counter -- every wave starts at once
                               // and the atomic's round trip would sit on a path a few iterations long
    int64_t wide_row;          // elements between successive tiles of a wide array
    int64_t out_row;           // ... of an output array
    int64_t byte_row;          // bytes between successive tiles of a byte array
    // Small rasters: the block that finishes last adds up the per-run partials itself and
    // writes the diagnostics vector (NULL: a separate kernel does, after this one)
    double* diag_out;          // 8 doubles
    unsigned* done_counter;    // blocks finished; the last block leaves it at 0 again
    int64_t nruns;             // partials to add (runs, or waves with work under the static schedule)
    // Mixed-precision forms, dynamic schedule: [runs][kCancelCap] pixels of the cancellation class
    // (mod16_mixed.hpp), as (piece of the run << 8 | lane << 2 | pixel of the lane), in the order
    // (piece, pixel, lane); the count travels in the run's partial (kCancelShift)
    uint16_t* cancel_list;
};
constexpr int kNoTile = 31;    // more pieces per "tile" than any raster has: one tile, plain arrays
constexpr int64_t kMaxPieces = int64_t(1) << 30;   // piece numbers are 32-bit in et_stream_kernel
static_assert(__builtin_offsetof(StreamArgs<double>, wide) == 0 &&
              __builtin_offsetof(StreamArgs<double>, bytes) == 128 &&
              __builtin_offsetof(StreamArgs<float>, wide) == 0 &&
              __builtin_offsetof(StreamArgs<float>, bytes) == 128,
              "et_stream_kernel reads wide[] / bytes[] at fixed kernel-argument offsets");

// One asm statement reads the whole slot and waits for it (lgkmcnt(0)): as
// ordinary LDS loads hipcc would put a full s_waitcnt vmcnt(0) in front of them
// (it pairs them with the LDS-DMA), which would also wait for the stores just
// issued. The returned reads are what allows the refill of the slot (WAR).
// Byte arrays land one dword per lane (sub-dword LDS-DMA, measured), 256 B per
// array behind the wide ones.
#define MOD16_RD(i, off) "ds_read_b128 %[w" #i "], %[base] offset:" #off "\n\t"
#define MOD16_RD14                                                                              \
    "ds_read_b128 %[w0], %[base]\n\t" MOD16_RD(1, 1024) MOD16_RD(2, 2048) MOD16_RD(3, 3072)     \
    MOD16_RD(4, 4096) MOD16_RD(5, 5120) MOD16_RD(6, 6144) MOD16_RD(7, 7168) MOD16_RD(8, 8192)   \
    MOD16_RD(9, 9216) MOD16_RD(10, 10240) MOD16_RD(11, 11264) MOD16_RD(12, 12288)               \
    MOD16_RD(13, 13312)
#define MOD16_W14(in)                                                                           \
    [w0] "=&v"(in[0]), [w1] "=&v"(in[1]), [w2] "=&v"(in[2]), [w3] "=&v"(in[3]),                 \
    [w4] "=&v"(in[4]), [w5] "=&v"(in[5]), [w6] "=&v"(in[6]), [w7] "=&v"(in[7]),                 \
    [w8] "=&v"(in[8]), [w9] "=&v"(in[9]), [w10] "=&v"(in[10]), [w11] "=&v"(in[11]),             \
    [w12] "=&v"(in[12]), [w13] "=&v"(in[13])

template <int NW, int NB> struct SlotRead;
template <> struct SlotRead<14, 1> {
    template <typename VT>
    static __device__ __forceinline__ void go(VT (&in)[14], unsigned (&b)[1], unsigned base, unsigned bad) {
        asm volatile(MOD16_RD14 "ds_read_b32 %[b0], %[bad]\n\ts_waitcnt lgkmcnt(0)"
                     : MOD16_W14(in), [b0] "=&v"(b[0])
                     : [base] "v"(base), [bad] "v"(bad)
                     : "memory");
    }
};
template <> struct SlotRead<14, 3> {
    template <typename VT>
    static __device__ __forceinline__ void go(VT (&in)[14], unsigned (&b)[3], unsigned base, unsigned bad) {
        asm volatile(MOD16_RD14 "ds_read_b32 %[b0], %[bad]\n\tds_read_b32 %[b1], %[bad] offset:256\n\t"
                     "ds_read_b32 %[b2], %[bad] offset:512\n\ts_waitcnt lgkmcnt(0)"
                     : MOD16_W14(in), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]), [b2] "=&v"(b[2])
                     : [base] "v"(base), [bad] "v"(bad)
                     : "memory");
    }
};
template <> struct SlotRead<15, 3> {
    template <typename VT>
    static __device__ __forceinline__ void go(VT (&in)[15], unsigned (&b)[3], unsigned base, unsigned bad) {
        asm volatile(MOD16_RD14 MOD16_RD(14, 14336)
                     "ds_read_b32 %[b0], %[bad]\n\tds_read_b32 %[b1], %[bad] offset:256\n\t"
                     "ds_read_b32 %[b2], %[bad] offset:512\n\ts_waitcnt lgkmcnt(0)"
                     : MOD16_W14(in), [w14] "=&v"(in[14]), [b0] "=&v"(b[0]), [b1] "=&v"(b[1]),
                       [b2] "=&v"(b[2])
                     : [base] "v"(base), [bad] "v"(bad)
                     : "memory");
    }
};
#undef MOD16_RD
#undef MOD16_RD14
#undef MOD16_W14

// ---- The slow branch of the domain guard (mod16_physics.hpp): the pixels the production
// arithmetic flagged, again, in the reference's operation order -- float64 whatever the storage
// type, like the FAST form. NOT inside the pipeline's loop: there its ~3000 instructions made the
// compiler hoist ~150 constants in front of the loop and keep them in registers all through it,
// and the loop -- which fills the scalar register file on its own -- spilled (+9 % instructions
// per iteration, +5.8 % kernel time, measured; a function call inside the loop did the same
// through the calling convention's register classes). The loop only records WHICH pieces had
// a flagged pixel -- one bit per piece in an otherwise unused field of the run's diagnostics
// partial -- and poisons the flagged pixels (NaN fPAR -> both totals NaN, counted as NaN, nothing
// added to the sums). Afterwards every flagged piece is revisited:
//   - large rasters (dynamic schedule): et_stream_redo_kernel, launched behind the pipeline
//     kernel, one wave per 64 runs: reads the masks, redoes the flagged pieces, corrects the
//     runs' partials before the fixed-order sum adds them up;
//   - small rasters (static schedule, latency-bound): every wave redoes its own flagged pieces
//     behind its loop, before it stores its partial -- no second dispatch.
// redo_piece: lane `lane`'s V pixels of piece `piece`: reads the inputs again from memory, asks
// the guard again (exactly the hot path's predicate, so exactly the flagged pixels are redone),
// stores the outputs element by element, and accumulates what the diagnostics must get back.
struct RedoAcc {
    double sum_d = 0.0, sum_n = 0.0, max_d = -__builtin_huge_val(), max_n = -__builtin_huge_val();
    unsigned num_d = 0, num_n = 0;      // flagged pixels whose true total is a number (per lane)
};

// redo_pixel: ONE pixel again (element offsets ow / oo / ob of the pixel in the wide, output and byte
// arrays), in float64 whatever the storage type; stores its outputs, returns the totals as stored.
//   LISTED = false  a pixel of a flagged piece: asks the guard again (exactly the hot path's predicate,
//                   so exactly the flagged pixels are redone) -> the reference's operation order; with
//                   `marked` (mixed-precision forms under the static schedule) also a pixel of the
//                   cancellation class (mod16_mixed.hpp, period_mixed), which the loop marked with
//                   kCancelPoison in its first output -> the FAST form's float64 arithmetic on the
//                   widened inputs (`tb`: its exp / log tables): what the FAST kernel stores for it;
//   LISTED = true   a pixel of a run's cancellation list (dynamic schedule): the FAST form, unasked.
// Returns false if the pixel needed nothing.
template <typename T, int MODE, bool LISTED>
__device__ __forceinline__ bool redo_pixel(const StreamArgs<T>& a, const double* lut, const double* tb,
                                           int64_t ow, int64_t oo, int64_t ob, bool marked, double& d, double& g) {
    typedef StreamSpec<MODE> S;
    constexpr int NW = S::NW, NB = S::NB;
    constexpr bool kSep6 = MODE == kStreamSep6 || MODE == kStreamSep6Mixed;
    constexpr bool kSep8 = MODE == kStreamSep8 || MODE == kStreamSep8Mixed;
    constexpr bool kPet = MODE == kStreamPet || MODE == kStreamPetMixed;
    constexpr bool kHoursArr = MODE == kStreamRawTotalHours || MODE == kStreamRawTotalHoursMixed;
    constexpr bool kTotal8 = kHoursArr || MODE == kStreamRawTotal || MODE == kStreamRawTotalMixed;
    constexpr bool kRawAny = MODE == kStreamRaw || MODE == kStreamRawTotal || MODE == kStreamRawTotalHours ||
                             MODE == kStreamRawMixed || MODE == kStreamRawTotalMixed ||
                             MODE == kStreamRawTotalHoursMixed;
    auto wide = [&](int k) -> double { return (double)a.wide[k][ow]; };
    PixelIn<double> x;
    RawIn<double> r;
    bool out = false;
    if constexpr (kRawAny) {
        r = RawIn<double>{wide(0), wide(1), wide(2), wide(3), wide(4), wide(5), wide(6),
                          wide(7), wide(8), wide(9), wide(10), wide(11), wide(12), wide(13),
                          (unsigned)a.bytes[NB > 1 ? 1 : 0][ob], (unsigned)a.bytes[NB > 2 ? 2 : 0][ob]};
        if constexpr (!LISTED) {
            if constexpr (stream_is_mixed(MODE)) out = raw_out_of_domain_f32(r);
            else out = raw_out_of_domain(r);
        }
    } else {
        x = PixelIn<double>{wide(0), wide(1), wide(2), wide(3), wide(4), wide(5), wide(6),
                            wide(7), wide(8), wide(9), wide(10), wide(11), wide(12), wide(13)};
        if constexpr (!LISTED) {
            if constexpr (stream_is_mixed(MODE)) out = out_of_domain_f32(x);
            else out = fast_out_of_domain(x);
        }
    }
    bool cancel = LISTED;
    if constexpr (!LISTED && stream_is_mixed(MODE) && sizeof(T) == 4) {
        if (marked && !out)
            cancel = __float_as_uint(__hip_atomic_load(reinterpret_cast<const float*>(a.out[0]) + oo, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT)) == kCancelPoison;
    }
    if (!out && !cancel) return false;
    if constexpr (kRawAny) x = raw_to_pixel_exact<double, true>(r);
    unsigned c = a.bytes[0][ob];
    c = c >= 13u ? 13u : c;
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
    PixelOut<double> o;
    if (cancel) {
        p.inv_dtmin = l[11 * kLutCols];
        p.inv_dvpd = l[12 * kLutCols];
        p.rbl_slope = l[13 * kLutCols];
        p.inv_beta = l[14 * kLutCols];
        o = et_pixel_fast<double, kPet>(x, p, tb);
    } else {
        if constexpr (!LISTED) o = et_pixel_exact<double, kPet, true>(x, p);
    }
    const double day = (o.canopy_d + o.soil_d) + o.trans_d;      // :792
    const double night = (o.canopy_n + o.soil_n) + o.trans_n;
    auto put = [&](int k, double val) { a.out[k][oo] = (T)val; };
    if constexpr (kSep6) {
        put(0, o.canopy_d); put(1, o.soil_d); put(2, o.trans_d);
        put(3, o.canopy_n); put(4, o.soil_n); put(5, o.trans_n);
    } else {
        put(0, day);
        put(1, night);
    }
    if constexpr (kPet) { put(2, o.pet_d); put(3, o.pet_n); }
    if constexpr (kSep8) {
        put(2, o.canopy_d); put(3, o.soil_d); put(4, o.trans_d);
        put(5, o.canopy_n); put(6, o.soil_n); put(7, o.trans_n);
    }
    if constexpr (kTotal8) {
#pragma clang fp contract(off)
        // tests/verification/verify2.py:113-115
        double h = a.hours;
        if constexpr (kHoursArr) h = wide(NW - 1);
        put(2, (day * h * 8.0 * 60.0 * 60.0) + (night * (24.0 - h) * 8.0 * 60.0 * 60.0));
    }
    d = (double)(T)day;      // as stored
    g = (double)(T)night;
    return true;
}

// redo_piece: lane `lane`'s V pixels of piece `piece` (see redo_pixel, LISTED = false), with what
// the diagnostics must get back accumulated per lane.
template <typename T, int MODE>
__device__ __forceinline__ void redo_piece(const StreamArgs<T>& a, const double* lut, const double* tb,
                                           int64_t piece, int lane, bool marked, RedoAcc& acc) {
    constexpr int V = 16 / (int)sizeof(T);
    if ((piece * 64 + lane) * V >= a.n) return;          // the ragged last piece
    const int64_t tile = piece >> a.tile_shift;
    const int64_t q = (piece - (tile << a.tile_shift)) * (int64_t)(64 * V) + (int64_t)lane * V;
    const int64_t ow = tile * a.wide_row + q, oo = tile * a.out_row + q, ob = tile * a.byte_row + q;
#pragma nounroll
    for (int j = 0; j < V; ++j) {
        double d, g;
        if (!redo_pixel<T, MODE, false>(a, lut, tb, ow + j, oo + j, ob + j, marked, d, g)) continue;
        if (d == d) { acc.sum_d += d; acc.num_d += 1u; acc.max_d = d > acc.max_d ? d : acc.max_d; }
        if (g == g) { acc.sum_n += g; acc.num_n += 1u; acc.max_n = g > acc.max_n ? g : acc.max_n; }
    }
}

// Folds the lanes' corrections (fixed butterfly order) into the 8 fields of a partial held one
// field per lane (lane k < 8 holds field k): sums += , NaN counts -= , maxima = max.
__device__ __forceinline__ double redo_fold(const RedoAcc& acc, int lane, double f) {
    double sd = acc.sum_d, sn = acc.sum_n, md = acc.max_d, mn = acc.max_n;
    double nd = (double)acc.num_d, nn = (double)acc.num_n;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        sd += __shfl_xor(sd, off, 64);
        sn += __shfl_xor(sn, off, 64);
        nd += __shfl_xor(nd, off, 64);
        nn += __shfl_xor(nn, off, 64);
        const double od = __shfl_xor(md, off, 64), on = __shfl_xor(mn, off, 64);
        md = od > md ? od : md;
        mn = on > mn ? on : mn;
    }
    return lane == 0 ? f + sd : lane == 1 ? f + sn : lane == 4 ? f - nd : lane == 5 ? f - nn
         : lane == 6 ? (md > f ? md : f) : lane == 7 ? (mn > f ? mn : f) : f;
}
constexpr int kFlagField = 2;        // the field of a diagnostics partial that carries the flags
constexpr int kFlagBits = 52;        // ... as an exact integer in a double
// Dynamic schedule, mixed-precision forms: the same field also counts the run's cancellation list
// (bits kCancelShift ...; piece flags then use the bits below, the last one standing for every later piece)
constexpr int kCancelShift = 44;
constexpr int kCancelCap = 32;       // entries per run (64 bytes = one partial's size, behind the partials)

// PITCHED: the wide arrays are equally spaced (one slab, as
// RasterEngine.alloc_raster lays them out): array k's address is wide[0] +
// k * pitch by two scalar adds instead of a pointer load.
// (no second __launch_bounds__ argument: asking for two waves per SIMD outright made hipcc pick a
// schedule 1.2 % slower for the same 188 registers -- profiles/r03_ab_bisect.txt)
// GUARD = false (MOD16_DOMAIN_TRUSTED, totals forms only): the caller vouches for the drivers; no
// domain test, no flag record, nothing revisited -- the loop of round 2.
// (the one exception, round 5: the raw-driver instances with per-pixel hours of daylight -- 15 wide
// inputs -- came out at 258 / 260 registers once the humidity moved into the period, i.e. ONE wave
// per SIMD; they alone are told to stay within two, and do so without a spill or a register parked
// in the accumulator file -- the float32-raster ones because their four pixels per thread are kept
// apart, see the loop; tests/test_abi.py::test_no_kernel_spills_vector_registers)
// (round 6: the float64 raw-driver instance without the 8-day total as well -- 252 registers in round 5,
// 258 once the ramps became clamps)
constexpr int stream_min_waves(int mode, bool f64) { return mode == kStreamRawTotalHours || (mode == kStreamRaw && f64) ? 2 : 1; }
#define MOD16_STREAM_BOUNDS __launch_bounds__(kBlock) \
    __attribute__((amdgpu_waves_per_eu(stream_min_waves(MODE, sizeof(T) == 8), 8)))
template <typename T, int MODE, bool PITCHED = false, bool GUARD = true>
__global__ void MOD16_STREAM_BOUNDS et_stream_kernel(const StreamArgs<T> a) {
    typedef StreamSpec<MODE> S;
    constexpr int V = 16 / (int)sizeof(T);
    constexpr int NW = S::NW, NB = S::NB, NOUT = S::NOUT;
    constexpr int kSlot = NW * 1024 + NB * 256;
    ignore_signalling_nans();
    constexpr bool RAW = MODE == kStreamRaw || MODE == kStreamRawTotal || MODE == kStreamRawTotalHours;
    constexpr int kTab = FastMath<double>::kTabDoubles;
    __shared__ double lut[MOD16_LUT_ROWS * kLutCols];
    __shared__ __attribute__((aligned(16))) double tab[kTab];
    __shared__ __attribute__((aligned(16))) char stage[(kBlock / 64) * kSlot];
    // (the tables are filled further down, BEHIND the issue of the first piece's DMA: the two
    // round trips to memory overlap -- a 1200 x 1200 launch is only six iterations long)
    const float* lutf = nullptr;       // mixed-precision forms: the table in float32 as well
    __shared__ float lut32[stream_is_mixed(MODE) ? MOD16_LUT_ROWS * kLutCols : 1];
    if constexpr (stream_is_mixed(MODE)) lutf = lut32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* const ws = stage + wave * kSlot;
    // Piece numbers are 32-bit (round 6; the launch refuses a raster of more than kMaxPieces): the
    // loop's scalar bookkeeping -- run state, the claim, three tile offsets -- is single adds, shifts
    // and 32 x 32 -> 64 multiplications instead of register pairs with carries (a scalar instruction
    // costs ~0.7 of a vector one at two waves per SIMD, and the loop held ~190 of them).
    const int64_t nvec_s = a.n / V;
    const int npiece = (int)((nvec_s + 63) >> 6);
    const int nfull = (int)(nvec_s >> 6);          // pieces whose 64 vectors all exist
    const int nrem = (int)(nvec_s & 63);           // vectors of the ragged piece behind them (0: none)
    const int nwaves = (int)gridDim.x * (kBlock / 64);
    const int rs = a.run_shift, rl = 1 << rs;
    int cbase = (int)((blockIdx.x * (kBlock / 64) + wave) << rs);
    // the run behind the current one. Static schedule (small rasters): the wave's next run, a fixed
    // stride on; dynamic: whatever the claim made in a run's first iteration returns, read in its
    // second (`claim_run`: the one test of the loop's head)
    const int sstride = a.static_sched ? nwaves << rs : 0;
    const int claim_run = a.static_sched ? -1 : 1;
    int next_base = a.static_sched ? cbase + sstride : npiece;
    unsigned long long ticket = 0;
    int run = 0;
    // lanes of piece p that hold a vector of the raster: 64, the ragged piece's, or none (wave-uniform)
    auto lanes_of = [&](int p) { return p < nfull ? 64 : p == nfull ? nrem : 0; };
    // element offsets of piece p in the wide arrays, the outputs and the byte arrays (wave-uniform,
    // in SGPRs). Rows between tiles fit 32 bits (launch_stream checks); plain arrays are ONE tile
    // (tile_shift = kNoTile = 31: tile 0), whose offset within is the one 64-bit quantity.
    struct Offs { int64_t w, o, b; };
    constexpr int kLogPiece = V == 2 ? 7 : V == 4 ? 8 : 6;       // log2 of the elements of a piece
    static_assert((64 * V) == (1 << kLogPiece), "elements per piece");
    auto offs_of = [&](int p) {
        const unsigned piece = (unsigned)__builtin_amdgcn_readfirstlane(p);
        const unsigned tile = piece >> a.tile_shift;
        const uint64_t q = (uint64_t)(piece - (tile << a.tile_shift)) << kLogPiece;
        return Offs{(int64_t)((uint64_t)tile * (uint64_t)(unsigned)a.wide_row + q),
                    (int64_t)((uint64_t)tile * (uint64_t)(unsigned)a.out_row + q),
                    (int64_t)((uint64_t)tile * (uint64_t)(unsigned)a.byte_row + q)};
    };
    const unsigned lane_elem = (unsigned)lane * (unsigned)V;
    auto advance = [&](int& cb, int& r) {
        if (((++r) >> rs) != 0) { r = 0; cb = next_base; next_base += sstride; }
    };
    int lanes = lanes_of(cbase + run);             // of the piece this iteration computes
    int64_t out_first = 0;                         // its offset in the outputs (found when its DMA was issued)
    double dsum_d = 0, dsum_n = 0, dmax_d = -__builtin_huge_val(), dmax_n = -__builtin_huge_val();
    float fmax_d = -__builtin_huge_valf(), fmax_n = -__builtin_huge_valf();   // mixed forms: maxima of float32 values
    unsigned nan_d = 0, nan_n = 0;
    auto vmax_f32 = [](float a, float b) {     // maxNum as one v_max_f32 (results of arithmetic: already quiet)
        float d;
        asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };

    // pieces of the current run (static schedule: of this wave) with a flagged pixel. Wave-uniform,
    // but kept in a VECTOR register pair (the asm hides the uniformity): it is written in a cold
    // branch and read at a flush only -- no reason to hold a scalar pair across the loop body,
    // where scalar registers are what hipcc runs out of
    unsigned long long flags = 0;
    asm volatile("" : "+v"(flags));
    // entries of the current run's cancellation list (mixed-precision forms, dynamic schedule);
    // wave-uniform and in a vector register for the same reason
    unsigned cancel_cnt = 0;
    constexpr bool kLists = stream_is_mixed(MODE) && GUARD;
    if constexpr (kLists) asm volatile("" : "+v"(cancel_cnt));
    // the launch's serial number (word [3] of the ticket counter's block; the last block of a launch
    // increments it): every run's partial carries it, and what runs behind this kernel counts the
    // runs that do -- a launch that met a stale ticket counter and processed only its first runs is
    // reported (kStatusIncomplete) instead of passing the previous step's numbers on. In a vector
    // register, like `flags`: read at a flush only.
    unsigned serial = 0;
    if (!a.static_sched)
        serial = __hip_atomic_load(reinterpret_cast<const unsigned*>(a.dyn_counter) + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" : "+v"(serial));
    int iters = 0;                     // iterations this wave has done
    // the 8 fields of a diagnostics partial, one per lane (lane k < 8: field k), from the
    // accumulators, which are reset: butterfly sums in a fixed order; field kFlagField = `flags`
    auto diag_fields = [&]() -> double {
        if constexpr (stream_is_mixed(MODE)) {
            dmax_d = (double)fmax_d;
            dmax_n = (double)fmax_n;
            fmax_d = fmax_n = -__builtin_huge_valf();
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dsum_d += __shfl_xor(dsum_d, off, 64);
            dsum_n += __shfl_xor(dsum_n, off, 64);
            dmax_d = FastMath<double>::vmax(dmax_d, __shfl_xor(dmax_d, off, 64));
            dmax_n = FastMath<double>::vmax(dmax_n, __shfl_xor(dmax_n, off, 64));
        }
        const double cnt_d = (double)__builtin_amdgcn_readfirstlane(nan_d);
        const double cnt_n = (double)__builtin_amdgcn_readfirstlane(nan_n);
        unsigned long long flag_word = flags;
        if constexpr (kLists)
            flag_word |= (unsigned long long)(cancel_cnt < (unsigned)kCancelCap ? cancel_cnt : (unsigned)kCancelCap) << kCancelShift;
        const double f = lane == 0 ? dsum_d : lane == 1 ? dsum_n : lane == kFlagField ? (double)flag_word
                       : lane == kSerialField ? launch_marker(serial)
                       : lane == 4 ? cnt_d : lane == 5 ? cnt_n : lane == 6 ? dmax_d : lane == 7 ? dmax_n : 0.0;
        dsum_d = dsum_n = 0.0;
        dmax_d = dmax_n = -__builtin_huge_val();
        nan_d = nan_n = 0;
        flags = 0;
        cancel_cnt = 0;
        return f;
    };
    struct Ptrs { const char* w[NW]; const char* b[NB]; };
    // scalar loads from the kernel-argument segment, ahead of the wait for the DMA
    typedef const __attribute__((address_space(4))) char* kptr_t;
    auto load_ptrs = [&](Ptrs& p) {
        kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        if constexpr (PITCHED) {
            p.w[0] = *reinterpret_cast<const char* const __attribute__((address_space(4)))*>(ka);
            // (the two steps between successive DMA bases, see issue())
            p.w[1] = reinterpret_cast<const char*>(
                *reinterpret_cast<const int64_t __attribute__((address_space(4)))*>(
                    ka + __builtin_offsetof(StreamArgs<T>, dma_step)));
            p.w[2] = reinterpret_cast<const char*>(
                *reinterpret_cast<const int64_t __attribute__((address_space(4)))*>(
                    ka + __builtin_offsetof(StreamArgs<T>, dma_step) + 8));
        } else {
#pragma unroll
            for (int k = 0; k < NW; ++k)
                p.w[k] = *reinterpret_cast<const char* const __attribute__((address_space(4)))*>(ka + 8 * k);
        }
#pragma unroll
        for (int k = 0; k < NB; ++k)
            p.b[k] = *reinterpret_cast<const char* const __attribute__((address_space(4)))*>(ka + 128 + 8 * k);
    };
    auto issue = [&](const Offs& of, const Ptrs& p) {
        unsigned lb = lane_elem * (unsigned)sizeof(T);
        unsigned wl = (unsigned)(uintptr_t)(lptr_t)ws;
        asm volatile("" : "+v"(lb));
        asm volatile("" : "+s"(wl));
        const int64_t first_b = of.w * (int64_t)sizeof(T);
        const int64_t first = of.b;
        // The instruction's immediate offset (< 4096) is added to BOTH addresses of an LDS-DMA load:
        // four successive LDS slots (1 KiB each) are reached from ONE value of M0 by the offsets 0,
        // 1024, 2048, 3072, with the global base moved back by the same amount -- M0 is written
        // (and its hazard slot spent) 4 times per piece instead of 14 (round 6).
        if constexpr (PITCHED) {
            // (base and steps: read again in this iteration, load_ptrs -- which also keeps
            // the 14 addresses from being hoisted). Base k+1 = base k + pitch - 1024 inside a
            // group of four slots, + pitch + 3072 into the next group: the host precomputes both.
            const int64_t step_in = (int64_t)(uintptr_t)p.w[1], step_over = (int64_t)(uintptr_t)p.w[2];
            const char* pk = p.w[0] + first_b;
            static_for<0, NW>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                __builtin_amdgcn_global_load_lds((gptr_t)(pk + lb), (lptr_t)(uintptr_t)(wl + (k & ~3) * 1024),
                                                 16, (k & 3) * 1024, kDmaNt);
                pk += (k & 3) == 3 ? step_over : step_in;
            });
        } else {
            static_for<0, NW>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                __builtin_amdgcn_global_load_lds((gptr_t)((p.w[k] + (first_b - (k & 3) * 1024)) + lb),
                                                 (lptr_t)(uintptr_t)(wl + (k & ~3) * 1024), 16, (k & 3) * 1024, kDmaNt);
            });
        }
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if constexpr (V == 2)
                __builtin_amdgcn_global_load_lds((gptr_t)((p.b[k] + first) + (lb >> 3)),
                                                 (lptr_t)(uintptr_t)(wl + NW * 1024 + k * 256), 2, 0, kDmaNt);
            else
                __builtin_amdgcn_global_load_lds((gptr_t)((p.b[k] + first) + (lb >> 2)),
                                                 (lptr_t)(uintptr_t)(wl + NW * 1024 + k * 256), 4, 0, kDmaNt);
        }
    };
    {
        Ptrs p;
        load_ptrs(p);
        const Offs of = offs_of(cbase + run);
        out_first = of.o;
        if (lane < lanes) issue(of, p);
    }
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) {
        const double x = a.lut64[i];
        lut[i] = x;
        // (the float32 copy of the mixed form holds rbl_max - rbl_min where the float64 forms keep the
        // ramp's slope: mod16_mixed.hpp, kLutDrbl)
        if constexpr (stream_is_mixed(MODE))
            lut32[i] = i / kLutCols == kLutDrbl ? (float)(a.lut64[9 * kLutCols + i % kLutCols] - a.lut64[8 * kLutCols + i % kLutCols])
                                                : (float)x;
    }
    for (int i = threadIdx.x; i < kTab; i += kBlock) tab[i] = a.tab[i];
    __syncthreads();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const GuardConsts gc = RAW ? raw_guard_consts() : guard_consts();      // (three register pairs held through the loop)
#pragma nounroll
    for (; cbase + run < npiece; ) {
        Ptrs ptrs;
#ifdef MOD16_PRIO   // experiment: the memory phase of a wave outranks its partner's arithmetic
        __builtin_amdgcn_s_setprio(MOD16_PRIO);
#endif
        load_ptrs(ptrs);
        // everything but the NOUT stores of the previous iteration (and its
        // diagnostics flush, if any) must be complete: this iteration's DMA and,
        // one iteration after a claim, the claim's atomic
        // (behind a flush the partial's store is one more: once per run the wait then covers the
        // oldest output store as well -- cheaper than a two-way branch in every iteration)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NOUT) : "memory");
        if (__builtin_expect(run == claim_run, 0)) {
            asm volatile("" : "+v"(ticket));
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ticket);
            const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ticket >> 32));
            // WHATEVER 64 bits sat in the counter (round 6): a ticket beyond the raster's pieces says
            // "nothing left" -- compared unsigned and clamped before the shift, so that no value can
            // overflow into a base the loop guard accepts (round 5's injected 0x3f3f... did: a wild
            // DMA read and a wild store). A stale ticket INSIDE the raster only skips runs, which the
            // markers report (kStatusIncomplete). Scalar, once per run: not in the per-piece path.
            // (the bound is in RUNS, one past the raster's: base >= npiece, and no shift can overflow)
            const unsigned past = ((unsigned)npiece >> rs) + 1u;
            const unsigned tk = (hi != 0u || lo > past) ? past : lo;
            next_base = (int)(((unsigned)nwaves + tk) << rs);
        }
        typedef typename Vec<T, V>::type VT;
        VT in[NW];
        unsigned bits[NB];
        SlotRead<NW, NB>::go(in, bits, (unsigned)(uintptr_t)(lptr_t)ws + lane * 16u,
                             (unsigned)(uintptr_t)(lptr_t)ws + NW * 1024u + lane * 4u);
        if (__builtin_expect(run == 0 && lane == 0 && !a.static_sched, 0)) {
            const unsigned long long one = 1;
            asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0"
                         : "=v"(ticket) : "v"(a.dyn_counter), "v"(one) : "memory");
        }
        int cb_n = cbase;
        int run_n = run;
        advance(cb_n, run_n);
        const int lanes_n = lanes_of(cb_n + run_n);
        const Offs of_n = offs_of(cb_n + run_n);
        if (lane < lanes_n) issue(of_n, ptrs);
        asm volatile("" ::: "memory");
#ifdef MOD16_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif

        if (lane < lanes) {   // only the last piece is ragged
            VT res[NOUT];
            // `bad`: one of this thread's pixels lies outside the domain of the production arithmetic
            // (mod16_physics.hpp, "domain guard"). Such a pixel gets a NaN fPAR, which makes both
            // of its totals NaN whatever else it holds, so the diagnostics below count it as NaN
            // and add nothing for it; what runs behind the loop computes it again in the
            // reference's operation order and puts its true values into the outputs and the
            // diagnostics (redo_piece). A raster without such pixels executes the instructions it
            // did before the guard existed, plus the guard's (15 per pixel) and one select.
            bool bad = false;
            double gmax = 0.0;          // float64 forms: the largest guard magnitude of this thread's pixels
            if constexpr (stream_is_mixed(MODE)) {
                static_assert(V == 4 || !stream_is_mixed(MODE), "the mixed form is for float32 rasters");
                // class codes of the four pixels first (the range check's side effect would
                // otherwise sit between the two pairs and keep their arithmetic apart)
                unsigned cls_of[V];
                bool cls_bad = false;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const unsigned c = (bits[0] >> (8 * j)) & 0xffu;
                    cls_bad = cls_bad | (c >= 13u);
                    cls_of[j] = c >= 13u ? 13u : c;
                }
                if (__builtin_expect(cls_bad, 0)) atomicOr(a.status, kStatusClassRange);
#pragma unroll
                for (int jj = 0; jj < V; jj += 2) {   // pairs of pixels: packed float32 arithmetic
                    float pin[14][2];
#pragma unroll
                    for (int k = 0; k < 14; ++k) { pin[k][0] = in[k][jj]; pin[k][1] = in[k][jj + 1]; }
                    const unsigned c0 = cls_of[jj], c1 = cls_of[jj + 1];
                    Parts2 pd, pn;
                    bool cancel2[2] = {false, false};
                    constexpr bool kRawMixed = MODE == kStreamRawMixed || MODE == kStreamRawTotalMixed ||
                                               MODE == kStreamRawTotalHoursMixed;
                    if constexpr (kRawMixed) {
                        unsigned fp[2] = {(bits[1] >> (8 * jj)) & 0xffu, (bits[1] >> (8 * jj + 8)) & 0xffu};
                        const unsigned lx[2] = {(bits[2] >> (8 * jj)) & 0xffu, (bits[2] >> (8 * jj + 8)) & 0xffu};
                        float din[14][2];
                        Humid2 hum[2];
                        const unsigned b2 = raw_pair_out_of_domain(pin);
                        bad |= b2 != 0u;
                        fp[0] = (b2 & 1u) ? 255u : fp[0];        // a fill code: fPAR = NaN
                        fp[1] = (b2 & 2u) ? 255u : fp[1];
                        raw_pair_mixed(pin, fp, lx, tab, din, hum);
                        et_pair_mixed_parts<false, true>(din, lut + c0, lut + c1, kLutCols, tab, pd, pn, nullptr, lutf + c0, lutf + c1, hum,
                                                         GUARD ? cancel2 : nullptr);
                    } else {
                        if constexpr (GUARD) {
                            const unsigned b2 = pair_out_of_domain(pin);
                            bad |= b2 != 0u;
                            pin[12][0] = (b2 & 1u) ? __builtin_nanf("") : pin[12][0];
                            pin[12][1] = (b2 & 2u) ? __builtin_nanf("") : pin[12][1];
                        }
                        et_pair_mixed_parts<MODE == kStreamPetMixed, true>(pin, lut + c0, lut + c1, kLutCols, tab, pd, pn, nullptr, lutf + c0, lutf + c1,
                                                                           nullptr, GUARD ? cancel2 : nullptr);
                    }
                    f2 day2 = pd.total, night2 = pn.total;                   // :792
                    if constexpr (GUARD) {
                        // the cancellation class (mod16_mixed.hpp; about one wave-iteration in ten holds such a
                        // pixel): it is MARKED -- its first output holds kCancelPoison, both totals count as NaN
                        // in the run's diagnostics -- and what runs behind the loop puts the float64 result in
                        // its place. Dynamic schedule: the pixel joins its run's list, in the order (piece,
                        // pixel, lane), which et_stream_redo_kernel works off with the pixels of 64 runs side by
                        // side in the lanes of a wave; a run whose list is full leaves the pixel its float32
                        // value. Static schedule (small rasters): the piece is flagged like one with a pixel
                        // outside the domain and the wave revisits it behind its loop (redo_piece).
                        if (__builtin_expect(__any(cancel2[0] | cancel2[1]), 0)) {
                            const unsigned long long b0 = __ballot(cancel2[0]), b1 = __ballot(cancel2[1]);
                            const unsigned more = (unsigned)__builtin_popcountll(b0) + (unsigned)__builtin_popcountll(b1);
                            if (a.static_sched | (cancel_cnt + more > (unsigned)kCancelCap)) {
                                bad |= cancel2[0] | cancel2[1];      // (a full list: the piece is flagged instead)
                            } else {
                                // (the list's address: read from the kernel arguments here, in the cold
                                // branch, rather than held in a scalar pair all through the loop)
                                kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
                                asm volatile("" : "+s"(ka));
                                uint16_t* list = *reinterpret_cast<uint16_t* const __attribute__((address_space(4)))*>(
                                                     ka + __builtin_offsetof(StreamArgs<T>, cancel_list)) +
                                                 ((int64_t)(cbase >> rs) * kCancelCap + cancel_cnt);
                                const unsigned r0 = __builtin_amdgcn_mbcnt_hi((unsigned)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b0, 0u));
                                const unsigned r1 = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u));
                                const unsigned ent = (unsigned)((run << 8) | (lane << 2) | jj);
                                if (cancel2[0]) list[r0] = (uint16_t)ent;
                                if (cancel2[1]) list[(unsigned)__builtin_popcountll(b0) + r1] = (uint16_t)(ent + 1u);
                                cancel_cnt += more;
                            }
                            const bool mark[2] = {cancel2[0], cancel2[1]};
                            const f2 poison = splat(__uint_as_float(kCancelPoison));
                            day2 = sel(mark[0], mark[1], poison, day2);
                            night2 = sel(mark[0], mark[1], poison, night2);
                            if constexpr (MODE == kStreamSep6Mixed) pd.canopy = sel(mark[0], mark[1], poison, pd.canopy);
                        }
                    }
                    auto put = [&](int k, f2 v) { res[k][jj] = v.x; res[k][jj + 1] = v.y; };
                    if constexpr (MODE == kStreamSep6Mixed) {
                        put(0, pd.canopy); put(1, pd.soil); put(2, pd.trans);
                        put(3, pn.canopy); put(4, pn.soil); put(5, pn.trans);
                    } else {
                        put(0, day2); put(1, night2);
                    }
                    if constexpr (MODE == kStreamPetMixed) { put(2, pd.pet); put(3, pn.pet); }
                    if constexpr (MODE == kStreamRawTotalMixed || MODE == kStreamRawTotalHoursMixed) {
                        // tests/verification/verify2.py:113-115
                        f2 h = splat((float)a.hours);
                        if constexpr (MODE == kStreamRawTotalHoursMixed) h = f2{in[14][jj], in[14][jj + 1]};
                        const f2 k8 = splat(8.f * 3600.f);
                        // (the contraction written out: every instance of the kernel rounds alike)
                        put(2, __builtin_elementwise_fma(day2 * h, k8, (night2 * (splat(24.f) - h)) * k8));
                    }
                    if constexpr (MODE == kStreamSep8Mixed) {
                        put(2, pd.canopy); put(3, pd.soil); put(4, pd.trans);
                        put(5, pn.canopy); put(6, pn.soil); put(7, pn.trans);
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        // NaN -> 0 while still float32 (one v_cndmask, not two), sums in float64
                        const float d = day2[e], g = night2[e];
                        const bool dn = d != d, gn = g != g;
                        nan_d += (unsigned)__builtin_popcountll(__ballot(dn));
                        nan_n += (unsigned)__builtin_popcountll(__ballot(gn));
                        dsum_d += (double)(dn ? 0.f : d);
                        dsum_n += (double)(gn ? 0.f : g);
                        fmax_d = vmax_f32(fmax_d, d);
                        fmax_n = vmax_f32(fmax_n, g);
                    }
                }
            } else {
            // class codes of the V pixels first: the range check has a side effect (the status
            // word), and between the pixels it would keep hipcc from interleaving their
            // (independent) arithmetic
            unsigned cls_of[V];
            bool cls_bad = false;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const unsigned c = (bits[0] >> (8 * j)) & 0xffu;
                cls_bad = cls_bad | (c >= 13u);
                cls_of[j] = c >= 13u ? 13u : c;
            }
            if (__builtin_expect(cls_bad, 0)) atomicOr(a.status, kStatusClassRange);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                PixelIn<double> x;
                double dav_d = 1.0, dav_n = 1.0;       // raw forms: the humidity quotients' denominators
                if constexpr (RAW) {
                    RawIn<double> r = {(double)in[0][j], (double)in[1][j], (double)in[2][j],
                                       (double)in[3][j], (double)in[4][j], (double)in[5][j],
                                       (double)in[6][j], (double)in[7][j], (double)in[8][j],
                                       (double)in[9][j], (double)in[10][j], (double)in[11][j],
                                       (double)in[12][j], (double)in[13][j],
                                       (bits[1] >> (8 * j)) & 0xffu, (bits[2] >> (8 * j)) & 0xffu};
                    // (a lane mask here, not the magnitude kept in a register pair as below: the raw-driver
                    // instances have no vector registers to spare)
                    const bool out = raw_guard_value(r, gc) >= gc.huge;
                    bad |= out;
                    x = raw_to_pixel_fast(r, tab, dav_d, dav_n);
                    x.fpar = out ? __builtin_nan("") : x.fpar;
                } else {
                    x = PixelIn<double>{(double)in[0][j], (double)in[1][j], (double)in[2][j],
                                        (double)in[3][j], (double)in[4][j], (double)in[5][j],
                                        (double)in[6][j], (double)in[7][j], (double)in[8][j],
                                        (double)in[9][j], (double)in[10][j], (double)in[11][j],
                                        (double)in[12][j], (double)in[13][j]};
                    if constexpr (GUARD) {
                        // the guard's magnitude stays in a vector register: tested here for this
                        // pixel's select (the high word decides: 0x7ff80000'xxxxxxxx is a quiet NaN --
                        // one v_cndmask), and once per iteration, as the maximum over the thread's
                        // pixels, for the flag record
                        const double gm = fast_guard_value(x, gc);
                        gmax = FastMath<double>::vmax(gmax, gm);
                        x.fpar = __hiloint2double(gm >= gc.huge ? 0x7ff80000 : __double2hiint(x.fpar), __double2loint(x.fpar));
                    }
                }
                const unsigned c = cls_of[j];
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
                // (the totals instances hold the pixel function's busiest constants in vector registers)
                typedef typename std::conditional<MODE == kStreamTotals && std::is_same<T, double>::value, KPin<kPinLevel>, KLit>::type KP;
                PixelOut<double> o = et_pixel_fast<double, MODE == kStreamPet, KP, RAW>(x, p, tab, dav_d, dav_n);
                const double day = (o.canopy_d + o.soil_d) + o.trans_d;      // :792
                const double night = (o.canopy_n + o.soil_n) + o.trans_n;
                if constexpr (MODE == kStreamSep6) {
                    res[0][j] = (T)o.canopy_d; res[1][j] = (T)o.soil_d; res[2][j] = (T)o.trans_d;
                    res[3][j] = (T)o.canopy_n; res[4][j] = (T)o.soil_n; res[5][j] = (T)o.trans_n;
                } else {
                    res[0][j] = (T)day;
                    res[1][j] = (T)night;
                }
                if constexpr (MODE == kStreamPet) {
                    res[2][j] = (T)o.pet_d;
                    res[3][j] = (T)o.pet_n;
                }
                if constexpr (MODE == kStreamSep8) {
                    res[2][j] = (T)o.canopy_d; res[3][j] = (T)o.soil_d; res[4][j] = (T)o.trans_d;
                    res[5][j] = (T)o.canopy_n; res[6][j] = (T)o.soil_n; res[7][j] = (T)o.trans_n;
                }
                if constexpr (MODE == kStreamRawTotal || MODE == kStreamRawTotalHours) {
#pragma clang fp contract(off)
                    // tests/verification/verify2.py:113-115
                    double h = a.hours;
                    if constexpr (MODE == kStreamRawTotalHours) h = (double)in[14][j];
                    res[2][j] = (T)((day * h * 8.0 * 60.0 * 60.0) +
                                    (night * (24.0 - h) * 8.0 * 60.0 * 60.0));
                }
                {
                    const double d = (double)(T)day, g = (double)(T)night;
                    const bool dn = d != d, gn = g != g;
                    nan_d += (unsigned)__builtin_popcountll(__ballot(dn));
                    nan_n += (unsigned)__builtin_popcountll(__ballot(gn));
                    dsum_d += dn ? 0.0 : d;
                    dsum_n += gn ? 0.0 : g;
                    dmax_d = FastMath<double>::vmax(dmax_d, d);
                    dmax_n = FastMath<double>::vmax(dmax_n, g);
                }
                // float32 raster, float64 arithmetic, raw drivers: four pixels per thread interleaved
                // need more than 256 registers (one wave per SIMD, or spills inside the counted-vmcnt
                // loop); one pixel after the other they fit
                // (round 6, second session: with the loop's scalar bookkeeping in 32 bits the potential-ET and
                // component instances were scheduled into 258 - 260 as well: every float32-raster instance
                // of the float64 arithmetic but the totals one, which stays below 240 on its own)
                if constexpr (V == 4 && MODE != kStreamTotals) __builtin_amdgcn_sched_barrier(0);
            }
            }
            const int64_t first = out_first;
#pragma unroll
            for (int k = 0; k < NOUT; ++k)
                __builtin_nontemporal_store(res[k], reinterpret_cast<VT*>((a.out[k] + first) + lane_elem));
            // a flagged pixel in this piece: remember the piece (see redo_piece above)
            if constexpr (GUARD) {
                if constexpr (!stream_is_mixed(MODE) && !RAW) bad = gmax >= gc.huge;
                if (__builtin_expect(__any(bad), 0)) {
                    const int bit = a.static_sched ? iters : run;
                    const int last = a.static_sched ? kFlagBits - 1 : kCancelShift - 1;
                    flags |= 1ull << (bit < last ? bit : last);
                }
            }
        }
        ++iters;
        // diagnostics partial: one per run (dynamic schedule: which wave computes a run is not
        // fixed, the run's pixels are) or one per wave (static schedule: the wave's runs are;
        // stored behind the loop, once the wave has revisited its flagged pieces)
        const bool flushed = !a.static_sched && (cb_n + run_n >= npiece || run_n == 0);
        if (__builtin_expect(flushed, 0)) {
            const double f = diag_fields();
            // (agent scope = written through to memory: the block that adds the partials up
            // may sit behind another L2)
            if (lane < kDiag)
                __hip_atomic_store(a.diag_partial + (int64_t)(cbase >> rs) * kDiag + lane, f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        cbase = cb_n;
        run = run_n;
        lanes = lanes_n;
        out_first = of_n.o;
    }
    // What runs behind the loop reads the kernel arguments again (through a pointer the compiler
    // cannot see through): the pointers redo_piece needs would otherwise be live -- in scalar
    // registers, i.e. spilled to vector-register lanes -- all through the loop, and the loop's own
    // scalars with them (float64 totals instance: 45 -> 21 v_readlane_b32 per iteration, raw
    // drivers 76 -> 52; 0.7-0.9 % on the raw-driver and mixed instances, nothing on the float64
    // totals instance, whose time follows the package power -- DESIGN.md section 6)
    StreamArgs<T> late;
    {
        typedef const __attribute__((address_space(4))) char* kptr_t;
        kptr_t ka = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        __builtin_memcpy(&late, (const void*)ka, sizeof(late));
    }
    // -- dynamic schedule: the launch leaves its ticket counter at zero again. Every block counts
    // itself finished once all its waves have left the loop (none of them claims any more); the
    // block that counts last resets the ticket and the count. The next launch that uses this
    // counter starts behind this kernel, so it finds zero -- without a memset in front of every
    // launch: as a node of a captured graph that memset was seen to run LATE when replays were
    // queued back to back between two torch.distributed barriers (RCCL on the null stream):
    // replays then found the previous launch's final ticket, claimed nothing and left the
    // outputs of the step before in place (round 4; tools/scratch notes in DESIGN.md section 7).
    if (!late.static_sched) {
        // (a wave claims one run ahead: its last claim may still be on its way to memory -- it must
        // have RETURNED before the block counts itself finished, or it could land behind the reset)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned* fin = reinterpret_cast<unsigned*>(late.dyn_counter + 1);
            if (__hip_atomic_fetch_add(fin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(late.dyn_counter, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(fin, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // the next launch on this counter is another launch: its runs carry another marker
                // (every wave of this one has left the loop: all of them read the old number)
                __hip_atomic_fetch_add(fin + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // -- static schedule (small rasters): the wave's flagged pieces again in the reference's
    // operation order, then its one partial. Iteration i of wave w was piece
    // ((w + (i >> rs) nwaves) << rs) + (i & (rl - 1)); the last flag bit stands for every
    // iteration from there on.
    if (late.static_sched) {
        const int first_cbase = (int)((blockIdx.x * (kBlock / 64) + wave) << rs);
        // ONE partial per BLOCK: the block's four waves add theirs up in LDS (fixed order:
        // wave 0 + wave 1 + ...), so the block that sums the partials of the whole launch
        // afterwards reads a quarter of them -- one round trip to memory instead of two on
        // the path behind the last block's last store (1200 x 1200: 512 partials)
        __shared__ double wave_part[kBlock / 64][kDiag];
        double f = lane < 6 ? 0.0 : -__builtin_huge_val();        // a wave without work adds nothing
        if (first_cbase < npiece) {
            const unsigned long long mine =      // (diag_fields() stores and clears them)
                ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(flags >> 32)) << 32) | __builtin_amdgcn_readfirstlane((unsigned)flags);
            f = diag_fields();
            if constexpr (GUARD) if (__builtin_expect(mine != 0ull, 0)) {
                RedoAcc acc;
                const int64_t w0 = first_cbase >> rs;
#pragma nounroll
                for (int i = 0; i < iters; ++i) {
                    const int bit = i < kFlagBits - 1 ? i : kFlagBits - 1;
                    if (!((mine >> bit) & 1ull)) continue;
                    const int64_t piece = ((w0 + (int64_t)(i >> rs) * nwaves) << rs) + (i & (rl - 1));
                    redo_piece<T, MODE>(a, lut, tab, piece, lane, true, acc);
                }
                f = redo_fold(acc, lane, f);
            }
        }
        if (lane < kDiag) wave_part[wave][lane] = (lane == kFlagField || lane == kSerialField) ? 0.0 : f;
        __syncthreads();
        if (wave == 0 && lane < kDiag) {
            double g = wave_part[0][lane];
#pragma unroll
            for (int w = 1; w < kBlock / 64; ++w) {
                const double o = wave_part[w][lane];
                g = lane < 6 ? g + o : (o > g ? o : g);
            }
            __hip_atomic_store(late.diag_partial + (int64_t)blockIdx.x * kDiag + lane, g,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // -- diagnostics of a small raster, finished in this launch: every block counts itself
    // done once its partials are in memory; the block that counts last adds all of them up in
    // a fixed order (thread t: partials t, t + 256, ...; then the wave and block trees), so
    // the sums depend on n and the device only -- not on which block was last. Partials are
    // stored and loaded at agent scope (through the L2s, which are per XCD) and the stores
    // are waited for before the count: no cache write-back or invalidate is needed, and a
    // full release fence here (an L2 write-back per block, 512 blocks ending together)
    // measured 19 us on a 1200 x 1200 raster -- more than the two dispatches it replaces.
    // (MOD16_NO_FUSED_FINAL: compiled out, for instruction counts of the loop -- tools/isa_count.py)
#ifndef MOD16_NO_FUSED_FINAL
    if (late.diag_out) {
        __shared__ int last_block;
        __shared__ double fin[kBlock / 64][kDiag];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            last_block = __hip_atomic_fetch_add(late.done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
        __syncthreads();
        if (last_block) {
            double acc[kDiag] = {0, 0, 0, 0, 0, 0, -__builtin_huge_val(), -__builtin_huge_val()};
            // four partials in flight per thread
            for (int64_t b0 = threadIdx.x; b0 < late.nruns; b0 += 4 * kBlock) {
                double o[4][kDiag];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int64_t b = b0 + j * kBlock;
#pragma unroll
                    for (int k = 0; k < kDiag; ++k)
                        o[j][k] = b < late.nruns ? __hip_atomic_load(late.diag_partial + b * kDiag + k, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT)
                                              : (k < 6 ? 0.0 : -__builtin_huge_val());
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) diag_merge(acc, o[j]);
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                double o[kDiag];
#pragma unroll
                for (int k = 0; k < kDiag; ++k) o[k] = __shfl_down(acc[k], off, 64);
                diag_merge(acc, o);
            }
            if (lane == 0)
                for (int k = 0; k < kDiag; ++k) fin[wave][k] = acc[k];
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int w = 1; w < kBlock / 64; ++w) {
                    double o[kDiag];
                    for (int k = 0; k < kDiag; ++k) o[k] = fin[w][k];
                    diag_merge(acc, o);
                }
                late.diag_out[0] = acc[0];
                late.diag_out[1] = acc[1];
                late.diag_out[2] = (double)late.n - acc[4];
                late.diag_out[3] = (double)late.n - acc[5];
                late.diag_out[4] = acc[4];
                late.diag_out[5] = acc[5];
                late.diag_out[6] = acc[6];
                late.diag_out[7] = acc[7];
                __hip_atomic_store(late.done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#endif
}

// The runs' cancellation lists of a mixed-precision launch (dynamic schedule; see the loop), behind
// the pipeline kernel: a wave takes 64 runs at a time (lane l reads the flag field of run r0 + l), and
// the pixels of all 64 lists are computed side by side in the lanes of the wave -- entry e of the
// concatenated lists by lane e % 64: one gather and one pass through the float64 pixel function per 64
// pixels (going piece by piece, as the domain guard's pass does, cost 4.6 ms on the global grid: nine
// runs in ten hold such a pixel) -- and every run's owner lane then adds ITS pixels' totals to its
// partial in list order: the sums do not depend on which lane computed what. Bound by the gather:
// every listed pixel costs fifteen 64-byte sectors of HBM traffic where the pipeline read 65 bytes
// (global grid, one pixel in 250 listed: 0.36 ms next to the pipeline's 10.6).
template <typename T, int MODE>
__global__ void __launch_bounds__(kBlock) et_stream_cancel_kernel(const StreamArgs<T> a) {
    constexpr int V = 16 / (int)sizeof(T);
    __shared__ double lut[MOD16_LUT_ROWS * kLutCols];
    // who[wave][e] = (owner lane << 8 | entry of its list) of entry e of the 64 runs' concatenated
    // lists; vals[wave][lane] = the totals lane `lane` computed in this batch
    __shared__ uint16_t who[kBlock / 64][64 * kCancelCap];
    __shared__ double vals[kBlock / 64][64][2];
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) lut[i] = a.lut64[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + wv;
    const int rs = a.run_shift;
    for (int64_t r0 = wave * 64; r0 < a.nruns; r0 += nwaves * 64) {
        const int64_t mine = r0 + lane;
        const double fl = mine < a.nruns ? a.diag_partial[mine * kDiag + kFlagField] : 0.0;
        const int cnt = (int)((unsigned long long)fl >> kCancelShift);          // (<= kCancelCap)
        if (!__any(cnt != 0)) continue;
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            incl += lane >= off ? up : 0;
        }
        const int excl = incl - cnt, total = __shfl(incl, 63, 64);
        for (int k = 0; k < cnt; ++k) who[wv][excl + k] = (uint16_t)((lane << 8) | k);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (one wave: its LDS accesses are in order)
        double sum_d = 0.0, sum_n = 0.0, max_d = -__builtin_huge_val(), max_n = -__builtin_huge_val();
        unsigned num_d = 0, num_n = 0;
        for (int base = 0; base < total; base += 64) {
            const int e = base + lane;
            double d = __builtin_nan(""), g = __builtin_nan("");
            if (e < total) {
                const unsigned w = who[wv][e];
                const int64_t run = r0 + (w >> 8);
                const unsigned ent = a.cancel_list[run * kCancelCap + (w & 255u)];
                const int64_t piece = (run << rs) + (ent >> 8);
                const int64_t tile = piece >> a.tile_shift;
                const int64_t q = (piece - (tile << a.tile_shift)) * (int64_t)(64 * V) + (int64_t)((ent >> 2) & 63u) * V + (ent & 3u);
                redo_pixel<T, MODE, true>(a, lut, a.tab, tile * a.wide_row + q, tile * a.out_row + q, tile * a.byte_row + q, false, d, g);
            }
            vals[wv][lane][0] = d;
            vals[wv][lane][1] = g;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the owner adds its run's pixels of this batch, in list order
            const int lo = excl > base ? excl : base, hi = incl < base + 64 ? incl : base + 64;
            for (int i = lo; i < hi; ++i) {
                const double vd = vals[wv][i - base][0], vg = vals[wv][i - base][1];
                if (vd == vd) { sum_d += vd; num_d += 1u; max_d = vd > max_d ? vd : max_d; }
                if (vg == vg) { sum_n += vg; num_n += 1u; max_n = vg > max_n ? vg : max_n; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (before the next batch overwrites vals)
        }
        if (cnt != 0) {
            double* part = a.diag_partial + mine * kDiag;
            part[0] += sum_d;
            part[1] += sum_n;
            part[4] -= (double)num_d;
            part[5] -= (double)num_n;
            part[6] = max_d > part[6] ? max_d : part[6];
            part[7] = max_n > part[7] ? max_n : part[7];
        }
    }
}

// The flagged pieces of a large raster (dynamic schedule), behind the pipeline kernel (and, mixed-
// precision forms, behind et_stream_cancel_kernel): a wave takes 64 runs at a time (lane l reads the
// flag field of run r0 + l), and for every run with a piece flag revisits its flagged pieces -- pixels
// outside the domain of the production arithmetic; mixed-precision forms: also the marked pixels of a
// run whose list was full (redo_piece) -- and corrects the run's partial: it is the only one touching
// that run here, and the fixed-order sum over the partials runs behind this kernel. A raster
// without flagged pixels costs this kernel one 8-byte load per run (global grid: 3.5 MB).
template <typename T, int MODE>
__global__ void __launch_bounds__(kBlock) et_stream_redo_kernel(const StreamArgs<T> a) {
    constexpr int V = 16 / (int)sizeof(T);
    ignore_signalling_nans();       // the guard is asked again: the same answer as in the pipeline kernel
    __shared__ double lut[MOD16_LUT_ROWS * kLutCols];
    for (int i = threadIdx.x; i < MOD16_LUT_ROWS * kLutCols; i += kBlock) lut[i] = a.lut64[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * (kBlock / 64);
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int64_t npiece = (a.n / V + 63) / 64;
    const int rs = a.run_shift;
    // every run's partial must carry the marker of the launch in front of this kernel (which has
    // incremented the serial number since): a run without it was not processed
    const double marker = launch_marker(__hip_atomic_load(reinterpret_cast<const unsigned*>(a.dyn_counter) + 3,
                                                          __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1u);
    for (int64_t r0 = wave * 64; r0 < a.nruns; r0 += nwaves * 64) {
        const int64_t mine = r0 + lane;
        const double fl = mine < a.nruns ? a.diag_partial[mine * kDiag + kFlagField] : 0.0;
        const double mk = mine < a.nruns ? a.diag_partial[mine * kDiag + kSerialField] : marker;
        if (__any(mk != marker) && lane == 0) atomicOr(a.status, kStatusIncomplete);
        const unsigned long long flbits = (unsigned long long)fl & ((1ull << kCancelShift) - 1ull);
        unsigned long long any = __ballot(flbits != 0ull);
        while (any) {
            const int src = __builtin_ctzll(any);
            any &= any - 1ull;
            const int64_t run = r0 + src;
            const unsigned long long flags = (unsigned long long)__shfl(fl, src, 64) & ((1ull << kCancelShift) - 1ull);
            RedoAcc acc;
#pragma nounroll
            for (int i = 0; i < (1 << rs); ++i) {
                const int bit = i < kCancelShift - 1 ? i : kCancelShift - 1;
                const int64_t piece = (run << rs) + i;
                if (((flags >> bit) & 1ull) && piece < npiece)
                    redo_piece<T, MODE>(a, lut, a.tab, piece, lane, stream_is_mixed(MODE), acc);
            }
            double* part = a.diag_partial + run * kDiag;
            const double f = lane < kDiag ? part[lane] : 0.0;
            const double g = redo_fold(acc, lane, f);
            if (lane < kDiag) part[lane] = g;
        }
    }
}

}  // namespace mod16
