// mtq_fast.hip — K1 for bf16 STORAGE: fused BFP{8,4,2} quantise + per-tile reductions in exact
// integer arithmetic (gfx950 packed-16 VALU, v_dot2 / v_sad), staged through LDS by LDS-DMA.
//
// Why integers are exact here: a bf16 value is ±m·2^(e−134) with an 8-bit m.  Inside a
// shared-exponent group (max exponent E) every value within 14 binades of the maximum (the "main" class of
// include/mtq.h's summation order) is an integer multiple of 2^(E−148):  x = ±(128·a + b)·2^(E−148),
// a = m·2^(7−d) for d = E−e ≤ 7 (else 0), b = m·2^(14−d) for 8 ≤ d ≤ 14 (else 0).  All three BFP roundings
// act on `a` alone (values with d ≥ 8 quantise to 0 in every format), y = q·2^(15−mb) in units of `a`, so
// every float32 term the reference forms (x*x, y*y, x*y, |x−y|) is exact and the group sums are small integers:
//     Σy = Σ±y · 2^(E−141)            Σy² = Σq² · 2^(2(E−126−mb))        Σxy = Σa·q · 2^(2E−267−mb)
//     Σ|x−y| = (128·Σ|a−y| + Σb) · 2^(E−148)      max|x−y| = max(128·max|a−y|, max b) · 2^(E−148)
//     Σx = (128·Σ±a + Σ±b) · 2^(E−148)            Σx² = (16384·Σa² + Σb²) · 2^(2E−296)
// An exact integer times a power of two is exactly the float64 the literal route (sequential float64 sum
// of the float32 terms) produces for the main class, so the records are bit-identical to tile_stats_generic
// and the oracle.  Tail-class values (d ≥ 15; rare) are added in a short divergent loop.  A group whose E is
// outside [80,180] (float32 products could under/overflow; also denormal-only, Inf/NaN groups) marks its tile; marked
// tiles are recomputed by the literal route in a follow-up launch (tile_stats_redo_flagged, mtq_kernels.hip).
// An all-zero group marks nothing: it contributes ±0.0 to every sum.
//
// Mapping: a wave owns a 32-row × 128-column unit (4 tiles, 8 KiB).  8 LDS-DMA instructions
// (global_load_lds_dwordx4, 1 KiB each, 256-B contiguous row segments) fill a wave-private LDS image;
// 16 lanes serve one tile, lane j takes rows 2j, 2j+1 (4 groups, 2 × ds_read_b128 each, XOR-swizzled so the
// reads are bank-conflict free).  A lane sums its 4 groups sequentially; the per-lane float64 partials meet in
// an LDS scratch (14 sums per tile, balanced tree over the 16 lanes); the three maxima go through lane
// permutes.  The 4 finished records of a unit leave as two coalesced wave-stores.
// No MFMA, no block barrier: waves never share data.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mtq.h"
#include "mtq_device.hpp"
#include "mtq_error.hpp"

namespace mtq {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ us2 as_us2(uint32_t x) { return __builtin_bit_cast(us2, x); }
__device__ __forceinline__ s2 as_s2(uint32_t x) { return __builtin_bit_cast(s2, x); }
__device__ __forceinline__ uint32_t as_u32(us2 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ uint32_t as_u32(s2 x) { return __builtin_bit_cast(uint32_t, x); }

#ifndef MTQ_FAST_WAVES
#define MTQ_FAST_WAVES 2
#endif
// waves per block.  Waves never share data, so any block size works; it decides what a co-resident kernel displaces: a scan wave
// of csrc/mtq_scan.hip (126 VGPRs on one SIMD) keeps a whole 4-wave block off its CU but only half of two 2-wave blocks — measured
// inside the bench (round 2, two interleaved runs): 4 waves 735 M tiles/s (K1 2.72 ms), 2 waves 777 M (2.55 ms), 1 wave 759 M (2.52 ms).
constexpr int kFastWaves = MTQ_FAST_WAVES;
constexpr int kUnitTiles = 4;                        // tiles per wave unit
constexpr int kUnitCols = kUnitTiles * kTile;        // 128
constexpr int kInBytes = kTile * kUnitCols * 2;      // 8192 B of bf16 per unit
// Σx of a tile the exact route could not take is overwritten with this NaN pattern (low mantissa bits set: no
// float32-derived NaN can carry it); tile_stats_redo_flagged recomputes exactly those tiles by the literal route.
constexpr unsigned long long kRedoMagic = 0x7FF8C0DE5EED0001ull;
constexpr int kSums = 14;                            // Σx, Σx², 3 × (Σy, Σy², Σxy, Σ|d|)
constexpr int kScratchStride = 17;                   // 16 lanes + 1 pad (doubles)
constexpr int kRecDoubles = kUnitTiles * (2 + 5 * kNumFmt);             // 4 records of up to 22 doubles

// bijection on 4 bits with bit0 = bit2^bit3: makes the XOR swizzle conflict-free for the lane groups
// ds_read_b128 is serviced in ({0-3,12-15,20-27}, ...; MI355X_MICROARCH.md §LDS).
__device__ __forceinline__ uint32_t swz(uint32_t j) { return (((j >> 2) ^ (j >> 3)) & 1u) | ((j & 3u) << 1) | (j & 8u); }

struct Fmt8 { static constexpr uint32_t sh = 8, half = 0x007F007Fu, keep = 0xFF00FF00u, sat = 0x7F007F00u; static constexpr bool onebit = false; };
struct Fmt4 { static constexpr uint32_t sh = 12, half = 0x07FF07FFu, keep = 0xF000F000u, sat = 0x70007000u; static constexpr bool onebit = false; };
struct Fmt2 { static constexpr uint32_t sh = 14, half = 0x1FFF1FFFu, keep = 0xC000C000u, sat = 0x40004000u; static constexpr bool onebit = true; };

struct FmtAcc {           // integer group sums of one BFP format
    int ssy;              // Σ ±y
    uint32_t sq2, saq, sad;
    uint32_t dmax, dmin;  // packed running max / min of (a − y) + 0x8000 (unsigned halves)
};

// One element pair of one BFP format.  kSum: Σy, Σy², Σxy (what the greedy search reads of a format); kErr: Σ|x−y|, max|x−y| (what a
// map's mae / atol columns, the mae / atol searches and the threshold rule read).
template <typename F, bool kSum, bool kErr>
__device__ __forceinline__ void fmt_step(uint32_t a, uint32_t abias, uint32_t sgn, FmtAcc &A)
{
    // RNE of `a` to a multiple of G = 2^sh, saturating at (2^mb − 1)·G (quantization_formats.py:133-141)
    const uint32_t lsb = (a >> F::sh) & 0x00010001u;
    const uint32_t t = a + F::half + lsb;                       // no carry between the halves (max 0x9F80)
    const uint32_t y = as_u32(__builtin_elementwise_min(as_us2(t & F::keep), as_us2(F::sat)));
    if constexpr (kSum) {
        const uint32_t q = y >> F::sh;                          // low sh bits of each half are zero: no cross-talk
        A.ssy = __builtin_amdgcn_sdot2(as_s2(y), as_s2(sgn), A.ssy, false);
        if constexpr (F::onebit) A.sq2 += q;   // q ∈ {0,1}: Σq² = Σq, kept as two packed 16-bit counters (≤ 8 each), folded at the end
        else A.sq2 = __builtin_amdgcn_udot2(as_us2(q), as_us2(q), A.sq2, false);
        A.saq = __builtin_amdgcn_udot2(as_us2(a), as_us2(q), A.saq, false);
    }
    if constexpr (kErr) {
        A.sad = __builtin_amdgcn_sad_u16(a, y, A.sad);
        // a − y per half, biased by 0x8000 so that one 32-bit subtract serves both halves (no borrow: a|0x8000 ≥ y)
        const uint32_t db = abias - y;
        A.dmax = as_u32(__builtin_elementwise_max(as_us2(A.dmax), as_us2(db)));
        A.dmin = as_u32(__builtin_elementwise_min(as_us2(A.dmin), as_us2(db)));
    }
}

__device__ __forceinline__ uint32_t fmt_maxabs(const FmtAcc &A)
{
    // biased extremes → max |a − y| over both halves: max(dmax − 0x8000, 0x8000 − dmin)
    const uint32_t up = max(A.dmax & 0xFFFFu, A.dmax >> 16) - 0x8000u;
    const uint32_t dn = 0x8000u - min(A.dmin & 0xFFFFu, A.dmin >> 16);
    return max(up, dn);
}

// v_pk_lshrrev_b16: per-half logical right shift; every caller keeps the shift amounts at 15 or below.  (Round 2 had this as inline
// asm; the compiler then put an `s_nop 0` — four issue cycles — behind each of the three per element pair.)
__device__ __forceinline__ uint32_t pk_lshr(uint32_t v, uint32_t sh) { return as_u32(as_us2(v) >> as_us2(sh)); }

// Per-group result handed to the tree: the 14 float64 terms of the group and its 3 float32 maxima.
struct GroupOut {
    double term[kSums];
    float mx[3];
    bool bad;   // outside the exact route's preconditions: the tile is redone by the literal fix-up kernel
};

// What an instantiation evaluates (bit 0 bfp8, 1 bfp4, 2 bfp2).  SUMS: formats whose Σy, Σy², Σxy are formed; ERRS: formats whose
// Σ|x−y|, max|x−y| are; XS: Σx, Σx² too.  The 14 group terms in record order: Σx, Σx², then per format Σy, Σy², Σxy, Σ|x−y|.
//   mtq_tile_stats (bf16 storage, mask 0xE): SUMS = ERRS = 7, XS;   mtq_tile_stats_partial of the streamed greedy search: SUMS 3, ERRS 1, XS;
//   mtq_tile_stats_listed behind it: SUMS 4, ERRS 6, no XS.
__host__ __device__ constexpr bool term_needed(uint32_t sums, uint32_t errs, bool xs, int s)
{
    return s < 2 ? xs : ((((s - 2) & 3) < 3 ? sums : errs) >> ((s - 2) >> 2)) & 1u;
}
__host__ __device__ constexpr int terms_needed(uint32_t sums, uint32_t errs, bool xs)
{
    int n = 0;
    for (int s = 0; s < kSums; ++s) n += term_needed(sums, errs, xs, s) ? 1 : 0;
    return n;
}
__host__ __device__ constexpr int term_at(uint32_t sums, uint32_t errs, bool xs, int i)
{
    int n = 0;
    for (int s = 0; s < kSums; ++s) {
        if (!term_needed(sums, errs, xs, s)) continue;
        if (n == i) return s;
        ++n;
    }
    return 0;
}
// One reduce pass takes up to 6 statistics: 4 tiles × 6 × 17 doubles of scratch + the record image (704 B) inside HALF the input image
constexpr int kPassMax = 6;
constexpr int kHalfBytes = kInBytes / 2;             // the rows 2j of a unit (16 x 256 B) / the rows 2j+1

template <uint32_t SUMS, uint32_t ERRS, bool XS, typename Reload>
__device__ __forceinline__ void fast_group(const uint32_t w[8], GroupOut &G, Reload reload)
{
    constexpr uint32_t ANY = SUMS | ERRS;
    constexpr bool need_sgn = XS || SUMS != 0u, need_b = XS || ERRS != 0u;
    uint32_t ab[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ab[i] = w[i] & 0x7FFF7FFFu;
    us2 m01 = __builtin_elementwise_max(as_us2(ab[0]), as_us2(ab[1])), m23 = __builtin_elementwise_max(as_us2(ab[2]), as_us2(ab[3]));
    us2 m45 = __builtin_elementwise_max(as_us2(ab[4]), as_us2(ab[5])), m67 = __builtin_elementwise_max(as_us2(ab[6]), as_us2(ab[7]));
    const uint32_t mxp = as_u32(__builtin_elementwise_max(__builtin_elementwise_max(m01, m23), __builtin_elementwise_max(m45, m67)));
    const uint32_t E = max(mxp & 0xFFFFu, mxp >> 16) >> 7;       // shared exponent (quantization_formats.py:118-119)
    const uint32_t Ep = E | (E << 16);

    int sxa = 0, sxb = 0;
    uint32_t sb2 = 0u, sbs = 0u, bmaxp = 0u, dor = 0u, sa2c = 0u;
    uint64_t sa2 = 0ull;
    FmtAcc A8 = {0, 0u, 0u, 0u, 0x80008000u, 0x80008000u}, A4 = A8, A2 = A8; // biased extremes start at δ = 0: a phantom zero never changes max |δ|
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t e = pk_lshr(ab[i], 0x00070007u);          // per-half exponent field (the sign bits are already cleared)
        const uint32_t d = as_u32(__builtin_elementwise_min(as_us2(Ep - e), as_us2(0x000F000Fu))); // min(E−e, 15) per half
        dor |= d + 0x00010001u;                                 // bit 4 of a half set ⇔ that element is in the tail class
        const uint32_t m = (ab[i] & 0x007F007Fu) | 0x00800080u;
        const uint32_t a = as_u32(as_us2(m) * as_us2(pk_lshr(0x00800080u, d)));                 // m·2^(7−d), 0 for d ≥ 8
        uint32_t b = 0u, sgn = 0u;
        if constexpr (need_b) b = as_u32(as_us2(m) * as_us2(pk_lshr(0x40004000u, d) & 0x007F007Fu));   // m·2^(14−d), 8 ≤ d ≤ 14
        if constexpr (need_sgn) sgn = as_u32(as_s2(w[i]) >> (short)15) | 0x00010001u;                  // ±1 per half
        if constexpr (XS) {
            sxa = __builtin_amdgcn_sdot2(as_s2(a), as_s2(sgn), sxa, false);
            sxb = __builtin_amdgcn_sdot2(as_s2(b), as_s2(sgn), sxb, false);
            sb2 = __builtin_amdgcn_udot2(as_us2(b), as_us2(b), sb2, false);                    // b ≤ 0x3FC0: 16 squares fit
            // Σa²: four squares (a ≤ 0x7F80) fit 32 bits; every second pair the chunk moves to the 64-bit sum
            sa2c = __builtin_amdgcn_udot2(as_us2(a), as_us2(a), (i & 1) ? sa2c : 0u, false);
            if (i & 1) sa2 += sa2c;
        }
        if constexpr (ERRS != 0u) {
            sbs = __builtin_amdgcn_udot2(as_us2(b), as_us2(0x00010001u), sbs, false);
            bmaxp = as_u32(__builtin_elementwise_max(as_us2(bmaxp), as_us2(b)));
        }
        const uint32_t abias = a | 0x80008000u;
        if constexpr (ANY & 1u) fmt_step<Fmt8, (SUMS & 1u) != 0, (ERRS & 1u) != 0>(a, abias, sgn, A8);
        if constexpr (ANY & 2u) fmt_step<Fmt4, (SUMS & 2u) != 0, (ERRS & 2u) != 0>(a, abias, sgn, A4);
        if constexpr (ANY & 4u) fmt_step<Fmt2, (SUMS & 4u) != 0, (ERRS & 4u) != 0>(a, abias, sgn, A2);
    }
    // exact route needs every float32 term normal and finite: E in [80, 180]; anything else marks the tile — except an
    // all-zero group (pruned weights, padding), which contributes nothing: its (garbage) integers are scaled by
    // 2^−3000 → ±0.0, and adding ±0.0 changes no accumulator.
    const bool zero_group = mxp == 0u;
    G.bad = !((E >= 80u) & (E <= 180u)) && !zero_group;
    const uint32_t bmax = max(bmaxp & 0xFFFFu, bmaxp >> 16);
    const int Ei = zero_group ? -3000 : (int)E;
    const int e1 = Ei - 148, e2 = 2 * Ei - 296;                 // 2^(E−148), 2^(2E−296)
    if constexpr (XS) {
        G.term[0] = __builtin_ldexp((double)(sxa * 128 + sxb), e1);
        G.term[1] = __builtin_ldexp(__builtin_fma((double)sa2, 16384.0, (double)sb2), e2);          // exact (< 2^49)
    } else {
        G.term[0] = G.term[1] = 0.0;
    }
    const float sf = zero_group ? 0.0f : __uint_as_float(((E - 21u) & 0xFFu) << 23); // 2^(E−148) as float32 (masked: garbage-safe when bad)
    const FmtAcc *A[3] = {&A8, &A4, &A2};
    const int mbs[3] = {7, 3, 1};
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        G.term[2 + 4 * f] = G.term[3 + 4 * f] = G.term[4 + 4 * f] = G.term[5 + 4 * f] = 0.0;
        G.mx[f] = 0.0f;
        if (SUMS & (1u << f)) {
            G.term[2 + 4 * f] = __builtin_ldexp((double)A[f]->ssy, e1 + 7);                       // Σ±y · 2^(E−141)
            const uint32_t sq2 = f == 2 ? (A[f]->sq2 & 0xFFFFu) + (A[f]->sq2 >> 16) : A[f]->sq2;
            G.term[3 + 4 * f] = __builtin_ldexp((double)sq2, 2 * (Ei - 126 - mbs[f]));            // Σq² · 2^(2(E−126−mb))
            G.term[4 + 4 * f] = __builtin_ldexp((double)A[f]->saq, 2 * Ei - 267 - mbs[f]);        // Σa·q · 2^(2E−267−mb)
        }
        if (ERRS & (1u << f)) {
            G.term[5 + 4 * f] = __builtin_ldexp((double)((A[f]->sad << 7) + sbs), e1);            // (128·Σ|a−y| + Σb) · 2^(E−148)
            G.mx[f] = (float)max(fmt_maxabs(*A[f]) << 7, bmax) * sf; // integer < 2^23: exact
        }
    }
    // tail class (more than 14 binades below the maximum; y = 0 in every BFP format): summed separately in
    // index order and added once — S = S_main + S_tail (include/mtq.h).  Zeros contribute nothing and are skipped.
    if (__builtin_expect((dor & 0x00100010u) != 0u, 0)) {
        double tx = 0.0, tx2 = 0.0, tab = 0.0;
        float tmx = 0.0f;
        uint32_t w2[8];
        reload(w2);                                             // the group again, from the LDS image (keeps w[] dead after the pair loop)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t a2 = w2[i] & 0x7FFF7FFFu;
            const uint32_t dd = Ep - ((a2 >> 7) & 0x00FF00FFu);
            if ((dd & 0xFFFFu) > 14u && (a2 & 0xFFFFu) != 0u) {
                const float xv = __uint_as_float(w2[i] << 16), p = xv * xv, av = fabsf(xv);
                tx += (double)xv; tx2 += (double)p; tab += (double)av; tmx = fmaxf(tmx, av);
            }
            if ((dd >> 16) > 14u && (a2 >> 16) != 0u) {
                const float xv = __uint_as_float(w2[i] & 0xFFFF0000u), p = xv * xv, av = fabsf(xv);
                tx += (double)xv; tx2 += (double)p; tab += (double)av; tmx = fmaxf(tmx, av);
            }
        }
        if constexpr (XS) { G.term[0] += tx; G.term[1] += tx2; }
#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (!(ERRS & (1u << f))) continue;
            G.term[5 + 4 * f] += tab; G.mx[f] = fmaxf(G.mx[f], tmx);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same group in the FLOAT domain (round 4; -DMTQ_K1_INTDOM selects the packed-integer form above instead).
//
// Nothing is decoded: the bf16 pair is widened to two float32 values (one shift, one and) and every BFP rounding is done where the
// value stands.  With P = 2^(E−127) (the group maximum's power of two), a format of M mantissa bits has the step q = P·2^(1−M):
//     r = x + C,  C = 1.5·2^23·q      r is x rounded to the nearest multiple of q, ties to the even multiple (C is an even multiple
//     y = r − C   (exact)             of q and |x| < 2P ≪ C/3, so r stays in C's binade) — the reference's RNE of the aligned mantissa
//     y = med3(y, −(2^M−1)·q, (2^M−1)·q)   its saturating round-up (quantization_formats.py:133-141)
// (bf16 input has 8 significant bits, so the alignment shift of :126-131 loses nothing for d = E−e ≤ 16 and values further down
// round to 0 either way.)  Which float32 accumulations are EXACT — a sum of ≤ 16 integers below 2^24 times one power of two, hence equal
// to the float64 sum of include/mtq.h's order — with u = 2^(E−148), every main-class value (d ≤ 14) an integer multiple of u below 2^22·u:
//     Σy (≤ 2^(M+4) steps), Σy² (≤ 2^(2M+4) steps²): always;
//     bfp8: y ≠ 0 ⇒ d ≤ 7 ⇒ δ = x − y is a multiple of 2^7·u, |δ| ≤ 2^15·u ⇒ Σδ·y ≤ 2^19 units of 2^22·u²: exact, and
//           Σxy = Σy² + Σδ·y (both exact, joined in float64; Σx·y itself may need 26 bits);
//     bfp4, bfp2: y ≠ 0 ⇒ d ≤ 3 (1) ⇒ x·y is a multiple of 2^30 (2^34)·u² below 2^44·u² ⇒ Σx·y ≤ 18 bits: exact directly;
//     Σ|x−y|: |δ| ≤ 2^15·u (bfp8), 2^19·u (bfp4), 2^21·u (bfp2), integers in u ⇒ sums of 16 ≤ 2^19, 2^23 — bfp2 needs 2^25 and
//           is kept in two accumulators of 8 elements (≤ 2^24 each) joined in float64;
//     max|x−y|: a maximum is exact.
// Σx and Σx² need up to 26 / 48 bits: they are float64 chains over the elements (v_cvt_f64_f32, v_add_f64, v_fma_f64 — x² is exact in
// float64, so the fused form equals the literal add of the float32 product), exact for the main class (< 2^53·u²).
// A lane whose group holds a non-zero TAIL element (d ≥ 15: |x| < P·2^−14; about 6e-5 of Gaussian elements) redoes Σx, Σx² and
// Σ|x−y| of that group in the contract's main + tail form in a divergent loop (the maxima and the y-side sums are unaffected: y = 0
// there).  Groups with E outside [80,180] or a NaN mark the tile for the literal fix-up, all-zero groups contribute ±0, as above.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float u2f(uint32_t v) { return __uint_as_float(v); }
__device__ __forceinline__ uint32_t f2u(float v) { return __float_as_uint(v); }
__device__ __forceinline__ void max3_abs(float &m, float a, float b) { asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b)); }
__device__ __forceinline__ void min3_abs(float &m, float a, float b) { asm("v_min3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b)); }

template <int M> struct FmtF {   // per-group constants of one BFP format as multiples of P = 2^(E−127)
    static constexpr float kC = M == 7 ? 196608.0f : (M == 3 ? 3145728.0f : 12582912.0f);   // 1.5·2^(24−M)
    static constexpr float kHi = M == 7 ? 1.984375f : (M == 3 ? 1.75f : 1.0f);              // (2^M − 1)·2^(1−M)
};

struct FAcc {   // float32 group sums of one BFP format (packed: element 2i in .x, 2i+1 in .y)
    f2 sy, sy2, sxy;     // Σy, Σy², bfp8: Σδ·y, bfp4 / bfp2: Σx·y
    f2 sd;               // bfp8 beside Σx: Σδ (Σx = Σy + Σδ, both exact: |Σδ| ≤ 2^19·u)
    float sad, sad2;     // Σ|x−y| (bfp2: elements 0..7 and 8..15)
    float mx;            // max|x−y|
};

template <int M, bool kSum, bool kErr, bool kSd = false>
__device__ __forceinline__ void f32_step(f2 x, float C, float hi, int i, FAcc &A)
{
    const f2 r = (x + C) - C;
    f2 y;
    y.x = __builtin_amdgcn_fmed3f(r.x, -hi, hi);
    y.y = __builtin_amdgcn_fmed3f(r.y, -hi, hi);
    if constexpr (kSum) {
        A.sy += y;
        A.sy2 = __builtin_elementwise_fma(y, y, A.sy2);
    }
    if constexpr (kErr || (kSum && M == 7)) {
        const f2 d = x - y;
        if constexpr (kSd) A.sd += d;
        if constexpr (kSum && M == 7) A.sxy = __builtin_elementwise_fma(d, y, A.sxy);
        if constexpr (kErr) {
            if (M == 1 && i >= 4) { A.sad2 += __builtin_fabsf(d.x); A.sad2 += __builtin_fabsf(d.y); }
            else { A.sad += __builtin_fabsf(d.x); A.sad += __builtin_fabsf(d.y); }
            max3_abs(A.mx, d.x, d.y);
        }
    }
    if constexpr (kSum && M != 7) A.sxy = __builtin_elementwise_fma(x, y, A.sxy);
}

// the group's terms of one format from its float32 sums
template <int M, bool kSum, bool kErr>
__device__ __forceinline__ void f32_terms(const FAcc &A, double *t, float &mx)
{
    if constexpr (kSum) {
        const double sy2 = (double)(A.sy2.x + A.sy2.y), sxy = (double)(A.sxy.x + A.sxy.y);
        t[0] = (double)(A.sy.x + A.sy.y);
        t[1] = sy2;
        t[2] = M == 7 ? sy2 + sxy : sxy;
    }
    if constexpr (kErr) {
        t[3] = M == 1 ? (double)A.sad + (double)A.sad2 : (double)A.sad;
        mx = A.mx;
    }
}

template <uint32_t SUMS, uint32_t ERRS, bool XS, typename Reload>
__device__ __forceinline__ void fast_group_f32(const uint32_t w[8], GroupOut &G, Reload reload)
{
    constexpr uint32_t ANY = SUMS | ERRS;
    constexpr bool XSD = XS && (SUMS & 1u) != 0;                   // Σx from bfp8's Σy + Σδ instead of a float64 chain
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i].x = u2f(w[i] << 16); x[i].y = u2f(w[i] & 0xFFFF0000u); }
    float amax = 0.0f, amin = __builtin_inff();
#pragma unroll
    for (int i = 0; i < 8; ++i) { max3_abs(amax, x[i].x, x[i].y); min3_abs(amin, x[i].x, x[i].y); }
    const uint32_t E = f2u(amax) >> 23;                           // shared exponent (quantization_formats.py:118-119); 255 for Inf (and NaN, should the maximum keep one)
    const bool zero_group = f2u(amax) == 0u, out_of_range = (E - 80u) > 100u;
    const float P = u2f((out_of_range ? 127u : E) << 23);          // finite constants whatever E is: a marked tile's numbers are never used
    const float c8 = P * FmtF<7>::kC, c4 = P * FmtF<3>::kC, c2 = P * FmtF<1>::kC;
    const float h8 = P * FmtF<7>::kHi, h4 = P * FmtF<3>::kHi, h2 = P;
    const float thr = P * 0x1p-14f;                                // smallest main-class magnitude (formed again inside the tail branch)

    double sx = 0.0, sx2 = 0.0;
    FAcc A8 = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, 0.0f, 0.0f, 0.0f}, A4 = A8, A2 = A8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (XS) {
            const double x0 = (double)x[i].x, x1 = (double)x[i].y;
            if constexpr (!XSD) sx += x0;
            sx2 = __builtin_fma(x0, x0, sx2);
            if constexpr (!XSD) sx += x1;
            sx2 = __builtin_fma(x1, x1, sx2);
        }
        if constexpr (ANY & 1u) f32_step<7, (SUMS & 1u) != 0, (ERRS & 1u) != 0, XSD>(x[i], c8, h8, i, A8);
        if constexpr (ANY & 2u) f32_step<3, (SUMS & 2u) != 0, (ERRS & 2u) != 0>(x[i], c4, h4, i, A4);
        if constexpr (ANY & 4u) f32_step<1, (SUMS & 4u) != 0, (ERRS & 4u) != 0>(x[i], c2, h2, i, A2);
    }
#pragma unroll
    for (int s = 0; s < kSums; ++s) G.term[s] = 0.0;
    G.mx[0] = G.mx[1] = G.mx[2] = 0.0f;
    if constexpr (XS) { G.term[0] = XSD ? (double)(A8.sy.x + A8.sy.y) + (double)(A8.sd.x + A8.sd.y) : sx; G.term[1] = sx2; }
    if constexpr (ANY & 1u) f32_terms<7, (SUMS & 1u) != 0, (ERRS & 1u) != 0>(A8, G.term + 2, G.mx[0]);
    if constexpr (ANY & 2u) f32_terms<3, (SUMS & 2u) != 0, (ERRS & 2u) != 0>(A4, G.term + 6, G.mx[1]);
    if constexpr (ANY & 4u) f32_terms<1, (SUMS & 4u) != 0, (ERRS & 4u) != 0>(A2, G.term + 10, G.mx[2]);
    // a NaN element: max3 / min3 pass it over, every sum it enters does not — probe one such sum per instantiation
    double probe = 0.0;
    if constexpr (XS) probe = sx2;
    else if constexpr (ERRS != 0u) probe = G.term[(ERRS & 1u) ? 5 : ((ERRS & 2u) ? 9 : 13)];
    else probe = G.term[(SUMS & 1u) ? 4 : ((SUMS & 2u) ? 8 : 12)];
    G.bad = (out_of_range && !zero_group) || probe != probe;

    // tail class present (a zero element alone also gets here — the loop then changes nothing but the order of a sum of exact terms).
    // A ROLLED loop over the group's words, read again from the LDS image one at a time, with the constants formed again from P: what the
    // branch keeps alive beside the fast path's registers decides the kernel's occupancy (unrolled: 149 VGPRs for <3,1>, 126 without it).
    if (__builtin_expect(amin < thr && !zero_group && !out_of_range, 0)) {
        double mx_ = 0.0, mx2 = 0.0, tx = 0.0, tx2 = 0.0, tab = 0.0, m8 = 0.0, m4 = 0.0, m2 = 0.0;
#pragma nounroll
        for (int i = 0; i < 8; ++i) {
            const uint32_t wi = reload(i);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float xv = h == 0 ? u2f(wi << 16) : u2f(wi & 0xFFFF0000u);
                const double xd = (double)xv;
                if (__builtin_fabsf(xv) < P * 0x1p-14f) {          // tail (zeros included): y = 0 in every BFP format, |x − y| = |x|
                    tx += xd; tx2 = __builtin_fma(xd, xd, tx2); tab += __builtin_fabs(xd);
                } else {
                    mx_ += xd; mx2 = __builtin_fma(xd, xd, mx2);
                    if constexpr (ERRS & 1u) m8 += (double)__builtin_fabsf(xv - __builtin_amdgcn_fmed3f((xv + P * FmtF<7>::kC) - P * FmtF<7>::kC, -(P * FmtF<7>::kHi), P * FmtF<7>::kHi));
                    if constexpr (ERRS & 2u) m4 += (double)__builtin_fabsf(xv - __builtin_amdgcn_fmed3f((xv + P * FmtF<3>::kC) - P * FmtF<3>::kC, -(P * FmtF<3>::kHi), P * FmtF<3>::kHi));
                    if constexpr (ERRS & 4u) m2 += (double)__builtin_fabsf(xv - __builtin_amdgcn_fmed3f((xv + P * FmtF<1>::kC) - P * FmtF<1>::kC, -P, P));
                }
            }
        }
        if constexpr (XS) { G.term[0] = mx_ + tx; G.term[1] = mx2 + tx2; }
        if constexpr (ERRS & 1u) G.term[5] = m8 + tab;
        if constexpr (ERRS & 2u) G.term[9] = m4 + tab;
        if constexpr (ERRS & 4u) G.term[13] = m2 + tab;
    }
}

// ---------------------------------------------------------------------------------------------
// The kernel.  The 4 groups of a lane run in a ROLLED loop that reads each group from the LDS image just
// before use (a fully unrolled, software-pipelined form needed 230 VGPRs and ran no faster: the kernel is
// VALU-issue bound).  The reduce scratch and the record image
// overlay the input image (8 KiB of LDS per wave) and the next unit's DMA is issued once the records sit in
// registers.  Latency is hidden by occupancy instead of by a software pipeline.
// ---------------------------------------------------------------------------------------------
constexpr int kRolledWaveLds = kInBytes; // 8192 B: input image, then (scratch | records 704 B)

// LDS-DMA through inline asm (the compiler then does not drain vmcnt before unrelated LDS reads; waits are ours).
__device__ __forceinline__ void glds16(const void *sbase, uint32_t voff, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
// the same with a per-lane 64-bit address (the listed form: the four tiles of a unit lie anywhere)
__device__ __forceinline__ void glds16v(const void *vaddr, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(vaddr), "s"(lds_addr) : "memory");
}

// waves per SIMD an instantiation is compiled and launched for: the three-format kernel needs 146 VGPRs (3 waves); every smaller
// evaluation fits 128 (4 waves: 8 KiB of LDS per wave allows 5).  -DMTQ_ROLLED_WAVES_PER_SIMD_FORCE=n overrides both.
__host__ __device__ constexpr int rolled_waves(uint32_t sums, uint32_t errs)
{
#ifdef MTQ_ROLLED_WAVES_PER_SIMD_FORCE
    return MTQ_ROLLED_WAVES_PER_SIMD_FORCE;
#elif defined(MTQ_K1_INTDOM)
    return (sums == 7u && errs == 7u) ? 3 : 4;
#else
    // float-domain form: the registers an instantiation wants without a cap (hipcc 7.2: <1,0> 89, <1,1> 101, <6,6> 127, <3,1> 130,
    // <3,3> 143, <7,7> 174) decide; 8 KiB of LDS per wave allows 5
    const uint32_t any = sums | errs;
    const int n = (int)((any & 1u) + ((any >> 1) & 1u) + ((any >> 2) & 1u));
    // <3,1> fits 128 registers (4 waves) — and then runs no faster alone (1.31 against 1.33 ms per 128 x 4096^2) and costs the streamed
    // search 15 %: a search wave (176 registers, csrc/mtq_scan.hip) is placed where ONE retiring K1 wave leaves room beside three
    // waves of <= 168, but needs two to retire at once beside four of 128 (bench: 1134 against 982 M tiles/s)
    if (n >= 2) return 3;
    return errs == 0u ? 5 : 4;
#endif
}

struct PlaceArgs { uint32_t eval_mask; int o_bf16, o8, o4, o2; };
__device__ __forceinline__ void place_stat(const PlaceArgs pa, double *rec_t, int sidx, double r)
{
    if (sidx == 0) {
        rec_t[0] = r;
        if (pa.eval_mask & 1u) {
            const double z = __builtin_fabs(r) * 0.0;
            rec_t[pa.o_bf16] = r; rec_t[pa.o_bf16 + 3] = z; rec_t[pa.o_bf16 + 4] = z;
        }
    } else if (sidx == 1) {
        rec_t[1] = r;
        if (pa.eval_mask & 1u) { rec_t[pa.o_bf16 + 1] = r; rec_t[pa.o_bf16 + 2] = r; }
    } else {
        const int f = (sidx - 2) >> 2, k = (sidx - 2) & 3;
        const int o = f == 0 ? pa.o8 : (f == 1 ? pa.o4 : pa.o2);
        if (pa.eval_mask & (2u << f)) rec_t[o + k] = r;
    }
}

// Arguments of the listed form (LISTED): the unit's four tiles come from a device list instead of the unit counter.
struct ListedArgs {
    const uint32_t *list, *n_list;   // tensor * tiles + tile; the list's length on the device
    uint32_t cap;
    uint32_t *redo, *n_redo;         // tiles the exact route cannot take are appended here (the caller runs them through the direct listed kernel)
    uint32_t tiles_w32;              // tiles per tile row
};

template <uint32_t SUMS, uint32_t ERRS, bool XS, bool LISTED>
__global__ __launch_bounds__(kFastWaves * 64, rolled_waves(SUMS, ERRS)) void tile_stats_bf16_rolled(
    const uint16_t *__restrict__ x, int64_t stride, int64_t ld, int tiles_w, int64_t tiles, int units_w, int units_per_tensor,
    int total_units, uint32_t fmt_mask, uint32_t eval_mask, uint32_t part_mask, int rec, double *__restrict__ stats, unsigned *__restrict__ work,
    unsigned launch_id, int units_per_wave, ListedArgs la, unsigned *__restrict__ mark, int self_reset)
{
    // fmt_mask: the record LAYOUT (which slots exist); eval_mask ⊂ fmt_mask: the slots this launch writes; part_mask ⊂ eval_mask: those
    // of them that only get Σy, Σy², Σxy.  Slots of the layout that are not written hold NaN afterwards (mtq_tile_stats_partial).
    // LISTED: nothing but the statistics the instantiation forms is written, each straight to its place in the tile's record.
    constexpr int nsum = terms_needed(SUMS, ERRS, XS);
    // the reduce runs in passes of at most kPassMax statistics: its scratch and the record image overlay ONE HALF of the input image
    constexpr int npass = (nsum + kPassMax - 1) / kPassMax, per = (nsum + npass - 1) / npass;
    constexpr int scratch_doubles = kUnitTiles * per * kScratchStride;
    static_assert((scratch_doubles + kRecDoubles) * 8 <= kHalfBytes, "reduce scratch + record image must fit in half the input image");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char *in = lds + wave * kRolledWaveLds;
    const uint32_t in_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)in;
    const uint32_t t = lane >> 4, j = lane & 15;

    // The image is two halves of 4 KiB: the unit's rows 2j (16 row segments of 256 B, what a lane's groups 0 and 1 read) in one, the rows
    // 2j+1 (groups 2 and 3) in the other — and the halves swap roles from unit to unit (`par`), so that the NEXT unit's even rows are
    // fetched into the half the odd rows have just been read out of, while the reduce of this unit still works in the other half:
    //     ... g2 g3 | DMA(next even rows -> B) | reduce in A, records out | DMA(next odd rows -> A) | next unit: g0 g1 read B ...
    // The rows a unit needs first have then been under way since before the previous unit's reduce, the others since its end with two
    // groups' arithmetic still ahead of them (round 3 issued all eight pieces at the end: a quarter of a wave's cycles went to waiting
    // for them, SQ_WAIT_ANY).  Piece i (1 KiB, one LDS-DMA instruction): even (i < 4) or odd (i >= 4) rows 8(i&3) + 2*rho + (i>>2),
    // rho = lane >> 4, at row slot 4(i&3) + rho of its half; chunk XOR-swizzled by the slot (= the j that reads it), as before.
    uint32_t dma_off[8], dma_tile[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t rho = lane >> 4;
        const uint32_t c = (lane & 15u) ^ swz(4u * (i & 3) + rho);
        const uint32_t row = 8u * (i & 3) + 2u * rho + (uint32_t)(i >> 2);
        dma_off[i] = LISTED ? (uint32_t)(row * ld * 2) + (c & 3u) * 16u : (uint32_t)(row * ld * 2) + c * 16u;
        dma_tile[i] = c >> 2;                                                 // which of the unit's four tiles this lane's chunk belongs to
    }
    const uint32_t rd_row = j * 256u;                                         // row slot j of either half
    const uint32_t c0 = (4u * t) ^ swz(j);                                    // chunk (4t+kl)^swz(j) = c0 ^ kl

    // Units are claimed from a device counter (zero at launch, reset by the follow-up kernel) rather than by a fixed
    // stride: when other work (a copy kernel, a profiler's blit) keeps part of the persistent grid from being resident,
    // the resident waves take the whole queue instead of the launch waiting for the late blocks' fixed shares.
    // One counter per group of blocks (kWorkGroups counters on separate 128-B lines; ≈ 12 ns per same-address atomic on
    // MI355X, so a single queue would serialise the launch): group g of G = min(blocks, kWorkGroups) owns units g, g + G, …
    // (The listed form strides: its launches are short, and it has no follow-up kernel to reset a counter.)
    const int groups = (int)min(gridDim.x, (unsigned)kWorkGroups), group = (int)(blockIdx.x % (unsigned)groups);
    unsigned *queue = LISTED ? nullptr : work + group * kWorkStride;
    uint32_t n_listed = 0u;
    if constexpr (LISTED) {
        n_listed = min(*la.n_list, la.cap);
        total_units = (int)((n_listed + 3u) >> 2);
    }
    int stride_next = (int)(blockIdx.x * kFastWaves) + wave;                  // listed form: this wave's next unit
    // A claim is two steps: the atomic is ISSUED at the top of a unit and its value READ after the unit's groups.  Through the
    // compiler's atomic the value cannot stay in flight: its wait-count pass puts `s_waitcnt vmcnt(0)` behind the atomic (the atomic
    // optimizer broadcasts the value at once) or, with the optimizer off, at the top of the next unit (the destination register is
    // written again there) — a device-atomic round trip at the head of every unit in round 3, and with it a wait for every piece of
    // the image.  So the in-loop claim is inline assembly the compiler's counters do not see: lane 0 alone issues it (EXEC narrowed
    // and restored inside the statement), nothing reads `v` before claim_value's own statement, which waits first; volatile asm
    // statements keep their order, the kernel has no scratch (checked in the build's resource report: a spill of `v` would read it early).
    auto claim_first = [&]() -> int {
        if constexpr (LISTED) {
            const int u = stride_next;
            stride_next += (int)gridDim.x * kFastWaves;
            return u;
        } else {
            unsigned v = 0u;
            if (lane == 0) v = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane(v);
            return k < 0x01000000u ? group + (int)k * groups : 0x7FFFFFFF;
        }
    };
    auto claim_issue = [&]() -> unsigned {
        unsigned v;
        if constexpr (LISTED) {
            v = (unsigned)stride_next;
            stride_next += (int)gridDim.x * kFastWaves;
        } else {
            unsigned long long saved;
            asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %0, %2, %3, %4 sc0\n\ts_mov_b64 exec, %1"
                         : "=&v"(v), "=&s"(saved) : "v"(0u), "v"(1u), "s"(queue) : "memory");
        }
        return v;
    };
    auto claim_value = [&](unsigned v) -> int {
        if constexpr (LISTED) return (int)v;
        unsigned k;
        asm volatile("s_waitcnt vmcnt(0)\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(k) : "v"(v) : "memory");
        return k < 0x01000000u ? group + (int)k * groups : 0x7FFFFFFF;
    };
    const int o_bf16 = 2, o8 = 2 + 5 * __builtin_popcount(fmt_mask & 1u), o4 = 2 + 5 * __builtin_popcount(fmt_mask & 3u),
              o2 = 2 + 5 * __builtin_popcount(fmt_mask & 7u);
    const bool holes = !LISTED && (eval_mask & ~part_mask & MTQ_MASK_ALL) != (fmt_mask & MTQ_MASK_ALL);   // some slot (or part of one) is not written

    // Where a unit lives is worked out ONCE, when the unit is claimed: its first row segment (the DMA base) and its records' place.  (Round 3
    // divided by the launch's runtime tensor and row geometry in every issue_dma and again for the records: six 32-bit divisions per
    // unit, each a dozen vector instructions — 5 % of the kernel; the listed form did two per LANE and call.)
    struct Where { const unsigned char *base; double *out; unsigned long long mine; };
    // listed form: the global tile this lane's 16-lane group serves in unit u (a short last unit repeats the list's last tile)
    auto listed_tile = [&](int u, uint32_t q) -> uint32_t { return la.list[min((uint32_t)u * 4u + q, n_listed - 1u)]; };
    auto locate = [&](int u) -> Where {
        Where w{nullptr, nullptr, 0ull};
        if constexpr (LISTED) {
            // lane q < 4 works out where tile q of the unit starts; issue_rows lets every lane pick the start of the tile its chunk belongs to
            const uint32_t gt = listed_tile(u, (uint32_t)lane & 3u);
            const uint32_t b = gt / (uint32_t)tiles, tt = gt - b * (uint32_t)tiles;
            const uint32_t tr = tt / la.tiles_w32, tc = tt - tr * la.tiles_w32;
            w.mine = (unsigned long long)(uintptr_t)(x + (int64_t)b * stride + ((int64_t)tr * kTile) * ld + (int64_t)tc * kTile);
        } else {
            const int b = u / units_per_tensor, r = u - b * units_per_tensor, tr = r / units_w, uc = r - tr * units_w;
            w.base = reinterpret_cast<const unsigned char *>(x + (int64_t)b * stride + ((int64_t)tr * kTile) * ld + (int64_t)uc * kUnitCols);
            w.out = stats + ((int64_t)b * tiles + (int64_t)tr * tiles_w + uc * kUnitTiles) * rec;
        }
        return w;
    };
    // the four pieces of a unit's even (odd = 0) or odd (odd = 1) rows into the half at LDS address `half`
    auto issue_rows = [&](const Where &w, int odd, uint32_t half) {
        if constexpr (LISTED) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t tl = odd ? dma_tile[4 + i] : dma_tile[i], off = odd ? dma_off[4 + i] : dma_off[i];
                const unsigned lo = (unsigned)__shfl((int)(unsigned)w.mine, (int)tl, 64), hi = (unsigned)__shfl((int)(unsigned)(w.mine >> 32), (int)tl, 64);
                const unsigned long long src = (((unsigned long long)hi << 32) | lo) + off;
                glds16v(reinterpret_cast<const void *>((uintptr_t)src), half + i * 1024);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) glds16(w.base, odd ? dma_off[4 + i] : dma_off[i], half + i * 1024);
        }
    };
    // lane j of a tile's 16 lanes reduces one statistic and puts it where the record wants it (place_stat: everything by value — as a
    // by-reference closure the compiler kept it in scratch memory: 56 B per lane, a scratch and a flat load per call, 70 % more HBM writes)
    const PlaceArgs pa{eval_mask, o_bf16, o8, o4, o2};
    auto tree16 = [&](const double *row) -> double {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = row[k];
#pragma unroll
        for (int stp = 1; stp < 16; stp <<= 1)
#pragma unroll
            for (int k = 0; k < 16; k += 2 * stp) v[k] = v[k] + v[k + stp];
        return v[0];
    };

    // A wave retires after units_per_wave units (0: never — a fully persistent grid): the launch then consists of more blocks than
    // are resident at once and slots keep opening up, so that the one-wave-per-tensor scan kernels of earlier chunks
    // (csrc/mtq_scan.hip, other streams) are placed within a block's lifetime instead of waiting for the whole launch to drain.
    int left = units_per_wave > 0 ? units_per_wave : 0x7FFFFFFF;
    int u = claim_first();
    --left;
    uint32_t par = 0u;                                                        // the half the current unit's even rows are in
    Where here{nullptr, nullptr, 0ull}, next{nullptr, nullptr, 0ull};
    if (u < total_units) { here = locate(u); issue_rows(here, 0, in_addr); issue_rows(here, 1, in_addr + kHalfBytes); }
    while (u < total_units) {
        unsigned char *half_a = in + par * kHalfBytes, *half_b = in + (par ^ 1u) * kHalfBytes;
        double *scratch = reinterpret_cast<double *>(half_a);                  // overlays the even rows after groups 0 and 1 have read them
        double *recbuf = scratch + scratch_doubles;
        // everything but the wave's four youngest vector-memory operations is done: the even rows landed (and the previous records retired);
        // the four youngest are the odd rows' pieces — the last thing a unit issues
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        const bool more = left > 0;
        const unsigned claimed = more ? claim_issue() : 0u;                   // in flight behind this unit's arithmetic
        --left;

        double acc[kSums];
#pragma unroll
        for (int s = 0; s < kSums; ++s) acc[s] = 0.0;
        float mx[3] = {0.0f, 0.0f, 0.0f};
        bool bad = false;
#pragma nounroll
        for (int g = 0; g < 4; ++g) {                                         // rows 2j (g = 0,1) then 2j+1 (g = 2,3)
            if (g == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the odd rows landed (and this unit's claim returned)
            const uint32_t kl = 2u * (g & 1);
            const unsigned char *rowp = (g < 2 ? half_a : half_b) + rd_row;
            const uint4 lo = *reinterpret_cast<const uint4 *>(rowp + ((c0 ^ kl) << 4));
            const uint4 hi = *reinterpret_cast<const uint4 *>(rowp + ((c0 ^ (kl | 1u)) << 4));
            const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            GroupOut G;
#ifdef MTQ_K1_INTDOM
            fast_group<SUMS, ERRS, XS>(w, G, [&](uint32_t w2[8]) {
                const uint4 l2 = *reinterpret_cast<const uint4 *>(rowp + ((c0 ^ kl) << 4));
                const uint4 h2 = *reinterpret_cast<const uint4 *>(rowp + ((c0 ^ (kl | 1u)) << 4));
                w2[0] = l2.x; w2[1] = l2.y; w2[2] = l2.z; w2[3] = l2.w; w2[4] = h2.x; w2[5] = h2.y; w2[6] = h2.z; w2[7] = h2.w;
            });
#else
            fast_group_f32<SUMS, ERRS, XS>(w, G, [&](int i) -> uint32_t {   // word i of the group, from the LDS image
                return *reinterpret_cast<const uint32_t *>(rowp + ((c0 ^ (kl | (uint32_t)(i >> 2))) << 4) + 4 * (i & 3));
            });
#endif
#pragma unroll
            for (int s = 0; s < kSums; ++s)
                if (term_needed(SUMS, ERRS, XS, s)) acc[s] = acc[s] + G.term[s];   // 0.0 + t = t exactly: sequential ((g0+g1)+g2)+g3
            // (plain v_max_f32: fmaxf makes the compiler canonicalise both operands first — two more slow-class instructions per maximum;
            //  the values are maxima of finite magnitudes, or the tile is marked and they are never used)
#pragma unroll
            for (int f = 0; f < 3; ++f)
                if (ERRS & (1u << f)) asm("v_max_f32 %0, %0, %1" : "+v"(mx[f]) : "v"(G.mx[f]));
            bad |= G.bad;
        }

#pragma unroll
        for (int f = 0; f < 3; ++f) {
            if (!(ERRS & (1u << f))) continue;
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) { const float o = __shfl_xor(mx[f], sft, 16); asm("v_max_f32 %0, %0, %1" : "+v"(mx[f]) : "v"(o)); }
        }
        const unsigned long long bad_lanes = __ballot(bad);
        const bool tile_bad = ((bad_lanes >> (16 * t)) & 0xFFFFull) != 0ull;
        const int u_next = more ? claim_value(claimed) : 0x7FFFFFFF;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // every lane's image reads are done: both halves are free
        if (u_next < total_units) { next = locate(u_next); issue_rows(next, 0, in_addr + (par ^ 1u) * kHalfBytes); }   // EARLY: the next unit's even rows, into the half the odd rows were in

        // the reduce, `per` statistics at a time: lane j < per of a tile's 16 adds the 16 row pairs' partials of one statistic by the balanced tree
        auto reduce_pass = [&](int pass, auto &&sink) {
#pragma unroll
            for (int i = 0; i < per; ++i)
                if (pass * per + i < nsum) scratch[(t * per + i) * kScratchStride + j] = acc[term_at(SUMS, ERRS, XS, pass * per + i)];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if ((int)j < per && pass * per + (int)j < nsum) {
                const double r = tree16(scratch + (t * per + j) * kScratchStride);
                int sidx = 0;
#pragma unroll
                for (int i = 0; i < per; ++i) sidx = (int)j == i ? term_at(SUMS, ERRS, XS, min(pass * per + i, nsum - 1)) : sidx;
                sink(sidx, r);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the scratch is consumed (next pass / the records / the refill)
        };

        if constexpr (LISTED) {
            // every statistic straight to its place in the tile's own record; a tile of the list's padding (a short last unit) writes nothing
            const bool real = (uint32_t)u * 4u + t < n_listed;
            const uint32_t gt = listed_tile(u, t);
            double *rec_g = stats + (int64_t)gt * rec;
#pragma unroll
            for (int pass = 0; pass < npass; ++pass)
                reduce_pass(pass, [&](int sidx, double r) { if (real && !tile_bad) place_stat(pa, rec_g, sidx, r); });
            if (j == 15 && real && !tile_bad) {
                if (ERRS & 1u) rec_g[o8 + 4] = (double)mx[0];
                if (ERRS & 2u) rec_g[o4 + 4] = (double)mx[1];
                if (ERRS & 4u) rec_g[o2 + 4] = (double)mx[2];
            }
            if (j == 0 && real && tile_bad) la.redo[atomicAdd(la.n_redo, 1u)] = gt;   // the direct listed kernel takes it (literal route)
            if (u_next < total_units) issue_rows(next, 1, in_addr + par * kHalfBytes);   // LATE: the next unit's odd rows — the unit's last vector-memory operations
            u = u_next;
            here = next;
            par ^= 1u;
            continue;
        }

        double *rec_t = recbuf + t * rec;
        const int nrec = kUnitTiles * rec;
        if (holes) {   // slots of the layout this launch does not write: NaN, so that a reader of an unwritten statistic cannot go unnoticed
            const double poison = __longlong_as_double(0x7FF8000000000BADll);
            if (lane < nrec) recbuf[lane] = poison;
            if (lane + 64 < nrec) recbuf[lane + 64] = poison;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int pass = 0; pass < npass; ++pass) reduce_pass(pass, [&](int sidx, double r) { place_stat(pa, rec_t, sidx, r); });
        if (j == 15) {
            if ((ERRS & 1u) && (eval_mask & ~part_mask & 2u)) rec_t[o8 + 4] = (double)mx[0];
            if ((ERRS & 2u) && (eval_mask & ~part_mask & 4u)) rec_t[o4 + 4] = (double)mx[1];
            if ((ERRS & 4u) && (eval_mask & ~part_mask & 8u)) rec_t[o2 + 4] = (double)mx[2];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tile_bad && j == 0) {
            rec_t[0] = __longlong_as_double((long long)kRedoMagic);
            work[kWorkStamp] = launch_id;                                         // tells the follow-up kernel there is something to redo
            if (mark) *mark = launch_id;                                          // … and the caller's own word, where the fix-up is the caller's launch (mtq_tile_stats_partial_begin / _end)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

        double *out = here.out;
        double r0 = 0.0, r1 = 0.0;
        if (lane < nrec) r0 = recbuf[lane];
        if (lane + 64 < nrec) r1 = recbuf[lane + 64];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // records are in registers: this half may be refilled
        if (lane < nrec) out[lane] = r0;
        if (lane + 64 < nrec) out[lane + 64] = r1;
        if (u_next < total_units) issue_rows(next, 1, in_addr + par * kHalfBytes);   // LATE: the next unit's odd rows — the unit's last vector-memory operations
        u = u_next;
        here = next;
        par ^= 1u;
    }
    if constexpr (!LISTED) {
        // The launch resets its own unit counters: every wave of the grid ends here once, and the one that completes the count knows that
        // no other wave will touch the counters again (each wave's last claim returned before it counted itself).
        if (self_reset && lane == 0) {
            const unsigned waves = gridDim.x * (unsigned)kFastWaves;
            if (__hip_atomic_fetch_add(work + kWorkDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == waves - 1u) {
                for (int g = 0; g < kWorkGroups; ++g) __hip_atomic_store(work + g * kWorkStride, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(work + kWorkDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

} // namespace mtq

using namespace mtq;

static int fast_cus()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 0;
        cus = p.multiProcessorCount;
    }
    return cus;
}

// Launcher used by mtq_tile_stats_batched / mtq_tile_stats_partial when the input qualifies (mtq_kernels.hip decides).
// fmt_mask: record layout; eval_mask ⊂ fmt_mask: slots to write; part_mask ⊂ eval_mask: slots that only get Σy, Σy², Σxy
// (served for the combinations instantiated below; any other part_mask is widened to full slots, which is always allowed).
extern "C" int mtq_launch_tile_stats_bf16_fast(const void *x, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                               int64_t ld, uint32_t fmt_mask, uint32_t eval_mask, uint32_t part_mask, double *stats, void *stream,
                                               mtq::WorkSlot *work_out, unsigned launch_id, unsigned *mark, int self_reset)
{
    const int64_t th = rows / kTile, tw = cols / kTile, tiles = th * tw;
    const int64_t units_w = cols / kUnitCols, upt = th * units_w, total = count * upt;
    if (total > INT32_MAX / 2 || 64 * ld > (int64_t)UINT32_MAX) return fail(MTQ_ERR_INVALID, "tensor batch too large for one fast launch");
    const int rec = 2 + 5 * __builtin_popcount(fmt_mask & MTQ_MASK_ALL);
    const int cus = fast_cus();
    if (cus == 0) return fail(MTQ_ERR_HIP, "hipGetDeviceProperties failed");
    eval_mask &= fmt_mask & MTQ_MASK_ALL;
    part_mask &= eval_mask & 0xEu;
    const uint32_t bfp = (eval_mask >> 1) & 7u;
    uint32_t part = (part_mask >> 1) & 7u;
    if (!((bfp == 3u && part == 2u) || (bfp == 1u && part == 1u) || (bfp == 2u && part == 2u))) { part = 0u; part_mask = 0u; }
    const uint32_t sums = bfp, errs = bfp & ~part;
    const int64_t need = (total + kFastWaves - 1) / kFastWaves;
    static int wps_env = -1;                              // MTQ_K1_WAVES: waves per SIMD the grid is sized for (experiments)
    if (wps_env < 0) { const char *e = getenv("MTQ_K1_WAVES"); wps_env = e ? atoi(e) : 0; }
    const int wps = wps_env > 0 ? wps_env : rolled_waves(sums, errs);
    const int64_t max_blocks = (int64_t)cus * wps * 4 / kFastWaves; // resident blocks: that many waves on each of a CU's 4 SIMDs
    // MTQ_K1_UNITS_PER_WAVE (default 8; 0 = persistent waves): with a bound, the grid is what the units need at that many per wave,
    // rounded up to whole counter groups plus one spare block per group (a block that finds its group's queue empty exits at once)
    static int upw = -1;
    if (upw < 0) { const char *e = getenv("MTQ_K1_UNITS_PER_WAVE"); upw = e ? atoi(e) : 8; if (upw < 0) upw = 0; }
    int64_t want = need < max_blocks ? need : max_blocks;
    if (upw > 0 && need > max_blocks) {
        const int64_t by_quota = (total + (int64_t)kFastWaves * upw - 1) / ((int64_t)kFastWaves * upw);
        want = ((by_quota + kWorkGroups - 1) / kWorkGroups + 1) * kWorkGroups;
        if (want < max_blocks) want = max_blocks;
    }
    const unsigned blocks = (unsigned)want;
    const dim3 grid(blocks), block(kFastWaves * 64);
    static int lds_pad = -1;                              // MTQ_K1_LDS_PAD: extra LDS bytes per block (experiments on what fits beside K1)
    if (lds_pad < 0) { const char *e = getenv("MTQ_K1_LDS_PAD"); lds_pad = e ? atoi(e) : 0; }
    const size_t lds_bytes = kFastWaves * kRolledWaveLds + (size_t)lds_pad;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *xp = static_cast<const uint16_t *>(x);
    if (int rc = work_counter_acquire(stream, work_out)) return rc;   // `st` now waits for the slot's previous launch to have reset it
    unsigned *work = work_out->counters;
    const ListedArgs none{};
#define MTQ_LAUNCH_FAST(S, E) \
    hipLaunchKernelGGL((tile_stats_bf16_rolled<S, E, true, false>), grid, block, lds_bytes, st, xp, stride_elems, ld, (int)tw, tiles, (int)units_w, (int)upt, (int)total, fmt_mask, eval_mask, part_mask, rec, stats, work, launch_id, (need > max_blocks ? upw : 0), none, mark, self_reset)
    switch (sums | (errs << 4)) { // one instantiation per evaluated subset: what is not asked for costs nothing
    case 0x11: MTQ_LAUNCH_FAST(1u, 1u); break;
    case 0x22: MTQ_LAUNCH_FAST(2u, 2u); break;
    case 0x33: MTQ_LAUNCH_FAST(3u, 3u); break;
    case 0x44: MTQ_LAUNCH_FAST(4u, 4u); break;
    case 0x55: MTQ_LAUNCH_FAST(5u, 5u); break;
    case 0x66: MTQ_LAUNCH_FAST(6u, 6u); break;
    case 0x77: MTQ_LAUNCH_FAST(7u, 7u); break;
    case 0x01: MTQ_LAUNCH_FAST(1u, 0u); break;           // partial records: the last evaluated format without Σ|x−y|, max|x−y|
    case 0x02: MTQ_LAUNCH_FAST(2u, 0u); break;
    case 0x13: MTQ_LAUNCH_FAST(3u, 1u); break;
    default:                                             // no BFP format evaluated: the launcher's callers never ask for that (they abandon the slot on any error: once, there)
        return fail(MTQ_ERR_INVALID, "the bf16 fast kernel needs at least one BFP format to evaluate");
    }
#undef MTQ_LAUNCH_FAST
    return check_launch("mtq_tile_stats (bf16 fast)");
}

// The listed form (mtq_tile_stats_listed): → MTQ_OK when launched, 1 when this input / mask combination is not served here (the caller
// uses the direct listed kernel), < 0 on error.  redo / n_redo: device list the kernel appends the tiles to that the exact route cannot
// take (n_redo zeroed by the caller on the same stream).
extern "C" int mtq_launch_tile_stats_bf16_listed(const void *x, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                                 uint32_t fmt_mask, uint32_t full_mask, uint32_t err_mask, const uint32_t *list, const uint32_t *n_list,
                                                 uint32_t cap, uint32_t *redo, uint32_t *n_redo, double *stats, void *stream)
{
    const uint32_t sums = (full_mask >> 1) & 7u, errs = ((full_mask | err_mask) >> 1) & 7u;
    if (!((sums == 4u && errs == 6u) || (sums == 2u && errs == 3u))) return 1;
    if (rows % kTile != 0 || cols % kTile != 0 || 64 * ld > (int64_t)UINT32_MAX) return 1;
    const int64_t th = rows / kTile, tw = cols / kTile, tiles = th * tw;
    const int rec = 2 + 5 * __builtin_popcount(fmt_mask & MTQ_MASK_ALL);
    const int cus = fast_cus();
    if (cus == 0) return fail(MTQ_ERR_HIP, "hipGetDeviceProperties failed");
    const int64_t need = (((int64_t)cap + 3) / 4 + kFastWaves - 1) / kFastWaves;
    static int lw = -1;                                  // MTQ_LISTED_WAVES: waves per SIMD the listed grid is sized for
    if (lw < 0) { const char *e = getenv("MTQ_LISTED_WAVES"); lw = e ? atoi(e) : 0; }
    // 4 waves per SIMD (the listed instantiations take 112–116 registers): the launch waits on 64-byte row segments, not on the issue port — 0.280 against 0.314 ms at 3
    const int64_t max_blocks = (int64_t)cus * (lw > 0 ? lw : 4) * 4 / kFastWaves;
    const dim3 grid((unsigned)(need < max_blocks ? need : max_blocks)), block(kFastWaves * 64);
    const size_t lds_bytes = kFastWaves * kRolledWaveLds;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint16_t *xp = static_cast<const uint16_t *>(x);
    const ListedArgs la{list, n_list, cap, redo, n_redo, (uint32_t)tw};
    const uint32_t eval = (full_mask | err_mask) & fmt_mask;
#define MTQ_LAUNCH_LISTED_FAST(S, E) \
    hipLaunchKernelGGL((tile_stats_bf16_rolled<S, E, false, true>), grid, block, lds_bytes, st, xp, stride_elems, ld, (int)tw, tiles, 0, 0, 0, fmt_mask, eval, 0u, rec, stats, (unsigned *)nullptr, 0u, 0, la, (unsigned *)nullptr, 0)
    if (sums == 4u) MTQ_LAUNCH_LISTED_FAST(4u, 6u);
    else MTQ_LAUNCH_LISTED_FAST(2u, 3u);
#undef MTQ_LAUNCH_LISTED_FAST
    return check_launch("mtq_tile_stats_listed (bf16 fast)");
}
