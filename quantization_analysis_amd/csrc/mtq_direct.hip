// mtq_direct.hip — K1 for float32 STORAGE (gpt2, dequantised DeepSeek tensors) and for bf16 inputs the LDS-staged
// kernel of mtq_fast.hip cannot take (ragged shapes, unaligned rows): one wave64 per 32×32 tile, lane ℓ owns the
// shared-exponent group (row ℓ>>1, half ℓ&1) — the mapping of tile_stats_generic — with the group's arithmetic
// reduced to what float32 input needs, and the cross-lane reduction done through LDS instead of 22 × 6 shuffles.
//
// Same records as the literal route, bit for bit.  Per main-class element (within 14 binades of the group's shared
// exponent E; include/mtq.h's summation order) the literal route does, per BFP format: uint32 decode / align / round /
// re-encode (~25 ops), four float32 terms, four float64 accumulations and a float64 max.  Here:
//   * the aligned 24-bit mantissa is ONE multiply + truncating convert: a = trunc(|x|·2^(150−E)) (exact scaling);
//   * q = RNE of `a` to M bits with saturation in 5 integer ops; |y| = float(q)·2^(E−127−(M−1)) (exact, E ≥ 80);
//   * x and y share their sign, so x·y = |x|·|y| and |x−y| = ||x|−|y||: the float32 products / differences the
//     reference rounds are formed by the same float32 instructions on the magnitudes;
//   * Σy and Σy² of a BFP format are sums of ≤ 16 integers (|q| ≤ 127, q² ≤ 16129) times one power of two: exact
//     in float32, so they are accumulated with float32 add / fma and widened once per group;
//   * Σxy and Σ|x−y| keep their float64 accumulators and the element order of the literal route;
//   * tail-class elements are masked to +0 in the main pass (adding +0.0 changes no accumulator) and, when a lane
//     holds a non-zero one (rare), added afterwards in index order into separate tail sums: S = S_main + S_tail.
// A group whose E lies outside [80,180] (and is not all-zero) marks its tile with kRedoMagic; tile_stats_redo_flagged
// recomputes marked tiles by the literal route.  All-zero groups (ragged edges, padding) contribute nothing and do not mark.
//
// Cross-lane: every lane writes its ≤ 18 float64 group sums to a wave-private LDS table (odd stride: conflict-free);
// lanes (k, h) — statistic k, half h of the tile — add 8 row pairs in the documented order (4 groups of a row pair
// sequentially, balanced tree over row pairs), one lane exchange joins the halves.  Maxima stay in float32 lane permutes.
// Loads are direct (4 × 16 B per lane for float32); the next tile's group is fetched before the current one is processed.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mtq.h"
#include "mtq_device.hpp"
#include "mtq_error.hpp"

#ifndef MTQ_DIRECT_WAVES_PER_SIMD
#define MTQ_DIRECT_WAVES_PER_SIMD 3
#endif

namespace mtq {

constexpr int kDirectWaves = 2;                                   // waves per block (independent waves; 2 instead of 4 for the same reason as mtq_fast.hip's kFastWaves)
constexpr unsigned long long kRedoMagicDirect = 0x7FF8C0DE5EED0001ull; // same pattern as mtq_fast.hip / mtq_kernels.hip
constexpr int kMaxSums = 2 + 4 * kNumFmt;                         // Σx, Σx², 4 × (Σy, Σy², Σxy, Σ|d|)

__host__ __device__ constexpr int popc4(uint32_t m) { return (int)((m & 1u) + ((m >> 1) & 1u) + ((m >> 2) & 1u) + ((m >> 3) & 1u)); }
__host__ __device__ constexpr int direct_pad(uint32_t fm) { return (2 + 4 * popc4(fm)) | 1; }  // odd stride in doubles

__device__ __forceinline__ float u2f(uint32_t v) { return __uint_as_float(v); }
__device__ __forceinline__ uint32_t f2u(float v) { return __float_as_uint(v); }

typedef float f2 __attribute__((ext_vector_type(2)));

// m = max(m, |a|, |b|) in one instruction (inputs are never NaN on the exact route).
__device__ __forceinline__ void max3_abs(float &m, float a, float b) { asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(m) : "v"(a), "v"(b)); }

// One BFP format of two neighbouring main-class elements.  xs = the values truncated to the group's 24-bit window
// (±a·2^(E−150), 0 for masked elements), xm = the values themselves, C = 1.5·2^(E−103−M): adding and subtracting C
// rounds xs to the nearest multiple of the format's step 2^(E−126−M), ties to the even multiple — the reference's RNE on
// the truncated aligned mantissa (quantization_formats.py:133-140; C is an even multiple of the step and |xs| < C/3, so
// the sum stays inside one binade of C) — and the clamp is its saturating round-up (:141).  y carries x's sign.
__device__ __forceinline__ void bfp_pair(f2 xs, f2 xm, float C, float ymax, f2 &sy, f2 &sy2, double &sxy, double &sab, float &mx)
{
    const f2 r = (xs + C) - C;
    f2 ys;
    ys.x = __builtin_amdgcn_fmed3f(r.x, -ymax, ymax);
    ys.y = __builtin_amdgcn_fmed3f(r.y, -ymax, ymax);
    sy += ys;                                                    // exact: |Σ q| < 2^11 steps
    sy2 = __builtin_elementwise_fma(ys, ys, sy2);                // exact: Σ q² < 2^18 steps²
    const f2 p = xm * ys;                                        // float32 products (mixed_tile_greedy.py:161), ≥ 0
    sxy += (double)p.x;
    sxy += (double)p.y;
    const f2 d = xm - ys;                                        // :163
    sab += (double)fabsf(d.x);
    sab += (double)fabsf(d.y);
    max3_abs(mx, d.x, d.y);
}

// bf16 candidates of two float32 values: the hardware's RNE convert equals the integer form of
// quantization_formats.py:29-45 for every finite normal value (all a main-class element can be).
__device__ __forceinline__ f2 bf16_round_pair(f2 x)
{
    uint32_t pk;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(x.x), "v"(x.y));
    f2 y;
    y.x = u2f(pk << 16);
    y.y = u2f(pk & 0xFFFF0000u);
    return y;
}

// Group sums of one lane.  s[0..1] = Σx, Σx²; s[2+4j .. 5+4j] = Σy, Σy², Σxy, Σ|x−y| of the j-th requested format
// (ascending format code); mx[j] its max|x−y|.  `bad` = the exact route does not apply (the tile is redone).
template <uint32_t FM, bool kBf16Storage>
__device__ __forceinline__ void direct_group(const uint32_t (&u)[kGroup], double (&s)[kMaxSums], float (&mx)[kNumFmt], bool &bad)
{
    constexpr bool f0 = (FM & 1u) != 0, f8 = (FM & 2u) != 0, f4 = (FM & 4u) != 0, f2_ = (FM & 8u) != 0;
    constexpr int j0 = 0, j8 = popc4(FM & 1u), j4 = popc4(FM & 3u), j2 = popc4(FM & 7u);

    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) m = max(m, u[i] & 0x7FFFFFFFu);
    const uint32_t E = m >> 23;                                   // shared exponent (:118-119)
    const bool out_of_range = (E - 80u) > 100u;
    bad = out_of_range && m != 0u;
    const uint32_t Es = out_of_range ? 127u : E;                  // keeps the constants finite; results unused / all zero then
    const float k_align = u2f((277u - Es) << 23);                 // 2^(150−E): x → units of the window's last bit
    const float k_back = u2f((Es - 23u) << 23);                   // 2^(E−150)
    const float tail_thr = u2f((Es - 14u) << 23);                 // smallest main-class magnitude 2^(E−14−127)
    const float c8 = u2f(((Es + 17u) << 23) | 0x400000u), c4 = u2f(((Es + 21u) << 23) | 0x400000u), c2 = u2f(((Es + 23u) << 23) | 0x400000u);
    const float ymax8 = 127.0f * u2f((Es - 6u) << 23), ymax4 = 7.0f * u2f((Es - 2u) << 23), ymax2 = u2f(Es << 23);

    double sx = 0.0, sx2 = 0.0, y0 = 0.0, y02 = 0.0, xy0 = 0.0, ab0 = 0.0;
    double xy8 = 0.0, ab8 = 0.0, xy4 = 0.0, ab4 = 0.0, xy2 = 0.0, ab2 = 0.0;
    f2 sy8 = {0.0f, 0.0f}, sy82 = {0.0f, 0.0f}, sy4 = {0.0f, 0.0f}, sy42 = {0.0f, 0.0f}, sy2 = {0.0f, 0.0f}, sy22 = {0.0f, 0.0f};
    float m0 = 0.0f, m8 = 0.0f, m4 = 0.0f, m2 = 0.0f;
    uint32_t tail_or = 0u;
#pragma unroll
    for (int i = 0; i < kGroup; i += 2) {
        const uint32_t ua = fabsf(u2f(u[i])) < tail_thr ? 0u : u[i];          // tail class (zeros included) → +0 in the main pass
        const uint32_t ub = fabsf(u2f(u[i + 1])) < tail_thr ? 0u : u[i + 1];
        tail_or |= (u[i] ^ ua) | (u[i + 1] ^ ub);
        const f2 xm = {u2f(ua), u2f(ub)};
        sx += (double)xm.x;
        sx += (double)xm.y;
        const f2 xx = xm * xm;
        sx2 += (double)xx.x;
        sx2 += (double)xx.y;
        if (f0 && !kBf16Storage) {                                             // bf16 candidate of a float32 value (:29-45)
            const f2 yv = bf16_round_pair(xm);
            y0 += (double)yv.x;
            y0 += (double)yv.y;
            const f2 yy = yv * yv, xy = xm * yv, dd = xm - yv;
            y02 += (double)yy.x;
            y02 += (double)yy.y;
            xy0 += (double)xy.x;
            xy0 += (double)xy.y;
            ab0 += (double)fabsf(dd.x);
            ab0 += (double)fabsf(dd.y);
            max3_abs(m0, dd.x, dd.y);
        }
        if (f8 || f4 || f2_) {
            const f2 t = xm * k_align;                                         // exact scaling; integer part = aligned mantissa man >> d (:121-131)
            f2 at;
            at.x = __builtin_truncf(t.x);
            at.y = __builtin_truncf(t.y);
            const f2 xs = at * k_back;                                         // x truncated to the group's 24-bit window
            if (f8) bfp_pair(xs, xm, c8, ymax8, sy8, sy82, xy8, ab8, m8);
            if (f4) bfp_pair(xs, xm, c4, ymax4, sy4, sy42, xy4, ab4, m4);
            if (f2_) bfp_pair(xs, xm, c2, ymax2, sy2, sy22, xy2, ab2, m2);
        }
    }
    if (kBf16Storage) { y0 = sx; y02 = sx2; xy0 = sx2; }                       // y == x: the same float32 terms in the same order

    if ((tail_or << 1) != 0u) {                                                // a non-zero tail element in this lane (divergent, rare)
        double tx = 0.0, tx2 = 0.0, ty0 = 0.0, ty02 = 0.0, txy0 = 0.0, tab0 = 0.0, tab = 0.0;
        float tmx = 0.0f;
#pragma unroll
        for (int i = 0; i < kGroup; ++i) {
            const float xv = u2f(u[i]);
            if (fabsf(xv) < tail_thr) {                                        // zeros add +0.0 everywhere: harmless
                tx += (double)xv;
                const float xx = xv * xv;
                tx2 += (double)xx;
                const float ax = fabsf(xv);
                tab += (double)ax;                                             // every BFP format gives y = +0 this far below the maximum
                tmx = fmaxf(tmx, ax);
                if (f0 && !kBf16Storage) {
                    const float yv = u2f(bf16_round_bits(u[i]));
                    ty0 += (double)yv;
                    ty02 += (double)(yv * yv);
                    txy0 += (double)(xv * yv);
                    const float df = fabsf(xv - yv);
                    tab0 += (double)df;
                    m0 = fmaxf(m0, df);
                }
            }
        }
        sx = sx + tx; sx2 = sx2 + tx2;
        if (kBf16Storage) { y0 = sx; y02 = sx2; xy0 = sx2; }
        else { y0 = y0 + ty0; y02 = y02 + ty02; xy0 = xy0 + txy0; ab0 = ab0 + tab0; }
        ab8 = ab8 + tab; ab4 = ab4 + tab; ab2 = ab2 + tab;
        m8 = fmaxf(m8, tmx); m4 = fmaxf(m4, tmx); m2 = fmaxf(m2, tmx);
    }

    s[0] = sx; s[1] = sx2;
    if (f0) { s[2 + 4 * j0] = y0; s[3 + 4 * j0] = y02; s[4 + 4 * j0] = xy0; s[5 + 4 * j0] = ab0; mx[j0] = m0; }
    if (f8) { s[2 + 4 * j8] = (double)(sy8.x + sy8.y); s[3 + 4 * j8] = (double)(sy82.x + sy82.y); s[4 + 4 * j8] = xy8; s[5 + 4 * j8] = ab8; mx[j8] = m8; }
    if (f4) { s[2 + 4 * j4] = (double)(sy4.x + sy4.y); s[3 + 4 * j4] = (double)(sy42.x + sy42.y); s[4 + 4 * j4] = xy4; s[5 + 4 * j4] = ab4; mx[j4] = m4; }
    if (f2_) { s[2 + 4 * j2] = (double)(sy2.x + sy2.y); s[3 + 4 * j2] = (double)(sy22.x + sy22.y); s[4 + 4 * j2] = xy2; s[5 + 4 * j2] = ab2; mx[j2] = m2; }
}

template <typename T, uint32_t FM>
__global__ __launch_bounds__(kDirectWaves * 64, MTQ_DIRECT_WAVES_PER_SIMD) void tile_stats_direct(
    const T *__restrict__ x, int64_t stride, int64_t rows, int64_t cols, int64_t ld, uint32_t tiles_w, uint32_t tiles,
    uint32_t total_tiles, double *__restrict__ stats, int vec_ok, unsigned *__restrict__ work, unsigned launch_id, int tiles_per_wave,
    const RaggedTable tb)
{
    constexpr int nf = popc4(FM), nsum = 2 + 4 * nf, pad = direct_pad(FM), rec = 2 + 5 * nf;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *part = reinterpret_cast<double *>(lds) + wave * (64 * pad);          // [64 groups][pad]
    const uint32_t groups = min(gridDim.x, (unsigned)kWorkGroups), group = blockIdx.x % groups;
    unsigned *queue = work + group * kWorkStride;                        // tiles come from per-group device counters (see mtq_fast.hip)
    auto claim = [&]() -> uint32_t {
        unsigned v = 0u;
        if (lane == 0) v = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned k = (unsigned)__builtin_amdgcn_readfirstlane(v);
        return k < 0x02000000u ? group + k * groups : 0xFFFFFFFFu;
    };

    auto fetch = [&](uint32_t gt, uint32_t (&u)[kGroup]) {
        if (tb.n) {   // a ragged batch (mtq_tile_stats_ragged): the tile's matrix from the table in the kernel arguments
            const TileSite w = ragged_site(tb, gt);
            const uint32_t tr = w.t / w.tiles_w, tc = w.t - tr * w.tiles_w;
            Loader<T>::group(static_cast<const T *>(w.x), (int64_t)tr * kTile + (lane >> 1), (int64_t)tc * kTile + (lane & 1) * kGroup, w.rows,
                             w.cols, w.ld, w.vec_ok != 0, u);
            return;
        }
        const uint32_t b = gt / tiles, t = gt - b * tiles;
        const uint32_t tr = t / tiles_w, tc = t - tr * tiles_w;
        Loader<T>::group(x + (int64_t)b * stride, (int64_t)tr * kTile + (lane >> 1), (int64_t)tc * kTile + (lane & 1) * kGroup, rows,
                         cols, ld, vec_ok != 0, u);
    };

    // a wave retires after tiles_per_wave tiles (0: never), so that co-resident kernels of other streams — the device scans of
    // earlier batches — are placed within a block's lifetime (mtq_fast.hip, MTQ_K1_UNITS_PER_WAVE)
    int left = tiles_per_wave > 0 ? tiles_per_wave : 0x7FFFFFFF;
    uint32_t gt = claim();
    --left;
    uint32_t nxt[kGroup];
    if (gt < total_tiles) fetch(gt, nxt);
    while (gt < total_tiles) {
        uint32_t u[kGroup];
#pragma unroll
        for (int i = 0; i < kGroup; ++i) u[i] = nxt[i];
        const uint32_t gt_next = left > 0 ? claim() : 0xFFFFFFFFu;
        --left;
        if (gt_next < total_tiles) fetch(gt_next, nxt);                         // in flight while this tile is processed

        double s[kMaxSums];
        float mx[kNumFmt];
        bool bad;
        direct_group<FM, sizeof(T) == 2>(u, s, mx, bad);

#pragma unroll
        for (int k = 0; k < nsum; ++k) part[lane * pad + k] = s[k];
#pragma unroll
        for (int j = 0; j < nf; ++j) {
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], sft, 64));
        }
        const bool tile_bad = __ballot(bad) != 0ull;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // every lane's group sums are in the table

        const int k = lane & 31, h = lane >> 5;                                  // statistic k over row pairs 8h .. 8h+7
        double r = 0.0;
        if (k < nsum) {
            const double *col = part + (32 * h) * pad + k;
            double q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)                                           // the 4 groups of a row pair, sequentially
                q[j] = ((col[(4 * j) * pad] + col[(4 * j + 1) * pad]) + col[(4 * j + 2) * pad]) + col[(4 * j + 3) * pad];
            r = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])); // balanced tree over the row pairs
        }
        r = r + __shfl_xor(r, 32, 64);                                            // the two halves of the tile (top tree level)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // table consumed: the next tile may overwrite it

        double *out = stats + (int64_t)gt * rec;
        if (lane < nsum) {
            const int slot = lane < 2 ? lane : 2 + 5 * ((lane - 2) >> 2) + ((lane - 2) & 3);
            if (tile_bad && lane == 0) {
                r = __longlong_as_double((long long)kRedoMagicDirect);
                work[kWorkStamp] = launch_id;                                     // tells the follow-up kernel there is something to redo
            }
            out[slot] = r;
        } else if (lane >= 32 && lane < 32 + nf) {
            float v = mx[0];
#pragma unroll
            for (int j = 1; j < nf; ++j) v = (lane - 32 == j) ? mx[j] : v;
            out[2 + 5 * (lane - 32) + 4] = (double)v;
        }
        gt = gt_next;
    }
}

// K1 for a LIST of tiles (mtq_tile_stats_listed): the same per-tile arithmetic and summation order, the tile taken from a device list
// (tensor * tiles + tile) whose length is a device counter — the candidates the split greedy search hands over before its last pass
// (csrc/mtq_scan.hip, phase 1).  Of every evaluated format j (ascending code) only the statistics named in its byte of `wmask` (bit k =
// statistic k of Σy, Σy², Σxy, Σ|x−y|, max|x−y|) are written, at the format's slot offset in the record layout (its byte of `obase`);
// Σx, Σx² and every other slot of the record are left as they are.  A tile the exact route cannot take (a group outside its exponent
// range) is evaluated on the spot by the literal route: no follow-up kernel.
// W = waves per block: kDirectWaves for a list that fills the chip; 1 — compiled for 5 waves per SIMD, i.e. 96 registers, the literal
// branch spilling — for the few-or-none tiles the LDS-staged listed kernel hands back: that launch sits in the search chain of every
// batch, and a 152-register wave is only placed where K1's blocks leave that much of a SIMD (0.26 ms per step in the round-4 trace).
template <typename T, uint32_t FM, int W = kDirectWaves>
__global__ __launch_bounds__(W * 64, W == 1 ? 5 : MTQ_DIRECT_WAVES_PER_SIMD) void tile_stats_listed(
    const T *__restrict__ x, int64_t stride, int64_t rows, int64_t cols, int64_t ld, uint32_t tiles_w, uint32_t tiles,
    const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list, uint32_t cap, double *__restrict__ stats, int vec_ok, int rec,
    uint32_t obase, uint32_t wmask)
{
    constexpr int nf = popc4(FM), nsum = 2 + 4 * nf, pad = direct_pad(FM);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *part = reinterpret_cast<double *>(lds) + wave * (64 * pad);          // [64 groups][pad]
    const uint32_t n = min(*n_list, cap);
    const uint32_t step = gridDim.x * W;

    auto fetch = [&](uint32_t gt, uint32_t (&u)[kGroup]) {
        const uint32_t b = gt / tiles, t = gt - b * tiles;
        const uint32_t tr = t / tiles_w, tc = t - tr * tiles_w;
        Loader<T>::group(x + (int64_t)b * stride, (int64_t)tr * kTile + (lane >> 1), (int64_t)tc * kTile + (lane & 1) * kGroup, rows,
                         cols, ld, vec_ok != 0, u);
    };

    uint32_t k = blockIdx.x * W + wave;
    uint32_t gt = k < n ? list[k] : 0u;
    uint32_t nxt[kGroup];
    if (k < n) fetch(gt, nxt);
    while (k < n) {
        uint32_t u[kGroup];
#pragma unroll
        for (int i = 0; i < kGroup; ++i) u[i] = nxt[i];
        const uint32_t k_next = k + step;
        const uint32_t gt_next = k_next < n ? list[k_next] : 0u;
        if (k_next < n) fetch(gt_next, nxt);                                     // in flight while this tile is processed

        double s[kMaxSums];
        float mx[kNumFmt];
        bool bad;
        direct_group<FM, sizeof(T) == 2>(u, s, mx, bad);
        double *out = stats + (int64_t)gt * rec;
        if (__ballot(bad) != 0ull) {                                             // wave-uniform, rare: the literal route, here and now
            double acc[2 + 5 * kNumFmt];
            tile_terms_literal(u, FM, acc);
            if (lane == 0) {
                int j = 0;
#pragma unroll
                for (int f = 0; f < kNumFmt; ++f) {
                    if (!(FM & (1u << f))) continue;
                    const uint32_t o = (obase >> (8 * j)) & 0xFFu, w = (wmask >> (8 * j)) & 0xFFu;
#pragma unroll
                    for (int q = 0; q < 5; ++q) if (w & (1u << q)) out[o + q] = acc[2 + 5 * f + q];
                    ++j;
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < nsum; ++q) part[lane * pad + q] = s[q];
#pragma unroll
            for (int j = 0; j < nf; ++j) {
#pragma unroll
                for (int sft = 1; sft < 64; sft <<= 1) mx[j] = fmaxf(mx[j], __shfl_xor(mx[j], sft, 64));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // every lane's group sums are in the table
            const int kk = lane & 31, h = lane >> 5;                              // statistic kk over row pairs 8h .. 8h+7
            double r = 0.0;
            if (kk < nsum) {
                const double *col = part + (32 * h) * pad + kk;
                double q[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)                                       // the 4 groups of a row pair, sequentially
                    q[j] = ((col[(4 * j) * pad] + col[(4 * j + 1) * pad]) + col[(4 * j + 2) * pad]) + col[(4 * j + 3) * pad];
                r = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7])); // balanced tree over the row pairs
            }
            r = r + __shfl_xor(r, 32, 64);                                        // the two halves of the tile (top tree level)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // table consumed: the next tile may overwrite it
            if (lane >= 2 && lane < nsum) {
                const int j = (lane - 2) >> 2, q = (lane - 2) & 3;
                const uint32_t o = (obase >> (8 * j)) & 0xFFu, w = (wmask >> (8 * j)) & 0xFFu;
                if (w & (1u << q)) out[o + q] = r;
            } else if (lane >= 32 && lane < 32 + nf) {
                const int j = lane - 32;
                float v = mx[0];
#pragma unroll
                for (int jj = 1; jj < nf; ++jj) v = (j == jj) ? mx[jj] : v;
                const uint32_t o = (obase >> (8 * j)) & 0xFFu, w = (wmask >> (8 * j)) & 0xFFu;
                if (w & 16u) out[o + 4] = (double)v;
            }
        }
        k = k_next;
        gt = gt_next;
    }
}

template <typename T, int W = kDirectWaves>
static void launch_listed(uint32_t fm, dim3 grid, hipStream_t st, const T *x, int64_t stride, int64_t rows, int64_t cols, int64_t ld,
                          uint32_t tiles_w, uint32_t tiles, const uint32_t *list, const uint32_t *n_list, uint32_t cap, double *stats, int vec_ok,
                          int rec, uint32_t obase, uint32_t wmask)
{
    const dim3 block(W * 64);
#define MTQ_LAUNCH_LISTED(M) \
    case M: hipLaunchKernelGGL((tile_stats_listed<T, M, W>), grid, block, (size_t)W * 64 * direct_pad(M) * sizeof(double), st, x, \
                               stride, rows, cols, ld, tiles_w, tiles, list, n_list, cap, stats, vec_ok, rec, obase, wmask); break;
    switch (fm) { // BFP subsets only: the bf16 slot has no use for a late evaluation
        MTQ_LAUNCH_LISTED(2u) MTQ_LAUNCH_LISTED(4u) MTQ_LAUNCH_LISTED(6u) MTQ_LAUNCH_LISTED(8u) MTQ_LAUNCH_LISTED(10u)
        MTQ_LAUNCH_LISTED(12u) MTQ_LAUNCH_LISTED(14u)
    default: break;
    }
#undef MTQ_LAUNCH_LISTED
}

template <typename T>
static void launch_direct(uint32_t fm, dim3 grid, hipStream_t st, const T *x, int64_t stride, int64_t rows, int64_t cols, int64_t ld,
                          uint32_t tiles_w, uint32_t tiles, uint32_t total, double *stats, int vec_ok, unsigned *work, unsigned launch_id, int tiles_per_wave,
                          const RaggedTable &tb)
{
    const dim3 block(kDirectWaves * 64);
#define MTQ_LAUNCH_DIRECT(M) \
    case M: hipLaunchKernelGGL((tile_stats_direct<T, M>), grid, block, (size_t)kDirectWaves * 64 * direct_pad(M) * sizeof(double), st, x, \
                               stride, rows, cols, ld, tiles_w, tiles, total, stats, vec_ok, work, launch_id, tiles_per_wave, tb); break;
    switch (fm) { // one instantiation per requested format subset: unrequested formats cost nothing
        MTQ_LAUNCH_DIRECT(1u) MTQ_LAUNCH_DIRECT(2u) MTQ_LAUNCH_DIRECT(3u) MTQ_LAUNCH_DIRECT(4u) MTQ_LAUNCH_DIRECT(5u)
        MTQ_LAUNCH_DIRECT(6u) MTQ_LAUNCH_DIRECT(7u) MTQ_LAUNCH_DIRECT(8u) MTQ_LAUNCH_DIRECT(9u) MTQ_LAUNCH_DIRECT(10u)
        MTQ_LAUNCH_DIRECT(11u) MTQ_LAUNCH_DIRECT(12u) MTQ_LAUNCH_DIRECT(13u) MTQ_LAUNCH_DIRECT(14u) MTQ_LAUNCH_DIRECT(15u)
    default: break;
    }
#undef MTQ_LAUNCH_DIRECT
}

} // namespace mtq

using namespace mtq;

// Launcher used by mtq_tile_stats_batched for every input the bf16 LDS-staged kernel does not take (mtq_kernels.hip
// decides and follows up with tile_stats_redo_flagged).  fmt_mask != 0, count * tiles < 2^31.
// Grid of a direct launch over `total` tiles and the per-wave tile quota that goes with it (0: waves never retire).
static int direct_grid(int64_t total, dim3 &grid, int &quota)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail(MTQ_ERR_HIP, "hipGetDeviceProperties failed");
        cus = p.multiProcessorCount;
    }
    const int64_t need = (total + kDirectWaves - 1) / kDirectWaves;
    const int64_t max_blocks = (int64_t)cus * MTQ_DIRECT_WAVES_PER_SIMD * 4 / kDirectWaves;   // resident blocks: that many waves on each of a CU's 4 SIMDs
    // MTQ_K1_UNITS_PER_WAVE (mtq_fast.hip; default 8) x 16 tiles: a wave of this kernel restarts more expensively (its prefetch chain),
    // measured on 8 x 4096² float32: 32 tiles per wave cost 10 % of the launch, 128 tiles 1 %
    static int tpw = -1;
    if (tpw < 0) { const char *e = getenv("MTQ_K1_UNITS_PER_WAVE"); tpw = 16 * (e ? atoi(e) : 8); if (tpw < 0) tpw = 0; }
    int64_t want = need < max_blocks ? need : max_blocks;
    quota = 0;
    if (tpw > 0 && need > max_blocks) {
        const int64_t by_quota = (total + (int64_t)kDirectWaves * tpw - 1) / ((int64_t)kDirectWaves * tpw);
        want = ((by_quota + kWorkGroups - 1) / kWorkGroups + 1) * kWorkGroups;
        if (want < max_blocks) want = max_blocks;
        quota = tpw;
    }
    grid = dim3((unsigned)want);
    return MTQ_OK;
}

// Launcher used by mtq_tile_stats_batched for every input the bf16 LDS-staged kernel does not take (mtq_kernels.hip
// decides and follows up with tile_stats_redo_flagged).  fmt_mask != 0, count * tiles < 2^31.
extern "C" int mtq_launch_tile_stats_direct(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                            int64_t ld, uint32_t fmt_mask, double *stats, int vec_ok, void *stream, mtq::WorkSlot *work_out, unsigned launch_id)
{
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw, total = count * tiles;
    if (total >= ((int64_t)1 << 31) || (fmt_mask & MTQ_MASK_ALL) == 0) return fail(MTQ_ERR_INVALID, "direct tile_stats launch out of range");
    dim3 grid;
    int quota = 0;
    if (int rc = direct_grid(total, grid, quota)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = work_counter_acquire(stream, work_out)) return rc;   // `st` now waits for the slot's previous launch to have reset it
    unsigned *work = work_out->counters;
    static const RaggedTable none{};   // n = 0: a uniform batch
    if (in_dtype == MTQ_DTYPE_BF16)
        launch_direct<uint16_t>(fmt_mask & MTQ_MASK_ALL, grid, st, static_cast<const uint16_t *>(x), stride_elems, rows, cols, ld, (uint32_t)tw,
                                (uint32_t)tiles, (uint32_t)total, stats, vec_ok, work, launch_id, quota, none);
    else
        launch_direct<float>(fmt_mask & MTQ_MASK_ALL, grid, st, static_cast<const float *>(x), stride_elems, rows, cols, ld, (uint32_t)tw,
                             (uint32_t)tiles, (uint32_t)total, stats, vec_ok, work, launch_id, quota, none);
    return check_launch("mtq_tile_stats (direct)");
}

// The same kernel over a ragged batch (mtq_tile_stats_ragged, mtq_kernels.hip): tb.total tiles numbered through the table's matrices.
int mtq_launch_tile_stats_direct_ragged(const mtq::RaggedTable &tb, int in_dtype, uint32_t fmt_mask, double *stats, void *stream,
                                        mtq::WorkSlot *work_out, unsigned launch_id)
{
    if (tb.n == 0 || tb.n > (uint32_t)kRaggedMax || tb.total == 0 || tb.total >= (1u << 31) || (fmt_mask & MTQ_MASK_ALL) == 0)
        return fail(MTQ_ERR_INVALID, "ragged tile_stats launch out of range");
    dim3 grid;
    int quota = 0;
    if (int rc = direct_grid(tb.total, grid, quota)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (int rc = work_counter_acquire(stream, work_out)) return rc;
    unsigned *work = work_out->counters;
    if (in_dtype == MTQ_DTYPE_BF16)
        launch_direct<uint16_t>(fmt_mask & MTQ_MASK_ALL, grid, st, nullptr, 0, 0, 0, 0, 1u, 1u, tb.total, stats, 0, work, launch_id, quota, tb);
    else
        launch_direct<float>(fmt_mask & MTQ_MASK_ALL, grid, st, nullptr, 0, 0, 0, 0, 1u, 1u, tb.total, stats, 0, work, launch_id, quota, tb);
    return check_launch("mtq_tile_stats_ragged (direct)");
}

extern "C" int mtq_launch_tile_stats_bf16_listed(const void *x, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                                 uint32_t fmt_mask, uint32_t full_mask, uint32_t err_mask, const uint32_t *list, const uint32_t *n_list,
                                                 uint32_t cap, uint32_t *redo, uint32_t *n_redo, double *stats, void *stream);

// mtq_tile_stats_listed (include/mtq.h): argument checks here.  bf16 storage with 16-byte aligned rows and the mask pairs the streamed
// search produces go through the exact-integer kernel of mtq_fast.hip (four listed tiles per wave unit), which hands the tiles it cannot
// take to the kernel above through `scratch`; everything else goes through the kernel above directly.
extern "C" int mtq_tile_stats_listed(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                     uint32_t layout_mask, uint32_t full_mask, uint32_t err_mask, const uint32_t *listed, const uint32_t *n_listed,
                                     int64_t capacity, uint32_t *scratch, double *stats, void *stream)
{
    if (!x || !listed || !n_listed || !stats) return fail(MTQ_ERR_INVALID, "null argument");
    if (in_dtype != MTQ_DTYPE_BF16 && in_dtype != MTQ_DTYPE_F32) return fail(MTQ_ERR_INVALID, "unknown input dtype");
    if (count <= 0 || rows <= 0 || cols <= 0 || ld < cols || capacity <= 0) return fail(MTQ_ERR_INVALID, "shapes must be positive and ld >= cols");
    if ((layout_mask & ~MTQ_MASK_ALL) != 0 || ((full_mask | err_mask) & ~layout_mask) != 0 || (full_mask & err_mask) != 0)
        return fail(MTQ_ERR_INVALID, "full_mask and err_mask must be disjoint subsets of layout_mask");
    const uint32_t fm = full_mask | err_mask;
    if (fm == 0u || (fm & 1u)) return fail(MTQ_ERR_INVALID, "the listed evaluation takes BFP formats (and at least one)");
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw;
    if (count * tiles >= ((int64_t)1 << 32) || tw > INT32_MAX) return fail(MTQ_ERR_INVALID, "too many tiles for a listed launch");
    if (int rc = require_device()) return rc;
    const int rec = 2 + 5 * popc4(layout_mask);
    uint32_t obase = 0u, wmask = 0u;
    int j = 0;
    for (int f = 1; f < kNumFmt; ++f) {
        if (!(fm & (1u << f))) continue;
        obase |= (uint32_t)(2 + 5 * popc4(layout_mask & ((1u << f) - 1u))) << (8 * j);
        wmask |= ((full_mask & (1u << f)) ? 0x1Fu : 0x18u) << (8 * j);   // all five, or Σ|x−y| and max|x−y|
        ++j;
    }
    const int64_t esz = in_dtype == MTQ_DTYPE_BF16 ? 2 : 4;
    const int vec_ok = (reinterpret_cast<uintptr_t>(x) & 15u) == 0 && (ld * esz) % 16 == 0 && (stride_elems * esz) % 16 == 0;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return fail(MTQ_ERR_HIP, "hipGetDeviceProperties failed");
        cus = p.multiProcessorCount;
    }
    // the list's length is only known on the device: a grid that fills the chip once, every wave striding through the list
    const int64_t need = (capacity + kDirectWaves - 1) / kDirectWaves;
    const int64_t max_blocks = (int64_t)cus * MTQ_DIRECT_WAVES_PER_SIMD * 4 / kDirectWaves;
    dim3 grid((unsigned)(need < max_blocks ? need : max_blocks));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const uint32_t cap = (uint32_t)(capacity < (int64_t)UINT32_MAX ? capacity : (int64_t)UINT32_MAX);
    bool handed_back = false;
    if (in_dtype == MTQ_DTYPE_BF16 && vec_ok && scratch && getenv("MTQ_LISTED_DIRECT") == nullptr) {
        // scratch[0]: the redo list's length, scratch[1..]: the list
        if (hipMemsetAsync(scratch, 0, sizeof(uint32_t), st) != hipSuccess) return fail(MTQ_ERR_HIP, "hipMemsetAsync failed");
        const int rc = mtq_launch_tile_stats_bf16_listed(x, count, stride_elems, rows, cols, ld, layout_mask, full_mask, err_mask, listed, n_listed, cap,
                                                         scratch + 1, scratch, stats, stream);
        if (rc < 0) return rc;
        if (rc == 0) {   // what is left for the kernel above: the tiles the exact route handed back — few or none: a small grid strides through them
            listed = scratch + 1;
            n_listed = scratch;
            grid = dim3(grid.x < 128u ? grid.x : 128u);
            handed_back = true;
        }
    }
    if (handed_back)
        launch_listed<uint16_t, 1>(fm, grid, st, static_cast<const uint16_t *>(x), stride_elems, rows, cols, ld, (uint32_t)tw, (uint32_t)tiles, listed,
                                   n_listed, cap, stats, vec_ok, rec, obase, wmask);
    else if (in_dtype == MTQ_DTYPE_BF16)
        launch_listed<uint16_t>(fm, grid, st, static_cast<const uint16_t *>(x), stride_elems, rows, cols, ld, (uint32_t)tw, (uint32_t)tiles, listed,
                                n_listed, cap, stats, vec_ok, rec, obase, wmask);
    else
        launch_listed<float>(fm, grid, st, static_cast<const float *>(x), stride_elems, rows, cols, ld, (uint32_t)tw, (uint32_t)tiles, listed, n_listed,
                             cap, stats, vec_ok, rec, obase, wmask);
    return check_launch("mtq_tile_stats_listed");
}
