// mtq_device.hpp — gfx950 device helpers shared by the kernels: the literal uint32 BFP / bf16
// quantize→dequantize of one value inside a shared-exponent group (reference:
// quantization_formats.py:29-45, 71-81, 115-158).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mtq {

constexpr int kTile = 32;    // tile edge (compression_algorithms/tile_utils.py:32)
constexpr int kGroup = 16;   // shared-exponent group length (quantization_formats.py:115)
constexpr int kNumFmt = 4;   // bf16, bfp8, bfp4, bfp2 (tile_utils.py:8)

__device__ __forceinline__ constexpr int mant_bits(int fmt) { return fmt == 1 ? 7 : (fmt == 2 ? 3 : 1); }

// quantization_formats.py:29-45: RNE to bf16 on the raw word (uint32 wrap, no NaN special case), widened back.
__device__ __forceinline__ uint32_t bf16_round_bits(uint32_t u)
{
    const uint32_t r = u + (0x7FFFu + ((u >> 16) & 1u));
    return r & 0xFFFF0000u;
}

// One element of quantize_dequantize_bfp_ttnn (quantization_formats.py:118-158) given the group's
// shared (max) biased exponent.  M = mantissa bits kept (7 / 3 / 1).
template <int M>
__device__ __forceinline__ uint32_t bfp_elem_bits(uint32_t u, uint32_t shared)
{
    constexpr uint32_t shift = 24u - M;
    constexpr uint32_t round_mask = (1u << shift) - 1u;
    constexpr uint32_t tie = 1u << (shift - 1u);
    constexpr uint32_t qmax = (1u << M) - 1u;
    const uint32_t e = (u >> 23) & 0xFFu;
    const uint32_t d = shared - e;                       // :126
    uint32_t man = (1u << 23) | (u & 0x007FFFFFu);      // :121,125
    man = d > 31u ? 0u : (man >> d);                     // :127-131
    const uint32_t rv = man & round_mask;                // :136
    man >>= shift;                                       // :137
    const uint32_t up = (rv > tie) | ((rv == tie) & (man & 1u)); // :138-139
    man = min(man + up, qmax);                           // :140-141 (saturating round-up)
    man = e == 0u ? 0u : man;                            // :145 zero / denormal input → code 0
    if (man == 0u) return 0u;                            // :143,155: sign cleared, exp_out = 0
    const uint32_t msb = 31u - (uint32_t)__clz((int)man); // :77
    const uint32_t sc = (uint32_t)(M - 1) - msb;         // :78
    const uint32_t ms = (man << (sc + 1u)) & qmax;       // :80
    const uint32_t exp_out = shared - sc;                // :154 (wraps below sc, kept)
    return (u & 0x80000000u) | (exp_out << 23) | (ms << (23u - M)); // :158
}

// Same arithmetic with the mantissa width as a run-time value (1..7): lets lanes holding different BFP formats
// share one instruction stream (K3, where neighbouring tiles of a wave use different formats).
__device__ __forceinline__ uint32_t bfp_elem_bits_rt(uint32_t u, uint32_t shared, uint32_t M)
{
    const uint32_t shift = 24u - M, round_mask = (1u << shift) - 1u, tie = 1u << (shift - 1u), qmax = (1u << M) - 1u;
    const uint32_t e = (u >> 23) & 0xFFu;
    const uint32_t d = shared - e;
    uint32_t man = (1u << 23) | (u & 0x007FFFFFu);
    man = d > 31u ? 0u : (man >> d);
    const uint32_t rv = man & round_mask;
    man >>= shift;
    const uint32_t up = (rv > tie) | ((rv == tie) & (man & 1u));
    man = min(man + up, qmax);
    man = e == 0u ? 0u : man;
    if (man == 0u) return 0u;
    const uint32_t msb = 31u - (uint32_t)__clz((int)man);
    const uint32_t sc = (M - 1u) - msb;
    const uint32_t ms = (man << (sc + 1u)) & qmax;
    const uint32_t exp_out = shared - sc;
    return (u & 0x80000000u) | (exp_out << 23) | (ms << (23u - M));
}

// format code → y bits with NO divergence between the mixed-tile formats: one BFP evaluation with a per-lane
// mantissa width, bf16 / fp0 by select.
__device__ __forceinline__ uint32_t quant_elem_bits_mixed(int fmt, uint32_t u, uint32_t shared)
{
    const uint32_t M = fmt == 1 ? 7u : (fmt == 2 ? 3u : 1u);
    const uint32_t b = bfp_elem_bits_rt(u, shared, M);
    return fmt == 0 ? bf16_round_bits(u) : ((fmt >= 1 && fmt <= 3) ? b : 0u);
}

__device__ __forceinline__ uint32_t quant_elem_bits(int fmt, uint32_t u, uint32_t shared)
{
    switch (fmt) {
    case 0: return bf16_round_bits(u);
    case 1: return bfp_elem_bits<7>(u, shared);
    case 2: return bfp_elem_bits<3>(u, shared);
    case 3: return bfp_elem_bits<1>(u, shared);
    default: return 0u; // fp0, quantization_formats.py:167-168
    }
}

// numpy's max propagates NaN (tile_utils.py:56).
__device__ __forceinline__ double nanmax(double m, double d) { return (d > m || d != d) ? d : m; }

// Load one 16-element group as raw fp32 words; elements outside the matrix read as +0.0
// (tile_utils.py:109-113).  `fast` = whole group in bounds and 16-byte aligned.
template <typename T> struct Loader;

template <> struct Loader<uint16_t> { // bf16 storage
    static __device__ __forceinline__ void group(const uint16_t *__restrict__ p, int64_t row, int64_t col0,
                                                  int64_t rows, int64_t cols, int64_t ld, bool vec_ok, uint32_t u[kGroup])
    {
        if (row < rows && col0 + kGroup <= cols && vec_ok) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p + row * ld + col0);
            const uint4 a = q[0], b = q[1];
            const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { u[2 * i] = w[i] << 16; u[2 * i + 1] = w[i] & 0xFFFF0000u; }
        } else {
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
                u[i] = (row < rows && col0 + i < cols) ? ((uint32_t)p[row * ld + col0 + i] << 16) : 0u;
        }
    }
};

template <> struct Loader<float> {
    static __device__ __forceinline__ void group(const float *__restrict__ p, int64_t row, int64_t col0,
                                                  int64_t rows, int64_t cols, int64_t ld, bool vec_ok, uint32_t u[kGroup])
    {
        if (row < rows && col0 + kGroup <= cols && vec_ok) {
            const uint4 *q = reinterpret_cast<const uint4 *>(p + row * ld + col0);
#pragma unroll
            for (int i = 0; i < 4; ++i) { const uint4 a = q[i]; u[4 * i] = a.x; u[4 * i + 1] = a.y; u[4 * i + 2] = a.z; u[4 * i + 3] = a.w; }
        } else {
#pragma unroll
            for (int i = 0; i < kGroup; ++i)
                u[i] = (row < rows && col0 + i < cols) ? __float_as_uint(p[row * ld + col0 + i]) : 0u;
        }
    }
};

// A RAGGED batch (include/mtq.h, mtq_tile_stats_ragged): up to kRaggedMax matrices of one storage type and any shapes whose tiles are
// numbered through — matrix j's tiles, row-major, are first[j] .. first[j] + tiles_j − 1 of the launch.  The table rides in the kernel
// arguments (1.2 KB): a wave finds the matrix of a tile with scalar compares, no upload precedes the launch.  n = 0: the launch is a uniform
// batch described by the kernel's other arguments.
constexpr int kRaggedMax = 24;
struct RaggedSeg { const void *x; int64_t rows, cols, ld; uint32_t tiles_w, first; int32_t vec_ok, pad_; };
struct RaggedTable { uint32_t n, total; RaggedSeg seg[kRaggedMax]; };
struct TileSite { const void *x; int64_t rows, cols, ld; uint32_t tiles_w, t; int vec_ok; };

__device__ __forceinline__ TileSite ragged_site(const RaggedTable &tb, uint32_t gt)
{
    uint32_t b = 0;
    for (uint32_t i = 1; i < tb.n; ++i) b += gt >= tb.seg[i].first ? 1u : 0u;
    const RaggedSeg &s = tb.seg[b];
    return TileSite{s.x, s.rows, s.cols, s.ld, s.tiles_w, gt - s.first, s.vec_ok};
}

__device__ __forceinline__ uint32_t group_shared_exp(const uint32_t u[kGroup])
{
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) m = max(m, u[i] & 0x7F800000u); // :118-119 (max exponent field)
    return m >> 23;
}


// Stats terms of ONE shared-exponent group by the literal route: uint32 quantisation per format,
// float32 terms (x*x, y*y, x*y, |x-y|) summed sequentially in float64.  t[0..1] = Σx, Σx²;
// t[2+5f .. 6+5f] = Σy, Σy², Σxy, Σ|x−y|, max|x−y| of format f (bf16, bfp8, bfp4, bfp2); formats
// outside fmt_mask are left 0.  Reference: mixed_tile_greedy.py:158-164,250-254 (terms),
// quantization_formats.py:84-164 (y).  Shared by the generic kernel and by the fast kernel's fallback.
__device__ __forceinline__ void group_terms_literal(const uint32_t u[kGroup], uint32_t fmt_mask, double t[2 + 5 * kNumFmt])
{
    const uint32_t shared = group_shared_exp(u);
    // main class: within 14 binades of the shared exponent; tail class: the rest (zeros/denormals included).
    // Each class is summed sequentially in index order, then S = S_main + S_tail (include/mtq.h).
    uint32_t tailmask = 0u;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) tailmask |= ((shared - ((u[i] >> 23) & 0xFFu)) > 14u ? 1u : 0u) << i;
    double sx = 0.0, sx2 = 0.0, tx = 0.0, tx2 = 0.0;
#pragma unroll
    for (int i = 0; i < kGroup; ++i) {
        const float xv = __uint_as_float(u[i]);
        const float p = xv * xv;
        if (tailmask & (1u << i)) { tx += (double)xv; tx2 += (double)p; }
        else { sx += (double)xv; sx2 += (double)p; }
    }
    t[0] = sx + tx;
    t[1] = sx2 + tx2;
#pragma unroll
    for (int f = 0; f < kNumFmt; ++f) {
        double sy = 0.0, sy2 = 0.0, sxy = 0.0, sab = 0.0, mx = 0.0, ty = 0.0, ty2 = 0.0, txy = 0.0, tab = 0.0;
        if (fmt_mask & (1u << f)) { // wave-uniform
#pragma unroll
            for (int i = 0; i < kGroup; ++i) {
                const float xv = __uint_as_float(u[i]);
                const float yv = __uint_as_float(quant_elem_bits(f, u[i], shared));
                const float p2 = yv * yv, pxy = xv * yv, df = fabsf(xv - yv);
                if (tailmask & (1u << i)) { ty += (double)yv; ty2 += (double)p2; txy += (double)pxy; tab += (double)df; }
                else { sy += (double)yv; sy2 += (double)p2; sxy += (double)pxy; sab += (double)df; }
                mx = nanmax(mx, (double)df);
            }
        }
        t[2 + 5 * f] = sy + ty; t[3 + 5 * f] = sy2 + ty2; t[4 + 5 * f] = sxy + txy; t[5 + 5 * f] = sab + tab; t[6 + 5 * f] = mx;
    }
}

// One tile by the literal route, complete in every lane: lane ℓ holds the group (row ℓ>>1, half ℓ&1) in u; the 4 groups of a row
// pair (lanes 4j..4j+3) are added sequentially, the 16 row pairs by a balanced tree (the order include/mtq.h documents).  acc is indexed by
// FORMAT (static); formats outside fmt_mask are left as group_terms_literal left them.
__device__ __forceinline__ void tile_terms_literal(const uint32_t u[kGroup], uint32_t fmt_mask, double acc[2 + 5 * kNumFmt])
{
    const int lane = threadIdx.x & 63;
    group_terms_literal(u, fmt_mask, acc);
#pragma unroll
    for (int k = 0; k < 2 + 5 * kNumFmt; ++k) {
        const bool live = k < 2 || (fmt_mask & (1u << ((k - 2) / 5))); // wave-uniform
        if (live) {
            const bool is_max = k >= 2 && ((k - 2) % 5) == 4;
            double v = acc[k];
            {   // the 4 groups of a row pair (lanes 4j..4j+3) sequentially, identically in every lane of the quad
                const int q0 = lane & ~3;
                const double a = __shfl(v, q0, 64), b = __shfl(v, q0 + 1, 64), c = __shfl(v, q0 + 2, 64), d = __shfl(v, q0 + 3, 64);
                v = is_max ? nanmax(nanmax(nanmax(a, b), c), d) : ((a + b) + c) + d;
            }
#pragma unroll
            for (int s = 4; s < 64; s <<= 1) { // 16 row pairs: balanced tree
                const double o = __shfl_xor(v, s, 64);
                v = is_max ? nanmax(v, o) : v + o;
            }
            acc[k] = v;
        }
    }
}

} // namespace mtq
