// mtq_host.cpp — host side of libmtq_hip.so: error state, device probe, and the decision steps that
// run on HOST copies of the per-tile stats records (include/mtq.h, HOST section): the sequential
// greedy scan, per-tile scores, threshold assignment and tensor-level columns.
//
// Build with -ffp-contract=off: the reference evaluates these formulas as separately rounded
// Python-float (IEEE double) operations (mixed_tile_greedy.py:176-190).
#include <hip/hip_runtime_api.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mtq.h"
#include "mtq_decide.hpp"
#include "mtq_error.hpp"

namespace mtq {

static thread_local char g_err[512] = "";

int fail(int code, const char *msg)
{
    std::snprintf(g_err, sizeof g_err, "%s", msg);
    return code;
}

int failf(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int require_device()
{
    static int cached = 1; // 1 = unknown
    if (cached == 1) {
        int n = 0;
        const hipError_t e = hipGetDeviceCount(&n);
        cached = (e == hipSuccess && n > 0) ? MTQ_OK : MTQ_ERR_NO_DEVICE;
    }
    if (cached != MTQ_OK) return fail(MTQ_ERR_NO_DEVICE, "no usable HIP device (libmtq_hip has no CPU fallback)");
    return MTQ_OK;
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return failf(MTQ_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return MTQ_OK;
}

// ------------------------------------------------------------------------------------------------

// load5 for a handle's slot width: slim records carry Σy, Σy², Σxy only (Σ|d| and max read as 0).
static inline Sums5 load5w(const double *r, int slot, int w)
{
    if (w == 5 || slot < 0) return load5(r, slot);
    const double *b = r + 2 + 3 * slot;
    return {b[0], b[1], b[2], 0.0, 0.0};
}

// sums5: pointer form of load5 (mtq_decide.hpp) for loops that want the record slot in place.
static inline const double *sums5(const double *r, int slot, double buf[5])
{
    if (slot >= 0) return r + 2 + 5 * slot;
    const double z = std::fabs(r[0]) * 0.0; // 0, or NaN when Σx is not finite (what the kernel stores)
    buf[0] = r[0]; buf[1] = r[1]; buf[2] = r[1]; buf[3] = z; buf[4] = z;
    return buf;
}

// pcc_value (:176-190) with the x-only subexpressions (mean_x, am2) taken from the handle: the same IEEE operations
// in the same order, evaluated once instead of at every step.
static inline double pcc_hoisted(double n, double mean_x, double am2, double sy, double sy2, double sxy, double sab, bool *degenerate)
{
    if (n == 0.0) return 1.0;
    const double mean_y = sy / n;
    double bm2 = sy2 - n * mean_y * mean_y;
    if (bm2 < 0.0) bm2 = 0.0;
    const double denom = std::sqrt(am2 * bm2);
    if (denom == 0.0) { *degenerate = true; return sab == 0.0 ? 1.0 : 0.0; } // the one place Σ|d| decides (slim records do not carry it)
    return (sxy - n * mean_x * mean_y) / denom;
}

#if defined(__x86_64__)
// pcc_hoisted for eight candidates at once (AVX-512: correctly rounded division and square root per lane, the same
// operations in the same order as the scalar form — the bits of every lane are the scalar result's).  Returns the mask of
// lanes with value >= thr; *special gets the lanes whose denominator is zero (the scalar path decides those: Σ|d|).
__attribute__((target("avx512f"))) static inline unsigned pcc_good8(__m512d cy, __m512d cy2, __m512d cxy, double n, double mean_x, double am2, double thr,
                                                                     unsigned *special)
{
    const __m512d zero = _mm512_setzero_pd(), N = _mm512_set1_pd(n);
    const __m512d mean_y = _mm512_div_pd(cy, N);
    __m512d bm2 = _mm512_sub_pd(cy2, _mm512_mul_pd(_mm512_mul_pd(N, mean_y), mean_y));
    bm2 = _mm512_mask_blend_pd(_mm512_cmp_pd_mask(bm2, zero, _CMP_LT_OQ), bm2, zero);
    const __m512d denom = _mm512_sqrt_pd(_mm512_mul_pd(_mm512_set1_pd(am2), bm2));
    *special = _mm512_cmp_pd_mask(denom, zero, _CMP_EQ_OQ);
    const __m512d num = _mm512_sub_pd(cxy, _mm512_mul_pd(_mm512_mul_pd(N, _mm512_set1_pd(mean_x)), mean_y));
    return _mm512_cmp_pd_mask(_mm512_div_pd(num, denom), _mm512_set1_pd(thr), _CMP_GE_OQ);
}
static const bool g_avx512 = __builtin_cpu_supports("avx512f") && !std::getenv("MTQ_SCAN_SCALAR");
#else
static const bool g_avx512 = false;
#endif

} // namespace mtq

using namespace mtq;

struct mtq_greedy {
    int64_t T;
    uint32_t mask;
    int rec;
    int metric;
    double thr, n;
    const double *stats; // caller-owned, must outlive the handle
    double sum_x, sum_x2, sum_y, sum_y2, sum_xy, sum_abs;
    bool cur_valid;     // cur_value below is the metric of the current sums (no accepted move since it was computed)
    double cur_value;
    double mean_x, am2; // mean_x = sum_x / n and am2 = max(sum_x2 - n*mean_x*mean_x, 0): constant during the scan (:179,181,183)
    double max_abs;
    int64_t max_count;
    int w;              // doubles per format slot: 5, or 3 for slim records (MTQ_MASK_SLIM: no Σ|d|, no max)
    bool degenerate;    // slim records only: a decision needed Σ|d| (zero-variance case) — the result is not valid
    int slot4[MTQ_NUM_TILE_FORMATS]; // record slot of every format (−1: not in mask)
    std::vector<int8_t> assign;  // the CURRENT format of every tile; its sums are stats[t][2 + 5*slot4[assign[t]] ..]
    std::vector<uint8_t> fixed;
    int64_t counts[MTQ_NUM_TILE_FORMATS];
};

extern "C" int mtq_version(void) { return MTQ_VERSION; }
extern "C" const char *mtq_last_error(void) { return g_err; }

extern "C" int mtq_device_count(int *count)
{
    if (!count) return fail(MTQ_ERR_INVALID, "count is null");
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return fail(MTQ_ERR_NO_DEVICE, "no usable HIP device");
    }
    *count = n;
    return MTQ_OK;
}

extern "C" int mtq_greedy_create(mtq_greedy **out, const double *stats, int64_t tiles, uint32_t fmt_mask, int metric,
                                 double threshold, double elem_count, int base_fmt)
{
    if (!out || !stats) return fail(MTQ_ERR_INVALID, "null argument");
    if (tiles <= 0) return fail(MTQ_ERR_INVALID, "tiles must be positive");
    if (metric < MTQ_METRIC_PCC || metric > MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    const int bslot = slot_of(fmt_mask, base_fmt);
    if (!slot_ok(bslot)) return fail(MTQ_ERR_INVALID, "base format is not in fmt_mask");
    mtq_greedy *g = new (std::nothrow) mtq_greedy();
    if (!g) return fail(MTQ_ERR_INVALID, "out of memory");
    g->T = tiles;
    g->mask = fmt_mask & (MTQ_MASK_ALL | MTQ_MASK_BF16_IDENTITY);
    g->w = (fmt_mask & MTQ_MASK_SLIM) ? 3 : 5;
    g->degenerate = false;
    if (g->w == 3 && metric != MTQ_METRIC_PCC) { delete g; return fail(MTQ_ERR_INVALID, "slim records serve the pcc metric only"); }
    g->rec = 2 + g->w * popcount4(fmt_mask);
    g->metric = metric;
    g->thr = threshold;
    g->n = elem_count;
    g->stats = stats;
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) g->slot4[f] = slot_of(fmt_mask, f);
    g->assign.assign((size_t)tiles, (int8_t)base_fmt); // :99
    g->fixed.assign((size_t)tiles, 0);                 // :100
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) g->counts[f] = 0;
    g->counts[base_fmt] = tiles;                       // :102-103
    g->sum_x = g->sum_x2 = g->sum_y = g->sum_y2 = g->sum_xy = g->sum_abs = 0.0;
    // running globals accumulated in tile order (:147-174, :195-204, :208-218)
    double vb[5];
    for (int64_t t = 0; t < tiles; ++t) {
        const double *r = stats + t * g->rec;
        const Sums5 b = load5w(r, bslot, g->w);
        g->sum_x += r[0];
        g->sum_x2 += r[1];
        g->sum_y += b.y;
        g->sum_y2 += b.y2;
        g->sum_xy += b.xy;
        g->sum_abs += b.ab;
    }
    g->cur_valid = false;
    g->cur_value = 0.0;
    g->mean_x = elem_count != 0.0 ? g->sum_x / elem_count : 0.0;
    g->am2 = g->sum_x2 - elem_count * g->mean_x * g->mean_x;
    if (g->am2 < 0.0) g->am2 = 0.0;
    g->max_abs = 0.0;
    g->max_count = 0;
    if (metric == MTQ_METRIC_ATOL) { // :219-220 — the running maximum and its multiplicity (only the atol scan reads them)
        double m = sums5(stats, bslot, vb)[4];
        for (int64_t t = 1; t < tiles; ++t) m = nanmax(m, sums5(stats + t * g->rec, bslot, vb)[4]);
        int64_t c = 0;
        for (int64_t t = 0; t < tiles; ++t) c += (sums5(stats + t * g->rec, bslot, vb)[4] == m);
        g->max_abs = m;
        g->max_count = c;
    }
    *out = g;
    return MTQ_OK;
}

// One visit of the pcc scan (mixed_tile_greedy.py:237-278), the scalar form: also the fallback of the eight-wide pass below.
static inline void pcc_visit(mtq_greedy *g, int fmt, int slot, int64_t t)
{
    const double thr = g->thr, N = g->n;
    const int prev = g->assign[(size_t)t];
    const double *rt = g->stats + t * g->rec;
    if (prev == fmt) { // :237-241 — current_value is a pure function of the running sums: reuse it until a move is accepted
        if (!g->cur_valid) {
            g->cur_value = pcc_hoisted(N, g->mean_x, g->am2, g->sum_y, g->sum_y2, g->sum_xy, g->sum_abs, &g->degenerate);
            g->cur_valid = true;
        }
        if (!is_good(g->cur_value, MTQ_METRIC_PCC, thr)) g->fixed[(size_t)t] = 1;
        return;
    }
    const Sums5 cur = load5w(rt, g->slot4[prev], g->w), q = load5w(rt, slot, g->w); // the tile's CURRENT format and the candidate
    const double cy = g->sum_y + (q.y - cur.y);     // :259
    const double cy2 = g->sum_y2 + (q.y2 - cur.y2); // :260
    const double cxy = g->sum_xy + (q.xy - cur.xy); // :261
    const double cab = g->sum_abs + (q.ab - cur.ab); // :262
    if (is_good(pcc_hoisted(N, g->mean_x, g->am2, cy, cy2, cxy, cab, &g->degenerate), MTQ_METRIC_PCC, thr)) { // :264-276
        g->sum_y = cy; g->sum_y2 = cy2; g->sum_xy = cxy; g->sum_abs = cab; g->cur_valid = false;
        g->counts[prev]--;
        g->counts[fmt]++;
        g->assign[(size_t)t] = (int8_t)fmt;
    } else {
        g->fixed[(size_t)t] = 1; // :277-278
    }
}

#if defined(__x86_64__)
// The pcc pass, eight candidates per step.  Two divisions and a square root per visit keep the divider busy for ~15 cycles
// and make up most of a scalar visit; eight lanes share them here.  The visits stay sequential in effect by speculating on
// the outcome of the batch and keeping only the prefix the speculation was right for:
//   accept mode — lane i assumes lanes < i were accepted: its sums are the running sums plus the deltas of lanes 0..i, added
//                 one after the other exactly as the sequential scan adds them; everything up to the first rejected lane
//                 (that lane's rejection included) is what the sequential scan does;
//   reject mode — every lane assumes the running sums are unchanged; everything up to the first accepted lane (included) is
//                 what the sequential scan does.
// The mode follows the last outcome (long runs of either kind are the rule: the scan accepts until the metric reaches the
// threshold and mostly rejects afterwards).  Batches with an out-of-range id, a tile already in this format or a zero
// denominator, and the tail of the order, go through pcc_visit one by one.
enum { kScanFull = 0, kScanSlim = 1 };
template <int kMode, typename Idx>
__attribute__((target("avx512f"))) static int greedy_pass_pcc8(mtq_greedy *g, int fmt, int slot, const Idx *order, int64_t n)
{
    constexpr bool kSlim = kMode != kScanFull;
    const double thr = g->thr, N = g->n;
    const int w = g->w, rec = g->rec;
    // where Σy, Σy², Σxy of every format sit in a slim record (the identity bf16 reads Σx, Σx², Σx²): branch-free deltas
    int oy[MTQ_NUM_TILE_FORMATS], oy2[MTQ_NUM_TILE_FORMATS], oxy[MTQ_NUM_TILE_FORMATS];
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) {
        const int ps = g->slot4[f];
        oy[f] = ps >= 0 ? 2 + w * ps : 0;
        oy2[f] = ps >= 0 ? 3 + w * ps : 1;
        oxy[f] = ps >= 0 ? 4 + w * ps : 1;
    }
    const int so = oy[fmt];
    bool accept_mode = true;
    alignas(64) double cy[8], cy2[8], cxy[8], cab[8];
    int64_t moved[MTQ_NUM_TILE_FORMATS] = {0, 0, 0, 0};   // accepted moves by previous format (counts are adjusted once, at the end)
    int64_t k = 0;
    int rc = MTQ_OK;
    while (k < n) {
        int64_t tt[8];
        int prev[8];
        bool plain = n - k >= 8;
        for (int i = 0; plain && i < 8; ++i) {
            const int64_t t = order[k + i];
            if ((uint64_t)t >= (uint64_t)g->T) { plain = false; break; }
            const int p = g->assign[(size_t)t];
            if (p == fmt) { plain = false; break; }
            tt[i] = t;
            prev[i] = p;
        }
        if (k + 24 <= n) { // the records of the batch after next (random visiting order, 2 cache lines each)
            for (int i = 16; i < 24; ++i) {
                const int64_t ta = order[k + i];
                if ((uint64_t)ta >= (uint64_t)g->T) continue;
                const double *ra = g->stats + ta * rec;
                __builtin_prefetch(ra + so);
                __builtin_prefetch(ra + oy[g->assign[(size_t)ta]]);
                if (!kSlim) __builtin_prefetch(ra + so + 4);
            }
        }
        if (!plain) {
            const int64_t t = order[k];
            if ((uint64_t)t >= (uint64_t)g->T) { rc = fail(MTQ_ERR_INVALID, "order contains a tile id out of range"); break; }
            pcc_visit(g, fmt, slot, t);
            ++k;
            continue;
        }
        // deltas of the eight candidates, kept in registers (a 64-byte load of eight fresh scalar stores would not be forwarded)
        double dy[8], dy2[8], dxy[8], dab[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double *rt = g->stats + tt[i] * rec;
            if (kSlim) {
                const int p = prev[i];
                dy[i] = rt[so] - rt[oy[p]]; dy2[i] = rt[so + 1] - rt[oy2[p]]; dxy[i] = rt[oxy[fmt]] - rt[oxy[p]];
            } else {
                const Sums5 cur = load5w(rt, g->slot4[prev[i]], w), q = load5w(rt, slot, w);
                dy[i] = q.y - cur.y; dy2[i] = q.y2 - cur.y2; dxy[i] = q.xy - cur.xy; dab[i] = q.ab - cur.ab;
            }
        }
        __m512d vy, vy2, vxy;
        if (accept_mode) { // lane i: the running sums after lanes 0..i, added one after the other as the sequential scan adds them
            double a[8], b[8], c[8], sab = g->sum_abs;
            a[0] = g->sum_y + dy[0]; b[0] = g->sum_y2 + dy2[0]; c[0] = g->sum_xy + dxy[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) { a[i] = a[i - 1] + dy[i]; b[i] = b[i - 1] + dy2[i]; c[i] = c[i - 1] + dxy[i]; }
            vy = _mm512_set_pd(a[7], a[6], a[5], a[4], a[3], a[2], a[1], a[0]);
            vy2 = _mm512_set_pd(b[7], b[6], b[5], b[4], b[3], b[2], b[1], b[0]);
            vxy = _mm512_set_pd(c[7], c[6], c[5], c[4], c[3], c[2], c[1], c[0]);
            if (!kSlim) for (int i = 0; i < 8; ++i) { sab = sab + dab[i]; cab[i] = sab; }
        } else {           // lane i: the running sums plus its own delta
            vy = _mm512_add_pd(_mm512_set1_pd(g->sum_y), _mm512_set_pd(dy[7], dy[6], dy[5], dy[4], dy[3], dy[2], dy[1], dy[0]));
            vy2 = _mm512_add_pd(_mm512_set1_pd(g->sum_y2), _mm512_set_pd(dy2[7], dy2[6], dy2[5], dy2[4], dy2[3], dy2[2], dy2[1], dy2[0]));
            vxy = _mm512_add_pd(_mm512_set1_pd(g->sum_xy), _mm512_set_pd(dxy[7], dxy[6], dxy[5], dxy[4], dxy[3], dxy[2], dxy[1], dxy[0]));
            if (!kSlim) for (int i = 0; i < 8; ++i) cab[i] = g->sum_abs + dab[i];
        }
        _mm512_store_pd(cy, vy); _mm512_store_pd(cy2, vy2); _mm512_store_pd(cxy, vxy);   // for the lane that ends up as the running sums
        unsigned special = 0;
        const unsigned ok = pcc_good8(vy, vy2, vxy, N, g->mean_x, g->am2, thr, &special) & 0xFFu;
        if (special) { // a zero denominator in the batch: Σ|d| decides (and slim records flag the tensor) — one by one
            for (int i = 0; i < 8; ++i) pcc_visit(g, fmt, slot, tt[i]);
            k += 8;
            continue;
        }
        auto take = [&](int i) { // lane i's candidate is accepted last: its sums become the running sums
            g->sum_y = cy[i]; g->sum_y2 = cy2[i]; g->sum_xy = cxy[i]; g->cur_valid = false;
            if (!kSlim) g->sum_abs = cab[i];
        };
        auto move = [&](int i) { moved[prev[i]]++; g->assign[(size_t)tt[i]] = (int8_t)fmt; };
        if (accept_mode) {
            const int j = __builtin_ctz((~ok & 0xFFu) | 0x100u);   // first rejected lane, 8 when none
            for (int i = 0; i < j; ++i) move(i);
            if (j > 0) take(j - 1);
            if (j < 8) { g->fixed[(size_t)tt[j]] = 1; k += j + 1; if (j == 0) accept_mode = false; }
            else k += 8;
        } else {
            const int j = __builtin_ctz(ok | 0x100u);              // first accepted lane, 8 when none
            for (int i = 0; i < j; ++i) g->fixed[(size_t)tt[i]] = 1;
            if (j < 8) { move(j); take(j); k += j + 1; if (j == 0) accept_mode = true; }
            else k += 8;
        }
    }
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) { g->counts[f] -= moved[f]; g->counts[fmt] += moved[f]; }
    return rc;
}
#endif

// mtq_greedy_pass for a visiting order of 64-bit ids (the C ABI) or 32-bit ids (mtq_greedy_run: half the footprint of the
// shuffled array)
template <typename Idx>
static int greedy_pass_any(mtq_greedy *g, int fmt, const Idx *order, int64_t n)
{
    if (!g || (!order && n > 0)) return fail(MTQ_ERR_INVALID, "null argument");
    const int slot = slot_of(g->mask, fmt);
    if (!slot_ok(slot)) return fail(MTQ_ERR_INVALID, "format is not in the handle's fmt_mask");
#if defined(__x86_64__)
    if (g->metric == MTQ_METRIC_PCC && g_avx512 && g->n != 0.0)
        return g->w == 3 ? greedy_pass_pcc8<kScanSlim, Idx>(g, fmt, slot, order, n) : greedy_pass_pcc8<kScanFull, Idx>(g, fmt, slot, order, n);
#endif
    const double thr = g->thr, N = g->n;
    constexpr int64_t kAhead = 12; // the visiting order is random and a record is 2–3 cache lines: fetch ahead of the dependent arithmetic
    for (int64_t k = 0; k < n; ++k) {
        const int64_t t = order[k];
        if (t < 0 || t >= g->T) return fail(MTQ_ERR_INVALID, "order contains a tile id out of range");
        if (k + kAhead < n) {
            const int64_t ta = order[k + kAhead];
            if (ta >= 0 && ta < g->T && g->assign[(size_t)ta] != fmt) { // a tile already in this format is decided without its record
                const double *ra = g->stats + ta * g->rec;      // the identity slot lives in the record's first two doubles
                const int sa = slot >= 0 ? 2 + g->w * slot : 0, pa = g->slot4[g->assign[(size_t)ta]] >= 0 ? 2 + g->w * g->slot4[g->assign[(size_t)ta]] : 0;
                __builtin_prefetch(ra + sa);
                __builtin_prefetch(ra + sa + 4);
                __builtin_prefetch(ra + pa);
                __builtin_prefetch(ra + pa + 4);
            }
        }
        const int prev = g->assign[(size_t)t];
        const double *rt = g->stats + t * g->rec;
        bool accept;
        if (g->metric == MTQ_METRIC_PCC) {
            pcc_visit(g, fmt, slot, t);
            continue;
        } else if (g->metric == MTQ_METRIC_MAE) {
            if (prev == fmt) { // :280-284
                if (!is_good(N != 0.0 ? g->sum_abs / N : 0.0, MTQ_METRIC_MAE, thr)) g->fixed[(size_t)t] = 1;
                continue;
            }
            const double cab = g->sum_abs + (load5(rt, slot).ab - load5(rt, g->slot4[prev]).ab); // :293
            accept = is_good(N != 0.0 ? cab / N : 0.0, MTQ_METRIC_MAE, thr);
            if (accept) g->sum_abs = cab;
        } else {
            if (prev == fmt) { // :305-309
                if (!is_good(g->max_abs, MTQ_METRIC_ATOL, thr)) g->fixed[(size_t)t] = 1;
                continue;
            }
            const double new_max = load5(rt, slot).mx, old_max = load5(rt, g->slot4[prev]).mx;
            double cand_max = g->max_abs;
            int64_t cand_count = g->max_count;
            if (new_max > g->max_abs) { // :322-324
                cand_max = new_max;
                cand_count = 1;
            } else if (new_max == g->max_abs) { // :325-327
                if (old_max != g->max_abs) cand_count = g->max_count + 1;
            } else if (old_max == g->max_abs) { // :329
                if (g->max_count > 1) cand_count = g->max_count - 1; // :330-331
                else {                                                 // :333-336 full recount
                    auto cur_max = [&](int64_t j) { double vb[5]; return sums5(g->stats + j * g->rec, g->slot4[g->assign[(size_t)j]], vb)[4]; };
                    double m = (t == 0) ? new_max : cur_max(0);
                    for (int64_t j = 1; j < g->T; ++j) m = nanmax(m, j == t ? new_max : cur_max(j));
                    int64_t c = 0;
                    for (int64_t j = 0; j < g->T; ++j) c += ((j == t ? new_max : cur_max(j)) == m);
                    cand_max = m;
                    cand_count = c;
                }
            }
            accept = is_good(cand_max, MTQ_METRIC_ATOL, thr); // :337
            if (accept) { g->max_abs = cand_max; g->max_count = cand_count; }
        }
        if (accept) { // :264-276 / :295-301 / :337-344 — the tile's current sums are now the candidate's record slot
            g->counts[prev]--;
            g->counts[fmt]++;
            g->assign[(size_t)t] = (int8_t)fmt;
        } else {
            g->fixed[(size_t)t] = 1; // :277-278
        }
    }
    return MTQ_OK;
}

extern "C" int mtq_greedy_pass(mtq_greedy *g, int fmt, const int64_t *order, int64_t n) { return greedy_pass_any<int64_t>(g, fmt, order, n); }

extern "C" int mtq_greedy_assignment(const mtq_greedy *g, int8_t *assign)
{
    if (!g || !assign) return fail(MTQ_ERR_INVALID, "null argument");
    std::memcpy(assign, g->assign.data(), (size_t)g->T);
    return MTQ_OK;
}

extern "C" int mtq_greedy_fixed(const mtq_greedy *g, uint8_t *fixed)
{
    if (!g || !fixed) return fail(MTQ_ERR_INVALID, "null argument");
    std::memcpy(fixed, g->fixed.data(), (size_t)g->T);
    return MTQ_OK;
}

extern "C" int mtq_greedy_counts(const mtq_greedy *g, int64_t counts[4])
{
    if (!g || !counts) return fail(MTQ_ERR_INVALID, "null argument");
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) counts[f] = g->counts[f];
    return MTQ_OK;
}

extern "C" int mtq_greedy_value(const mtq_greedy *g, double *value)
{
    if (!g || !value) return fail(MTQ_ERR_INVALID, "null argument");
    if (g->metric == MTQ_METRIC_PCC) *value = pcc_from_moments(g->n, g->sum_x, g->sum_x2, g->sum_y, g->sum_y2, g->sum_xy, g->sum_abs);
    else if (g->metric == MTQ_METRIC_MAE) *value = g->n != 0.0 ? g->sum_abs / g->n : 0.0;
    else *value = g->max_abs;
    return MTQ_OK;
}

extern "C" void mtq_greedy_destroy(mtq_greedy *g) { delete g; }

extern "C" int mtq_tile_scores(const double *stats, int64_t tiles, uint32_t fmt_mask, int metric, double *scores)
{
    if (!stats || !scores) return fail(MTQ_ERR_INVALID, "null argument");
    if (metric < MTQ_METRIC_PCC || metric > MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    const int rec = 2 + 5 * popcount4(fmt_mask);
    int64_t row = 0;
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) { // ascending format code; the identity bf16 (if any) comes first
        const int slot = slot_of(fmt_mask, f);
        if (!slot_ok(slot)) continue;
        for (int64_t t = 0; t < tiles; ++t) scores[row * tiles + t] = tile_score(stats + t * rec, slot, metric);
        ++row;
    }
    return MTQ_OK;
}

extern "C" int mtq_threshold_assign(const double *stats, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                                    int metric, double threshold, double band, int8_t *map, int64_t *knife_ids,
                                    uint8_t *knife_near, int64_t knife_cap, int64_t *n_knife)
{
    if (!stats || !formats || !map) return fail(MTQ_ERR_INVALID, "null argument");
    if (n_formats <= 0 || n_formats > MTQ_NUM_TILE_FORMATS) return fail(MTQ_ERR_INVALID, "n_formats must be 1..4");
    if (metric < MTQ_METRIC_PCC || metric > MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    ThresholdPlan plan;
    if (!plan_threshold(fmt_mask, formats, n_formats, plan)) return fail(MTQ_ERR_INVALID, "a requested format is not in fmt_mask");
    const int rec = 2 + 5 * popcount4(fmt_mask);
    const double thr32 = (double)(float)threshold; // NumPy >= 2 compares np.float32 score with float32(threshold) (metrics.py:30-33, NEP 50)
    int64_t nk = 0;
    for (int64_t t = 0; t < tiles; ++t) {
        unsigned near;
        map[t] = (int8_t)threshold_decide(stats + t * rec, plan, metric, thr32, band, near);
        if (near) {
            if (nk < knife_cap) {
                if (knife_ids) knife_ids[nk] = t;
                if (knife_near) knife_near[nk] = (uint8_t)near;
            }
            ++nk;
        }
    }
    if (n_knife) *n_knife = nk;
    return MTQ_OK;
}

extern "C" int mtq_columns_from_stats(const double *stats, int64_t tiles, uint32_t fmt_mask, const int8_t *map,
                                      double elem_count, double out[9])
{
    if (!stats || !map || !out) return fail(MTQ_ERR_INVALID, "null argument");
    const int rec = 2 + 5 * popcount4(fmt_mask);
    double sx = 0, sx2 = 0, sy = 0, sy2 = 0, sxy = 0, sab = 0, mx = 0;
    for (int64_t t = 0; t < tiles; ++t) {
        const int slot = slot_of(fmt_mask, map[t]);
        if (!slot_ok(slot)) return fail(MTQ_ERR_INVALID, "map names a format that is not in fmt_mask");
        double vb[5];
        const double *r = stats + t * rec, *b = sums5(r, slot, vb);
        sx += r[0]; sx2 += r[1]; sy += b[0]; sy2 += b[1]; sxy += b[2]; sab += b[3];
        mx = nanmax(mx, b[4]);
    }
    const double sums[7] = {sx, sx2, sy, sy2, sxy, sab, mx};
    return mtq_columns_from_sums(sums, elem_count, out);
}

extern "C" int mtq_columns_from_sums(const double sums[7], double elem_count, double out[9])
{
    if (!sums || !out) return fail(MTQ_ERR_INVALID, "null argument");
    out[0] = pcc_from_moments(elem_count, sums[0], sums[1], sums[2], sums[3], sums[4], sums[5]);
    out[1] = elem_count != 0.0 ? sums[5] / elem_count : 0.0;
    out[2] = sums[6];
    for (int k = 0; k < 6; ++k) out[3 + k] = sums[k];
    return MTQ_OK;
}

// ------------------------------------------------------------------------------------------------
// NumPy-compatible visiting order (mixed_tile_greedy.py:222-231 uses np.random.default_rng(seed) and
// rng.permutation(candidates)).  Bit-compatible restatement of NumPy's published algorithms:
// SeedSequence (numpy/random/bit_generator.pyx: hashmix/mix pool of 4 uint32) → PCG64 (128-bit LCG,
// XSL-RR output, numpy/random/src/pcg64) → Generator.permutation = Fisher–Yates from the top with
// random_interval's masked rejection sampling on 32-bit draws (numpy/random/_generator.pyx,
// src/distributions/distributions.c).  Pinned against NumPy itself in tests/test_capi_host.py.
// permutation(array) == array[permutation(len(array))] for the same generator state.
// ------------------------------------------------------------------------------------------------
struct mtq_rng {
    unsigned __int128 state, inc;
    bool has32;
    uint32_t u32;
};

namespace {

const unsigned __int128 kPcgMult = ((unsigned __int128)0x2360ED051FC65DA4ull << 64) | 0x4385DF649FCCF645ull;

inline void pcg_step(mtq_rng *r) { r->state = r->state * kPcgMult + r->inc; }

inline uint64_t pcg_output(unsigned __int128 state)   // XSL-RR of a state
{
    const uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(state >> 122);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}

inline uint64_t pcg_next64(mtq_rng *r)
{
    pcg_step(r);
    return pcg_output(r->state);
}

inline uint32_t pcg_next32(mtq_rng *r)
{
    if (r->has32) { r->has32 = false; return r->u32; }
    const uint64_t n = pcg_next64(r);
    r->has32 = true;
    r->u32 = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

void rng_seed(mtq_rng *r, uint64_t seed)
{
    // SeedSequence(seed): entropy = little-endian uint32 words of the integer (at least one word)
    uint32_t ent[2];
    int n_ent = 1;
    ent[0] = (uint32_t)seed;
    if (seed >> 32) { ent[1] = (uint32_t)(seed >> 32); n_ent = 2; }
    uint32_t hc = 0x43b0d7e5u; // INIT_A
    auto hashmix = [&](uint32_t v) { v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16; return v; };
    auto mix = [](uint32_t x, uint32_t y) { uint32_t t = 0xca01f9ddu * x - 0x4973f715u * y; t ^= t >> 16; return t; };
    uint32_t pool[4];
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
    // generate_state(4, uint64) = 8 uint32 words
    uint32_t hb = 0x8b51f9ddu, w[8];
    for (int i = 0; i < 8; ++i) { uint32_t v = pool[i & 3]; v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; w[i] = v; }
    const uint64_t u0 = w[0] | ((uint64_t)w[1] << 32), u1 = w[2] | ((uint64_t)w[3] << 32);
    const uint64_t u2 = w[4] | ((uint64_t)w[5] << 32), u3 = w[6] | ((uint64_t)w[7] << 32);
    const unsigned __int128 initstate = ((unsigned __int128)u0 << 64) | u1, initseq = ((unsigned __int128)u2 << 64) | u3;
    r->state = 0;
    r->inc = (initseq << 1) | 1u;
    pcg_step(r);
    r->state += initstate;
    pcg_step(r);
    r->has32 = false;
    r->u32 = 0;
}

// Generator.permutation(arr) in place (Fisher–Yates from the top, random_interval's masked rejection on buffered 32-bit
// draws; permutation(arr) == arr[permutation(len(arr))]).  The visiting order is a third of the scan's time when done
// draw by draw (an unpredictable rejection branch per element), so the draws of a block are generated first and then
// consumed without branches: a rejected draw swaps an element with itself and leaves i where it is.  Draws generated
// beyond the last one consumed are handed back by replaying the generator from the block's start.
template <bool kSwap, typename Idx>
void rng_shuffle_impl(mtq_rng *r, int64_t n, Idx *arr)
{
    if (n < 2) return;
    int64_t i = n - 1;
    uint64_t mask = ~0ull >> __builtin_clzll((uint64_t)i);
    if ((uint64_t)i > 0xFFFFFFFFull) { // 64-bit draws: the plain loop (never reached by the scan: tiles < 2^31)
        for (; i >= 1; --i) {
            while ((mask >> 1) >= (uint64_t)i) mask >>= 1;
            uint64_t v;
            if ((uint64_t)i <= 0xFFFFFFFFull) { while ((v = (pcg_next32(r) & mask)) > (uint64_t)i) {} }
            else { while ((v = (pcg_next64(r) & mask)) > (uint64_t)i) {} }
            if (kSwap) { const Idx t = arr[i]; arr[i] = arr[(int64_t)v]; arr[(int64_t)v] = t; }
        }
        return;
    }
    constexpr int kBlock = 128;                       // 64-bit outputs per block
    uint32_t buf[2 * kBlock + 1];
    uint32_t m32 = (uint32_t)mask, ii = (uint32_t)i;
    while (ii >= 1u) {
        int cnt = 0;
        const int pend = r->has32 ? 1 : 0;            // a buffered high half is the next draw
        if (pend) buf[cnt++] = r->u32;
        const unsigned __int128 state0 = r->state;
        for (int k = 0; k < kBlock; ++k) { const uint64_t o = pcg_next64(r); buf[cnt++] = (uint32_t)o; buf[cnt++] = (uint32_t)(o >> 32); }
        int p = 0;
        while (p < cnt && ii >= 1u) {
            // the mask (smallest 2^k − 1 >= ii) only changes when ii crosses a power of two: inside a level the loop-carried
            // chain is the compare and the conditional decrement alone
            while ((m32 >> 1) >= ii) m32 >>= 1;
            const uint32_t lo = m32 >> 1;                 // ii in (lo, m32] keeps this mask; lo == 0 at the last level (ii == 1)
            while (p < cnt && ii > lo) {
                const uint32_t v = buf[p++] & m32;
                const bool ok = v <= ii;
                if (kSwap) {
                    const uint32_t j = ok ? v : ii;   // rejected: swap with itself
                    const Idx a = arr[ii], b = arr[j];
                    arr[ii] = b; arr[j] = a;
                }
                ii -= ok ? 1u : 0u;
            }
        }
        if (p == cnt) { r->has32 = false; continue; } // the whole block was consumed, its last high half included
        // finished inside the block: hand the unused draws back by replaying from the block's start
        const int fresh = p - pend;                   // halves taken from this block's own outputs
        r->state = state0;
        if (fresh < 0) continue;                      // not even the pending half was needed (cannot happen with ii >= 1 on entry)
        if (fresh == 0) { r->has32 = false; continue; } // only the pending half was used
        uint64_t last = 0;
        for (int k = 0; k < (fresh + 1) / 2; ++k) last = pcg_next64(r);
        r->has32 = (fresh & 1) != 0;                  // an odd count leaves the last output's high half buffered
        r->u32 = (uint32_t)(last >> 32);
    }
}

void rng_shuffle(mtq_rng *r, int64_t n, int64_t *arr) { rng_shuffle_impl<true, int64_t>(r, n, arr); }
void rng_shuffle(mtq_rng *r, int64_t n, int32_t *arr) { rng_shuffle_impl<true, int32_t>(r, n, arr); }
// The same draws without the array: advances the generator exactly as a shuffle of n elements would.
void rng_skip_shuffle(mtq_rng *r, int64_t n) { rng_shuffle_impl<false, int64_t>(r, n, nullptr); }

void rng_permutation(mtq_rng *r, int64_t n, int64_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = i;
    rng_shuffle(r, n, out);
}

// Generator.integers(0, high, size=n, dtype=int64) for high − 1 < 2^32 − 1: Lemire's multiply-shift rejection
// on buffered 32-bit draws (numpy/random/src/distributions/distributions.c, buffered_bounded_lemire_uint32;
// Generator.integers never takes the masked form).  high == 1 consumes nothing.
void rng_integers(mtq_rng *r, uint32_t high, int64_t n, int64_t *out)
{
    const uint32_t rng = high - 1u;
    if (rng == 0u) { for (int64_t i = 0; i < n; ++i) out[i] = 0; return; }
    const uint32_t rng_excl = high;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t m = (uint64_t)pcg_next32(r) * rng_excl;
        uint32_t leftover = (uint32_t)m;
        if (leftover < rng_excl) {
            const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
            while (leftover < threshold) { m = (uint64_t)pcg_next32(r) * rng_excl; leftover = (uint32_t)m; }
        }
        out[i] = (int64_t)(m >> 32);
    }
}

} // namespace

extern "C" int mtq_rng_create(mtq_rng **out, uint64_t seed)
{
    if (!out) return fail(MTQ_ERR_INVALID, "null argument");
    mtq_rng *r = new (std::nothrow) mtq_rng();
    if (!r) return fail(MTQ_ERR_INVALID, "out of memory");
    rng_seed(r, seed);
    *out = r;
    return MTQ_OK;
}

extern "C" int mtq_rng_permutation(mtq_rng *r, int64_t n, int64_t *out)
{
    if (!r || (!out && n > 0) || n < 0) return fail(MTQ_ERR_INVALID, "bad argument");
    rng_permutation(r, n, out);
    return MTQ_OK;
}

extern "C" int mtq_rng_integers(mtq_rng *r, int64_t high, int64_t n, int64_t *out)
{
    if (!r || (!out && n > 0) || n < 0) return fail(MTQ_ERR_INVALID, "bad argument");
    if (high < 1 || high >= 0xFFFFFFFFll) return fail(MTQ_ERR_INVALID, "high must be in [1, 2^32 - 2]");
    rng_integers(r, (uint32_t)high, n, out);
    return MTQ_OK;
}

extern "C" void mtq_rng_destroy(mtq_rng *r) { delete r; }

// The whole search of one tensor (mixed_tile_greedy.py:95-346) on host records: H1 passes in `formats`
// order, visiting order from the NumPy-compatible generator above.
extern "C" int mtq_greedy_run(const double *stats, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                              int metric, double threshold, double elem_count, uint64_t seed, int8_t *map,
                              int64_t counts[4], double out[9])
{
    if (!stats || !formats || !map) return fail(MTQ_ERR_INVALID, "null argument");
    if (n_formats <= 0 || n_formats > MTQ_NUM_TILE_FORMATS) return fail(MTQ_ERR_INVALID, "n_formats must be 1..4");
    if (seed == 0) return fail(MTQ_ERR_INVALID, "seed 0 means 'draw a random seed' in the reference; resolve it before calling");
    mtq_greedy *g = nullptr;
    if (int rc = mtq_greedy_create(&g, stats, tiles, fmt_mask, metric, threshold, elem_count, formats[0])) return rc;
    mtq_rng rng;
    rng_seed(&rng, seed);
    if (tiles > INT32_MAX) { mtq_greedy_destroy(g); return fail(MTQ_ERR_INVALID, "more than 2^31 tiles in one tensor"); }
    std::vector<int32_t> cand((size_t)tiles);   // 32-bit tile ids: the shuffled array of a 4096² tensor is 64 KB instead of 128
    int rc = MTQ_OK;
    for (int f = 0; f < n_formats && rc == MTQ_OK; ++f) {
        int64_t n = 0;
        for (int64_t t = 0; t < tiles; ++t) if (!g->fixed[(size_t)t]) cand[(size_t)n++] = (int32_t)t; // np.where(~fixed)[0], :228
        if (n == 0) break;                                                                     // :229-230
        if (f == 0 && n == tiles) {
            // The pass of the base format: every tile already has it, so each visit only asks whether the current value passes
            // (:237-241) — the same answer for all of them, whatever the order.  The generator still advances as the
            // permutation would have (the later passes' orders depend on it).
            rng_skip_shuffle(&rng, n);
            rc = greedy_pass_any<int32_t>(g, formats[0], cand.data(), 1);     // evaluates the current value once, on tile 0
            if (rc == MTQ_OK && g->fixed[0]) std::fill(g->fixed.begin(), g->fixed.end(), (uint8_t)1);
            continue;
        }
        rng_shuffle(&rng, n, cand.data());                                                    // order = rng.permutation(candidates), :231
        rc = greedy_pass_any<int32_t>(g, formats[f], cand.data(), n);
    }
    if (rc == MTQ_OK && g->w == 3 && g->degenerate)
        rc = fail(MTQ_ERR_UNSUPPORTED, "zero-variance tensor: the decision needs the full records (Σ|x−y|), not the slim ones");
    if (rc == MTQ_OK) {
        std::memcpy(map, g->assign.data(), (size_t)tiles);
        if (counts) for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) counts[f] = g->counts[f];
        if (out) {
            if (g->w == 3) for (int k = 0; k < 9; ++k) out[k] = std::nan(""); // columns of slim scans come from mtq_column_sums_device
            else rc = mtq_columns_from_stats(stats, tiles, fmt_mask, map, elem_count, out);
        }
    }
    mtq_greedy_destroy(g);
    return rc;
}

// Process-wide pool of scan threads.  The streamed driver calls mtq_greedy_run_batch once per chunk of tensors, a few
// calls in flight at a time; spawning and joining a thread per worker per call cost about as much as scanning a tensor
// (≈ 0.35 ms per call with 8–16 workers), so the threads are created once and sleep on a condition variable between
// batches.  The singleton itself is never destroyed (no destructor of ours runs at process exit); its threads are
// joined by mtq_shutdown (host_shutdown below), which the Python binding calls from an atexit hook, and started
// again by the next batch if there is one.
namespace {
class ScanPool {
public:
    static ScanPool &instance() { static ScanPool *p = new ScanPool(); return *p; }
    // Every GENERATION of workers has its own stop flag (captured by its threads): shutdown() retires the current generation and
    // joins it, and an ensure() racing with that join starts a new generation instead of clearing the flag the retiring threads are
    // about to read (one shared flag: they went back to sleep and the join never returned).
    void ensure(int n)
    {
        std::lock_guard<std::mutex> lock(mu_);
        if (!gen_) gen_ = std::make_shared<std::atomic<bool>>(false);
        auto gen = gen_;
        while ((int)threads_.size() < n) threads_.emplace_back([this, gen] { loop(gen); });
    }
    void submit(std::function<void()> fn)
    {
        { std::lock_guard<std::mutex> lock(mu_); queue_.push_back(std::move(fn)); }
        cv_.notify_one();
    }
    void shutdown()   // the queue is drained first: a batch in flight on another thread completes
    {
        std::vector<std::thread> mine;
        {
            std::lock_guard<std::mutex> lock(mu_);
            if (gen_) gen_->store(true);
            gen_.reset();                                    // the next ensure() starts a fresh generation
            mine.swap(threads_);
        }
        cv_.notify_all();
        for (auto &t : mine) if (t.joinable()) t.join();
    }
private:
    void loop(std::shared_ptr<std::atomic<bool>> stop)
    {
        for (;;) {
            std::function<void()> fn;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop->load() || !queue_.empty(); });
                if (queue_.empty()) return;              // retired and nothing left to do
                fn = std::move(queue_.front());
                queue_.pop_front();
            }
            fn();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> queue_;
    std::vector<std::thread> threads_;
    std::shared_ptr<std::atomic<bool>> gen_;
};

struct BatchState {
    std::atomic<int64_t> next{0};
    std::atomic<int> first_error{MTQ_OK};
    std::mutex mu;
    std::condition_variable cv;
    int active = 0;      // helpers inside work()
    bool closed = false; // set by the caller once every tensor has been claimed: helpers that start later return at once
    std::string message;
};
} // namespace

// `job(i)` for i in [0, count) on up to n_threads host threads (the calling thread and n_threads − 1 pool threads pull indices
// from a shared counter); the first failing job's status and message are reported.
static int run_batch(int64_t count, int n_threads, const char *what, const std::function<int(int64_t)> &job)
{
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, count));
    auto st = std::make_shared<BatchState>();
    auto work = [=]() {
        for (;;) {
            const int64_t i = st->next.fetch_add(1);
            if (i >= count) break;
            const int rc = job(i);
            if (rc != MTQ_OK) {
                int expected = MTQ_OK;
                if (st->first_error.compare_exchange_strong(expected, rc)) {
                    std::lock_guard<std::mutex> lock(st->mu);
                    st->message = mtq_last_error();
                }
            }
        }
    };
    if (nt > 1) {
        ScanPool &pool = ScanPool::instance();
        pool.ensure(nt - 1);
        for (int t = 1; t < nt; ++t)
            pool.submit([st, work]() {
                {
                    std::lock_guard<std::mutex> lock(st->mu);
                    if (st->closed) return;          // the batch finished before this helper got a thread: the buffers may be gone
                    ++st->active;
                }
                work();
                std::lock_guard<std::mutex> lock(st->mu);
                if (--st->active == 0) st->cv.notify_all();
            });
    }
    work();                                           // returns when every tensor has been claimed
    {
        std::unique_lock<std::mutex> lock(st->mu);
        st->closed = true;
        st->cv.wait(lock, [&] { return st->active == 0; });
    }
    if (st->first_error.load() != MTQ_OK) return fail(st->first_error.load(), st->message.empty() ? what : st->message.c_str());
    return MTQ_OK;
}

// mtq_greedy_run over `count` equally sized tensors on up to `n_threads` host threads (tensors are independent).
extern "C" int mtq_greedy_run_batch(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats,
                                    int n_formats, int metric, double threshold, double elem_count, const uint64_t *seeds,
                                    int8_t *maps, int64_t *counts, double *outs, int n_threads)
{
    if (!stats || !formats || !seeds || !maps) return fail(MTQ_ERR_INVALID, "null argument");
    if (count <= 0 || tiles <= 0) return fail(MTQ_ERR_INVALID, "count and tiles must be positive");
    const int rec = 2 + ((fmt_mask & MTQ_MASK_SLIM) ? 3 : 5) * popcount4(fmt_mask);
    return run_batch(count, n_threads, "mtq_greedy_run_batch failed", [=](int64_t i) {
        return mtq_greedy_run(stats + i * tiles * rec, tiles, fmt_mask, formats, n_formats, metric, threshold, elem_count, seeds[i],
                              maps + i * tiles, counts ? counts + 4 * i : nullptr, outs ? outs + 9 * i : nullptr);
    });
}

// mtq_shutdown's part of this file (mtq_error.hpp): the scan threads are joined; a later batch starts them again.
void mtq::host_shutdown() { ScanPool::instance().shutdown(); }
