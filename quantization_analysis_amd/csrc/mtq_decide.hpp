// mtq_decide.hpp — the arithmetic of the decisions taken on stats records (per-tile scores, the threshold rule, the
// moment form of Pearson's r), written once for the host functions of mtq_host.cpp and the device kernels of
// mtq_decide.hip.  Every operation is an individually rounded IEEE double operation in the order the reference writes it
// (-ffp-contract=off on both sides; gfx950's double division and square root are the compiler's correctly rounded
// expansions), so a decision is the same bit pattern wherever it is evaluated — tests/test_hip_kernels.py compares the
// two sides bit for bit.
#pragma once
#include <stdint.h>

#include "../../include/mtq.h"

#if defined(__HIP__)
#define MTQ_HD __host__ __device__
#else
#define MTQ_HD
#endif

namespace mtq {

MTQ_HD inline int popcount4(uint32_t m) { return __builtin_popcount(m & MTQ_MASK_ALL); }

constexpr int kVirtualSlot = -2; // bf16 as the identity (MTQ_MASK_BF16_IDENTITY): no record slot, sums come from Σx, Σx²

// slot of format f among the set bits of mask, kVirtualSlot for the identity bf16, or -1 when f is not available
MTQ_HD inline int slot_of(uint32_t mask, int f)
{
    if (f < 0 || f >= MTQ_NUM_TILE_FORMATS) return -1;
    if (f == 0 && (mask & MTQ_MASK_BF16_IDENTITY) && !(mask & 1u)) return kVirtualSlot;
    if (!(mask & (1u << f))) return -1;
    return __builtin_popcount(mask & ((1u << f) - 1u));
}
MTQ_HD inline bool slot_ok(int slot) { return slot >= 0 || slot == kVirtualSlot; }

// The five sums (Σy, Σy², Σxy, Σ|d|, max|d|) of one format of record r by value; for the identity bf16 the values K1
// writes for bf16 storage (y == x): [Σx, Σx², Σx², z, z] with z = |Σx|·0 (0, or NaN when Σx is not finite).
struct Sums5 { double y, y2, xy, ab, mx; };
MTQ_HD inline Sums5 load5(const double *r, int slot)
{
    if (slot >= 0) { const double *b = r + 2 + 5 * slot; return {b[0], b[1], b[2], b[3], b[4]}; }
    const double z = __builtin_fabs(r[0]) * 0.0;
    return {r[0], r[1], r[1], z, z};
}

MTQ_HD inline double nanmax(double m, double d) { return (d > m || d != d) ? d : m; }

// metrics.py:30-33
MTQ_HD inline bool is_good(double v, int metric, double thr) { return metric == MTQ_METRIC_PCC ? v >= thr : v <= thr; }

// mixed_tile_greedy.py:176-190; every operation individually rounded, evaluation order as written there.
MTQ_HD inline double pcc_from_moments(double n, double sx, double sx2, double sy, double sy2, double sxy, double sab)
{
    if (n == 0.0) return 1.0;
    const double mean_x = sx / n;
    const double mean_y = sy / n;
    double am2 = sx2 - n * mean_x * mean_x;
    double bm2 = sy2 - n * mean_y * mean_y;
    if (am2 < 0.0) am2 = 0.0;
    if (bm2 < 0.0) bm2 = 0.0;
    const double denom = __builtin_sqrt(am2 * bm2);
    if (denom == 0.0) return sab == 0.0 ? 1.0 : 0.0;
    return (sxy - n * mean_x * mean_y) / denom;
}

// Per-tile score of one record slot, n = 1024 (tile_utils.py:46-57 on the raw sums).
MTQ_HD inline double tile_score(const double *r, int slot, int metric)
{
    const Sums5 b = load5(r, slot);
    if (metric == MTQ_METRIC_MAE) return b.ab / 1024.0;
    if (metric == MTQ_METRIC_ATOL) return b.mx;
    return pcc_from_moments(1024.0, r[0], r[1], b.y, b.y2, b.xy, b.ab);
}

// mixed_tile_threshold.py:111-123 for one tile: formats in ascending bytes (order[], their record slots in slots[]), the
// first whose score passes, else `best` (the highest-bytes one).  The reference compares float32 scores with a
// float32-rounded threshold (NumPy >= 2, NEP 50): thr32 = (double)(float)threshold.  `near` gets bit c for every looked-at
// format code c whose score lies within `band`·max(1, |thr32|) of thr32 (its decision is inside the float32 noise band — the
// reference's float32 mean / Pearson carry a RELATIVE error, so for mae thresholds far above 1 the band scales with the
// threshold: the caller decides exactly those with the literal float32 expression; formats behind the chosen one were not
// looked at).
struct ThresholdPlan { int order[MTQ_NUM_TILE_FORMATS], slots[MTQ_NUM_TILE_FORMATS], n, best; };
MTQ_HD inline int threshold_decide(const double *r, const ThresholdPlan &p, int metric, double thr32, double band, unsigned &near)
{
    int chosen = p.best;
    near = 0u;
    const double a32 = __builtin_fabs(thr32), width = band * (a32 > 1.0 ? a32 : 1.0);
    for (int i = 0; i < p.n; ++i) {
        const double s = tile_score(r, p.slots[i], metric);
        if (__builtin_fabs(s - thr32) <= width) near |= 1u << p.order[i];
        if (is_good((double)(float)s, metric, thr32)) { chosen = p.order[i]; break; }
    }
    return chosen;
}

// Host: sort the requested formats by bytes per element (stable, tile_utils.py:9-14 / mixed_tile_threshold.py:112-115).
// Returns false when a format is not available under fmt_mask.
inline bool plan_threshold(uint32_t fmt_mask, const int *formats, int n_formats, ThresholdPlan &p)
{
    static const double bytes_per_elem[MTQ_NUM_TILE_FORMATS] = {2.0, 1.088, 0.50097, 0.25097};
    p.n = n_formats;
    for (int i = 0; i < n_formats; ++i) {
        if (!slot_ok(slot_of(fmt_mask, formats[i]))) return false;
        p.order[i] = formats[i];
    }
    for (int i = 1; i < n_formats; ++i)
        for (int j = i; j > 0 && bytes_per_elem[p.order[j]] < bytes_per_elem[p.order[j - 1]]; --j) { const int t = p.order[j]; p.order[j] = p.order[j - 1]; p.order[j - 1] = t; }
    p.best = p.order[0];                                   // best_precision = FIRST maximum in that order (:115)
    for (int i = 1; i < n_formats; ++i) if (bytes_per_elem[p.order[i]] > bytes_per_elem[p.best]) p.best = p.order[i];
    for (int i = 0; i < n_formats; ++i) p.slots[i] = slot_of(fmt_mask, p.order[i]);
    for (int i = n_formats; i < MTQ_NUM_TILE_FORMATS; ++i) { p.order[i] = p.best; p.slots[i] = -1; }
    return true;
}

} // namespace mtq
