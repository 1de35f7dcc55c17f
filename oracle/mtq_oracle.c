/*
 * oracle/mtq_oracle.c — CPU restatement of the reference's mixed-tile search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under quantization_analysis_amd/ may import,
 * link or call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED.  tests/golden/ holds vectors produced by importing the
 * reference in the build container (tests/golden/make_golden.py); tests/test_oracle_golden.py
 * checks every function below against them.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; contraction must stay off:
 * the reference forms float32 products and then sums them in float64).
 *
 * Citations are file:line in the reference repository (johanna-rock/quantization_analysis).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 32
#define GROUP 16
#define NFMT 4 /* bf16, bfp8, bfp4, bfp2 — compression_algorithms/tile_utils.py:8 */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* quantization_formats.py:29-45 — RNE to bf16 on the raw bits, widen back. No NaN special case. */
static inline uint32_t orc_bf16_bits(uint32_t u)
{
    uint32_t lsb = (u >> 16) & 1u;
    uint32_t r = u + (0x7FFFu + lsb); /* uint32 wrap, as numpy */
    return (r >> 16) << 16;
}

/* quantization_formats.py:71-81 — decode table entry computed on the fly (man in [1, 2^m)). */
static inline void orc_decode(uint32_t man, int m, uint32_t *shift_cnt, uint32_t *man_shifted)
{
    int msb = 31 - __builtin_clz(man);
    uint32_t s = (uint32_t)((m - 1) - msb);
    *shift_cnt = s;
    *man_shifted = (man << (s + 1)) & ((1u << m) - 1u);
}

/* quantization_formats.py:115-158 for ONE aligned group of 16 raw fp32 words. */
static void orc_bfp_group(const uint32_t in[GROUP], int m, uint32_t out[GROUP])
{
    uint32_t shared = 0;
    for (int i = 0; i < GROUP; ++i) {
        uint32_t e = (in[i] >> 23) & 0xFFu; /* :118 */
        if (e > shared) shared = e;         /* :119 */
    }
    const uint32_t shift = 24u - (uint32_t)m;       /* :133 */
    const uint32_t round_mask = (1u << shift) - 1u; /* :134 */
    const uint32_t tie = 1u << (shift - 1u);        /* :135 */
    const uint32_t qmax = (1u << m) - 1u;
    for (int i = 0; i < GROUP; ++i) {
        uint32_t u = in[i];
        uint32_t e = (u >> 23) & 0xFFu;
        uint32_t sign = u >> 31;                        /* :122 */
        uint32_t man = (1u << 23) | (u & 0x007FFFFFu);  /* :121,125 */
        uint32_t d = shared - e;                        /* :126 */
        man = (d > 31u) ? 0u : (man >> d);              /* :127-131: 24-bit value, >>31 then more ⇒ 0 */
        uint32_t rv = man & round_mask;                 /* :136 */
        man >>= shift;                                  /* :137 */
        uint32_t guard = man & 1u;                      /* :138 */
        uint32_t up = (rv > tie) || (rv == tie && guard == 1u); /* :139 */
        man += up;                                      /* :140 */
        if (man > qmax) man = qmax;                     /* :141 saturate, not renormalise */
        if (man == 0u) sign = 0u;                       /* :143 */
        if (e == 0u) { man = 0u; sign = 0u; }           /* :145 zero/denormal input → code 0 */
        uint32_t bits = 0u;
        if (man != 0u) {
            uint32_t sc, ms;
            orc_decode(man, m, &sc, &ms);               /* :150-152 */
            uint32_t exp_out = shared - sc;             /* :154 (uint32 wrap for shared < sc) */
            bits = (sign << 31) | (exp_out << 23) | (ms << (23u - (uint32_t)m)); /* :157-158 */
        } else {
            bits = sign << 31;                          /* :155: exp_out = 0, man_shifted = 0, sign = 0 */
        }
        out[i] = bits;
    }
}

static int orc_mant_bits(int fmt) { return fmt == 1 ? 7 : fmt == 2 ? 3 : 1; }

/*
 * Quantize→dequantize a row-major (rows × cols) float32 matrix, groups = 16 contiguous
 * columns aligned from column 0, a partial last group is completed with +0.0
 * (quantization_formats.py:101-107,115; the 32×32 tile only adds zero padding).
 * fmt: 0 bf16, 1 bfp8, 2 bfp4, 3 bfp2, 4 fp0 (quantization_formats.py:171-194).
 */
void orc_quantize(const float *x, int64_t rows, int64_t cols, int fmt, float *y)
{
    if (fmt == 4) { memset(y, 0, sizeof(float) * (size_t)(rows * cols)); return; } /* :167-168 */
    if (fmt == 0) {
        for (int64_t i = 0; i < rows * cols; ++i) y[i] = u2f(orc_bf16_bits(f2u(x[i])));
        return;
    }
    const int m = orc_mant_bits(fmt);
    for (int64_t r = 0; r < rows; ++r) {
        for (int64_t c0 = 0; c0 < cols; c0 += GROUP) {
            uint32_t in[GROUP], out[GROUP];
            for (int i = 0; i < GROUP; ++i)
                in[i] = (c0 + i < cols) ? f2u(x[r * cols + c0 + i]) : 0u;
            orc_bfp_group(in, m, out);
            for (int i = 0; i < GROUP && c0 + i < cols; ++i) y[r * cols + c0 + i] = u2f(out[i]);
        }
    }
}

/* numpy's max propagates NaN (tile_utils.py:56, mixed_tile_greedy.py:215). */
static inline double orc_nanmax(double m, double d) { return (d > m || d != d) ? d : m; }

/*
 * Per-tile raw sums (replaces mixed_tile_greedy.py:147-174,192-220,245-254 and
 * feeds tile_utils.py:46-57).  x is the UNPADDED row-major (rows × cols) matrix produced by
 * reshape_to_2d_with_padding's 2-D flatten (tile_utils.py:91-107); pads are +0.0 (:109-113).
 * Tile id = tr*tiles_w + tc (mixed_tile_greedy.py:89-93).
 *
 * Record layout per tile, doubles: [Σx, Σx², then for each format bit set in fmt_mask,
 * ascending bit (bf16=0,bfp8=1,bfp4=2,bfp2=3): Σy, Σy², Σxy, Σ|x−y|, max|x−y|].
 *
 * Every term is a float32 expression (x*x, y*y, x*y, |x−y| rounded to float32) summed in
 * float64, as `np.sum(x_view * y_view, dtype=np.float64)` does (mixed_tile_greedy.py:158-164).
 * SUMMATION ORDER (ours; numpy's pairwise order is not reproduced — SURVEY §7.3-3): within one
 * 16-element group, the elements within 14 binades of the group's maximum exponent ("main") sequentially,
 * the remaining ones ("tail", zeros included) sequentially, then main + tail; the four groups of a row pair (rows 2j, 2j+1; row-major) sequentially;
 * the 16 row-pair sums of a tile by a balanced binary tree over j (adjacent pairs first).
 * The HIP kernels follow the same order, so stats compare bit-for-bit.
 */
void orc_tile_stats(const float *x, int64_t rows, int64_t cols, uint32_t fmt_mask, double *stats)
{
    const int64_t th = (rows + TILE - 1) / TILE, tw = (cols + TILE - 1) / TILE;
    int fmts[NFMT], nf = 0;
    for (int f = 0; f < NFMT; ++f) if (fmt_mask & (1u << f)) fmts[nf++] = f;
    const int rec = 2 + 5 * nf;
    double (*lane)[2 + 5 * NFMT] = malloc(sizeof(double) * 64 * (2 + 5 * NFMT));
    for (int64_t tr = 0; tr < th; ++tr) for (int64_t tc = 0; tc < tw; ++tc) {
        for (int l = 0; l < 64; ++l) {
            const int64_t r = tr * TILE + (l >> 1), c0 = tc * TILE + (l & 1) * GROUP;
            uint32_t in[GROUP], out[GROUP];
            float xv[GROUP];
            for (int i = 0; i < GROUP; ++i) {
                xv[i] = (r < rows && c0 + i < cols) ? x[r * cols + c0 + i] : 0.0f;
                in[i] = f2u(xv[i]);
            }
            double *a = lane[l];
            /* main class: elements within 14 binades of the group's maximum exponent; tail class: the rest
             * (zeros and denormals included).  Each class is summed sequentially in index order, then
             * S = S_main + S_tail. */
            uint32_t E = 0;
            for (int i = 0; i < GROUP; ++i) { uint32_t e = (in[i] >> 23) & 0xFFu; if (e > E) E = e; }
            int tail[GROUP];
            for (int i = 0; i < GROUP; ++i) tail[i] = (E - ((in[i] >> 23) & 0xFFu)) > 14u;
            double sx[2] = {0.0, 0.0}, sx2[2] = {0.0, 0.0};
            for (int i = 0; i < GROUP; ++i) { float p = xv[i] * xv[i]; sx[tail[i]] += (double)xv[i]; sx2[tail[i]] += (double)p; }
            a[0] = sx[0] + sx[1]; a[1] = sx2[0] + sx2[1];
            for (int k = 0; k < nf; ++k) {
                if (fmts[k] == 0) for (int i = 0; i < GROUP; ++i) out[i] = orc_bf16_bits(in[i]);
                else orc_bfp_group(in, orc_mant_bits(fmts[k]), out);
                double sy[2] = {0.0, 0.0}, sy2[2] = {0.0, 0.0}, sxy[2] = {0.0, 0.0}, sab[2] = {0.0, 0.0}, mx = 0.0;
                for (int i = 0; i < GROUP; ++i) {
                    float yv = u2f(out[i]);
                    float p2 = yv * yv, pxy = xv[i] * yv, df = fabsf(xv[i] - yv);
                    const int c = tail[i];
                    sy[c] += (double)yv; sy2[c] += (double)p2; sxy[c] += (double)pxy; sab[c] += (double)df;
                    mx = orc_nanmax(mx, (double)df);
                }
                double *b = a + 2 + 5 * k;
                b[0] = sy[0] + sy[1]; b[1] = sy2[0] + sy2[1]; b[2] = sxy[0] + sxy[1]; b[3] = sab[0] + sab[1]; b[4] = mx;
            }
        }
        /* lanes 4j..4j+3 = the four groups of rows 2j, 2j+1: summed sequentially into lane 4j */
        for (int l = 0; l < 64; l += 4)
            for (int g = 1; g < 4; ++g) {
                double *a = lane[l], *b = lane[l + g];
                a[0] += b[0]; a[1] += b[1];
                for (int k = 0; k < nf; ++k) {
                    double *p = a + 2 + 5 * k, *q = b + 2 + 5 * k;
                    p[0] += q[0]; p[1] += q[1]; p[2] += q[2]; p[3] += q[3];
                    p[4] = orc_nanmax(p[4], q[4]);
                }
            }
        /* balanced binary tree over the 16 row pairs */
        for (int step = 4; step < 64; step <<= 1)
            for (int l = 0; l < 64; l += 2 * step) {
                double *a = lane[l], *b = lane[l + step];
                a[0] += b[0]; a[1] += b[1];
                for (int k = 0; k < nf; ++k) {
                    double *p = a + 2 + 5 * k, *q = b + 2 + 5 * k;
                    p[0] += q[0]; p[1] += q[1]; p[2] += q[2]; p[3] += q[3];
                    p[4] = orc_nanmax(p[4], q[4]);
                }
            }
        memcpy(stats + (tr * tw + tc) * rec, lane[0], sizeof(double) * (size_t)rec);
    }
    free(lane);
}

/* mixed_tile_greedy.py:176-190 — moment form of Pearson in float64 (Python float semantics). */
static double orc_pcc_value(double n, double sum_x, double sum_x2, double sy, double sy2, double sxy, double sab)
{
    if (n == 0.0) return 1.0;
    double mean_x = sum_x / n, mean_y = sy / n;
    double am2 = sum_x2 - n * mean_x * mean_x;
    double bm2 = sy2 - n * mean_y * mean_y;
    if (am2 < 0.0) am2 = 0.0;
    if (bm2 < 0.0) bm2 = 0.0;
    double denom = sqrt(am2 * bm2);
    if (denom == 0.0) return sab == 0.0 ? 1.0 : 0.0;
    return (sxy - n * mean_x * mean_y) / denom;
}

/* compression_algorithms/metrics.py:30-33 */
static inline int orc_good(double v, int metric, double thr) { return metric == 0 ? v >= thr : v <= thr; }

/*
 * Greedy scan state (mixed_tile_greedy.py:133-220 initialisation, :227-346 one pass per call).
 * metric: 0 pcc, 1 mae, 2 atol.  All per-tile arrays are caller-owned, length T.
 */
typedef struct {
    int64_t T;
    int metric;
    double threshold;
    double n; /* elem_count = float(xf.size), :134 */
    double sum_x, sum_x2, sum_y, sum_y2, sum_xy, sum_abs;
    double max_abs;
    int64_t max_abs_count;
    double *t_sy, *t_sy2, *t_sxy, *t_sab, *t_max; /* current per-tile values */
    int8_t *assign;
    uint8_t *fixed;
    int64_t counts[NFMT];
} orc_greedy_state;

/* stats_base: [T][rec] records, fmt_slot = position of base format among the mask's formats. */
void orc_greedy_init(orc_greedy_state *s, const double *stats, int rec, int base_slot, int base_idx)
{
    const int64_t T = s->T;
    s->sum_x = s->sum_x2 = s->sum_y = s->sum_y2 = s->sum_xy = s->sum_abs = 0.0;
    for (int f = 0; f < NFMT; ++f) s->counts[f] = 0;
    s->counts[base_idx] = T; /* :102-103 */
    for (int64_t t = 0; t < T; ++t) { /* tile order, :147-174 */
        const double *r = stats + t * rec, *b = r + 2 + 5 * base_slot;
        s->sum_x += r[0]; s->sum_x2 += r[1];
        s->sum_y += b[0]; s->sum_y2 += b[1]; s->sum_xy += b[2]; s->sum_abs += b[3];
        s->t_sy[t] = b[0]; s->t_sy2[t] = b[1]; s->t_sxy[t] = b[2]; s->t_sab[t] = b[3]; s->t_max[t] = b[4];
        s->assign[t] = (int8_t)base_idx; /* :99 */
        s->fixed[t] = 0;                 /* :100 */
    }
    /* :219-220 */
    double m = T > 0 ? s->t_max[0] : 0.0;
    for (int64_t t = 1; t < T; ++t) m = orc_nanmax(m, s->t_max[t]);
    int64_t c = 0;
    for (int64_t t = 0; t < T; ++t) c += (s->t_max[t] == m);
    s->max_abs = m; s->max_abs_count = c;
}

/* One `for fmt in tile_formats` iteration given `order = rng.permutation(candidates)` (:231-346). */
void orc_greedy_pass(orc_greedy_state *s, const double *stats, int rec, int slot, int fmt_idx,
                     const int64_t *order, int64_t n_order)
{
    const double thr = s->threshold, n = s->n;
    for (int64_t k = 0; k < n_order; ++k) {
        const int64_t t = order[k];
        const int prev = s->assign[t];
        const double *q = stats + t * rec + 2 + 5 * slot;
        if (s->metric == 0) {
            double cur = orc_pcc_value(n, s->sum_x, s->sum_x2, s->sum_y, s->sum_y2, s->sum_xy, s->sum_abs); /* :237 */
            if (prev == fmt_idx) { if (!orc_good(cur, 0, thr)) s->fixed[t] = 1; continue; } /* :238-241 */
            double cy = s->sum_y + (q[0] - s->t_sy[t]);     /* :259 */
            double cy2 = s->sum_y2 + (q[1] - s->t_sy2[t]);  /* :260 */
            double cxy = s->sum_xy + (q[2] - s->t_sxy[t]);  /* :261 */
            double cab = s->sum_abs + (q[3] - s->t_sab[t]); /* :262 */
            double cand = orc_pcc_value(n, s->sum_x, s->sum_x2, cy, cy2, cxy, cab);
            if (orc_good(cand, 0, thr)) { /* :264-276 */
                s->sum_y = cy; s->sum_y2 = cy2; s->sum_xy = cxy; s->sum_abs = cab;
                s->t_sy[t] = q[0]; s->t_sy2[t] = q[1]; s->t_sxy[t] = q[2]; s->t_sab[t] = q[3];
                s->counts[prev]--; s->counts[fmt_idx]++; s->assign[t] = (int8_t)fmt_idx;
            } else s->fixed[t] = 1; /* :277-278 */
        } else if (s->metric == 1) {
            double cur = n != 0.0 ? s->sum_abs / n : 0.0; /* :280 */
            if (prev == fmt_idx) { if (!orc_good(cur, 1, thr)) s->fixed[t] = 1; continue; }
            double cab = s->sum_abs + (q[3] - s->t_sab[t]); /* :293 */
            double cand = n != 0.0 ? cab / n : 0.0;
            if (orc_good(cand, 1, thr)) {
                s->sum_abs = cab; s->t_sab[t] = q[3];
                s->counts[prev]--; s->counts[fmt_idx]++; s->assign[t] = (int8_t)fmt_idx;
            } else s->fixed[t] = 1;
        } else {
            double cur = s->max_abs; /* :305 */
            if (prev == fmt_idx) { if (!orc_good(cur, 2, thr)) s->fixed[t] = 1; continue; }
            double new_max = q[4], old_max = s->t_max[t];
            double cand_max = s->max_abs; int64_t cand_count = s->max_abs_count;
            if (new_max > s->max_abs) { cand_max = new_max; cand_count = 1; }        /* :322-324 */
            else if (new_max == s->max_abs) { if (old_max != s->max_abs) cand_count = s->max_abs_count + 1; } /* :325-327 */
            else if (old_max == s->max_abs) {                                          /* :329 */
                if (s->max_abs_count > 1) cand_count = s->max_abs_count - 1;           /* :330-331 */
                else {                                                                 /* :333-336 */
                    double m = (t == 0) ? new_max : s->t_max[0];
                    for (int64_t j = 1; j < s->T; ++j) m = orc_nanmax(m, j == t ? new_max : s->t_max[j]);
                    int64_t c = 0;
                    for (int64_t j = 0; j < s->T; ++j) c += ((j == t ? new_max : s->t_max[j]) == m);
                    cand_max = m; cand_count = c;
                }
            }
            if (orc_good(cand_max, 2, thr)) { /* :337-344 */
                s->t_max[t] = new_max; s->max_abs = cand_max; s->max_abs_count = cand_count;
                s->counts[prev]--; s->counts[fmt_idx]++; s->assign[t] = (int8_t)fmt_idx;
            } else s->fixed[t] = 1;
        }
    }
}

/* Current global metric of the state (diagnostics for tests). */
double orc_greedy_value(const orc_greedy_state *s)
{
    if (s->metric == 0) return orc_pcc_value(s->n, s->sum_x, s->sum_x2, s->sum_y, s->sum_y2, s->sum_xy, s->sum_abs);
    if (s->metric == 1) return s->n != 0.0 ? s->sum_abs / s->n : 0.0;
    return s->max_abs;
}

size_t orc_greedy_state_size(void) { return sizeof(orc_greedy_state); }
