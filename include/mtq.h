/*
 * mtq.h — C ABI of libmtq_hip.so, the MI355X (gfx950) backend of the mixed-tile
 * quantization-format search path.
 *
 * The reference (johanna-rock/quantization_analysis) is pure Python and has no FFI; its seam for
 * this path is the `--backend` selector (wq:57-62) → `Quantizer.quantize` (compression_algorithms/
 * quantizer.py:13-34) and `CompressionAlgorithm.run` (compression_algorithms/base.py:36-44).  Each
 * entry point below names the reference code it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types in signatures (`stream` is a hipStream_t passed as void*,
 *     NULL = the default stream).
 *   - pointers named x / y / stats / map in the DEVICE section are device pointers owned by the caller;
 *     calls are asynchronous on `stream`.  Pointers in the HOST section are host pointers.
 *   - every function returns MTQ_OK (0) or a negative mtq_status; mtq_last_error() gives a
 *     thread-local message.  Nothing throws across the boundary.  There is NO CPU fallback: device
 *     entry points fail with MTQ_ERR_NO_DEVICE when no gfx950 device is usable.
 *   - matrices are row-major (rows × cols) with a leading dimension `ld` in ELEMENTS; this is the 2-D
 *     flatten of compression_algorithms/tile_utils.py:91-107 (a 1-D vector of n elements is passed as
 *     its zero-filled ceil(n/32) × 32 matrix, an N-D tensor as prod(shape[:-1]) × shape[-1]).
 *     32×32 tiles are numbered tile = tr * tiles_w + tc (mixed_tile_greedy.py:89-93); elements outside
 *     rows × cols read as +0.0 (tile_utils.py:109-113).
 */
#ifndef MTQ_H
#define MTQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTQ_VERSION 143 /* 0.1.4.3: + ragged batches (MtqMatrix, mtq_tile_stats_ragged, mtq_threshold_enqueue_ragged / _columns_ragged); 0.1.4.2: - chain records, + mtq_threshold_enqueue / _columns; 0.1.4.1: + partial / listed K1 (mtq_tile_stats_partial, mtq_tile_stats_listed), the search in phases with shared visiting orders, mtq_shutdown, mtq_knife_tiles_device */

typedef enum {
    MTQ_OK = 0,
    MTQ_ERR_INVALID = -1,     /* bad argument (shape, dtype, mask, alignment, null pointer) */
    MTQ_ERR_HIP = -2,         /* a HIP runtime call failed; message carries hipGetErrorString */
    MTQ_ERR_NO_DEVICE = -3,   /* no usable gfx950 device */
    MTQ_ERR_UNSUPPORTED = -4  /* recognised but not built (e.g. unknown format code) */
} mtq_status;

/* input element types */
enum { MTQ_DTYPE_BF16 = 0, MTQ_DTYPE_F32 = 1 };

/* format codes = index into MIXED_TILE_FORMATS (tile_utils.py:8) = value stored in assignment maps;
 * FP0 is quantize-only (quantization_formats.py:167-168). */
enum { MTQ_FMT_BF16 = 0, MTQ_FMT_BFP8 = 1, MTQ_FMT_BFP4 = 2, MTQ_FMT_BFP2 = 3, MTQ_FMT_FP0 = 4 };
#define MTQ_NUM_TILE_FORMATS 4
#define MTQ_MASK_ALL 0xFu
/* HOST functions only: "the records hold no bf16 slot; the bf16 candidate is the identity" — true for bf16 STORAGE, where
 * K1's bf16 slot is [Σx, Σx², Σx², |Σx|·0, |Σx|·0].  Set together with a mask WITHOUT bit 0 (records written by K1 for
 * mask & 0xE: 17 instead of 22 doubles per tile cross PCIe); format 0 is then synthesised from Σx, Σx² on the host. */
#define MTQ_MASK_BF16_IDENTITY 0x10u
/* mtq_greedy_* with metric pcc only: "slim" records [Σx, Σx², {Σy, Σy², Σxy} per format bit] — 3 instead of 5 doubles per
 * format (written by mtq_pack_slim_records from K1's records).  Σ|x−y| and max|x−y| feed no pcc decision except the
 * degenerate zero-variance case: a scan that reaches it fails with MTQ_ERR_UNSUPPORTED and the caller repeats that tensor
 * with full records; the mae / atol columns of the result come from mtq_column_sums_device on the full device records. */
#define MTQ_MASK_SLIM 0x20u

/* metrics (compression_algorithms/metrics.py:19-39) */
enum { MTQ_METRIC_PCC = 0, MTQ_METRIC_MAE = 1, MTQ_METRIC_ATOL = 2 };

/* ------------------------------------------------------------------ library */

int mtq_version(void);
const char *mtq_last_error(void);
/* Number of visible HIP devices; MTQ_ERR_NO_DEVICE if the runtime reports none. */
int mtq_device_count(int *count);
/* Releases what the library keeps between calls: joins the scan threads, drains the devices it used and frees its device tables and
 * events.  Idempotent; the library sets itself up again on the next call that needs any of it.  Nothing of this is ever done from a
 * static destructor: call it before the process tears the HIP runtime down (the Python binding registers it with atexit). */
int mtq_shutdown(void);

/* ------------------------------------------------------------------ DEVICE: kernels */

/* Doubles per tile record for a format mask: 2 + 5 * popcount(mask & 0xF); 2 + 3 * popcount under MTQ_MASK_SLIM. */
size_t mtq_stats_record_doubles(uint32_t fmt_mask);

/*
 * K1 tile_stats — fused BFP quantize + per-tile reduction; y never leaves registers.
 * Replaces, per candidate format, quantization_formats.py:84-164 (via quantizer.py:34) plus the
 * per-tile sums of mixed_tile_greedy.py:147-174,192-220,245-254,288-291,313-318 and is the input of the
 * per-tile scores of tile_utils.py:46-57.
 *
 * stats[tile][0..1] = Σx, Σx²; then for each bit set in fmt_mask (ascending: bf16, bfp8, bfp4, bfp2)
 * five doubles Σy, Σy², Σxy, Σ|x−y|, max|x−y|.  Every term is the float32 expression the reference
 * forms (x*x, y*y, x*y, |x−y|) summed in float64; summation order: inside a shared-exponent group the
 * elements within 14 binades of the shared exponent sequentially, the others (zeros included) sequentially,
 * then main + tail; the 4 groups of a row pair (rows 2j, 2j+1) sequentially, the 16 row pairs of a
 * tile by a balanced binary tree over j.
 * Sums run over the zero-padded 1024 elements (pads contribute exactly 0).
 */
int mtq_tile_stats(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld,
                   uint32_t fmt_mask, double *stats, void *stream);

/*
 * Same kernel over `count` equally shaped matrices at x + i*stride_elems (one launch);
 * stats holds count × tiles × record doubles.  Used for model-sized streams of tensors.
 */
int mtq_tile_stats_batched(const void *x, int in_dtype, int64_t count, int64_t stride_elems,
                           int64_t rows, int64_t cols, int64_t ld,
                           uint32_t fmt_mask, double *stats, void *stream);

/*
 * K1 with part of the record left for later (round 3) — same records, same layout (`layout_mask` names the slots that exist), but only
 * the five statistics of the formats in `full_mask` and Σy, Σy², Σxy of the formats in `sums_mask` are promised; every other statistic of
 * the layout is UNSPECIFIED afterwards (the LDS-staged bf16 kernel of csrc/mtq_fast.hip leaves NaN there, the other routes write the whole record).
 * Σx, Σx² are always written.  What the greedy search needs of a format before any tile has ended up in it is Σy, Σy², Σxy
 * (mixed_tile_greedy.py:245-261), and it only looks at format p for the tiles that accepted every earlier format (:227-231): the
 * streamed driver evaluates the last format of the list — and Σ|x−y|, max|x−y| of the one before it — for those tiles alone
 * (mtq_tile_stats_listed) between the search's last two passes.  full_mask and sums_mask are disjoint subsets of layout_mask.
 */
int mtq_tile_stats_partial(const void *x, int in_dtype, int64_t count, int64_t stride_elems,
                           int64_t rows, int64_t cols, int64_t ld,
                           uint32_t layout_mask, uint32_t full_mask, uint32_t sums_mask, double *stats, void *stream);

/* mtq_tile_stats_partial as two launches the caller places itself (bf16 storage in whole 32x128 units with 16-byte aligned rows;
 * MTQ_ERR_UNSUPPORTED otherwise): _begin launches the LDS-staged bf16 kernel alone — it resets its own unit counters, and stores *launch_id
 * (returned to the host) into the device word *mark if it meets a tile it cannot take; _end, given that id, recomputes those tiles by the
 * literal route (it returns at once when *mark holds another value).  The records are complete behind _end.  The streamed driver puts
 * _begin on its K1 stream, where nothing then sits between two K1 launches, and _end on the batch's search stream. */
int mtq_tile_stats_partial_begin(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                 uint32_t layout_mask, uint32_t full_mask, uint32_t sums_mask, double *stats, uint32_t *mark,
                                 uint32_t *launch_id_out, void *stream);
int mtq_tile_stats_partial_end(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                               uint32_t layout_mask, double *stats, const uint32_t *mark, uint32_t launch_id, void *stream);

/*
 * K1 for a list of tiles: what mtq_tile_stats_partial left out, for the tiles that turn out to need it.  listed[0 .. *n_listed) (device;
 * at most `capacity` entries are read) names tiles as tensor * tiles + tile; for each of them the five statistics of the formats in
 * full_mask and Σ|x−y|, max|x−y| of those in err_mask are written into the tile's record (layout `layout_mask`), bit for bit what
 * mtq_tile_stats writes there; nothing else of the record is touched.  BFP formats only.  scratch: device memory of capacity + 1 uint32
 * (bf16 storage goes through the LDS-staged exact kernel, four listed tiles per wave, and parks the few tiles that kernel cannot take
 * there for the one-wave-per-tile kernel), or NULL (every tile through the one-wave-per-tile kernel).  The list comes from phase 1 of a split search
 * (mtq_greedy_scan_device_ex), which is the only reader of these statistics before the map is final.
 */
int mtq_tile_stats_listed(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                          uint32_t layout_mask, uint32_t full_mask, uint32_t err_mask, const uint32_t *listed, const uint32_t *n_listed,
                          int64_t capacity, uint32_t *scratch, double *stats, void *stream);

/*
 * K2 quantize — materialise y = quantize→dequantize(x) for one format as float32.
 * Replaces quantize_weight_values (quantization_formats.py:171-194) behind Quantizer.quantize
 * (quantizer.py:34) for fmt in {bf16, bfp8, bfp4, bfp2, fp0}.  Bit-exact, including the
 * reference's saturating round-up, sign-of-zero, denormal→0 and uint32 wrap-around quirks.
 */
int mtq_quantize(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld,
                 int fmt, float *y, int64_t ldy, void *stream);

/*
 * K3 apply_assignment — y where each 32×32 tile uses the format its int8 map entry names.
 * Replaces the tile gather/scatter of mixed_tile_threshold.py:125-132, mixed_tile_greedy.py:273,348-352
 * and scripts/reconstruct_mixed_tile_assignment.py:82-94.  map is tiles_h × tiles_w, row-major, on device.
 */
int mtq_apply_assignment(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld,
                         const int8_t *map, float *y, int64_t ldy, void *stream);

/*
 * K5 dequant_fp8_block (loader) — float8-e4m3fn weights × float32 inverse block scales → float32: the on-load
 * dequantisation of DeepSeek-style checkpoints, `w.float() * scale_inv.repeat_interleave(block)` with
 * block = ceil(dim / scale_dim) (hf_model_utils.py:199-215, used at :273-281).  w: rows × cols bytes (ldw),
 * scale_inv: scale_rows × scale_cols float32 (contiguous), out: rows × cols float32 (ldo).  Bit-exact.
 */
int mtq_dequant_fp8_block(const void *w, const float *scale_inv, int64_t rows, int64_t cols, int64_t ldw,
                          int64_t scale_rows, int64_t scale_cols, float *out, int64_t ldo, void *stream);

/*
 * Slim copy of K1's records for the pcc greedy scan (MTQ_MASK_SLIM): out[t] = [Σx, Σx², {Σy, Σy², Σxy} per format bit of
 * fmt_mask], 2 + 3F doubles per tile, for `tiles` records back to back (any number of tensors).  Device to device; what
 * crosses PCIe afterwards is 88 instead of 136 B/tile at F = 3.
 */
int mtq_pack_slim_records(const double *stats, int64_t tiles, uint32_t fmt_mask, double *out, void *stream);

/* ------------------------------------------------------------------ HOST: decisions on stats records */

/*
 * H1 greedy scan — the sequential part of mixed_tile_greedy.py:133-346 on HOST copies of the stats.
 * The visiting order of every pass comes from the caller (numpy's Generator, :222-231), so the
 * NumPy PCG64 stream stays the caller's.
 */
typedef struct mtq_greedy mtq_greedy; /* opaque */

/* elem_count = float(xf.size) (:134).  stats/rec as written by mtq_tile_stats for fmt_mask. */
int mtq_greedy_create(mtq_greedy **out, const double *stats, int64_t tiles, uint32_t fmt_mask,
                      int metric, double threshold, double elem_count, int base_fmt);
/* One `for fmt in tile_formats` iteration (:227-346) over order[0..n). */
int mtq_greedy_pass(mtq_greedy *g, int fmt, const int64_t *order, int64_t n);
/* Copies of the scan state: assignment int8[tiles], fixed uint8[tiles], counts int64[4]. */
int mtq_greedy_assignment(const mtq_greedy *g, int8_t *assign);
int mtq_greedy_fixed(const mtq_greedy *g, uint8_t *fixed);
int mtq_greedy_counts(const mtq_greedy *g, int64_t counts[4]);
/* Current global metric value (pcc_value :176-190, mae :280, atol :305). */
int mtq_greedy_value(const mtq_greedy *g, double *value);
void mtq_greedy_destroy(mtq_greedy *g);

/*
 * NumPy-compatible generator for the visiting order: mtq_rng_create(seed) ≡ np.random.default_rng(seed),
 * successive mtq_rng_permutation(n) ≡ successive rng.permutation(n) (SeedSequence → PCG64 → Fisher–Yates with
 * masked rejection sampling, NumPy ≥ 1.17).  rng.permutation(a) == a[rng.permutation(len(a))].
 */
typedef struct mtq_rng mtq_rng; /* opaque */
int mtq_rng_create(mtq_rng **out, uint64_t seed);
int mtq_rng_permutation(mtq_rng *r, int64_t n, int64_t *out);
/* ≡ rng.integers(0, high, size=n, dtype=np.int64) for 1 <= high < 2^32 − 1 (mixed_tile_random.py:135: Lemire
 * rejection on buffered 32-bit draws; high == 1 draws nothing). */
int mtq_rng_integers(mtq_rng *r, int64_t high, int64_t n, int64_t *out);
void mtq_rng_destroy(mtq_rng *r);

/*
 * The whole greedy search of one tensor on host records (mixed_tile_greedy.py:95-346): passes in `formats`
 * order (formats[0] is the base format), candidates = np.where(~fixed)[0], order = permutation(candidates) from
 * default_rng(seed) (seed != 0).  map: int8[tiles]; counts: per MIXED_TILE_FORMATS entry; out[9] as
 * mtq_columns_from_stats.  Thread-safe; the streamed driver calls it from worker threads.
 */
int mtq_greedy_run(const double *stats, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                   int metric, double threshold, double elem_count, uint64_t seed, int8_t *map,
                   int64_t counts[4], double out[9]);

/*
 * mtq_greedy_run over `count` equally sized tensors (records contiguous: count × tiles × record doubles) on up to
 * n_threads host threads.  seeds[count]; maps int8[count][tiles]; counts int64[count][4]; outs double[count][9].
 */
int mtq_greedy_run_batch(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats,
                         int n_formats, int metric, double threshold, double elem_count, const uint64_t *seeds,
                         int8_t *maps, int64_t *counts, double *outs, int n_threads);

/*
 * Per-tile scores from the raw sums, n = 1024 (tile_utils.py:46-57 semantics on float64 moments):
 * pcc via the moment formula, mae = Σ|d|/1024, atol = max|d|.  scores is [formats][tiles], formats in ascending
 * code order: popcount(mask & 0xF) rows, plus a leading bf16 row under MTQ_MASK_BF16_IDENTITY.
 */
int mtq_tile_scores(const double *stats, int64_t tiles, uint32_t fmt_mask, int metric, double *scores);


/*
 * K4 threshold_assign — mixed_tile_threshold.py:111-123 / scripts/sweep_mixed_tile_threshold.py:145-155:
 * per tile the lowest-bytes format among `formats` whose score passes, else the highest-bytes one.
 * The reference compares float32 scores with a float32-rounded threshold (NumPy ≥ 2, NEP 50);
 * tiles with a looked-at format whose float64 score lies within `band` of float32(threshold) are listed in knife_ids
 * (capacity knife_cap), with the bit mask of those format codes in knife_near (nullable, same capacity), so the caller
 * can decide exactly those formats of those tiles with the literal float32 expression.  Formats behind the chosen one
 * were not looked at.
 */
int mtq_threshold_assign(const double *stats, int64_t tiles, uint32_t fmt_mask,
                         const int *formats, int n_formats, int metric, double threshold, double band,
                         int8_t *map, int64_t *knife_ids, uint8_t *knife_near, int64_t knife_cap, int64_t *n_knife);

/*
 * Tensor-level columns (pcc, mae, atol) of the reconstruction a map implies, from the raw sums in
 * float64 (replaces wq:683-687 for the hip backend; see DESIGN.md on the float32 column).
 * out[0..2] = pcc, mae, atol; out[3..8] = Σx, Σx², Σy, Σy², Σxy, Σ|d|.
 */
int mtq_columns_from_stats(const double *stats, int64_t tiles, uint32_t fmt_mask, const int8_t *map,
                           double elem_count, double out[9]);
/* The same columns from the seven sums themselves (Σx, Σx², Σy, Σy², Σxy, Σ|d|, max|d|), e.g. those of
 * mtq_column_sums_device copied to the host. */
int mtq_columns_from_sums(const double sums[7], double elem_count, double out[9]);

/* ------------------------------------------------------------------ DEVICE: decisions on device-resident records */

/*
 * The functions above read HOST copies of the records (the greedy scan is sequential and lives there).  The threshold
 * rule, the per-tile scores and the moments of a map are per-tile / reduction work and have DEVICE forms that read the
 * records where K1 wrote them: nothing but maps (1 B/tile), flags and a handful of doubles crosses PCIe.  Same arithmetic
 * (csrc/mtq_decide.hpp), same bits as the host forms — except the column sums, which add in a fixed tree order.
 * All pointers are device pointers; calls are asynchronous on `stream`.
 */

/* mtq_tile_scores on the device: scores[formats][tiles] (device), rows as documented for mtq_tile_scores. */
int mtq_tile_scores_device(const double *stats, int64_t tiles, uint32_t fmt_mask, int metric, double *scores, void *stream);

/* K4: mtq_threshold_assign on the device.  map: int8[tiles]; knife: uint8[tiles], bit c set where looked-at format code c
 * scores within `band` of float32(threshold) (0 for most tiles; the caller decides those formats of those tiles with the
 * literal float32 expression). */
int mtq_threshold_assign_device(const double *stats, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                                int metric, double threshold, double band, int8_t *map, uint8_t *knife, void *stream);


/* Σx, Σx², Σy, Σy², Σxy, Σ|d|, max|d| of the reconstruction `map` implies → scratch[0..6] (device), and the map's number of tiles per
 * format code 0..3 → scratch[7..10] (whole numbers as doubles: mixed_tile_threshold.py:133-135's bincount without the map leaving the
 * device first).  scratch must hold mtq_columns_scratch_doubles() doubles.  Σx is NaN when the map names a format that is not
 * available.  The caller turns the seven sums into pcc / mae / atol with the formulas of mtq_columns_from_stats. */
size_t mtq_columns_scratch_doubles(void);
int mtq_column_sums_device(const double *stats, int64_t tiles, uint32_t fmt_mask, const int8_t *map, double *scratch, void *stream);
/* The same for `count` equally sized tensors in one launch pair: records [count][tiles][rec], maps [count][tiles], scratch
 * [count][mtq_columns_scratch_doubles()] — tensor i's seven sums at scratch[i * mtq_columns_scratch_doubles()]. */
int mtq_column_sums_device_batched(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int8_t *maps,
                                   double *scratch, void *stream);

/* H1 on the device (csrc/mtq_scan.hip): mtq_greedy_run for `count` equally sized tensors whose FULL records
 * [count][tiles][2+5F] are where K1 wrote them — replaces mixed_tile_greedy.py:133-346 without moving the records: one wave64
 * per tensor performs the initial sums, NumPy's Generator.permutation of every pass (SeedSequence → PCG64, bit-compatible)
 * and the sequential accept / reject scan with the host scan's IEEE operations in the host scan's order, so maps[count][tiles]
 * (device, int8 codes) are the host's maps.  status[count] (device): 0 = done; 1 = a zero denominator turned up (the decision
 * needs Σ|x−y|: run mtq_greedy_run on that tensor's records); 2 = internal budget exhausted (same remedy).  Serves the pcc
 * metric, the mae metric (one running sum, Σ|x−y|: mixed_tile_greedy.py:280-301) and the atol metric (:305-344 — order-independent for
 * finite maxima, a per-tile walk down the format list: csrc/mtq_scan.hip greedy_atol; status 1 when a maximum is NaN), distinct
 * formats (fmt_mask may carry MTQ_MASK_BF16_IDENTITY), tiles <= MTQ_SCAN_DEVICE_MAX_TILES; anything else (repeated formats, more
 * tiles) returns MTQ_ERR_UNSUPPORTED and the caller uses the host scan.  seeds[count]: device array of non-zero seeds.  counts
 * [count][4] (device, may be NULL): tiles per format code of every finished map (np.bincount of the map).  scratch:
 * device memory of mtq_greedy_scan_scratch_bytes(count, tiles) bytes.  Asynchronous on `stream`. */
#define MTQ_SCAN_DEVICE_MAX_TILES (1 << 22)
size_t mtq_greedy_scan_scratch_bytes(int64_t count, int64_t tiles);
int mtq_greedy_scan_device(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                           int metric, double threshold, double elem_count, const uint64_t *seeds, int8_t *maps, int32_t *status,
                           int32_t *counts, void *scratch, size_t scratch_bytes, void *stream);

/* The same search with what round 3 added (mtq_greedy_scan_device is this with orders = NULL, phase = 0):
 *   orders  — NULL, or the buffer mtq_scan_orders_device filled for the seed that ALL `count` tensors share (seeds[] must hold that
 *             seed): every tensor of a model run is searched with one seed (mixed_tile_greedy.py:222-226), and a pass's permutation
 *             depends only on the generator state and the number of candidates, so the base pass's draws and the permutations of
 *             range(tiles) of passes 1 and 2 are computed once per launch; a tensor whose pass 1 rejects a tile shuffles its own
 *             candidates from the shared state, as before.  With orders the kernel runs two waves per tensor: the second gathers the
 *             passes' deltas ahead of the first.
 *   phase   — 0: the whole search.  1: every pass but the last, then the last pass's candidates (np.where(~fixed), :228) of every
 *             tensor are appended to listed[] as tensor * tiles + tile (n_listed: device counter, zeroed by the caller) and the
 *             search's state goes to carry (mtq_scan_carry_bytes(count) bytes); maps hold work-in-progress codes.  2: the last pass
 *             from carry, then the finished maps, status and counts.  Between 1 and 2 the caller fills in the statistics of the
 *             last format for the listed tiles (mtq_tile_stats_listed) — the search never reads them for any other tile.  Phases need
 *             n_formats >= 2 and the pcc or mae metric.
 */
size_t mtq_scan_carry_bytes(int64_t count);
int mtq_greedy_scan_device_ex(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                              int metric, double threshold, double elem_count, const uint64_t *seeds, int8_t *maps, int32_t *status,
                              int32_t *counts, void *scratch, size_t scratch_bytes, const void *orders, int phase, uint32_t *listed,
                              uint32_t *n_listed, void *carry, void *stream);
/* The visiting orders a launch's tensors share: the generator (SeedSequence(seed) → PCG64) after the base pass's draws, then
 * Generator.permutation(tiles) for pass 1 and — n_orders == 2 — once more for pass 2, with the generator states in between, into
 * `orders` (mtq_scan_orders_bytes(tiles) bytes, device).  Depends on nothing but seed and tiles: it can run beside K1. */
size_t mtq_scan_orders_bytes(int64_t tiles);
int mtq_scan_orders_device(uint64_t seed, int64_t tiles, int n_orders, void *orders, size_t orders_bytes, void *stream);

/* The knife-edge tiles of the threshold rule, prepared on the device for the host's literal float32 score (replaces the gather of
 * tiles and the per-format Quantizer.quantize calls of mixed_tile_threshold.py:97-110 for the tiles whose float64 score fell inside
 * the noise band — `near` is the mask array mtq_threshold_assign_device wrote, one int8 per tile of `count` equally shaped tensors):
 * list[0 .. cap) receives the flat ids (tensor * tiles + tile) of flagged tiles in no particular order, list[cap] their total number
 * (which may exceed cap: the caller then handles the batch another way); tiles_out[(p * cap + slot) * 1024 + r * 32 + c], p = 0 the
 * tile's own values as float32 (pads of ragged edge tiles +0.0, tile_utils.py:109-113), p = 1 + i the reconstruction in formats[i]
 * (codes 0..3, n_formats <= 4), for slot < min(list[cap], cap).  Asynchronous on `stream`; list and tiles_out are device memory
 * (tiles_out 16-byte aligned). */
int mtq_knife_tiles_device(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                           const int8_t *near, const int *formats, int n_formats, int64_t cap, int64_t *list, float *tiles_out, void *stream);

/* One batch of the streamed threshold driver as ONE call (round 4: behind Python every launch costs the driver 10–20 us, and a model's
 * small tensors — DeepSeek-R1 layer 0: seven tensors, 183 k tiles, 0.45 ms of K1 — were launch-bound at a dozen calls per batch):
 * mtq_tile_stats_batched(k1_mask) → mtq_threshold_assign_device(dec_mask: k1_mask, or k1_mask | MTQ_MASK_BF16_IDENTITY) into
 * both_dev[0 .. T) (maps) and both_dev[T .. 2T) (knife-edge masks), T = count * tiles → both of them into the pinned mirror both_host
 * (mtq_device_copy_2d) on `stream`; then, behind an event, on `side_stream` (NULL: on `stream`): mtq_knife_tiles_device(cap) and the list
 * (cap + 1 int64) into the pinned list_host.  Replaces the per-tensor body of wq:655-706 / mixed_tile_threshold.py:97-123 up to the
 * knife-edge decisions.  scratch and sums_host both non-NULL: mtq_threshold_columns under the maps as K4 left them follows on `stream`
 * at once — the batch's final sums unless its list names a knife-edge tile (the band is 2e-6 wide: rarely), in which case the caller
 * patches the maps and calls mtq_threshold_columns again; the host's look at the list is then off the GPU's critical path.
 * Everything asynchronous; the caller waits for an event of its own behind the call. */
int mtq_threshold_enqueue(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                          uint32_t k1_mask, uint32_t dec_mask, const int *formats, int n_formats, int metric, double threshold, double band,
                          double *stats, int8_t *both_dev, int8_t *both_host, int64_t cap, int64_t *list_dev, float *knife_dev,
                          int64_t *list_host, double *scratch, double *sums_host, void *stream, void *side_stream);
/* … and its second half: mtq_column_sums_device_batched under the (patched) maps, the seven sums of every tensor and behind them the
 * map's tile count per format code 0..3 (mixed_tile_threshold.py:133-135's bincount, as doubles) into the pinned sums_host[count][11]
 * (wq:683-706's columns come from the sums: mtq_columns_from_sums). */
int mtq_threshold_columns(const double *stats, int64_t count, int64_t tiles, uint32_t dec_mask, const int8_t *maps_dev, double *scratch,
                          double *sums_host, void *stream);

/* RAGGED batches (round 4): n <= MTQ_RAGGED_MAX matrices of ONE storage type and ANY shapes as one launch per stage — a model's odd
 * tensors (DeepSeek-R1 layer 0's five float32 projections of five shapes, its two norm vectors as (ceil(n/32), 32) matrices) cost a
 * launch chain each through the calls above, and the chain, not the arithmetic, was their time.  The batch's tiles are numbered
 * through: matrix j's tiles, row-major (tile_utils.py:96-113), follow matrix j-1's; records, maps, masks and list ids use that number.
 * Per tile the kernels, the arithmetic and the summation order are those of the per-matrix calls (mtq_tile_stats on float32 or on bf16
 * storage the LDS-staged kernel does not take: the direct kernel + its fix-up): same records, bit for bit; the column sums of matrix j
 * are formed in the order mtq_column_sums_device uses for that matrix alone.  Replaces the same reference lines as the calls they
 * generalise (wq:655-706, mixed_tile_threshold.py:97-123). */
#define MTQ_RAGGED_MAX 24
typedef struct MtqMatrix { const void *x; int64_t rows, cols, ld; } MtqMatrix;   /* device pointer, leading dimension in elements */
int mtq_tile_stats_ragged(const MtqMatrix *mats, int n, int in_dtype, uint32_t fmt_mask, double *stats, void *stream);
int mtq_knife_tiles_ragged(const MtqMatrix *mats, int n, int in_dtype, const int8_t *near, const int *formats, int n_formats, int64_t cap,
                           int64_t *list, float *tiles_out, void *stream);
int mtq_column_sums_device_ragged(const double *stats, const int64_t *tiles_per, int n, uint32_t fmt_mask, const int8_t *maps, double *scratch,
                                  void *stream);   /* scratch: n * mtq_columns_scratch_doubles(); tensor j's sums at scratch[j * that] */
int mtq_threshold_enqueue_ragged(const MtqMatrix *mats, int n, int in_dtype, uint32_t k1_mask, uint32_t dec_mask, const int *formats, int n_formats,
                                 int metric, double threshold, double band, double *stats, int8_t *both_dev, int8_t *both_host, int64_t cap,
                                 int64_t *list_dev, float *knife_dev, int64_t *list_host, double *scratch, double *sums_host, void *stream,
                                 void *side_stream);
int mtq_threshold_columns_ragged(const double *stats, const int64_t *tiles_per, int n, uint32_t dec_mask, const int8_t *maps_dev, double *scratch,
                                 double *sums_host, void *stream);

/* Results home without a copy engine (no reference counterpart: the reference's arrays are host arrays).  A kernel copies `rows` rows
 * of `width_bytes` bytes from src (pitch src_pitch) to dst (pitch dst_pitch) on `stream`; dst may be pinned host memory
 * (hipHostMalloc / torch pin_memory: mapped into the device's address space), in which case the stores cross PCIe from the kernel
 * and the data are on the host when an event recorded behind the call has completed.  The streamed driver brings maps, counts and
 * column sums back this way: an asynchronous device-to-host memcpy of a few MB was seen to hold the calling thread until the stream
 * had drained.  Pointers device-accessible, pitches >= width_bytes, rows >= 1. */
int mtq_device_copy_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream);

/* Diagnostics (no reference counterpart).  K1's persistent waves claim their units from device counters that come from a
 * per-device ring of slots (csrc/mtq_slot_ring.hpp): a slot is handed out again only behind the event recorded after its
 * previous launch's reset, so any number of launches may be pending on any streams.  This runs that bookkeeping against
 * mock event operations on the host (no GPU): 0 = every property holds, else the number of the first failed check. */
int mtq_selftest_slot_ring(void);
/* Shader-clock ticks the first tensor of the last mtq_greedy_scan_device launch spent per phase (tools/scan_device_bench.py):
 * [0] start-up + initial sums, [1] base pass + draw-only shuffle, then for pass p = 1..3: [2+3(p-1)] candidates + shuffle,
 * [3+3(p-1)] deltas, [4+3(p-1)] visits.  Synchronises the device. */
int mtq_debug_scan_ticks(uint64_t out[16]);

#ifdef __cplusplus
}
#endif
#endif /* MTQ_H */
