// mtq_decide.hip — K4 and the column sums on the DEVICE: decisions of the threshold algorithm, per-tile scores (sweep)
// and the tensor-level moments of a reconstruction (columns) taken directly from device-resident stats records, so that
// these algorithms never move the records (176 B/tile) over PCIe: a map is 1 B/tile, scores are 8 B/tile/format, the
// moments of a map are 7 doubles.  The arithmetic is mtq_decide.hpp, shared with the host functions: same IEEE
// operations, same bits (the column sums are the exception: a fixed tree order instead of the host's tile order —
// deterministic, equal to the host's to ~1e-16 relative).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mtq.h"
#include "mtq_decide.hpp"
#include "mtq_error.hpp"

namespace mtq {

struct SlotTable { int slot[MTQ_NUM_TILE_FORMATS]; };   // record slot per format code (−1 absent, kVirtualSlot identity bf16)

__global__ __launch_bounds__(256) void tile_scores_dev(const double *__restrict__ stats, int64_t tiles, int rec, SlotTable st, int metric,
                                                       double *__restrict__ scores)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    const double *r = stats + t * rec;
    int64_t row = 0;
#pragma unroll
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) {
        if (!slot_ok(st.slot[f])) continue;                 // launch-uniform
        scores[row * tiles + t] = tile_score(r, st.slot[f], metric);
        ++row;
    }
}

__global__ __launch_bounds__(256) void threshold_assign_dev(const double *__restrict__ stats, int64_t tiles, int rec, ThresholdPlan plan,
                                                            int metric, double thr32, double band, int8_t *__restrict__ map,
                                                            uint8_t *__restrict__ knife)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tiles) return;
    unsigned near;
    map[t] = (int8_t)threshold_decide(stats + t * rec, plan, metric, thr32, band, near);
    knife[t] = (uint8_t)near;
}

constexpr int kColBlocks = 256, kColVals = 11, kColMax = 6;   // Σx, Σx², Σy, Σy², Σxy, Σ|d|, max|d|, then the map's tile count per format code 0..3

// Stage 1: block b sums tiles b*256 + tid, + 256*blocks, … per thread in that order, then a fixed LDS tree over the 256
// threads.  Stage 2 (one block): the same tree over the block partials.  A map entry naming an unavailable format
// poisons the sums with NaN (the host wrapper reports it).
// blockIdx.y = tensor of a batch (records, maps and scratch areas of equal size back to back).
// A ragged batch (seg.n > 0): tensor j's records and map entries start at seg.first[j], and it is summed by the blocks a launch of its own
// would have had (min(ceil(tiles_j / 256), kColBlocks): the summation order — and so the bits — of mtq_column_sums_device on that tensor alone).
constexpr int kColSegs = 24;   // = mtq_device.hpp kRaggedMax
struct ColSegs { uint32_t n; uint32_t first[kColSegs + 1]; };

__global__ __launch_bounds__(256) void columns_partial_dev(const double *__restrict__ stats, int64_t tiles, int rec, SlotTable st,
                                                           const int8_t *__restrict__ map, double *__restrict__ partial, int64_t scratch_stride,
                                                           const ColSegs seg)
{
    __shared__ double red[kColVals][256];
    int64_t stride_blocks = gridDim.x;
    if (seg.n) {
        const int64_t first = seg.first[blockIdx.y];
        tiles = (int64_t)seg.first[blockIdx.y + 1] - first;
        stride_blocks = (tiles + 255) / 256 < kColBlocks ? (tiles + 255) / 256 : kColBlocks;
        if ((int64_t)blockIdx.x >= stride_blocks) return;    // block-uniform
        stats += first * rec;
        map += first;
    } else {
        stats += (int64_t)blockIdx.y * tiles * rec;
        map += (int64_t)blockIdx.y * tiles;
    }
    partial += (int64_t)blockIdx.y * scratch_stride;
    double v[kColVals] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < tiles; t += stride_blocks * 256) {
        const double *r = stats + t * rec;
        const int f = map[t];
        const int slot = (f >= 0 && f < MTQ_NUM_TILE_FORMATS) ? st.slot[f] : -1;
        if (!slot_ok(slot)) { v[0] = __builtin_nan(""); continue; }
        const Sums5 b = load5(r, slot);
        v[0] += r[0]; v[1] += r[1]; v[2] += b.y; v[3] += b.y2; v[4] += b.xy; v[5] += b.ab;
        v[kColMax] = nanmax(v[kColMax], b.mx);
#pragma unroll
        for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) v[7 + c] += f == c ? 1.0 : 0.0;   // integers below 2^53: exact in any order
    }
#pragma unroll
    for (int k = 0; k < kColVals; ++k) red[k][threadIdx.x] = v[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < kColVals; ++k)
                if (k != kColMax) red[k][threadIdx.x] = red[k][threadIdx.x] + red[k][threadIdx.x + s];
            red[kColMax][threadIdx.x] = nanmax(red[kColMax][threadIdx.x], red[kColMax][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < kColVals) partial[(int64_t)blockIdx.x * kColVals + threadIdx.x] = red[threadIdx.x][0];
}

__global__ __launch_bounds__(256) void columns_final_dev(const double *__restrict__ partial, int blocks, double *__restrict__ out,
                                                         int64_t scratch_stride, const ColSegs seg)
{
    __shared__ double red[kColVals][256];
    if (seg.n) {
        const int64_t tiles = (int64_t)seg.first[blockIdx.y + 1] - seg.first[blockIdx.y];
        blocks = (int)((tiles + 255) / 256 < kColBlocks ? (tiles + 255) / 256 : kColBlocks);
    }
    partial += (int64_t)blockIdx.y * scratch_stride;
    out += (int64_t)blockIdx.y * scratch_stride;
#pragma unroll
    for (int k = 0; k < kColVals; ++k) red[k][threadIdx.x] = (int)threadIdx.x < blocks ? partial[(int64_t)threadIdx.x * kColVals + k] : 0.0;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < kColVals; ++k)
                if (k != kColMax) red[k][threadIdx.x] = red[k][threadIdx.x] + red[k][threadIdx.x + s];
            red[kColMax][threadIdx.x] = nanmax(red[kColMax][threadIdx.x], red[kColMax][threadIdx.x + s]);
        }
        __syncthreads();
    }
    if (threadIdx.x < kColVals) out[threadIdx.x] = red[threadIdx.x][0];
}

static SlotTable slot_table(uint32_t fmt_mask)
{
    SlotTable st;
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) st.slot[f] = slot_of(fmt_mask, f);
    return st;
}

} // namespace mtq

using namespace mtq;

extern "C" int mtq_tile_scores_device(const double *stats, int64_t tiles, uint32_t fmt_mask, int metric, double *scores, void *stream)
{
    if (!stats || !scores) return fail(MTQ_ERR_INVALID, "null argument");
    if (tiles <= 0 || tiles > ((int64_t)1 << 38)) return fail(MTQ_ERR_INVALID, "tiles out of range");
    if (metric < MTQ_METRIC_PCC || metric > MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    if (int rc = require_device()) return rc;
    const int rec = 2 + 5 * popcount4(fmt_mask);
    hipLaunchKernelGGL(tile_scores_dev, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stats, tiles, rec,
                       slot_table(fmt_mask), metric, scores);
    return check_launch("mtq_tile_scores_device");
}

extern "C" int mtq_threshold_assign_device(const double *stats, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                                           int metric, double threshold, double band, int8_t *map, uint8_t *knife, void *stream)
{
    if (!stats || !formats || !map || !knife) return fail(MTQ_ERR_INVALID, "null argument");
    if (tiles <= 0 || tiles > ((int64_t)1 << 38)) return fail(MTQ_ERR_INVALID, "tiles out of range");
    if (n_formats <= 0 || n_formats > MTQ_NUM_TILE_FORMATS) return fail(MTQ_ERR_INVALID, "n_formats must be 1..4");
    if (metric < MTQ_METRIC_PCC || metric > MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    ThresholdPlan plan;
    if (!plan_threshold(fmt_mask, formats, n_formats, plan)) return fail(MTQ_ERR_INVALID, "a requested format is not in fmt_mask");
    if (int rc = require_device()) return rc;
    const int rec = 2 + 5 * popcount4(fmt_mask);
    hipLaunchKernelGGL(threshold_assign_dev, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stats, tiles,
                       rec, plan, metric, (double)(float)threshold, band, map, knife);
    return check_launch("mtq_threshold_assign_device");
}

extern "C" size_t mtq_columns_scratch_doubles(void) { return (size_t)kColVals * (kColBlocks + 1); }

extern "C" int mtq_column_sums_device_batched(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int8_t *maps,
                                              double *scratch, void *stream)
{
    if (!stats || !maps || !scratch) return fail(MTQ_ERR_INVALID, "null argument");
    if (tiles <= 0 || tiles > ((int64_t)1 << 38) || count <= 0 || count > 65535) return fail(MTQ_ERR_INVALID, "tiles / count out of range");
    if (int rc = require_device()) return rc;
    const int rec = 2 + 5 * popcount4(fmt_mask);
    const int blocks = (int)((tiles + 255) / 256 < kColBlocks ? (tiles + 255) / 256 : kColBlocks);
    const int64_t stride = (int64_t)kColVals * (kColBlocks + 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static const ColSegs uniform{};
    hipLaunchKernelGGL(columns_partial_dev, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, st, stats, tiles, rec, slot_table(fmt_mask), maps,
                       scratch + kColVals, stride, uniform);
    hipLaunchKernelGGL(columns_final_dev, dim3(1, (unsigned)count), dim3(256), 0, st, scratch + kColVals, blocks, scratch, stride, uniform);
    return check_launch("mtq_column_sums_device");
}

// Column sums of a ragged batch (include/mtq.h): tensor j's tiles_per[j] records and map entries follow tensor j-1's.
extern "C" int mtq_column_sums_device_ragged(const double *stats, const int64_t *tiles_per, int n, uint32_t fmt_mask, const int8_t *maps, double *scratch,
                                             void *stream)
{
    if (!stats || !tiles_per || !maps || !scratch) return fail(MTQ_ERR_INVALID, "null argument");
    if (n <= 0 || n > kColSegs) return fail(MTQ_ERR_INVALID, "a ragged batch holds 1..MTQ_RAGGED_MAX tensors");
    ColSegs seg{};
    int64_t first = 0, widest = 0;
    for (int j = 0; j < n; ++j) {
        if (tiles_per[j] <= 0 || first + tiles_per[j] >= ((int64_t)1 << 31)) return fail(MTQ_ERR_INVALID, "tiles out of range");
        seg.first[j] = (uint32_t)first;
        first += tiles_per[j];
        widest = tiles_per[j] > widest ? tiles_per[j] : widest;
    }
    seg.first[n] = (uint32_t)first;
    seg.n = (uint32_t)n;
    if (int rc = require_device()) return rc;
    const int rec = 2 + 5 * popcount4(fmt_mask);
    const int blocks = (int)((widest + 255) / 256 < kColBlocks ? (widest + 255) / 256 : kColBlocks);
    const int64_t stride = (int64_t)kColVals * (kColBlocks + 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(columns_partial_dev, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, st, stats, 0, rec, slot_table(fmt_mask), maps, scratch + kColVals,
                       stride, seg);
    hipLaunchKernelGGL(columns_final_dev, dim3(1, (unsigned)n), dim3(256), 0, st, scratch + kColVals, blocks, scratch, stride, seg);
    return check_launch("mtq_column_sums_device_ragged");
}

extern "C" int mtq_column_sums_device(const double *stats, int64_t tiles, uint32_t fmt_mask, const int8_t *map, double *scratch, void *stream)
{
    return mtq_column_sums_device_batched(stats, 1, tiles, fmt_mask, map, scratch, stream);
}

// One batch of the streamed threshold driver as one call (include/mtq.h): the launches ThresholdPipeline.enqueue issued one by one.
extern "C" int mtq_threshold_enqueue(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                     uint32_t k1_mask, uint32_t dec_mask, const int *formats, int n_formats, int metric, double threshold, double band,
                                     double *stats, int8_t *both_dev, int8_t *both_host, int64_t cap, int64_t *list_dev, float *knife_dev,
                                     int64_t *list_host, double *scratch, double *sums_host, void *stream, void *side_stream)
{
    if (!x || !formats || !stats || !both_dev || !both_host || !list_dev || !list_host) return fail(MTQ_ERR_INVALID, "null argument");
    if (count <= 0 || rows <= 0 || cols <= 0 || cap < 0) return fail(MTQ_ERR_INVALID, "count, rows, cols must be positive and cap non-negative");
    const int64_t tiles = ((rows + 31) / 32) * ((cols + 31) / 32), T = count * tiles;
    if (int rc = mtq_tile_stats_batched(x, in_dtype, count, stride_elems, rows, cols, ld, k1_mask, stats, stream)) return rc;
    if (int rc = mtq_threshold_assign_device(stats, T, dec_mask, formats, n_formats, metric, threshold, band, both_dev,
                                             reinterpret_cast<uint8_t *>(both_dev + T), stream)) return rc;
    if (int rc = mtq_device_copy_2d(both_host, (size_t)(2 * T), both_dev, (size_t)(2 * T), (size_t)(2 * T), 1, stream)) return rc;
    hipStream_t main_s = static_cast<hipStream_t>(stream), side = side_stream ? static_cast<hipStream_t>(side_stream) : main_s;
    if (side != main_s) {   // the listing waits for the masks; the event's resources go when it has completed
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(MTQ_ERR_HIP, "hipEventCreate failed");
        const bool ok = hipEventRecord(ev, main_s) == hipSuccess && hipStreamWaitEvent(side, ev, 0) == hipSuccess;
        (void)hipEventDestroy(ev);
        if (!ok) return fail(MTQ_ERR_HIP, "could not order the side stream behind the masks");
    }
    if (int rc = mtq_knife_tiles_device(x, in_dtype, count, stride_elems, rows, cols, ld, both_dev + T, formats, n_formats, cap, list_dev,
                                        cap ? knife_dev : nullptr, side)) return rc;
    if (int rc = mtq_device_copy_2d(list_host, (size_t)(cap + 1) * 8, list_dev, (size_t)(cap + 1) * 8, (size_t)(cap + 1) * 8, 1, side)) return rc;
    // the column sums under the maps as K4 left them: final unless the list names a knife-edge tile (the caller then patches and sums again)
    return scratch && sums_host ? mtq_threshold_columns(stats, count, tiles, dec_mask, both_dev, scratch, sums_host, stream) : MTQ_OK;
}

extern "C" int mtq_threshold_columns(const double *stats, int64_t count, int64_t tiles, uint32_t dec_mask, const int8_t *maps_dev, double *scratch,
                                     double *sums_host, void *stream)
{
    if (!sums_host) return fail(MTQ_ERR_INVALID, "null argument");
    if (int rc = mtq_column_sums_device_batched(stats, count, tiles, dec_mask, maps_dev, scratch, stream)) return rc;
    const size_t pitch = mtq_columns_scratch_doubles() * sizeof(double);
    return mtq_device_copy_2d(sums_host, kColVals * sizeof(double), scratch, pitch, kColVals * sizeof(double), (size_t)count, stream);
}

// The same two calls for a ragged batch (include/mtq.h): n matrices of one storage type and any shapes, their tiles numbered through.
extern "C" int mtq_threshold_enqueue_ragged(const MtqMatrix *mats, int n, int in_dtype, uint32_t k1_mask, uint32_t dec_mask, const int *formats, int n_formats,
                                            int metric, double threshold, double band, double *stats, int8_t *both_dev, int8_t *both_host, int64_t cap,
                                            int64_t *list_dev, float *knife_dev, int64_t *list_host, double *scratch, double *sums_host, void *stream,
                                            void *side_stream)
{
    if (!mats || !formats || !stats || !both_dev || !both_host || !list_dev || !list_host) return fail(MTQ_ERR_INVALID, "null argument");
    if (n <= 0 || n > kColSegs || cap < 0) return fail(MTQ_ERR_INVALID, "a ragged batch holds 1..MTQ_RAGGED_MAX matrices; cap must be non-negative");
    int64_t T = 0, tiles_per[kColSegs];
    for (int j = 0; j < n; ++j) {
        if (mats[j].rows <= 0 || mats[j].cols <= 0) return fail(MTQ_ERR_INVALID, "rows and cols must be positive");
        tiles_per[j] = ((mats[j].rows + 31) / 32) * ((mats[j].cols + 31) / 32);
        T += tiles_per[j];
    }
    if (int rc = mtq_tile_stats_ragged(mats, n, in_dtype, k1_mask, stats, stream)) return rc;
    if (int rc = mtq_threshold_assign_device(stats, T, dec_mask, formats, n_formats, metric, threshold, band, both_dev,
                                             reinterpret_cast<uint8_t *>(both_dev + T), stream)) return rc;
    if (int rc = mtq_device_copy_2d(both_host, (size_t)(2 * T), both_dev, (size_t)(2 * T), (size_t)(2 * T), 1, stream)) return rc;
    hipStream_t main_s = static_cast<hipStream_t>(stream), side = side_stream ? static_cast<hipStream_t>(side_stream) : main_s;
    if (side != main_s) {
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return fail(MTQ_ERR_HIP, "hipEventCreate failed");
        const bool ok = hipEventRecord(ev, main_s) == hipSuccess && hipStreamWaitEvent(side, ev, 0) == hipSuccess;
        (void)hipEventDestroy(ev);
        if (!ok) return fail(MTQ_ERR_HIP, "could not order the side stream behind the masks");
    }
    if (int rc = mtq_knife_tiles_ragged(mats, n, in_dtype, both_dev + T, formats, n_formats, cap, list_dev, cap ? knife_dev : nullptr, side)) return rc;
    if (int rc = mtq_device_copy_2d(list_host, (size_t)(cap + 1) * 8, list_dev, (size_t)(cap + 1) * 8, (size_t)(cap + 1) * 8, 1, side)) return rc;
    return scratch && sums_host ? mtq_threshold_columns_ragged(stats, tiles_per, n, dec_mask, both_dev, scratch, sums_host, stream) : MTQ_OK;
}

extern "C" int mtq_threshold_columns_ragged(const double *stats, const int64_t *tiles_per, int n, uint32_t dec_mask, const int8_t *maps_dev, double *scratch,
                                            double *sums_host, void *stream)
{
    if (!sums_host) return fail(MTQ_ERR_INVALID, "null argument");
    if (int rc = mtq_column_sums_device_ragged(stats, tiles_per, n, dec_mask, maps_dev, scratch, stream)) return rc;
    const size_t pitch = mtq_columns_scratch_doubles() * sizeof(double);
    return mtq_device_copy_2d(sums_host, kColVals * sizeof(double), scratch, pitch, kColVals * sizeof(double), (size_t)n, stream);
}
