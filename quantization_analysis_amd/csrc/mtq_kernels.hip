// mtq_kernels.hip — gfx950 kernels of the mixed-tile search path and their C-ABI launchers
// (include/mtq.h, DEVICE section).  Written for wave64 / CDNA4 only.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <mutex>

#include "../../include/mtq.h"
#include "mtq_device.hpp"
#include "mtq_error.hpp"
#include "mtq_slot_ring.hpp"

namespace mtq {

// ---------------------------------------------------------------------------------------------
// K1 (generic form): one wave64 per 32×32 tile, lane ℓ owns the shared-exponent group
// (row = ℓ>>1, half = ℓ&1).  Literal uint32 quantisation per candidate format, float32 terms,
// float64 accumulation; lane-sequential over the 16 elements, sequential over the 4 lanes of a row pair,
// xor-butterfly over the 16 row pairs (the order include/mtq.h documents).  Handles every input
// (fp32 or bf16 storage, specials, ragged edges); the bf16 fast kernel defers to the same
// arithmetic for groups it cannot take.
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void tile_stats_one(const T *__restrict__ x, int64_t gt, int64_t stride, int64_t rows, int64_t cols,
                                               int64_t ld, int tiles_w, int64_t tiles, uint32_t fmt_mask, int rec,
                                               double *__restrict__ stats, int vec_ok, const RaggedTable &tb)
{
    const int lane = threadIdx.x & 63;
    uint32_t u[kGroup];
    if (tb.n) {   // a ragged batch: the tile's matrix from the table (mtq_device.hpp)
        const TileSite w = ragged_site(tb, (uint32_t)gt);
        const int64_t tr = w.t / w.tiles_w, tc = w.t - tr * w.tiles_w;
        Loader<T>::group(static_cast<const T *>(w.x), tr * kTile + (lane >> 1), tc * kTile + (lane & 1) * kGroup, w.rows, w.cols, w.ld, w.vec_ok != 0, u);
    } else {
        const int64_t b = gt / tiles, t = gt - b * tiles;
        const int64_t tr = t / tiles_w, tc = t - tr * tiles_w;
        Loader<T>::group(x + b * stride, tr * kTile + (lane >> 1), tc * kTile + (lane & 1) * kGroup, rows, cols, ld, vec_ok != 0, u);
    }
    double acc[2 + 5 * kNumFmt]; // registers are indexed by FORMAT (static); the record is compacted when written
    tile_terms_literal(u, fmt_mask, acc);
    if (lane == 0) {
        double *out = stats + gt * rec;
        out[0] = acc[0];
        out[1] = acc[1];
        int o = 2;
#pragma unroll
        for (int f = 0; f < kNumFmt; ++f)
            if (fmt_mask & (1u << f)) {
#pragma unroll
                for (int j = 0; j < 5; ++j) out[o + j] = acc[2 + 5 * f + j];
                o += 5;
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tile_stats_generic(const T *__restrict__ x, int64_t count, int64_t stride,
                                                          int64_t rows, int64_t cols, int64_t ld, int tiles_w,
                                                          int64_t tiles, uint32_t fmt_mask, int rec,
                                                          double *__restrict__ stats, int vec_ok, const RaggedTable tb)
{
    const int64_t gt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); // global tile over the batch, one wave per tile
    if (gt >= (tb.n ? (int64_t)tb.total : count * tiles)) return;      // wave-uniform
    tile_stats_one<T>(x, gt, stride, rows, cols, ld, tiles_w, tiles, fmt_mask, rec, stats, vec_ok, tb);
}

// Follow-up of the exact-route kernels (mtq_fast.hip, mtq_direct.hip): every wave inspects 64 records, and recomputes
// by the literal route the tiles whose Σx carries kRedoMagic (groups outside the exact routes' preconditions).
constexpr unsigned long long kRedoMagicGeneric = 0x7FF8C0DE5EED0001ull;
template <typename T>
__global__ __launch_bounds__(256) void tile_stats_redo_flagged(const T *__restrict__ x, int64_t count, int64_t stride,
                                                               int64_t rows, int64_t cols, int64_t ld, int tiles_w,
                                                               int64_t tiles, uint32_t fmt_mask, int rec,
                                                               double *__restrict__ stats, int vec_ok, unsigned *__restrict__ work, unsigned launch_id,
                                                               const RaggedTable tb)
{
    if (blockIdx.x == 0 && threadIdx.x < kWorkGroups) work[threadIdx.x * kWorkStride] = 0u; // the launch's unit counters, ready for their next user
    if (work[kWorkStamp] != launch_id) return;                          // no tile of this launch was marked (the usual case): nothing to read
    // a grid of at most kRedoBlocks blocks strides over the records (round 3: the follow-up sits on the K1 stream between two K1 launches,
    // and a grid of one wave per 64 records — 8 192 blocks for a 128-tensor batch — took 20–55 µs to pass through the chip just to find nothing)
    const int lane = threadIdx.x & 63;
    const int64_t total = tb.n ? (int64_t)tb.total : count * tiles;
    for (int64_t first = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; first < total; first += (int64_t)gridDim.x * 4 * 64) {
        const int64_t mine = first + lane;
        bool flagged = false;
        if (mine < total) flagged = (unsigned long long)__double_as_longlong(stats[mine * rec]) == kRedoMagicGeneric;
        unsigned long long todo = __ballot(flagged);
        while (todo) {                                                     // wave-uniform loop over the flagged tiles
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            tile_stats_one<T>(x, first + k, stride, rows, cols, ld, tiles_w, tiles, fmt_mask, rec, stats, vec_ok, tb);
        }
    }
}
constexpr int64_t kRedoBlocks = 512;

// The fix-up as a launch of its own (mtq_tile_stats_partial_end): the same search for marked records, decided by the caller's word.
// Its waves must fit into the holes a running K1 leaves (one retiring K1 wave frees 112 VGPRs; a 184-VGPR wave waited for the whole of
// the next K1, and the search behind it with it — r3 trace): one wave per block, and a register budget below K1's.  The budget spills
// the literal route (taken by tiles with non-finite values only) and costs the usual launch, which finds no mark and returns, nothing.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 5))) void tile_stats_redo_marked(const uint16_t *__restrict__ x, int64_t count, int64_t stride, int64_t rows, int64_t cols, int64_t ld,
                                                              int tiles_w, int64_t tiles, uint32_t fmt_mask, int rec, double *__restrict__ stats, int vec_ok,
                                                              const unsigned *__restrict__ mark, unsigned launch_id)
{
    if (*mark != launch_id) return;                                     // no tile of that launch was marked (the usual case)
    const int lane = threadIdx.x & 63;
    const int64_t total = count * tiles;
    for (int64_t first = (int64_t)blockIdx.x * 64; first < total; first += (int64_t)gridDim.x * 64) {
        const int64_t mine = first + lane;
        bool flagged = false;
        if (mine < total) flagged = (unsigned long long)__double_as_longlong(stats[mine * rec]) == kRedoMagicGeneric;
        unsigned long long todo = __ballot(flagged);
        while (todo) {
            const int k = __builtin_ctzll(todo);
            todo &= todo - 1;
            tile_stats_one<uint16_t>(x, first + k, stride, rows, cols, ld, tiles_w, tiles, fmt_mask, rec, stats, vec_ok, RaggedTable{});
        }
    }
}

// ---------------------------------------------------------------------------------------------
// K2 / K3: one thread per shared-exponent group; fmt >= 0 → that format everywhere (K2),
// fmt < 0 → the format named by map[tile] (K3).  Output float32.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void quantize_groups(const T *__restrict__ x, int64_t rows, int64_t cols, int64_t ld,
                                                       int groups_w, int tiles_w, int fmt, const int8_t *__restrict__ map,
                                                       float *__restrict__ y, int64_t ldy, int vec_ok, int vec_ok_y)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = g / groups_w;
    if (row >= rows) return;
    const int64_t col0 = (g - row * groups_w) * kGroup;
    uint32_t u[kGroup];
    Loader<T>::group(x, row, col0, rows, cols, ld, vec_ok != 0, u);
    const uint32_t shared = group_shared_exp(u);
    int f = fmt;
    if (f < 0) f = map[(row / kTile) * tiles_w + col0 / kTile];
    uint32_t o[kGroup];
#pragma unroll
    for (int i = 0; i < kGroup; ++i) o[i] = quant_elem_bits(f, u[i], shared);
    float *yr = y + row * ldy + col0;
    if (col0 + kGroup <= cols && vec_ok_y) {
        uint4 *q = reinterpret_cast<uint4 *>(yr);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < kGroup; ++i)
            if (col0 + i < cols) yr[i] = __uint_as_float(o[i]);
    }
}


// ---------------------------------------------------------------------------------------------
// K5 (loader): float8-e4m3fn weights × float32 inverse block scales → float32, the dequantisation the
// reference's loader does on the host for DeepSeek-style checkpoints (hf_model_utils.py:199-215,273-281):
// out = w.float() * scale_inv.repeat_interleave(block)[..].  Block shape = ceil(dim / scale_dim) (:199-206).
// HBM bound (1 B read + 4 B written per element).  Coalesced form: one lane per 4 elements — a wave instruction
// reads 256 contiguous bytes and writes 1 KiB contiguous; blockIdx.y walks the rows, so no per-thread division by
// the row length; gfx950's OCP-fp8 convert (v_cvt_pk_f32_fp8) decodes two codes per instruction.  The scalar form
// takes unaligned / strided views and the ragged row ends.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float e4m3fn_to_f32(uint32_t b)
{
    const uint32_t s = (b & 0x80u) << 24, e = (b >> 3) & 0xFu, m = b & 7u;
    if (e == 0u) return __uint_as_float(s | __float_as_uint((float)m * 0.001953125f)); // subnormal: m * 2^-9 (exact)
    if (e == 15u && m == 7u) return __uint_as_float(s | 0x7FC00000u);                  // the single NaN encoding
    return __uint_as_float(s | ((e + 120u) << 23) | (m << 20));
}

typedef float k5f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void dequant_fp8_quads(const uint8_t *__restrict__ w, const float *__restrict__ scale, int64_t rows,
                                                         uint32_t cols, int64_t ldw, int64_t scale_cols, uint32_t bh, uint32_t bw,
                                                         int bw_shift, float *__restrict__ out, int64_t ldo)
{
    const uint32_t c0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (c0 >= cols) return;
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        const float *srow = scale + (row / bh) * scale_cols;                      // block-uniform: scalar unit
        const uint8_t *src = w + row * ldw + c0;
        float *dst = out + row * ldo + c0;
        if (c0 + 4u <= cols) {
            const uint32_t v = *reinterpret_cast<const uint32_t *>(src);
            const k5f2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)v, true);
            float4 o;
            if (bw_shift >= 0 && (bw & 3u) == 0u) {                               // 4 | block width: one scale for the quad
                const float sc = srow[c0 >> bw_shift];
                o.x = lo.x * sc; o.y = lo.y * sc; o.z = hi.x * sc; o.w = hi.y * sc;
            } else {
                o.x = lo.x * srow[c0 / bw]; o.y = lo.y * srow[(c0 + 1u) / bw]; o.z = hi.x * srow[(c0 + 2u) / bw]; o.w = hi.y * srow[(c0 + 3u) / bw];
            }
            *reinterpret_cast<float4 *>(dst) = o;
        } else {
            for (uint32_t i = 0; c0 + i < cols; ++i) dst[i] = e4m3fn_to_f32(src[i]) * srow[(c0 + i) / bw];
        }
    }
}

__global__ __launch_bounds__(256) void dequant_fp8_scalar(const uint8_t *__restrict__ w, const float *__restrict__ scale, int64_t rows,
                                                          int64_t cols, int64_t ldw, int64_t scale_cols, int bh, int bw,
                                                          float *__restrict__ out, int64_t ldo)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = g / cols;
    if (row >= rows) return;
    const int64_t c = g - row * cols;
    out[row * ldo + c] = e4m3fn_to_f32(w[row * ldw + c]) * scale[(row / bh) * scale_cols + c / bw];
}


// ---------------------------------------------------------------------------------------------
// K2 / K3, coalesced form: one lane per 4 elements — 8-byte (bf16) or 16-byte (fp32) loads and 16-byte
// stores, every wave instruction touching one contiguous 512 B / 1 KiB span; rows on blockIdx.y (no per-thread
// division); the 4 lanes of a shared-exponent group agree on the exponent with two lane exchanges.
// BFP values come from the float-domain form of the reference's rounding (mtq_direct.hip): truncate x to the group's
// 24-bit window (scale, v_trunc, scale back), add and subtract C = 1.5·2^(E−103−M) — RNE to the format's step, ties to
// the even multiple — and clamp to ±(2^M − 1) steps; the result IS the float the reference assembles from sign, exp_out
// and shifted mantissa whenever 24 <= E <= 231 (every constant normal, no exponent wrap).  Groups outside that range
// (denormal-only, Inf/NaN, huge) take the literal uint32 route; the two agree bit for bit on the F1/F2 vectors.
// Needs cols % 16 == 0 and 16-byte aligned rows; anything else takes quantize_groups.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void quantize_quads(const T *__restrict__ x, int64_t rows, uint32_t cols, int64_t ld, int tiles_w,
                                                      int fmt, const int8_t *__restrict__ map, float *__restrict__ y, int64_t ldy)
{
    const uint32_t col0 = (blockIdx.x * 256u + threadIdx.x) * 4u;
    const bool live = col0 < cols;                                       // the whole group is live or not (cols % 16 == 0)
    for (int64_t row = blockIdx.y; row < rows; row += gridDim.y) {
        uint32_t u[4] = {0u, 0u, 0u, 0u};
        if (live) {
            if constexpr (sizeof(T) == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(x + row * ld + col0);
                u[0] = v.x << 16; u[1] = v.x & 0xFFFF0000u; u[2] = v.y << 16; u[3] = v.y & 0xFFFF0000u;
            } else {
                const uint4 v = *reinterpret_cast<const uint4 *>(x + row * ld + col0);
                u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
            }
        }
        uint32_t m = max(max(u[0] & 0x7F800000u, u[1] & 0x7F800000u), max(u[2] & 0x7F800000u, u[3] & 0x7F800000u));
        m = max(m, (uint32_t)__shfl_xor((int)m, 1, 64));                 // lanes 4k..4k+3 hold one group
        m = max(m, (uint32_t)__shfl_xor((int)m, 2, 64));
        if (!live) continue;
        const uint32_t shared = m >> 23;
        const int f = fmt >= 0 ? fmt : (int)map[(row / kTile) * tiles_w + col0 / kTile];   // K2: launch-wide; K3: the tile's own
        uint4 o;
        if (f == 0) {                                                    // bf16: RNE on the raw word (:29-45), no group state
            o = make_uint4(bf16_round_bits(u[0]), bf16_round_bits(u[1]), bf16_round_bits(u[2]), bf16_round_bits(u[3]));
        } else if (f >= 1 && f <= 3 && shared - 24u <= 207u) {           // BFP, exponent in the float-domain range
            const uint32_t M = f == 1 ? 7u : (f == 2 ? 3u : 1u);
            const float k_align = __uint_as_float((277u - shared) << 23), k_back = __uint_as_float((shared - 23u) << 23);
            const float C = __uint_as_float(((shared + 24u - M) << 23) | 0x400000u);
            const float ymax = __uint_as_float((shared + 1u) << 23) - __uint_as_float((shared + 1u - M) << 23); // (2^M − 1)·2^(E−127−(M−1))
            uint32_t r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xs = __builtin_truncf(__uint_as_float(u[i]) * k_align) * k_back;
                r[i] = __float_as_uint(__builtin_amdgcn_fmed3f((xs + C) - C, -ymax, ymax));
            }
            o = make_uint4(r[0], r[1], r[2], r[3]);
        } else if (fmt >= 0) {
            o.x = quant_elem_bits(f, u[0], shared); o.y = quant_elem_bits(f, u[1], shared);
            o.z = quant_elem_bits(f, u[2], shared); o.w = quant_elem_bits(f, u[3], shared);
        } else {
            o.x = quant_elem_bits_mixed(f, u[0], shared); o.y = quant_elem_bits_mixed(f, u[1], shared);
            o.z = quant_elem_bits_mixed(f, u[2], shared); o.w = quant_elem_bits_mixed(f, u[3], shared);
        }
        *reinterpret_cast<uint4 *>(y + row * ldy + col0) = o;
    }
}

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int check_matrix(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld)
{
    if (!x) return fail(MTQ_ERR_INVALID, "x is null");
    if (in_dtype != MTQ_DTYPE_BF16 && in_dtype != MTQ_DTYPE_F32) return fail(MTQ_ERR_INVALID, "in_dtype must be MTQ_DTYPE_BF16 or MTQ_DTYPE_F32");
    if (rows <= 0 || cols <= 0) return fail(MTQ_ERR_INVALID, "rows and cols must be positive (empty tensors are handled by the caller)");
    if (ld < cols) return fail(MTQ_ERR_INVALID, "ld < cols");
    if (rows > (int64_t)1 << 40 || cols > (int64_t)1 << 30) return fail(MTQ_ERR_INVALID, "matrix too large");
    return MTQ_OK;
}

} // namespace mtq

using namespace mtq;

extern "C" int mtq_launch_tile_stats_bf16_fast(const void *x, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                               int64_t ld, uint32_t fmt_mask, uint32_t eval_mask, uint32_t part_mask, double *stats, void *stream,
                                               mtq::WorkSlot *work_out, unsigned launch_id, unsigned *mark, int self_reset);
extern "C" int mtq_launch_tile_stats_direct(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                            int64_t ld, uint32_t fmt_mask, double *stats, int vec_ok, void *stream, mtq::WorkSlot *work_out, unsigned launch_id);

unsigned mtq::next_launch_id()
{
    static std::atomic<unsigned> id{1u};
    unsigned v = id.fetch_add(1u);
    return v ? v : id.fetch_add(1u);   // 0 is what a fresh slot holds
}

// Per-device ring of zeroed unit counters (mtq_error.hpp, mtq_slot_ring.hpp).  A slot is handed to one K1 launch and set back
// to zero by that launch's follow-up kernel on the same stream; the slot's event, recorded behind that kernel, orders its next
// user after it whatever the stream.
namespace {
struct HipEventOps {
    typedef hipEvent_t event;
    typedef hipStream_t stream;
    static bool create(event &e) { return hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; }
    static bool record(event &e, stream s) { return hipEventRecord(e, s) == hipSuccess; }
    static bool wait(stream s, event &e) { return hipStreamWaitEvent(s, e, 0) == hipSuccess; }
    static void destroy(event &e) { (void)hipEventDestroy(e); }
};
constexpr int kMaxDev = 64;
constexpr size_t kSlotUnsigned = (size_t)kWorkGroups * kWorkStride;
struct DeviceRing {
    std::atomic<unsigned *> base{nullptr};
    SlotRing<HipEventOps, kWorkSlots> slots;
};
DeviceRing g_rings[kMaxDev];
std::mutex g_ring_mu;
} // namespace

int mtq::work_counter_acquire(void *stream, WorkSlot *out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return fail(MTQ_ERR_HIP, "hipGetDevice failed");
    DeviceRing &r = g_rings[dev];
    unsigned *base = r.base.load(std::memory_order_acquire);
    if (!base) {
        std::lock_guard<std::mutex> lock(g_ring_mu);
        base = r.base.load(std::memory_order_relaxed);
        if (!base) {
            unsigned *p = nullptr;
            if (hipMalloc(reinterpret_cast<void **>(&p), kWorkSlots * kSlotUnsigned * sizeof(unsigned)) != hipSuccess) return fail(MTQ_ERR_HIP, "could not allocate the work counters");
            if (hipMemset(p, 0, kWorkSlots * kSlotUnsigned * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                (void)hipFree(p);
                return fail(MTQ_ERR_HIP, "could not clear the work counters");
            }
            r.base.store(p, std::memory_order_release);
            base = p;
        }
    }
    const int i = r.slots.acquire(static_cast<hipStream_t>(stream));
    if (i < 0) return fail(MTQ_ERR_HIP, "no free work-counter slot (every slot is mid-launch on another thread, or hipStreamWaitEvent failed)");
    out->counters = base + (size_t)i * kSlotUnsigned;
    out->index = i;
    out->device = dev;
    return MTQ_OK;
}

void mtq::work_counter_release(const WorkSlot &slot, void *stream)
{
    if (slot.index < 0 || slot.device < 0) return;
    (void)g_rings[slot.device].slots.release(slot.index, static_cast<hipStream_t>(stream));
}

void mtq::work_counter_abandon(const WorkSlot &slot)
{
    if (slot.index < 0 || slot.device < 0) return;
    g_rings[slot.device].slots.abandon(slot.index);
}

// Everything the library holds on to between calls is released here, on request — never from a static destructor, where the HIP
// runtime may already be gone (a run under rocprofv3 ended in a SIGSEGV inside __cxa_finalize in round 2).  The Python binding
// registers it with atexit when it loads the library, i.e. after torch has registered its own exit hooks: it runs before them.
extern "C" int mtq_shutdown(void)
{
    host_shutdown();                                   // scan threads joined
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess) n_dev = 0;   // no runtime left: nothing device-side can be released, and nothing needs to be
    int cur = 0;
    const bool have_cur = n_dev > 0 && hipGetDevice(&cur) == hipSuccess;
    {
        std::lock_guard<std::mutex> lock(g_ring_mu);
        for (int d = 0; d < kMaxDev && d < n_dev; ++d) {
            unsigned *base = g_rings[d].base.exchange(nullptr, std::memory_order_acq_rel);
            if (!base) continue;
            if (hipSetDevice(d) == hipSuccess) {
                (void)hipDeviceSynchronize();          // no launch still counts units in the slots, no event is still pending
                g_rings[d].slots.reset();
                (void)hipFree(base);
            }
        }
    }
    if (n_dev > 0) scan_shutdown();                    // device jump-ahead tables
    if (have_cur) (void)hipSetDevice(cur);
    return MTQ_OK;
}

// The bookkeeping above against mock event operations (no GPU): 0 when every property holds, else the number of the
// first failed check.  Exported for tests/test_capi_host.py.
namespace {
struct MockOps {
    struct event { int id = -1; int recorded_on = -1; };
    typedef int stream;
    static int created, waits, last_wait_event, last_wait_stream;
    static bool create(event &e) { e.id = created++; return true; }
    static bool record(event &e, stream s) { e.recorded_on = s; return true; }
    static bool wait(stream s, event &e) { ++waits; last_wait_event = e.id; last_wait_stream = s; if (fail_next > 0) { --fail_next; return false; } return e.recorded_on >= 0; }
    static void destroy(event &e) { e.id = -1; e.recorded_on = -1; ++destroyed; }
    static int fail_next, destroyed;
};
int MockOps::fail_next = 0, MockOps::destroyed = 0;
int MockOps::created = 0, MockOps::waits = 0, MockOps::last_wait_event = -1, MockOps::last_wait_stream = -1;
} // namespace

namespace {
template <typename W>
__global__ __launch_bounds__(256) void copy_rows(W *__restrict__ dst, size_t dst_pitch_w, const W *__restrict__ src, size_t src_pitch_w, size_t width_w, size_t total)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / width_w, c = i - r * width_w;
        dst[r * dst_pitch_w + c] = src[r * src_pitch_w + c];
    }
}
template <typename W>
void launch_copy_rows(void *dst, size_t dp, const void *src, size_t sp, size_t width, size_t rows, hipStream_t s)
{
    const size_t total = width / sizeof(W) * rows;
    const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(copy_rows<W>, dim3(blocks), dim3(256), 0, s, static_cast<W *>(dst), dp / sizeof(W), static_cast<const W *>(src), sp / sizeof(W),
                       width / sizeof(W), total);
}
} // namespace

extern "C" int mtq_device_copy_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, void *stream)
{
    if (!dst || !src) return fail(MTQ_ERR_INVALID, "null argument");
    if (width_bytes == 0 || rows == 0) return fail(MTQ_ERR_INVALID, "width_bytes and rows must be positive");
    if (dst_pitch < width_bytes || src_pitch < width_bytes) return fail(MTQ_ERR_INVALID, "a pitch is smaller than the row");
    if (int rc = require_device()) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (rows > 1 && dst_pitch == width_bytes && src_pitch == width_bytes) { width_bytes *= rows; dst_pitch = src_pitch = width_bytes; rows = 1; }   // contiguous
    const uintptr_t all = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | width_bytes | (rows > 1 ? (dst_pitch | src_pitch) : 0);
    if (all % 16 == 0) launch_copy_rows<uint4>(dst, dst_pitch, src, src_pitch, width_bytes, rows, s);
    else if (all % 8 == 0) launch_copy_rows<unsigned long long>(dst, dst_pitch, src, src_pitch, width_bytes, rows, s);
    else launch_copy_rows<unsigned char>(dst, dst_pitch, src, src_pitch, width_bytes, rows, s);
    return check_launch("mtq_device_copy_2d");
}

// ---------------------------------------------------------------------------------------------
// The knife-edge tiles of the threshold rule, ready for the host's literal float32 score (mixed_tile_threshold.py:97-123 scores a
// tile from its float32 values and each format's reconstruction of them, tile_utils.py:46-57): a list of the flagged tiles
// (any order: every entry carries its tile id) and, per listed tile, its 32x32 values and every requested format's y as float32
// (pads of ragged edge tiles +0.0, as in the reference's padded view).  Two launches, no host round trip.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void knife_list(const int8_t *__restrict__ near, int64_t total, int64_t cap, long long *__restrict__ list)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total || near[i] == 0) return;
    const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long *>(list + cap), 1ull);   // list[cap]: how many were flagged
    if ((int64_t)slot < cap) list[slot] = i;
}

template <typename T>
__global__ __launch_bounds__(64) void knife_tiles(const T *__restrict__ x, int64_t stride, int64_t rows, int64_t cols, int64_t ld, int tiles_w, int64_t tiles,
                                                  const long long *__restrict__ list, int64_t cap, int4 fmts, int n_fmts, float *__restrict__ out, int vec_ok,
                                                  const RaggedTable tb)
{
    const int64_t listed = list[cap] < cap ? list[cap] : cap;
    const int64_t b = blockIdx.x;
    if (b >= listed) return;
    const int64_t id = list[b];
    const int lane = threadIdx.x;
    const int64_t r = lane >> 1, c0 = (int64_t)(lane & 1) * kGroup;
    uint32_t u[kGroup];
    if (tb.n) {   // a ragged batch: the listed id is a tile number of the launch (mtq_device.hpp)
        const TileSite w = ragged_site(tb, (uint32_t)__builtin_amdgcn_readfirstlane((int)id));
        Loader<T>::group(static_cast<const T *>(w.x), (int64_t)(w.t / w.tiles_w) * kTile + r, (int64_t)(w.t % w.tiles_w) * kTile + c0, w.rows, w.cols, w.ld,
                         w.vec_ok != 0, u);
    } else {
        const int64_t j = id / tiles, t = id - j * tiles;
        Loader<T>::group(x + j * stride, (t / tiles_w) * kTile + r, (t % tiles_w) * kTile + c0, rows, cols, ld, vec_ok != 0, u);
    }
    const uint32_t shared = group_shared_exp(u);
    const int f4[4] = {fmts.x, fmts.y, fmts.z, fmts.w};
    for (int p = 0; p <= n_fmts; ++p) {
        uint4 *q = reinterpret_cast<uint4 *>(out + ((int64_t)p * cap + b) * (kTile * kTile) + r * kTile + c0);
        uint32_t o[kGroup];
#pragma unroll
        for (int i = 0; i < kGroup; ++i) o[i] = p == 0 ? u[i] : quant_elem_bits_mixed(f4[p - 1], u[i], shared);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = make_uint4(o[4 * i], o[4 * i + 1], o[4 * i + 2], o[4 * i + 3]);
    }
}
} // namespace

static int knife_tiles_launch(const void *x, int in_dtype, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld, int64_t tw, int64_t tiles, int64_t total,
                              int vec_ok, const RaggedTable &tb, const int8_t *near, const int *formats, int n_formats, int64_t cap, int64_t *list,
                              float *tiles_out, void *stream)
{
    if (!near || !list || !formats) return fail(MTQ_ERR_INVALID, "null argument");
    if (cap < 0 || (cap > 0 && !tiles_out)) return fail(MTQ_ERR_INVALID, "cap must be non-negative and tiles_out set when cap > 0");
    if (n_formats < 1 || n_formats > 4) return fail(MTQ_ERR_INVALID, "n_formats must be 1..4");
    for (int i = 0; i < n_formats; ++i)
        if (formats[i] < 0 || formats[i] > 3) return fail(MTQ_ERR_INVALID, "format codes are 0..3 (bf16, bfp8, bfp4, bfp2)");
    if (!aligned16(tiles_out)) return fail(MTQ_ERR_INVALID, "tiles_out must be 16-byte aligned");
    if (int rc = require_device()) return rc;
    if (tw > INT32_MAX || total > ((int64_t)1 << 40) || cap > (1 << 22)) return fail(MTQ_ERR_INVALID, "too many tiles for one launch");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(list + cap, 0, sizeof(int64_t), s) != hipSuccess) return fail(MTQ_ERR_HIP, "hipMemsetAsync failed");
    hipLaunchKernelGGL(knife_list, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, near, total, cap, reinterpret_cast<long long *>(list));
    if (int rc = check_launch("mtq_knife_tiles_device (list)")) return rc;
    if (cap == 0) return MTQ_OK;
    int4 f = make_int4(formats[0], n_formats > 1 ? formats[1] : 0, n_formats > 2 ? formats[2] : 0, n_formats > 3 ? formats[3] : 0);
    if (in_dtype == MTQ_DTYPE_BF16)
        hipLaunchKernelGGL(knife_tiles<uint16_t>, dim3((unsigned)cap), dim3(64), 0, s, static_cast<const uint16_t *>(x), stride_elems, rows, cols, ld, (int)tw, tiles,
                           reinterpret_cast<const long long *>(list), cap, f, n_formats, tiles_out, vec_ok, tb);
    else
        hipLaunchKernelGGL(knife_tiles<float>, dim3((unsigned)cap), dim3(64), 0, s, static_cast<const float *>(x), stride_elems, rows, cols, ld, (int)tw, tiles,
                           reinterpret_cast<const long long *>(list), cap, f, n_formats, tiles_out, vec_ok, tb);
    return check_launch("mtq_knife_tiles_device");
}

static const RaggedTable kUniform{};   // n = 0: the kernel's other arguments describe a uniform batch

extern "C" int mtq_knife_tiles_device(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                                      const int8_t *near, const int *formats, int n_formats, int64_t cap, int64_t *list, float *tiles_out, void *stream)
{
    if (int rc = check_matrix(x, in_dtype, rows, cols, ld)) return rc;
    if (count <= 0) return fail(MTQ_ERR_INVALID, "count must be positive");
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw, total = count * tiles;
    const int64_t esz = in_dtype == MTQ_DTYPE_BF16 ? 2 : 4;
    const int vec_ok = aligned16(x) && (ld * esz) % 16 == 0 && (stride_elems * esz) % 16 == 0;
    return knife_tiles_launch(x, in_dtype, stride_elems, rows, cols, ld, tw, tiles, total, vec_ok, kUniform, near, formats, n_formats, cap, list, tiles_out, stream);
}

// The table of a ragged batch from the caller's matrices (include/mtq.h, MtqMatrix): every matrix checked as mtq_tile_stats checks one.
static int ragged_table(const MtqMatrix *mats, int n, int in_dtype, RaggedTable &tb)
{
    if (!mats) return fail(MTQ_ERR_INVALID, "mats is null");
    if (n <= 0 || n > kRaggedMax) return fail(MTQ_ERR_INVALID, "a ragged batch holds 1..MTQ_RAGGED_MAX matrices");
    const int64_t esz = in_dtype == MTQ_DTYPE_BF16 ? 2 : 4;
    int64_t first = 0;
    for (int j = 0; j < n; ++j) {
        const MtqMatrix &m = mats[j];
        if (int rc = check_matrix(m.x, in_dtype, m.rows, m.cols, m.ld)) return rc;
        const int64_t th = (m.rows + kTile - 1) / kTile, tw = (m.cols + kTile - 1) / kTile;
        if (tw > INT32_MAX || first + th * tw >= ((int64_t)1 << 31)) return fail(MTQ_ERR_INVALID, "too many tiles for one launch");
        RaggedSeg &s = tb.seg[j];
        s.x = m.x; s.rows = m.rows; s.cols = m.cols; s.ld = m.ld;
        s.tiles_w = (uint32_t)tw; s.first = (uint32_t)first;
        s.vec_ok = aligned16(m.x) && (m.ld * esz) % 16 == 0;
        s.pad_ = 0;
        first += th * tw;
    }
    tb.n = (uint32_t)n;
    tb.total = (uint32_t)first;
    return MTQ_OK;
}

extern "C" int mtq_knife_tiles_ragged(const MtqMatrix *mats, int n, int in_dtype, const int8_t *near, const int *formats, int n_formats, int64_t cap,
                                      int64_t *list, float *tiles_out, void *stream)
{
    RaggedTable tb{};
    if (int rc = ragged_table(mats, n, in_dtype, tb)) return rc;
    return knife_tiles_launch(nullptr, in_dtype, 0, 0, 0, 0, 1, 1, tb.total, 0, tb, near, formats, n_formats, cap, list, tiles_out, stream);
}

extern "C" int mtq_selftest_slot_ring(void)
{
    MockOps::created = MockOps::waits = 0;
    MockOps::destroyed = 0;
    static SlotRing<MockOps, 4> ring;                    // static: the mutexes are not movable; the test runs once per process
    int a[4];
    for (int k = 0; k < 4; ++k) a[k] = ring.acquire(10 + k);
    for (int k = 0; k < 4; ++k) if (a[k] != k) return 1;                 // round robin over fresh slots
    if (MockOps::waits != 0) return 2;                                    // fresh slots have nothing to wait for
    if (ring.acquire(20) != -1) return 3;                                 // all four held between acquire and release: refused, not shared
    if (!ring.release(1, 11)) return 4;
    const int b = ring.acquire(21);                                       // the only free slot, found wherever the cursor stands
    if (b != 1) return 5;
    if (MockOps::waits != 1 || MockOps::last_wait_stream != 21 || MockOps::last_wait_event != 0) return 6;   // its next user waits for its event
    for (int k = 0; k < 4; ++k) if (!ring.release(k, 30 + k)) return 7;
    if (MockOps::created != 4) return 8;                                  // one event per slot, created once
    const int before = MockOps::waits;
    for (int k = 0; k < 8; ++k) {                                         // 2 × N launches in a row: every reuse waits
        const int i = ring.acquire(40 + k);
        if (i < 0) return 9;
        if (!ring.release(i, 40 + k)) return 10;
    }
    if (MockOps::waits != before + 8) return 11;
    // a launch that could not be enqueued gives its slot back without a new event: the slot is free again, its old event still orders the next user
    const int c = ring.acquire(60);
    if (c < 0) return 12;
    ring.abandon(c);
    bool seen = false;
    for (int k = 0; k < 4 && !seen; ++k) {
        const int i = ring.acquire(70 + k);
        if (i < 0) return 13;
        seen = i == c;
        if (!ring.release(i, 70 + k)) return 14;
    }
    if (!seen) return 15;
    // a slot whose event cannot be waited for is passed over, not fatal
    const int w0 = MockOps::waits;
    MockOps::fail_next = 1;                               // the first slot tried cannot be waited for …
    const int got = ring.acquire(81);
    if (got < 0 || MockOps::fail_next != 0 || MockOps::waits != w0 + 2) return 17;   // … and the next one is handed out
    if (!ring.release(got, 81)) return 18;
    ring.reset();
    if (MockOps::destroyed != 4) return 19;
    return 0;
}

// MTQ_FORCE_GENERIC=1 routes every input through tile_stats_generic (A/B checks of the fast kernel).
static bool force_generic()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("MTQ_FORCE_GENERIC"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}

// The follow-up kernel zeroes the launch's unit counter; if it could not be launched, do it with a memset so that the
// slot's next user does not start from a stale count.
static int finish_counter_launch(const WorkSlot &work, hipStream_t s)
{
    const int rc = check_launch("mtq_tile_stats (redo flagged)");
    if (rc != MTQ_OK) (void)hipMemsetAsync(work.counters, 0, (size_t)kWorkGroups * kWorkStride * sizeof(unsigned), s);
    work_counter_release(work, s);   // the slot's next user is ordered behind the reset, on whatever stream it launches
    return rc;
}

extern "C" size_t mtq_stats_record_doubles(uint32_t fmt_mask)
{
    return 2 + ((fmt_mask & MTQ_MASK_SLIM) ? 3 : 5) * (size_t)__builtin_popcount(fmt_mask & MTQ_MASK_ALL);
}

// out[t] = [Σx, Σx², {Σy, Σy², Σxy} per format] of the full record t: one thread per output double.
__global__ __launch_bounds__(256) void pack_slim_records(const double *__restrict__ stats, int64_t tiles, int rec_in, int rec_out,
                                                         double *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= tiles * rec_out) return;
    const int64_t t = i / rec_out;
    const int k = (int)(i - t * rec_out);
    const int src = k < 2 ? k : 2 + 5 * ((k - 2) / 3) + (k - 2) % 3;
    out[i] = stats[t * rec_in + src];
}

extern "C" int mtq_pack_slim_records(const double *stats, int64_t tiles, uint32_t fmt_mask, double *out, void *stream)
{
    if (!stats || !out) return fail(MTQ_ERR_INVALID, "null argument");
    if (tiles <= 0 || tiles > ((int64_t)1 << 34)) return fail(MTQ_ERR_INVALID, "tiles out of range");
    if ((fmt_mask & ~MTQ_MASK_ALL) != 0 || (fmt_mask & MTQ_MASK_ALL) == 0) return fail(MTQ_ERR_INVALID, "fmt_mask must name stored formats only");
    if (int rc = require_device()) return rc;
    const int nf = __builtin_popcount(fmt_mask & MTQ_MASK_ALL), rec_in = 2 + 5 * nf, rec_out = 2 + 3 * nf;
    const int64_t n = tiles * rec_out;
    hipLaunchKernelGGL(pack_slim_records, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), stats, tiles, rec_in,
                       rec_out, out);
    return check_launch("mtq_pack_slim_records");
}

// K1 over a batch.  eval_mask / part_mask: mtq_tile_stats_partial (the whole layout, no partial slot, for mtq_tile_stats[_batched]).
static int tile_stats_launch(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols, int64_t ld,
                             uint32_t fmt_mask, uint32_t eval_mask, uint32_t part_mask, double *stats, void *stream)
{
    if (int rc = check_matrix(x, in_dtype, rows, cols, ld)) return rc;
    if (!stats) return fail(MTQ_ERR_INVALID, "stats is null");
    if (count <= 0) return fail(MTQ_ERR_INVALID, "count must be positive");
    if ((fmt_mask & ~MTQ_MASK_ALL) != 0) return fail(MTQ_ERR_INVALID, "fmt_mask has bits outside bf16|bfp8|bfp4|bfp2");
    if (int rc = require_device()) return rc;
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw;
    if (tw > INT32_MAX || count * tiles > ((int64_t)1 << 33)) return fail(MTQ_ERR_INVALID, "too many tiles for one launch");
    const int rec = (int)mtq_stats_record_doubles(fmt_mask);
    const int64_t esz = in_dtype == MTQ_DTYPE_BF16 ? 2 : 4;
    const int vec_ok = aligned16(x) && (ld * esz) % 16 == 0 && (stride_elems * esz) % 16 == 0;
    // bf16 storage, whole 32x128 units, 16-byte aligned rows, at least one BFP format → exact-integer fast kernel
    if (in_dtype == MTQ_DTYPE_BF16 && vec_ok && rows % kTile == 0 && cols % 128 == 0 && (fmt_mask & 0xEu) != 0 && !force_generic()) {
        WorkSlot work;
        const unsigned launch_id = next_launch_id();
        // the exact-integer kernel serves partial evaluation; every other route below writes the whole layout (always allowed)
        const uint32_t ev = (eval_mask & 0xEu) ? eval_mask : fmt_mask;
        if (int rc = mtq_launch_tile_stats_bf16_fast(x, count, stride_elems, rows, cols, ld, fmt_mask, ev, ev == eval_mask ? part_mask : 0u, stats, stream, &work, launch_id, nullptr, 0)) {
            work_counter_abandon(work);                  // the launch never happened: the slot keeps its previous user's event
            return rc;
        }
        const int64_t waves = (count * tiles + 63) / 64;
        hipLaunchKernelGGL(tile_stats_redo_flagged<uint16_t>, dim3((unsigned)std::min<int64_t>((waves + 3) / 4, kRedoBlocks)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint16_t *>(x), count, stride_elems, rows, cols, ld, (int)tw, tiles, fmt_mask, rec, stats, vec_ok, work.counters, launch_id, kUniform);
        return finish_counter_launch(work, static_cast<hipStream_t>(stream));
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    // float32 storage, ragged or unaligned bf16: one wave per tile on the reduced-arithmetic route (mtq_direct.hip)
    if ((fmt_mask & MTQ_MASK_ALL) != 0 && count * tiles < ((int64_t)1 << 31) && !force_generic()) {
        WorkSlot work;
        const unsigned launch_id = next_launch_id();
        if (int rc = mtq_launch_tile_stats_direct(x, in_dtype, count, stride_elems, rows, cols, ld, fmt_mask, stats, vec_ok, stream, &work, launch_id)) {
            work_counter_abandon(work);
            return rc;
        }
        const dim3 rgrid((unsigned)std::min<int64_t>(((count * tiles + 63) / 64 + 3) / 4, kRedoBlocks));
        if (in_dtype == MTQ_DTYPE_BF16)
            hipLaunchKernelGGL(tile_stats_redo_flagged<uint16_t>, rgrid, dim3(256), 0, s, static_cast<const uint16_t *>(x), count, stride_elems,
                               rows, cols, ld, (int)tw, tiles, fmt_mask, rec, stats, vec_ok, work.counters, launch_id, kUniform);
        else
            hipLaunchKernelGGL(tile_stats_redo_flagged<float>, rgrid, dim3(256), 0, s, static_cast<const float *>(x), count, stride_elems, rows,
                               cols, ld, (int)tw, tiles, fmt_mask, rec, stats, vec_ok, work.counters, launch_id, kUniform);
        return finish_counter_launch(work, s);
    }
    const int64_t blocks = (count * tiles + 3) / 4;
    if (in_dtype == MTQ_DTYPE_BF16)
        hipLaunchKernelGGL(tile_stats_generic<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(x), count,
                           stride_elems, rows, cols, ld, (int)tw, tiles, fmt_mask, rec, stats, vec_ok, kUniform);
    else
        hipLaunchKernelGGL(tile_stats_generic<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(x), count,
                           stride_elems, rows, cols, ld, (int)tw, tiles, fmt_mask, rec, stats, vec_ok, kUniform);
    return check_launch("mtq_tile_stats");
}

extern "C" int mtq_tile_stats_batched(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows,
                                      int64_t cols, int64_t ld, uint32_t fmt_mask, double *stats, void *stream)
{
    return tile_stats_launch(x, in_dtype, count, stride_elems, rows, cols, ld, fmt_mask, fmt_mask, 0u, stats, stream);
}

int mtq_launch_tile_stats_direct_ragged(const mtq::RaggedTable &tb, int in_dtype, uint32_t fmt_mask, double *stats, void *stream,
                                        mtq::WorkSlot *work_out, unsigned launch_id);

// K1 over a ragged batch (include/mtq.h): the direct kernel over the batch's tiles numbered through, then its fix-up — two launches whatever
// the number of matrices.
extern "C" int mtq_tile_stats_ragged(const MtqMatrix *mats, int n, int in_dtype, uint32_t fmt_mask, double *stats, void *stream)
{
    if (in_dtype != MTQ_DTYPE_BF16 && in_dtype != MTQ_DTYPE_F32) return fail(MTQ_ERR_INVALID, "in_dtype must be MTQ_DTYPE_F32 or MTQ_DTYPE_BF16");
    RaggedTable tb{};
    if (int rc = ragged_table(mats, n, in_dtype, tb)) return rc;
    if (!stats) return fail(MTQ_ERR_INVALID, "stats is null");
    if ((fmt_mask & ~MTQ_MASK_ALL) != 0) return fail(MTQ_ERR_INVALID, "fmt_mask has bits outside bf16|bfp8|bfp4|bfp2");
    if (int rc = require_device()) return rc;
    const int rec = (int)mtq_stats_record_doubles(fmt_mask);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((fmt_mask & MTQ_MASK_ALL) != 0 && !force_generic()) {
        WorkSlot work;
        const unsigned launch_id = next_launch_id();
        if (int rc = mtq_launch_tile_stats_direct_ragged(tb, in_dtype, fmt_mask, stats, stream, &work, launch_id)) {
            work_counter_abandon(work);
            return rc;
        }
        const dim3 rgrid((unsigned)std::min<int64_t>((((int64_t)tb.total + 63) / 64 + 3) / 4, kRedoBlocks));
        if (in_dtype == MTQ_DTYPE_BF16)
            hipLaunchKernelGGL(tile_stats_redo_flagged<uint16_t>, rgrid, dim3(256), 0, s, static_cast<const uint16_t *>(nullptr), 0, 0, 0, 0, 0, 1, 1, fmt_mask,
                               rec, stats, 0, work.counters, launch_id, tb);
        else
            hipLaunchKernelGGL(tile_stats_redo_flagged<float>, rgrid, dim3(256), 0, s, static_cast<const float *>(nullptr), 0, 0, 0, 0, 0, 1, 1, fmt_mask, rec,
                               stats, 0, work.counters, launch_id, tb);
        return finish_counter_launch(work, s);
    }
    const int64_t blocks = ((int64_t)tb.total + 3) / 4;
    if (in_dtype == MTQ_DTYPE_BF16)
        hipLaunchKernelGGL(tile_stats_generic<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(nullptr), 0, 0, 0, 0, 0, 1, 1,
                           fmt_mask, rec, stats, 0, tb);
    else
        hipLaunchKernelGGL(tile_stats_generic<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(nullptr), 0, 0, 0, 0, 0, 1, 1, fmt_mask,
                           rec, stats, 0, tb);
    return check_launch("mtq_tile_stats_ragged");
}

extern "C" int mtq_tile_stats_partial(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                      int64_t ld, uint32_t layout_mask, uint32_t full_mask, uint32_t sums_mask, double *stats, void *stream)
{
    if (((full_mask | sums_mask) & ~layout_mask) != 0 || (full_mask & sums_mask) != 0 || (layout_mask & ~MTQ_MASK_ALL) != 0)
        return fail(MTQ_ERR_INVALID, "full_mask and sums_mask must be disjoint subsets of layout_mask");
    if (sums_mask & 1u) return fail(MTQ_ERR_INVALID, "the bf16 slot has no partial form");
    return tile_stats_launch(x, in_dtype, count, stride_elems, rows, cols, ld, layout_mask, full_mask | sums_mask, sums_mask, stats, stream);
}

// mtq_tile_stats_partial in two launches the caller places itself: _begin is the exact-integer kernel alone (it resets its own unit
// counters and leaves its launch id in *mark when it meets a tile it cannot take), _end the literal fix-up of those tiles.  The
// streamed driver issues _begin on its K1 stream — nothing then sits between two K1 launches — and _end on the batch's search stream.
extern "C" int mtq_tile_stats_partial_begin(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                            int64_t ld, uint32_t layout_mask, uint32_t full_mask, uint32_t sums_mask, double *stats,
                                            uint32_t *mark, uint32_t *launch_id_out, void *stream)
{
    if (!mark || !launch_id_out) return fail(MTQ_ERR_INVALID, "null argument");
    if (((full_mask | sums_mask) & ~layout_mask) != 0 || (full_mask & sums_mask) != 0 || (layout_mask & ~MTQ_MASK_ALL) != 0 || (sums_mask & 1u))
        return fail(MTQ_ERR_INVALID, "full_mask and sums_mask must be disjoint subsets of layout_mask (and the bf16 slot has no partial form)");
    if (int rc = check_matrix(x, in_dtype, rows, cols, ld)) return rc;
    if (!stats || count <= 0) return fail(MTQ_ERR_INVALID, "stats is null or count is not positive");
    if (int rc = require_device()) return rc;
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw;
    if (tw > INT32_MAX || count * tiles > ((int64_t)1 << 33)) return fail(MTQ_ERR_INVALID, "too many tiles for one launch");
    const uint32_t eval = full_mask | sums_mask;
    const int vec_ok = aligned16(x) && (ld * 2) % 16 == 0 && (stride_elems * 2) % 16 == 0;
    if (!(in_dtype == MTQ_DTYPE_BF16 && vec_ok && rows % kTile == 0 && cols % 128 == 0 && (eval & 0xEu) != 0 && !force_generic()))
        return fail(MTQ_ERR_UNSUPPORTED, "the two-launch form serves bf16 storage in whole 32x128 units with 16-byte aligned rows: use mtq_tile_stats_partial");
    WorkSlot work;
    const unsigned launch_id = next_launch_id();
    if (int rc = mtq_launch_tile_stats_bf16_fast(x, count, stride_elems, rows, cols, ld, layout_mask, eval, sums_mask, stats, stream, &work, launch_id, mark, 1)) {
        work_counter_abandon(work);
        return rc;
    }
    work_counter_release(work, stream);   // the kernel has reset the counters when it ends: the slot's next user is ordered behind it
    *launch_id_out = launch_id;
    return MTQ_OK;
}

extern "C" int mtq_tile_stats_partial_end(const void *x, int in_dtype, int64_t count, int64_t stride_elems, int64_t rows, int64_t cols,
                                          int64_t ld, uint32_t layout_mask, double *stats, const uint32_t *mark, uint32_t launch_id, void *stream)
{
    if (!mark || !stats) return fail(MTQ_ERR_INVALID, "null argument");
    if (int rc = check_matrix(x, in_dtype, rows, cols, ld)) return rc;
    if (in_dtype != MTQ_DTYPE_BF16 || count <= 0 || (layout_mask & ~MTQ_MASK_ALL) != 0) return fail(MTQ_ERR_INVALID, "arguments do not belong to a mtq_tile_stats_partial_begin launch");
    if (int rc = require_device()) return rc;
    const int64_t th = (rows + kTile - 1) / kTile, tw = (cols + kTile - 1) / kTile, tiles = th * tw;
    const int rec = (int)mtq_stats_record_doubles(layout_mask);
    const int vec_ok = aligned16(x) && (ld * 2) % 16 == 0 && (stride_elems * 2) % 16 == 0;
    const int64_t waves = (count * tiles + 63) / 64;
    hipLaunchKernelGGL(tile_stats_redo_marked, dim3((unsigned)std::min<int64_t>(waves, 4 * kRedoBlocks)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<const uint16_t *>(x), count, stride_elems, rows, cols, ld, (int)tw, tiles, layout_mask, rec, stats, vec_ok, mark, launch_id);
    return check_launch("mtq_tile_stats_partial_end");
}

extern "C" int mtq_tile_stats(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld, uint32_t fmt_mask,
                              double *stats, void *stream)
{
    return mtq_tile_stats_batched(x, in_dtype, 1, 0, rows, cols, ld, fmt_mask, stats, stream);
}

static int launch_quantize(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld, int fmt, const int8_t *map,
                           float *y, int64_t ldy, void *stream, const char *what)
{
    if (int rc = check_matrix(x, in_dtype, rows, cols, ld)) return rc;
    if (!y) return fail(MTQ_ERR_INVALID, "y is null");
    if (ldy < cols) return fail(MTQ_ERR_INVALID, "ldy < cols");
    if (int rc = require_device()) return rc;
    const int64_t gw = (cols + kGroup - 1) / kGroup, tw = (cols + kTile - 1) / kTile;
    const int64_t groups = rows * gw;
    if (gw > INT32_MAX || groups > ((int64_t)1 << 38)) return fail(MTQ_ERR_INVALID, "matrix too large for one launch");
    const int64_t esz = in_dtype == MTQ_DTYPE_BF16 ? 2 : 4;
    const int vec_ok = aligned16(x) && (ld * esz) % 16 == 0;
    const int vec_ok_y = aligned16(y) && (ldy * 4) % 16 == 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec_ok && vec_ok_y && cols % kGroup == 0 && (in_dtype == MTQ_DTYPE_F32 || aligned16(x))) { // coalesced quad form
        const dim3 qgrid((unsigned)((cols + 1023) / 1024), (unsigned)(rows < 65535 ? rows : 65535));
        if (in_dtype == MTQ_DTYPE_BF16)
            hipLaunchKernelGGL(quantize_quads<uint16_t>, qgrid, dim3(256), 0, s, static_cast<const uint16_t *>(x), rows, (uint32_t)cols, ld,
                               (int)tw, fmt, map, y, ldy);
        else
            hipLaunchKernelGGL(quantize_quads<float>, qgrid, dim3(256), 0, s, static_cast<const float *>(x), rows, (uint32_t)cols, ld, (int)tw,
                               fmt, map, y, ldy);
        return check_launch(what);
    }
    const int64_t blocks = (groups + 255) / 256;
    if (in_dtype == MTQ_DTYPE_BF16)
        hipLaunchKernelGGL(quantize_groups<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint16_t *>(x), rows, cols,
                           ld, (int)gw, (int)tw, fmt, map, y, ldy, vec_ok, vec_ok_y);
    else
        hipLaunchKernelGGL(quantize_groups<float>, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const float *>(x), rows, cols, ld,
                           (int)gw, (int)tw, fmt, map, y, ldy, vec_ok, vec_ok_y);
    return check_launch(what);
}

extern "C" int mtq_quantize(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld, int fmt, float *y,
                            int64_t ldy, void *stream)
{
    if (fmt < MTQ_FMT_BF16 || fmt > MTQ_FMT_FP0) return fail(MTQ_ERR_UNSUPPORTED, "format code must be 0..4 (bf16,bfp8,bfp4,bfp2,fp0)");
    return launch_quantize(x, in_dtype, rows, cols, ld, fmt, nullptr, y, ldy, stream, "mtq_quantize");
}

extern "C" int mtq_apply_assignment(const void *x, int in_dtype, int64_t rows, int64_t cols, int64_t ld, const int8_t *map,
                                    float *y, int64_t ldy, void *stream)
{
    if (!map) return fail(MTQ_ERR_INVALID, "map is null");
    return launch_quantize(x, in_dtype, rows, cols, ld, -1, map, y, ldy, stream, "mtq_apply_assignment");
}

extern "C" int mtq_dequant_fp8_block(const void *w, const float *scale_inv, int64_t rows, int64_t cols, int64_t ldw,
                                     int64_t scale_rows, int64_t scale_cols, float *out, int64_t ldo, void *stream)
{
    if (!w || !scale_inv || !out) return fail(MTQ_ERR_INVALID, "null argument");
    if (rows <= 0 || cols <= 0 || scale_rows <= 0 || scale_cols <= 0) return fail(MTQ_ERR_INVALID, "shapes must be positive");
    if (ldw < cols || ldo < cols) return fail(MTQ_ERR_INVALID, "leading dimension < cols");
    if (int rc = require_device()) return rc;
    const int64_t bh = (rows + scale_rows - 1) / scale_rows, bw = (cols + scale_cols - 1) / scale_cols; // hf_model_utils.py:199-206
    if ((rows + bh - 1) / bh > scale_rows || (cols + bw - 1) / bw > scale_cols) return fail(MTQ_ERR_INVALID, "scale grid does not cover the tensor");
    if (cols > (int64_t)1 << 31 || rows * cols > ((int64_t)1 << 38) || bh > INT32_MAX || bw > INT32_MAX) return fail(MTQ_ERR_INVALID, "tensor too large");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool vec_ok = (reinterpret_cast<uintptr_t>(w) & 3u) == 0 && ldw % 4 == 0 && aligned16(out) && (ldo * 4) % 16 == 0;
    if (vec_ok) {
        const int bw_shift = (bw & (bw - 1)) == 0 ? __builtin_ctzll((unsigned long long)bw) : -1;
        const dim3 grid((unsigned)((cols + 1023) / 1024), (unsigned)(rows < 65535 ? rows : 65535));
        hipLaunchKernelGGL(dequant_fp8_quads, grid, dim3(256), 0, st, static_cast<const uint8_t *>(w), scale_inv, rows, (uint32_t)cols, ldw,
                           scale_cols, (uint32_t)bh, (uint32_t)bw, bw_shift, out, ldo);
    } else {
        hipLaunchKernelGGL(dequant_fp8_scalar, dim3((unsigned)((rows * cols + 255) / 256)), dim3(256), 0, st, static_cast<const uint8_t *>(w),
                           scale_inv, rows, cols, ldw, scale_cols, (int)bh, (int)bw, out, ldo);
    }
    return check_launch("mtq_dequant_fp8_block");
}
