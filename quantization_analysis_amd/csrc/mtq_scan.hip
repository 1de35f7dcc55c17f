// mtq_scan.hip — H1 on the device: the sequential mixed-tile-greedy scan (mixed_tile_greedy.py:133-346, pcc metric) run where
// K1 wrote the stats records, one wave64 per tensor.  Nothing but the finished maps (1 B/tile) crosses PCIe; the host neither
// scans nor shuffles.  Same IEEE double operations in the same order as the host scan (csrc/mtq_host.cpp), same NumPy
// generator stream (SeedSequence → PCG64 → Generator.permutation), hence the same maps bit for bit
// (tests/test_hip_kernels.py::test_device_scan_*).
//
// What is sequential in the reference and how a 64-lane wave keeps it sequential in effect:
//   * initial sums (:147-174): running sums in tile order — five dependent float64 chains, one lane each, fed from an LDS
//     staging block that all 64 lanes fill;
//   * visiting order (:222-231): Generator.permutation = Fisher–Yates from the top with masked rejection sampling on buffered
//     32-bit halves of PCG64 outputs.  A batch is 64 stream positions: lane l evaluates position l by LCG jump-ahead
//     (state_q = A_q·state + G_q·inc), acceptance `v <= i − (accepted before me)` is settled exactly (certain accepts, certain
//     rejects, the few in between one by one), the batch is cut where the mask level changes, unused positions are handed
//     back; the accepted steps' swaps are applied in parallel except those that share a position with another step of the
//     batch (found with byte counters in LDS), which are resolved in registers in step order (lane broadcasts).  Modelled and checked against NumPy
//     in tools/scan_model/shuffle_model.py;
//   * the scan (:234-278): 64 visits per round under a speculation on their outcome — accept mode: lane i holds the running
//     sums plus the deltas of visits 0..i added one after the other (a serial chain of float64 additions on three lanes,
//     through LDS: the order of the additions is the contract); reject mode: every lane holds the running sums plus its own
//     delta.  The prefix the speculation was right for is kept (csrc/mtq_host.cpp greedy_pass_pcc8, eight lanes there).
// Not handled here (status != 0, the caller falls back to the host scan for that tensor): a zero denominator anywhere in
// the search (the decision then needs Σ|x−y|, :186-189).  The entry point refuses other metrics, repeated formats and tensors
// of more than kMaxTiles tiles (callers use the host scan for those).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>
#include <stdlib.h>
#include <mutex>

#include "../../include/mtq.h"
#include "mtq_decide.hpp"
#include "mtq_error.hpp"

namespace mtq {
namespace {

constexpr int kScanMaxTilesLds = 32768;   // visiting order kept in LDS as 16-bit tile ids up to here (64 KiB: K1 blocks still fit beside the scan block;
                                          // 65536 was measured: 7.6 against 8.4 ms per 57 344-tile tensor alone, but K1 loses the CU and `wq` Llama-3-8B got slower)
constexpr int kTagSlots = 16384;          // conflict detection: one byte counter per hashed position
constexpr int kJump = 34;                 // jump-ahead table entries (q = 0..33)

// Diagnostics: shader-clock ticks tensor 0 of the last launch spent per phase (mtq_debug_scan_ticks): 0 start-up + initial sums,
// 1 base pass + draw-only shuffle, then per later pass p (1..3): 2+3(p−1) candidates + shuffle, +1 deltas, +2 visits.
} }
__device__ unsigned long long g_scan_ticks[16];
namespace mtq { namespace {

struct U128 { uint64_t lo, hi; };
__host__ __device__ inline U128 mul128(U128 a, U128 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t h = __umul64hi(a.lo, b.lo);
#else
    const uint64_t h = (uint64_t)(((unsigned __int128)a.lo * b.lo) >> 64);
#endif
    return {a.lo * b.lo, h + a.lo * b.hi + a.hi * b.lo};
}
__host__ __device__ inline U128 add128(U128 a, U128 b)
{
    const uint64_t lo = a.lo + b.lo;
    return {lo, a.hi + b.hi + (lo < a.lo ? 1ull : 0ull)};
}
__host__ __device__ inline uint64_t pcg_out(U128 s)   // XSL-RR (numpy/random/src/pcg64/pcg64.h)
{
    const uint64_t x = s.hi ^ s.lo;
    const unsigned rot = (unsigned)(s.hi >> 58);
    return (x >> rot) | (x << ((64u - rot) & 63u));
}
constexpr uint64_t kMultHi = 0x2360ED051FC65DA4ull, kMultLo = 0x4385DF649FCCF645ull;

// SeedSequence(seed) → PCG64 state and increment (csrc/mtq_host.cpp rng_seed, numpy/random/bit_generator.pyx)
__device__ inline void seed_pcg(uint64_t seed, U128 &state, U128 &inc)
{
    uint32_t ent[2];
    int n_ent = 1;
    ent[0] = (uint32_t)seed;
    ent[1] = 0u;
    if (seed >> 32) { ent[1] = (uint32_t)(seed >> 32); n_ent = 2; }
    uint32_t hc = 0x43b0d7e5u;
    uint32_t pool[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t v = i < n_ent ? ent[i] : 0u;
        v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16;
        pool[i] = v;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (s != d) {
                uint32_t v = pool[s];
                v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16;
                uint32_t t = 0xca01f9ddu * pool[d] - 0x4973f715u * v;
                t ^= t >> 16;
                pool[d] = t;
            }
    uint32_t hb = 0x8b51f9ddu, w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { uint32_t v = pool[i & 3]; v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; w[i] = v; }
    const uint64_t u0 = w[0] | ((uint64_t)w[1] << 32), u1 = w[2] | ((uint64_t)w[3] << 32);
    const uint64_t u2 = w[4] | ((uint64_t)w[5] << 32), u3 = w[6] | ((uint64_t)w[7] << 32);
    const U128 initstate = {u1, u0}, initseq = {u3, u2};
    const U128 mult = {kMultLo, kMultHi};
    inc = {(initseq.lo << 1) | 1ull, (initseq.hi << 1) | (initseq.lo >> 63)};
    state = {0ull, 0ull};
    state = add128(mul128(state, mult), inc);
    state = add128(state, initstate);
    state = add128(mul128(state, mult), inc);
}

// loads / stores that are served by L2 (this wave re-reads what it wrote: never through the CU's vector L1)
template <typename T>
__device__ inline T ld_l2(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double ld_l2(const double *p) { return __longlong_as_double((long long)ld_l2(reinterpret_cast<const unsigned long long *>(p))); }
__device__ inline void mem_wait() { __builtin_amdgcn_s_waitcnt(0); __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

// keeps the compiler from moving memory operations across it; emits nothing (the hardware runs one wave's LDS operations in order)
__device__ inline void compiler_fence() { asm volatile("" ::: "memory"); }
__device__ inline uint64_t below(int lane) { return lane >= 64 ? ~0ull : ((1ull << lane) - 1ull); }
__device__ inline double shfl_f64(double v, int src)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const unsigned lo = (unsigned)__shfl((int)(unsigned)u, src, 64), hi = (unsigned)__shfl((int)(unsigned)(u >> 32), src, 64);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ inline uint64_t readlane_u64(uint64_t u, int src)   // src wave-uniform
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, src), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

__device__ inline double readlane_f64(double v, int src) { return __longlong_as_double((long long)readlane_u64((uint64_t)__double_as_longlong(v), src)); }

// The visiting order of one tensor: 16-bit ids in LDS (tiles <= kScanMaxTilesLds) or 32-bit ids in global scratch.
struct OrderLds {
    uint16_t *p;   // plain accesses between compiler fences: a volatile pointer loses its LDS address space and every access becomes a FLAT instruction with its own wait
    __device__ uint32_t get(uint32_t i) const { return p[i]; }
    __device__ void set(uint32_t i, uint32_t v) const { p[i] = (uint16_t)v; }
    __device__ void sync() const { __builtin_amdgcn_s_waitcnt(0xC07F); /* lgkmcnt(0) */ __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }
    // LDS operations of one wave execute in program order: reads before dependent writes need only the compiler's own wait for the
    // loaded value, and a later read sees an earlier write without any wait
    __device__ void sync_reads() const { compiler_fence(); }
    __device__ void sync_writes() const { compiler_fence(); }
};
struct OrderGlobal {
    uint32_t *p;
    __device__ uint32_t get(uint32_t i) const { return ld_l2(p + i); }
    __device__ void set(uint32_t i, uint32_t v) const { p[i] = v; }
    __device__ void sync() const { mem_wait(); }
    __device__ void sync_reads() const { mem_wait(); }    // global memory: loads and stores of one wave may complete out of order
    __device__ void sync_writes() const { mem_wait(); }
};

struct Rng {
    U128 state;       // wave-uniform
    bool has32;       // a buffered high half is the next draw
    uint32_t u32;
    U128 a0, c0, a1, c1;   // per lane: jump multiplier / increment term for the lane's output index without / with a pending half
};

// Generator.permutation(candidates) in place on `ord[0..n)` (kSwap) or just the generator's advance (the base pass, where the
// order does not matter: csrc/mtq_host.cpp rng_skip_shuffle).  → false if the batch budget ran out (cannot happen in practice:
// every batch accepts at least one draw with probability 1 − 2^−64).
#ifdef MTQ_SCAN_PROFILE   // tools: ticks of block 0 inside the shuffle batches, slots 12 (draws + acceptance), 13 (swaps), 14 (hand-back), 15 (batches)
#define MTQ_SHUF_TICK(slot) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && lane == 0) g_scan_ticks[slot] += now_ - tick_; tick_ = now_; } while (0)
#else
#define MTQ_SHUF_TICK(slot) do { } while (0)
#endif

template <bool kSwap, typename Order>
__device__ bool wave_shuffle(Rng &r, const Order &ord, int n, uint32_t *cnt, int lane)
{
    if (n < 2) return true;
#ifdef MTQ_SCAN_PROFILE
    unsigned long long tick_ = clock64();
#endif
    int i0 = n - 1;
    int budget = 8 * n + 4096;
    while (i0 >= 1) {
        if (--budget < 0) return false;
        const uint32_t mask = 0xFFFFFFFFu >> __builtin_clz((unsigned)i0);
        const int lo = (int)(mask >> 1);
        const int pend = r.has32 ? 1 : 0;
        // ---- this lane's stream position
        const U128 A = pend ? r.a1 : r.a0, C = pend ? r.c1 : r.c0;
        const U128 s = add128(mul128(A, r.state), C);
        const uint64_t o = pcg_out(s);
        const int half = (lane - pend) & 1;
        uint32_t raw = half ? (uint32_t)(o >> 32) : (uint32_t)o;
        if (pend && lane == 0) raw = r.u32;
        const uint32_t v = raw & mask;
        // ---- acceptance: v <= i0 − (accepted before me)
        uint64_t okmask = __ballot((int)v <= i0 - lane);
        uint64_t amb = ~(okmask | __ballot(v > (uint32_t)i0));
        while (amb) {   // wave-uniform loop over the few undecided lanes, in lane order
            const int a = __builtin_ctzll(amb);
            const int P = __builtin_popcountll(okmask & below(a));
            const uint32_t va = (uint32_t)__builtin_amdgcn_readlane((int)v, a);
            if ((int)va <= i0 - P) okmask |= 1ull << a;
            amb &= amb - 1ull;
        }
        const int kmax = i0 - lo;                  // steps this mask level still has (i must stay above lo; lo == 0 at the last level)
        const int total = __builtin_popcountll(okmask);
        int c = 64, k = total;
        if (total >= kmax) {                       // cut just after the kmax-th accepted position: the rest belongs to the next level
            uint64_t mm = okmask;
            for (int q = 1; q < kmax; ++q) mm &= mm - 1ull;
            c = __builtin_ctzll(mm) + 1;
            k = kmax;
            okmask &= below(c);
        }
        MTQ_SHUF_TICK(12);
        if (kSwap && k > 0) {
            const bool mine = (okmask >> lane) & 1ull;
            const int tstep = __builtin_popcountll(okmask & below(lane));
            const uint32_t pi = (uint32_t)(i0 - tstep), pj = v;
            const uint32_t hi_ = pi & (kTagSlots - 1), hj = pj & (kTagSlots - 1);
            // both operands of every step are fetched now: the reads travel while the conflict detection below runs
            uint32_t vi = 0, vj = 0;
            if (mine) { vi = ord.get(pi); vj = ord.get(pj); }
            // positions touched by two steps of the batch: every step adds 1 to the byte counters of its two (hashed) positions, reads
            // them back — 2 or more means the position is shared (or aliased: harmless, the step is just taken out of the parallel
            // part) — and takes its additions back.  LDS operations of one wave execute in program order: one wait, for the read-back.
            const uint32_t wi = hi_ >> 2, wj = hj >> 2, si = (hi_ & 3u) * 8u, sj = (hj & 3u) * 8u;
            if (mine) { atomicAdd(&cnt[wi], 1u << si); if (pj != pi) atomicAdd(&cnt[wj], 1u << sj); }
            compiler_fence();
            bool flagged = false;
            if (mine) {
                const uint32_t ci = cnt[wi], cj = cnt[wj];   // both reads issued, one wait
                const uint32_t fi = ((ci >> si) & 0xFFu) >= 2u ? 1u : 0u, fj = (((cj >> sj) & 0xFFu) >= 2u ? 1u : 0u) & (pj != pi ? 1u : 0u);
                flagged = (fi | fj) != 0u;                   // no short circuit: the second read must not wait for the first
            }
            compiler_fence();
            if (mine) { atomicSub(&cnt[wi], 1u << si); if (pj != pi) atomicSub(&cnt[wj], 1u << sj); }
            const uint64_t fmask = __ballot(flagged);
            ord.sync_reads();
            // steps that share a position with another step: resolved in registers, in step order.  Every such lane holds the current
            // values of its two positions (vi, vj: nothing else in the batch touches them); step f puts lane f's vj at its pi and its
            // vi at its pj, and every lane that holds one of those positions — lane f included — takes the new value over.
            uint64_t fm = fmask;
            while (fm) {
                const int f = __builtin_ctzll(fm);
                const uint32_t pif = (uint32_t)__builtin_amdgcn_readlane((int)pi, f), pjf = (uint32_t)__builtin_amdgcn_readlane((int)pj, f);
                const uint32_t vif = (uint32_t)__builtin_amdgcn_readlane((int)vi, f), vjf = (uint32_t)__builtin_amdgcn_readlane((int)vj, f);
                if (flagged) {
                    const uint32_t nvi = pi == pif ? vjf : (pi == pjf ? vif : vi);
                    const uint32_t nvj = pj == pif ? vjf : (pj == pjf ? vif : vj);
                    vi = nvi; vj = nvj;
                }
                fm &= fm - 1ull;
            }
            // write-back: a step that shares nothing swaps its two values; the others hold the final values of their positions
            if (mine) {
                if (flagged) { ord.set(pi, vi); ord.set(pj, vj); }
                else { ord.set(pi, vj); ord.set(pj, vi); }
            }
            ord.sync_writes();
        }
        MTQ_SHUF_TICK(13);
        i0 -= k;
        // ---- hand the unused positions back: the generator stands after the c-th position
        const int used = c - pend;
        if (used > 0) {
            const int src = c - 1;                 // wave-uniform: lane broadcasts, not LDS permutes
            r.state.lo = readlane_u64(s.lo, src);
            r.state.hi = readlane_u64(s.hi, src);
            r.has32 = (used & 1) != 0;
            r.u32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(o >> 32), src);
        } else {
            r.has32 = false;
        }
        MTQ_SHUF_TICK(14);
#ifdef MTQ_SCAN_PROFILE
        if (blockIdx.x == 0 && lane == 0) g_scan_ticks[15] += 1;
#endif
    }
    return true;
}

// Visiting orders that every tensor of a launch shares (mtq_scan_orders_device): all tensors of a model run are searched with ONE seed
// (mixed_tile_greedy.py:222-226: default_rng(self.seed) per tensor), and a pass's permutation depends on nothing but the generator state
// and the number of candidates — which is the tile count T as long as every earlier pass accepted every tile (the usual course of the
// first passes: bfp8 is accepted everywhere at any threshold a user would ask for).  So the draws of the base pass, pass 1's permutation
// of range(T) and — on the assumption that pass 1 accepts everything, checked by every tensor — pass 2's are computed once per launch
// instead of once per tensor.  A tensor whose course differs falls back to its own shuffle from the shared generator state.
struct alignas(16) OrdersHdr {
    U128 state[3];            // generator state after the base pass's draws, after pass 1's permutation, after pass 2's
    U128 inc;
    uint32_t has32[3], u32[3];
    uint32_t tiles, n_orders;
    uint64_t seed;
    uint32_t ok[2];           // written by the two blocks of the orders kernel: 1 = that order is complete
    uint32_t pad[4];
};
static_assert(sizeof(OrdersHdr) == 128, "orders header layout");
__host__ __device__ inline size_t orders_stride(int64_t tiles) { return (size_t)((tiles + 63) & ~(int64_t)63); }

// What phase 1 of a split search hands to phase 2, per tensor
struct Carry {
    double Sy, Sy2, Sxy, sum_x, sum_x2;
    U128 state, inc;
    uint32_t has32, u32;
    int32_t status, done;
};
static_assert(sizeof(Carry) == 88, "carry layout");

struct ScanArgs {
    const double *stats;      // [count][tiles][rec]
    int64_t tiles;
    int rec;
    int n_formats;
    int fmt[MTQ_NUM_TILE_FORMATS];       // format codes in search order
    int oy[MTQ_NUM_TILE_FORMATS], oy2[MTQ_NUM_TILE_FORMATS], oxy[MTQ_NUM_TILE_FORMATS];   // record offsets of Σy, Σy², Σxy BY FORMAT CODE (identity bf16: 0, 1, 1)
    int oab[MTQ_NUM_TILE_FORMATS];       // … of Σ|x−y| (mae metric); kNoOffset for the identity bf16, whose Σ|x−y| is 0
    int omx[MTQ_NUM_TILE_FORMATS];       // … of max|x−y| (atol metric); kNoOffset for the identity bf16
    int metric;                          // MTQ_METRIC_PCC, MTQ_METRIC_MAE (scan kernels) or MTQ_METRIC_ATOL (greedy_atol)
    double thr, n;
    const uint64_t *seeds;    // [count]
    int8_t *maps;             // [count][tiles] out
    int32_t *status;          // [count] out: 0 ok, 1 zero denominator (host scan needed), 2 internal budget exhausted
    int32_t *counts;          // [count][4] out (may be null): tiles per format code in the finished map
    uint32_t *order_g;        // [count][tiles] scratch (tiles > kScanMaxTilesLds)
    double *delta;            // [count][tiles][4] scratch: Δ(Σy, Σy², Σxy) of the visit and the tile's previous code
    double *delta2;           // the same for pass 2 under the shared orders (filled by the helper wave while pass 1 is visited)
    const U128 *jump_a, *jump_g;   // [kJump]
    const unsigned char *orders;   // OrdersHdr + two index arrays, or null
    int phase;                // 0: the whole search; 1: every pass but the last, then the last pass's candidates → listed; 2: the last pass
    uint32_t *listed, *n_listed;
    Carry *carry;             // [count] (phases 1 and 2)
};

constexpr int kNoOffset = 0xFF;
__device__ inline double rec_at(const double *rt, uint32_t off) { return off == (uint32_t)kNoOffset ? 0.0 : rt[off]; }

// Lanes 0..2: the running sum `s` plus the staged deltas of visits 0..kN−1, added one after the other as the sequential scan adds them;
// prefix i → lds_p[i][lane].  Sixteen staged values are read ahead of the additions, which stay one dependent chain.
template <int kN>
__device__ inline void prefix_chain(const double *lds_d, double *lds_p, int lane, double s)
{
    double v[16], w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = lds_d[u * 4 + lane];
#pragma unroll
    for (int i0 = 0; i0 < kN; i0 += 16) {
        if (i0 + 16 < kN) {
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = lds_d[(i0 + 16 + u) * 4 + lane];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { s = s + v[u]; lds_p[(i0 + u) * 4 + lane] = s; }
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = w[u];
    }
}

// The same chain run by EVERY lane (lane l ≥ 3 repeats a fourth, unused column: rows are [4] wide): no exec-masked region, so that the
// compiler can interleave its dependent additions with independent work of the same basic block (visit_pass, the accepting run).
__device__ inline void prefix_chain_all(const double *lds_d, double *lds_p, int col, double s)
{
    double v[16], w[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = lds_d[u * 4 + col];
#pragma unroll
    for (int i0 = 0; i0 < 64; i0 += 16) {
        if (i0 + 16 < 64) {
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = lds_d[(i0 + 16 + u) * 4 + col];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { s = s + v[u]; lds_p[(i0 + u) * 4 + col] = s; }
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = w[u];
    }
}

// pcc_hoisted of csrc/mtq_host.cpp (mixed_tile_greedy.py:176-190 with the x-only terms hoisted): the same operations in the
// same order.  special: zero denominator.
__device__ inline bool pcc_good(double n, double mean_x, double am2, double thr, double sy, double sy2, double sxy, bool &special)
{
    const double mean_y = sy / n;
    double bm2 = sy2 - n * mean_y * mean_y;
    if (bm2 < 0.0) bm2 = 0.0;
    const double denom = __builtin_sqrt(am2 * bm2);
    special = denom == 0.0;
    const double val = (sxy - n * mean_x * mean_y) / denom;
    return val >= thr;
}

#ifdef MTQ_SCAN_PROFILE
#define MTQ_SCAN_STAMP(slot) do { const unsigned long long now_ = clock64(); if (b == 0 && lane == 0 && (slot) < 16) g_scan_ticks[slot] = now_ - tick_s; tick_s = now_; } while (0)
#else
#define MTQ_SCAN_STAMP(slot) do { } while (0)
#endif

// The map, 64 tiles at a time in tile order, for the loops that sift it (candidates, the listed tiles, the final counts): f(t0, byte of
// tile t0 + lane — 0x80, "fixed", past the end).  Eight blocks' bytes are in flight at once: one block per L2 round trip (≈ 750 cycles)
// made each of these loops 670 k cycles on a 57 344-tile tensor — the two at the end of phase 1 and the two of phase 2 together more
// than a pass's visits (round 4, found by adding up the shader-clock stamps: tools/scan_ticks.py).
template <typename F>
__device__ __forceinline__ void sift_map(const int8_t *map, int T, int lane, F &&f)
{
    for (int t00 = 0; t00 < T; t00 += 64 * 8) {
        uint8_t b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = (uint8_t)ld_l2(map + min(t00 + 64 * u + lane, T - 1));   // unconditional (a branch per load would put each in a block of its own)
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = t00 + 64 * u + lane < T ? b[u] : (uint8_t)0x80u;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t00 + 64 * u < T) f(t00 + 64 * u, b[u]);
    }
}

struct Offsets3 { uint32_t y, y2, xy; };   // packed by format code, one byte each (rec <= 22): a per-lane code picks its offset with a shift

// Deltas of every visit of a pass in visiting order (mixed_tile_greedy.py:259-261; mae :293): the candidate format's sums minus those of
// the format the tile holds.  Four rounds of 64 gathers in flight (latency-bound).  tile_at(k) → tile of visit k; prev_of(tile) → its code.
template <bool kMae, typename TileAt, typename PrevOf>
__device__ inline void gather_deltas(const double *st, int rec, Offsets3 po, int f, int nc, double *delta, int lane, TileAt tile_at, PrevOf prev_of)
{
    const uint32_t oy_f = (po.y >> (8 * f)) & 0xFFu, oy2_f = (po.y2 >> (8 * f)) & 0xFFu, oxy_f = (po.xy >> (8 * f)) & 0xFFu;
    uint32_t nt[4];                                       // the tile ids of the NEXT 256 visits: their loads and this iteration's record gathers are in flight together
#pragma unroll                                            // (round 4: the helper wave's gather of a 57 344-tile pass 1.2 M → 0.95 M cycles, no longer behind the initial sums)
    for (int u = 0; u < 4; ++u) {
        const int k = 64 * u + lane;
        nt[u] = k < nc ? tile_at((uint32_t)k) : 0u;
    }
    for (int k0 = 0; k0 < nc; k0 += 256) {
        uint32_t tt[4];
        int pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) tt[u] = nt[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 256 + 64 * u + lane;
            nt[u] = k < nc ? tile_at((uint32_t)k) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) pv[u] = prev_of(tt[u]);
        double dd[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const double *rt = st + (int64_t)tt[u] * rec;
            const int sh = 8 * pv[u];                     // the previous format's offsets out of the packed tables: no indexed load
            if (kMae) {                                      // :293 — the candidate's Σ|x−y| minus the tile's current one
                dd[u][0] = rec_at(rt, oy_f) - rec_at(rt, (po.y >> sh) & 0xFFu);
                dd[u][1] = dd[u][2] = 0.0;
            } else {
                dd[u][0] = rt[oy_f] - rt[(po.y >> sh) & 0xFFu];   // :259-261
                dd[u][1] = rt[oy2_f] - rt[(po.y2 >> sh) & 0xFFu];
                dd[u][2] = rt[oxy_f] - rt[(po.xy >> sh) & 0xFFu];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + 64 * u + lane;
            if (k < nc) {
                double *d = delta + (int64_t)k * 4;
                d[0] = dd[u][0]; d[1] = dd[u][1]; d[2] = dd[u][2]; d[3] = __longlong_as_double((long long)pv[u]);
            }
        }
    }
    mem_wait();
}

// The visits of one pass (:234-278), 64 per round under a speculation on their outcome (file header).  → 0, or 1 when a zero
// denominator turned up.  any_rej: some visit of the pass was rejected (its tile is fixed).
template <bool kMae, typename TileAt>
__device__ inline int visit_pass(TileAt tile_at, const double *delta, int nc, int f, int8_t *map, double *lds_d, double *lds_p, int lane, double n,
                                 double mean_x, double am2, double thr, double &Sy, double &Sy2, double &Sxy, bool &any_rej)
{
    constexpr bool mae = kMae;
    int k = 0, pk = -1;
    double pdy = 0.0, pdy2 = 0.0, pdxy = 0.0;
    int pprev = 0;
    uint32_t pt = 0;
    bool accept_mode = true;
#ifdef MTQ_SCAN_PROFILE   // tools: ticks of block 0 inside the visit rounds, slots 8 (window fetch), 9 (staging + prefix chains), 10 (pcc), 11 (decision + map), 15 (rounds)
    unsigned long long vt_ = clock64();
#define MTQ_VISIT_TICK(slot) do { const unsigned long long now_ = clock64(); if (blockIdx.x == 0 && lane == 0) g_scan_ticks[slot] += now_ - vt_; vt_ = now_; } while (0)
#else
#define MTQ_VISIT_TICK(slot) do { } while (0)
#endif
    int span = 16;   // visits an accept-mode round stages: its three serial chains cost per visit staged, and a run of acceptances is short
                     // where rejections are frequent — 16 after a rejection, 64 after a round that accepted all of its visits
    double *lds_d2 = lds_p + 64 * 4 + 64 * 5, *lds_p2 = lds_d2 + 64 * 4;   // second staging / prefix blocks (behind lds_i) for the accepting run
    const int col = min(lane, 3);
    auto load_window = [&](int k0, double &wy, double &wy2, double &wxy, int &wprev, uint32_t &wt) {
        const double *d = delta + (int64_t)(k0 + lane) * 4;
        wy = ld_l2(d); wy2 = ld_l2(d + 1); wxy = ld_l2(d + 2);
        wprev = (int)__double_as_longlong(ld_l2(d + 3));
        wt = tile_at((uint32_t)(k0 + lane));
    };
    while (k < nc) {
        // ---- the accepting run: while whole rounds of 64 visits are accepted — all of pass 1 as a rule, and pass 2 until the metric reaches
        // the threshold — the prefix chains of round r+1 (started from round r's last prefix, i.e. on the assumption that round r accepts
        // everything) are computed beside round r's 64 pcc evaluations: two chains of dependent float64 operations in one basic block instead
        // of one after the other.  A round is only ever committed whole from here; at the first rejection (or zero denominator) nothing of the
        // round has been applied and the general code below takes it again, visit by visit as before.  Same operations, same order.
        if (accept_mode && span == 64 && nc - k >= 64) {
            double c_y, c_y2, c_xy, n_y = 0.0, n_y2 = 0.0, n_xy = 0.0, f_y = 0.0, f_y2 = 0.0, f_xy = 0.0;
            int c_prev, n_prev = 0, f_prev = 0;
            uint32_t c_t, n_t = 0, f_t = 0;
            if (pk == k) { c_y = pdy; c_y2 = pdy2; c_xy = pdxy; c_prev = pprev; c_t = pt; }
            else load_window(k, c_y, c_y2, c_xy, c_prev, c_t);
            bool has_next = nc - (k + 64) >= 64;
            if (has_next) load_window(k + 64, n_y, n_y2, n_xy, n_prev, n_t);
            double *dA = lds_d, *pA = lds_p, *dB = lds_d2, *pB = lds_p2;
            dA[lane * 4 + 0] = c_y; dA[lane * 4 + 1] = c_y2; dA[lane * 4 + 2] = c_xy;
            compiler_fence();
            prefix_chain_all(dA, pA, col, col == 0 ? Sy : (col == 1 ? Sy2 : Sxy));
            compiler_fence();
            bool bailed = false;
            for (;;) {
                const bool has_far = has_next && nc - (k + 128) >= 64;
                if (has_far) load_window(k + 128, f_y, f_y2, f_xy, f_prev, f_t);   // two rounds ahead: consumed one iteration from now
                dB[lane * 4 + 0] = n_y; dB[lane * 4 + 1] = n_y2; dB[lane * 4 + 2] = n_xy;   // (zeros when there is no next round: the chain below then runs for nothing,
                compiler_fence();                                                  //  but unconditionally — a branch would put it in a basic block of its own)
                const double start = pA[63 * 4 + col];                             // the sums after this round's 64th visit, if all are accepted
                const double cy = pA[lane * 4 + 0], cy2 = pA[lane * 4 + 1], cxy = pA[lane * 4 + 2];
                prefix_chain_all(dB, pB, col, start);
                bool special = false;
                const bool good = mae ? cy / n <= thr : pcc_good(n, mean_x, am2, thr, cy, cy2, cxy, special);
                const uint64_t okm = __ballot(good), spm = __ballot(special);
                compiler_fence();
                if (okm != ~0ull || spm != 0ull) { bailed = true; break; }          // not a whole accepted round: the general code takes it
                map[c_t] = (int8_t)f;                                              // accepted (:264-276), all 64 of them
                Sy = readlane_f64(cy, 63); Sy2 = readlane_f64(cy2, 63); Sxy = readlane_f64(cxy, 63);
                k += 64;
                if (!has_next) break;
                c_y = n_y; c_y2 = n_y2; c_xy = n_xy; c_prev = n_prev; c_t = n_t;
                n_y = f_y; n_y2 = f_y2; n_xy = f_xy; n_prev = f_prev; n_t = f_t;
                has_next = has_far;
                double *tq = dA; dA = dB; dB = tq; tq = pA; pA = pB; pB = tq;
            }
            if (bailed) { pk = k; pdy = c_y; pdy2 = c_y2; pdxy = c_xy; pprev = c_prev; pt = c_t; }   // the round in hand is the general code's next window
            else pk = -1;
            if (k >= nc) break;
            if (!bailed) continue;
        }
        // ---- the rejecting run: once the metric sits at the threshold nearly every visit of a pass is rejected (pass 2 of the headline case:
        // 13 800 of 16 384 visits behind the 2 500 accepted ones), and a rejected visit changes nothing — so two windows (128 visits) are
        // decided per step against the same sums, every lane two independent pcc evaluations instead of one, and the windows of the next
        // step are in flight meanwhile.  Committed only when all 128 are rejected (and none has a zero denominator); otherwise nothing
        // has been applied and the general code takes the round visit by visit, the first window in hand.
        if (!accept_mode && nc - k >= 128) {
            double a_y, a_y2, a_xy, b_y, b_y2, b_xy, na_y = 0.0, na_y2 = 0.0, na_xy = 0.0, nb_y = 0.0, nb_y2 = 0.0, nb_xy = 0.0;
            int a_prev, b_prev, na_prev = 0, nb_prev = 0;
            uint32_t a_t, b_t, na_t = 0, nb_t = 0;
            if (pk == k) { a_y = pdy; a_y2 = pdy2; a_xy = pdxy; a_prev = pprev; a_t = pt; }
            else load_window(k, a_y, a_y2, a_xy, a_prev, a_t);
            load_window(k + 64, b_y, b_y2, b_xy, b_prev, b_t);
            bool bailed = false;
            for (;;) {
                const bool has_next = nc - (k + 128) >= 128;
                const int kn = has_next ? k + 128 : k;                                   // unconditional loads (see the accepting run)
                load_window(kn, na_y, na_y2, na_xy, na_prev, na_t);
                load_window(kn + 64, nb_y, nb_y2, nb_xy, nb_prev, nb_t);
                bool sp_a = false, sp_b = false;
                const double ay = Sy + a_y, ay2 = Sy2 + a_y2, axy = Sxy + a_xy, by = Sy + b_y, by2 = Sy2 + b_y2, bxy = Sxy + b_xy;
                const bool good_a = mae ? ay / n <= thr : pcc_good(n, mean_x, am2, thr, ay, ay2, axy, sp_a);
                const bool good_b = mae ? by / n <= thr : pcc_good(n, mean_x, am2, thr, by, by2, bxy, sp_b);
                if (__ballot(good_a || good_b || sp_a || sp_b) != 0ull) { bailed = true; break; }
                map[a_t] = (int8_t)(a_prev | 0x80);                                     // fixed (:277-278), all 128 of them
                map[b_t] = (int8_t)(b_prev | 0x80);
                any_rej = true;
                k += 128;
                if (!has_next) break;
                a_y = na_y; a_y2 = na_y2; a_xy = na_xy; a_prev = na_prev; a_t = na_t;
                b_y = nb_y; b_y2 = nb_y2; b_xy = nb_xy; b_prev = nb_prev; b_t = nb_t;
            }
            if (bailed) { pk = k; pdy = a_y; pdy2 = a_y2; pdxy = a_xy; pprev = a_prev; pt = a_t; }   // the first window is the general code's next one
            else pk = -1;
            if (k >= nc) break;
        }
        const int m = min(accept_mode ? span : 64, nc - k);
        const bool active = lane < m;
        double dy = 0.0, dy2 = 0.0, dxy = 0.0;
        int prev = 0;
        uint32_t t = 0;
        if (pk == k) {   // the window fetched ahead (the previous round consumed all of its visits)
            dy = pdy; dy2 = pdy2; dxy = pdxy; prev = pprev; t = pt;
        } else if (active) {
            const double *d = delta + (int64_t)(k + lane) * 4;
            dy = ld_l2(d); dy2 = ld_l2(d + 1); dxy = ld_l2(d + 2);
            prev = (int)__double_as_longlong(ld_l2(d + 3));
            t = tile_at((uint32_t)(k + lane));
        }
        pk = k + m;      // fetch the window after this one while this one is decided (useless only when the speculation fails)
        pdy = pdy2 = pdxy = 0.0; pprev = 0; pt = 0;
        if (pk + lane < nc) {
            const double *d = delta + (int64_t)(pk + lane) * 4;
            pdy = ld_l2(d); pdy2 = ld_l2(d + 1); pdxy = ld_l2(d + 2);
            pprev = (int)__double_as_longlong(ld_l2(d + 3));
            pt = tile_at((uint32_t)(pk + lane));
        }
        MTQ_VISIT_TICK(8);
        double cy, cy2, cxy;
        if (accept_mode) {   // lane i: the running sums after visits 0..i, added one after the other as the sequential scan adds them
            lds_d[lane * 4 + 0] = dy; lds_d[lane * 4 + 1] = dy2; lds_d[lane * 4 + 2] = dxy;
            compiler_fence();   // one wave: its LDS operations execute in program order
            if (lane < 3) {   // inactive visits staged +0 deltas: their prefixes are never read
                const double s0 = lane == 0 ? Sy : (lane == 1 ? Sy2 : Sxy);
                if (span == 16) prefix_chain<16>(lds_d, lds_p, lane, s0);   // both fully unrolled: a run-time bound cost more than the short chain saves
                else prefix_chain<64>(lds_d, lds_p, lane, s0);
            }
            compiler_fence();
            cy = lds_p[lane * 4 + 0]; cy2 = lds_p[lane * 4 + 1]; cxy = lds_p[lane * 4 + 2];
            compiler_fence();
        } else {             // lane i: the running sums plus its own delta
            cy = Sy + dy; cy2 = Sy2 + dy2; cxy = Sxy + dxy;
        }
        MTQ_VISIT_TICK(9);
        bool special = false;
        const bool good = mae ? cy / n <= thr : pcc_good(n, mean_x, am2, thr, cy, cy2, cxy, special);   // mae: is_good(cab / N) (:293-294)
        const uint64_t act = below(m);
        const uint64_t okm = __ballot(good && active) & act, spm = __ballot(special && active) & act;
        MTQ_VISIT_TICK(10);
        int j, take = -1;
        if (accept_mode) {
            const uint64_t rej = ~okm & act;
            j = rej ? __builtin_ctzll(rej) : m;            // first rejected visit
            if (spm & below(min(j + 1, 64))) return 1;      // a zero denominator among the visits this round settles
            if (lane < j) map[t] = (int8_t)f;              // accepted (:264-276)
            if (j > 0) take = j - 1;
            if (j < m) { if (lane == j) map[t] = (int8_t)(prev | 0x80); any_rej = true; k += j + 1; span = 16; if (j == 0) accept_mode = false; }   // fixed (:277-278)
            else { k += m; span = 64; }
        } else {
            j = okm ? __builtin_ctzll(okm) : m;            // first accepted visit
            if (spm & below(min(j + 1, 64))) return 1;
            if (lane < j) map[t] = (int8_t)(prev | 0x80);
            if (j > 0) any_rej = true;
            if (j < m) { if (lane == j) map[t] = (int8_t)f; take = j; k += j + 1; if (j == 0) { accept_mode = true; span = 16; } }
            else k += m;
        }
        if (take >= 0) { Sy = readlane_f64(cy, take); Sy2 = readlane_f64(cy2, take); Sxy = readlane_f64(cxy, take); }   // take is wave-uniform
        MTQ_VISIT_TICK(11);
#ifdef MTQ_SCAN_PROFILE
        if (blockIdx.x == 0 && lane == 0) g_scan_ticks[15] += 1;
#endif
    }
    mem_wait();
    return 0;
}

template <bool kMae, typename Order>
__device__ __forceinline__ void scan_tensor(const ScanArgs &a, const Order &ord, int b, unsigned char *lds, int lane, int wave, bool two_waves)
{
    const int T = (int)a.tiles;
    const int rec = a.rec;
    const double *st = a.stats + (int64_t)b * a.tiles * rec;
    int8_t *map = a.maps + (int64_t)b * a.tiles;
    double *delta = a.delta + (int64_t)b * a.tiles * 4;
    double *delta2 = a.delta2 + (int64_t)b * a.tiles * 4;
    double *lds_d = reinterpret_cast<double *>(lds);                 // [64][4] deltas / staging
    double *lds_p = lds_d + 64 * 4;                                    // [64][4] prefixes
    double *lds_i = lds_p + 64 * 4;                                    // [64][5] initial-sum staging; behind it [64][4] + [64][4]: visit_pass's second staging / prefix blocks
    uint32_t *cnt = reinterpret_cast<uint32_t *>(lds_i + 64 * 5 + 64 * 4 + 64 * 4);   // [kTagSlots / 4] byte counters of the shuffle's conflict detection
    uint32_t *helper_flag = cnt + kTagSlots / 4;                       // [2] set by the helper wave when the deltas of pass 1 / pass 2 are in memory
    constexpr bool mae = kMae;   // compiled per metric (a run-time flag cost the pcc scan 9 %): one running sum, Σ|x−y| (:280-301), carried where the pcc search carries Σy; Σy², Σxy idle
    const int base = a.fmt[0];
    // record offsets by format code, one byte each (rec <= 22): a per-lane code picks its offset with a shift, where indexing the
    // argument arrays by a per-lane value makes every gather wait for a table load
    const int *o1 = mae ? a.oab : a.oy;
    Offsets3 po;
    po.y = (uint32_t)o1[0] | ((uint32_t)o1[1] << 8) | ((uint32_t)o1[2] << 16) | ((uint32_t)o1[3] << 24);
    po.y2 = (uint32_t)a.oy2[0] | ((uint32_t)a.oy2[1] << 8) | ((uint32_t)a.oy2[2] << 16) | ((uint32_t)a.oy2[3] << 24);
    po.xy = (uint32_t)a.oxy[0] | ((uint32_t)a.oxy[1] << 8) | ((uint32_t)a.oxy[2] << 16) | ((uint32_t)a.oxy[3] << 24);

    // passes p_begin .. p_end−1 run in this launch; passes 1 .. n_sh may take their visiting order from the launch's shared orders
    const int p_end = a.phase == 1 ? a.n_formats - 1 : a.n_formats;
    const OrdersHdr *hdr = reinterpret_cast<const OrdersHdr *>(a.orders);
    int n_sh = 0;
    // … if they were drawn for this tensor's tile count AND seed (include/mtq.h asks the caller for that; a tensor whose seed differs shuffles for itself)
    if (hdr && a.phase != 2 && hdr->tiles == (uint32_t)T && hdr->seed == (uint64_t)a.seeds[b] && hdr->ok[0]) n_sh = (hdr->n_orders >= 2 && hdr->ok[1]) ? 2 : 1;
    n_sh = min(n_sh, p_end - 1);
    const uint32_t *P1 = reinterpret_cast<const uint32_t *>(a.orders + sizeof(OrdersHdr)), *P2 = P1 + orders_stride(T);

    // tile of visit k: out of the launch's shared order Pg, or out of this tensor's own order
    const uint32_t *Pg = nullptr;
    auto tile_at = [&](uint32_t k) -> uint32_t { return Pg ? Pg[k] : ord.get(k); };
    // code a tile holds when a pass comes to it: a known format (prev_code >= 0: the helper wave's speculation) or what the map says
    int prev_code = -1;
    auto prev_of = [&](uint32_t t) -> int { return prev_code >= 0 ? prev_code : (int)((uint8_t)ld_l2(map + t) & 0x7Fu); };

    if (two_waves) {   // both waves: the flags start at zero before either of them is set or read
        if (wave == 0 && lane < 2) helper_flag[lane] = 0u;
        __syncthreads();
    }
    auto wait_helper = [&](int p) {   // the visiting wave: until the helper wave has stored the deltas of pass p (it always gets there: it waits for nothing)
        while (__hip_atomic_load(helper_flag + (p - 1), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u) __builtin_amdgcn_s_sleep(8);
    };
    if (wave == 1) {
        // Helper wave: the deltas of the passes that use the shared orders, speculating that pass p finds every tile in format fmt[p−1]
        // (true while every earlier pass accepted every tile — the visiting wave checks, and gathers its own deltas otherwise).  Pass 1's
        // are gathered while the visiting wave forms the initial sums, pass 2's while it visits pass 1.
#pragma nounroll
        for (int p = 1; p <= 2; ++p) {
            if (p <= n_sh) {
                Pg = p == 1 ? P1 : P2;
                prev_code = a.fmt[p - 1];
                gather_deltas<kMae>(st, rec, po, a.fmt[p], T, p == 1 ? delta : delta2, lane, tile_at, prev_of);   // returns with its stores complete
            }
            // "pass p's deltas are in memory": a word in LDS the visiting wave polls — not a barrier, so that this wave can end (and give its
            // registers back to K1's waves) as soon as its gathers are done instead of waiting for the visiting wave to reach a rendezvous
            if (lane == 0) __hip_atomic_store(helper_flag + (p - 1), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }

    int status = 0;
    bool done = false;            // the search is over before the launch's last pass: nothing but the finish is left
#ifdef MTQ_SCAN_PROFILE
    if (b == 0 && lane == 0) for (int q = 8; q < 16; ++q) g_scan_ticks[q] = 0;
    unsigned long long tick_s = clock64();
#endif
    for (int i = lane; i < kTagSlots / 4; i += 64) cnt[i] = 0u;
    compiler_fence();

    Rng r;
    U128 inc;
    double sum_x = 0.0, sum_x2 = 0.0, Sy = 0.0, Sy2 = 0.0, Sxy = 0.0;
    const double n = a.n, thr = a.thr;
    auto set_jumps = [&]() {
        const int q0 = (lane >> 1) + 1, q1 = lane >= 1 ? ((lane - 1) >> 1) + 1 : 1;
        r.a0 = a.jump_a[q0]; r.c0 = mul128(a.jump_g[q0], inc);
        r.a1 = a.jump_a[q1]; r.c1 = mul128(a.jump_g[q1], inc);
    };
    int p_begin = 1;

    if (a.phase != 2) {
        for (int t = lane; t < T; t += 64) map[t] = (int8_t)base;        // :99
        // ---- generator
        if (n_sh >= 1) { inc = hdr->inc; r.state = hdr->state[0]; r.has32 = hdr->has32[0] != 0u; r.u32 = hdr->u32[0]; }   // as it stands after the base pass's draws
        else { seed_pcg(a.seeds[b], r.state, inc); r.has32 = false; r.u32 = 0u; }
        set_jumps();

        // ---- initial sums in tile order (:147-174): chains Σx, Σx², Σy, Σy², Σxy on lanes 0..4
        const int off5[5] = {0, 1, mae ? a.oab[base] : a.oy[base], a.oy2[base], a.oxy[base]};
        // The chains' additions are ≈ 6 cycles each; what the wave waits for is the records: 136-byte strided gathers out of HBM or the Infinity
        // Cache (K1 wrote them a batch ago), as many in flight as the wave holds registers for.  kNc chains, kAhead blocks of 64 tiles ahead:
        //   any base format: five values per tile, four blocks ahead (22 cycles per tile on a 57 344-tile tensor, tools/scan_ticks.py);
        //   the identity bf16 base of bf16 storage (round 4): Σy, Σy², Σxy are the SAME additions of the same values as Σx, Σx², Σx²
        //   (mtq_decide.hpp load5) — two values per tile, ten blocks ahead, three chains copied from the two.
        auto init_sums = [&](auto nc_tag, auto ahead_tag, const int *offs) -> double {
            constexpr int kNc = decltype(nc_tag)::value, kAhead = decltype(ahead_tag)::value;
            double acc = 0.0;
            double nx[kAhead][kNc];
#pragma unroll
            for (int r = 0; r < kAhead; ++r) {
#pragma unroll
                for (int c = 0; c < kNc; ++c) nx[r][c] = 0.0;
                if (64 * r + lane < T) {
                    const double *rt = st + (int64_t)(64 * r + lane) * rec;
#pragma unroll
                    for (int c = 0; c < kNc; ++c) nx[r][c] = rec_at(rt, (uint32_t)offs[c]);
                }
            }
            for (int t00 = 0; t00 < T; t00 += 64 * kAhead) {
#pragma unroll
                for (int r = 0; r < kAhead; ++r) {
                    const int t0 = t00 + 64 * r;
                    if (t0 < T) {
                        const int m = min(64, T - t0);
                        if (lane < m) {
#pragma unroll
                            for (int c = 0; c < kNc; ++c) lds_i[lane * kNc + c] = nx[r][c];
                        }
                        if (t0 + 64 * kAhead + lane < T) {   // the block kAhead ahead takes this block's registers
                            const double *rt = st + (int64_t)(t0 + 64 * kAhead + lane) * rec;
#pragma unroll
                            for (int c = 0; c < kNc; ++c) nx[r][c] = rec_at(rt, (uint32_t)offs[c]);
                        }
                        compiler_fence();
                        if (lane < kNc) {
                            if (m == 64) {   // sixteen staged values ahead: the additions stay one dependent chain in tile order, the LDS reads do not wait for it
                                double v[16], w[16];
#pragma unroll
                                for (int u = 0; u < 16; ++u) v[u] = lds_i[u * kNc + lane];
#pragma unroll
                                for (int i0 = 0; i0 < 64; i0 += 16) {
                                    if (i0 + 16 < 64) {
#pragma unroll
                                        for (int u = 0; u < 16; ++u) w[u] = lds_i[(i0 + 16 + u) * kNc + lane];
                                    }
#pragma unroll
                                    for (int u = 0; u < 16; ++u) acc = acc + v[u];
#pragma unroll
                                    for (int u = 0; u < 16; ++u) v[u] = w[u];
                                }
                            } else {
                                for (int i = 0; i < m; ++i) acc = acc + lds_i[i * kNc + lane];
                            }
                        }
                        compiler_fence();
                    }
                }
            }
            return acc;
        };
        const bool identity_base = !mae && off5[2] == 0 && off5[3] == 1 && off5[4] == 1;
        double acc;
        if (identity_base) {
            acc = init_sums(std::integral_constant<int, 2>{}, std::integral_constant<int, 10>{}, off5);
            const double s1 = shfl_f64(acc, 1), s0 = shfl_f64(acc, 0);
            acc = lane == 0 || lane == 2 ? s0 : s1;      // lanes 2, 3, 4: Σy = Σx, Σy² = Σxy = Σx²
        } else {
            acc = init_sums(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, off5);
        }
        sum_x = shfl_f64(acc, 0); sum_x2 = shfl_f64(acc, 1);
        Sy = shfl_f64(acc, 2); Sy2 = shfl_f64(acc, 3); Sxy = shfl_f64(acc, 4);
    } else {
        const Carry &c = a.carry[b];
        Sy = c.Sy; Sy2 = c.Sy2; Sxy = c.Sxy; sum_x = c.sum_x; sum_x2 = c.sum_x2;
        r.state = c.state; inc = c.inc; r.has32 = c.has32 != 0u; r.u32 = c.u32;
        status = c.status; done = c.done != 0;
        set_jumps();
        p_begin = a.n_formats - 1;
    }
    const double mean_x = sum_x / n;
    double am2 = sum_x2 - n * mean_x * mean_x;
    if (am2 < 0.0) am2 = 0.0;
    MTQ_SCAN_STAMP(0);

    if (a.phase != 2) {
        // ---- pass of the base format: every tile already has it, one question for all of them (:237-241)
        bool special = false;
        const bool good = mae ? Sy / n <= thr : pcc_good(n, mean_x, am2, thr, Sy, Sy2, Sxy, special);   // mae: is_good(sum_abs / N) (:280-284)
        if (special) status = 1;
        done = !good;                                                  // every tile is fixed at once: no later pass has a candidate
        if (n_sh == 0 && !wave_shuffle<false>(r, ord, T, cnt, lane)) status = 2;   // the generator advances as the permutation would have
        if (two_waves) wait_helper(1);                                  // the helper wave's deltas of pass 1 are in memory
    }
    MTQ_SCAN_STAMP(1);

    bool pristine = a.phase != 2;     // every pass so far accepted every tile: the next pass's candidates are all T tiles, in tile order
    for (int p = p_begin; p < p_end && status == 0 && !done; ++p) {
        const int f = a.fmt[p];
        if (p == 2 && two_waves && n_sh >= 2) wait_helper(2);             // the helper wave's deltas of pass 2 are in memory
        bool any_rej = false;
        const bool use_sh = pristine && p <= n_sh;
        int nc = T;
        double *dl = delta;
        if (use_sh) {
            // order = rng.permutation(arange(T)) from the shared state: computed once per launch; the deltas came from the helper wave
            Pg = p == 1 ? P1 : P2;
            dl = p == 1 ? delta : delta2;
            MTQ_SCAN_STAMP(2 + 3 * (p - 1));
        } else {
            Pg = nullptr;
            // candidates = np.where(~fixed)[0] (:228): in tile order
            nc = 0;
            sift_map(map, T, lane, [&](int t0, uint8_t code) {
                const bool cand = (code & 0x80u) == 0u;
                const uint64_t bm = __ballot(cand);
                if (cand) ord.set((uint32_t)(nc + __builtin_popcountll(bm & below(lane))), (uint32_t)(t0 + lane));
                nc += __builtin_popcountll(bm);
            });
            ord.sync();
            if (nc == 0) break;                                            // :229-230
            if (!wave_shuffle<true>(r, ord, nc, cnt, lane)) { status = 2; break; }   // order = rng.permutation(candidates), :231
            MTQ_SCAN_STAMP(2 + 3 * (p - 1));
            // deltas of every visit of the pass (a tile is visited once per pass, so its previous format is what the map holds now)
            gather_deltas<kMae>(st, rec, po, f, nc, delta, lane, tile_at, prev_of);
        }
        MTQ_SCAN_STAMP(3 + 3 * (p - 1));
        status = visit_pass<kMae>(tile_at, dl, nc, f, map, lds_d, lds_p, lane, n, mean_x, am2, thr, Sy, Sy2, Sxy, any_rej);
        if (use_sh) { r.state = hdr->state[p]; r.has32 = hdr->has32[p] != 0u; r.u32 = hdr->u32[p]; }   // the generator as it stands after this pass's permutation
        if (any_rej) pristine = false;
        MTQ_SCAN_STAMP(4 + 3 * (p - 1));
    }
    mem_wait();

    if (a.phase == 1) {
        // the last pass's candidates (:228) → the launch's list of tiles whose remaining statistics are evaluated now (mtq_tile_stats_listed)
        int nc = 0;
        if (status == 0 && !done) {
            sift_map(map, T, lane, [&](int, uint8_t code) { nc += __builtin_popcountll(__ballot((code & 0x80u) == 0u)); });
            unsigned first = 0u;
            if (lane == 0 && nc > 0) first = atomicAdd(a.n_listed, (unsigned)nc);
            first = (unsigned)__builtin_amdgcn_readfirstlane((int)first);
            int at = 0;
            if (nc > 0)
                sift_map(map, T, lane, [&](int t0, uint8_t code) {
                    const bool cand = (code & 0x80u) == 0u;
                    const uint64_t bm = __ballot(cand);
                    if (cand) a.listed[first + (unsigned)(at + __builtin_popcountll(bm & below(lane)))] = (uint32_t)((int64_t)b * T + t0 + lane);
                    at += __builtin_popcountll(bm);
                });
        }
        if (lane == 0) {
            Carry &c = a.carry[b];
            c.Sy = Sy; c.Sy2 = Sy2; c.Sxy = Sxy; c.sum_x = sum_x; c.sum_x2 = sum_x2;
            c.state = r.state; c.inc = inc; c.has32 = r.has32 ? 1u : 0u; c.u32 = r.u32;
            c.status = status; c.done = (done || nc == 0) ? 1 : 0;       // no candidate: :229-230
        }
        return;
    }

    int per_fmt[MTQ_NUM_TILE_FORMATS] = {0, 0, 0, 0};
    sift_map(map, T, lane, [&](int t0, uint8_t raw) {
        const int t = t0 + lane;
        int code = -1;
        if (t < T) { code = (int)(raw & 0x7Fu); map[t] = (int8_t)code; }
#pragma unroll
        for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) per_fmt[c] += __builtin_popcountll(__ballot(code == c));
    });
    if (lane == 0) {
        a.status[b] = status;
        if (a.counts)
            for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) a.counts[b * MTQ_NUM_TILE_FORMATS + c] = per_fmt[c];
    }
}

// The atol search needs no scan.  Its global metric is max over tiles of max|x−y|, so a replacement is accepted iff the new tile maximum
// and every OTHER tile's current maximum are within the threshold (mixed_tile_greedy.py:305-344 keeps that maximum with a multiplicity and
// an occasional recount; for finite values the candidate it forms IS max(new, others)).  If any tile's maximum under the first format
// exceeds the threshold, the first format's own pass fixes every tile (:305-309) and the map stays at the first format; otherwise every
// other tile is always within the threshold and tile t walks down the format list while its own maximum stays within it — whatever the
// visiting order, so no generator either (checked against the reference's three atol maps of F4 and, seed after seed, against the
// host scan).  A NaN among the maxima leaves that equivalence: status 1, the host scan takes the tensor.
__global__ __launch_bounds__(256) void greedy_atol(ScanArgs a)
{
    __shared__ int s_bad, s_nan, s_cnt[MTQ_NUM_TILE_FORMATS];
    const int b = blockIdx.x;
    const int64_t T = a.tiles;
    const double *st = a.stats + (int64_t)b * T * a.rec;
    int8_t *map = a.maps + (int64_t)b * T;
    if (threadIdx.x == 0) { s_bad = 0; s_nan = 0; }
    if (threadIdx.x < MTQ_NUM_TILE_FORMATS) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const int base = a.fmt[0];
    bool bad = false, nan = false;
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        const double *rt = st + t * a.rec;
        bad |= !(rec_at(rt, (uint32_t)a.omx[base]) <= a.thr);
        for (int p = 0; p < a.n_formats; ++p) { const double v = rec_at(rt, (uint32_t)a.omx[a.fmt[p]]); nan |= v != v; }
    }
    if (bad) s_bad = 1;
    if (nan) s_nan = 1;
    __syncthreads();
    const bool all_base = s_bad != 0;
    int mine[MTQ_NUM_TILE_FORMATS] = {0, 0, 0, 0};
    for (int64_t t = threadIdx.x; t < T; t += 256) {
        const double *rt = st + t * a.rec;
        int code = base;
        if (!all_base)
            for (int p = 1; p < a.n_formats; ++p) {
                if (!(rec_at(rt, (uint32_t)a.omx[a.fmt[p]]) <= a.thr)) break;   // rejected: fixed at the format it has (:277-278)
                code = a.fmt[p];
            }
        map[t] = (int8_t)code;
#pragma unroll
        for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) mine[c] += code == c;
    }
#pragma unroll
    for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) if (mine[c]) atomicAdd(&s_cnt[c], mine[c]);
    __syncthreads();
    if (threadIdx.x == 0) {
        a.status[b] = s_nan ? 1 : 0;
        if (a.counts)
            for (int c = 0; c < MTQ_NUM_TILE_FORMATS; ++c) a.counts[b * MTQ_NUM_TILE_FORMATS + c] = s_cnt[c];
    }
}

// One block per tensor: wave 0 searches; wave 1 (launched when the tensors share their visiting orders) gathers deltas ahead of it.
#ifndef MTQ_SCAN_SETPRIO
#define MTQ_SCAN_SETPRIO 0
#endif
#ifdef MTQ_SCAN_WAVES_PER_EU   // experiments: a register budget for the search kernels (3 → 168 VGPRs, 4 → 128), tools/r3_env_ab.sh with MTQ_LIB
#define MTQ_SCAN_BUDGET __attribute__((amdgpu_waves_per_eu(MTQ_SCAN_WAVES_PER_EU, MTQ_SCAN_WAVES_PER_EU)))
#else
#define MTQ_SCAN_BUDGET
#endif

__global__ __launch_bounds__(128) MTQ_SCAN_BUDGET void greedy_scan_pcc_lds(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (MTQ_SCAN_SETPRIO) __builtin_amdgcn_s_setprio(MTQ_SCAN_SETPRIO);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    OrderLds ord{reinterpret_cast<uint16_t *>(lds)};
    if (a.metric == MTQ_METRIC_MAE) scan_tensor<true>(a, ord, blockIdx.x, lds + 2 * ((a.tiles + 7) & ~(int64_t)7), lane, wave, blockDim.x == 128);
    else scan_tensor<false>(a, ord, blockIdx.x, lds + 2 * ((a.tiles + 7) & ~(int64_t)7), lane, wave, blockDim.x == 128);
}

__global__ __launch_bounds__(128) MTQ_SCAN_BUDGET void greedy_scan_pcc_global(ScanArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    if (MTQ_SCAN_SETPRIO) __builtin_amdgcn_s_setprio(MTQ_SCAN_SETPRIO);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    OrderGlobal ord{a.order_g + (int64_t)blockIdx.x * a.tiles};
    if (a.metric == MTQ_METRIC_MAE) scan_tensor<true>(a, ord, blockIdx.x, lds, lane, wave, blockDim.x == 128);
    else scan_tensor<false>(a, ord, blockIdx.x, lds, lane, wave, blockDim.x == 128);
}

constexpr size_t kFixedLds = (64 * 4 + 64 * 4 + 64 * 5 + 64 * 4 + 64 * 4) * sizeof(double) + kTagSlots + 16;

// The launch's shared orders: block w (one wave) computes order w+1.  Block 0: the base pass's draws, then pass 1's permutation of
// range(T); block 1: the draws of the base pass and of pass 1, then pass 2's permutation of range(T).
struct OrdersArgs {
    uint64_t seed;
    int64_t tiles;
    int n_orders;
    unsigned char *out;
    const U128 *jump_a, *jump_g;
};

template <typename Order>
__device__ void make_order(const OrdersArgs &a, const Order &ord, uint32_t *cnt, uint32_t *P, int which, int lane, bool copy_out)
{
    const int T = (int)a.tiles;
    OrdersHdr *hdr = reinterpret_cast<OrdersHdr *>(a.out);
    for (int i = lane; i < kTagSlots / 4; i += 64) cnt[i] = 0u;
    compiler_fence();
    Rng r;
    U128 inc;
    seed_pcg(a.seed, r.state, inc);
    r.has32 = false;
    r.u32 = 0u;
    const int q0 = (lane >> 1) + 1, q1 = lane >= 1 ? ((lane - 1) >> 1) + 1 : 1;
    r.a0 = a.jump_a[q0]; r.c0 = mul128(a.jump_g[q0], inc);
    r.a1 = a.jump_a[q1]; r.c1 = mul128(a.jump_g[q1], inc);
    bool ok = wave_shuffle<false>(r, ord, T, cnt, lane);                 // the base pass's draws
    if (which == 0 && lane == 0) {
        hdr->state[0] = r.state; hdr->has32[0] = r.has32 ? 1u : 0u; hdr->u32[0] = r.u32;
        hdr->inc = inc; hdr->tiles = (uint32_t)T; hdr->n_orders = (uint32_t)a.n_orders; hdr->seed = a.seed;
    }
    if (which == 1) ok = wave_shuffle<false>(r, ord, T, cnt, lane) && ok;   // pass 1's draws
    for (int t = lane; t < T; t += 64) ord.set((uint32_t)t, (uint32_t)t);
    ord.sync();
    ok = wave_shuffle<true>(r, ord, T, cnt, lane) && ok;
    ord.sync();
    if (copy_out)
        for (int t = lane; t < T; t += 64) P[t] = ord.get((uint32_t)t);
    if (lane == 0) {
        hdr->state[1 + which] = r.state; hdr->has32[1 + which] = r.has32 ? 1u : 0u; hdr->u32[1 + which] = r.u32;
        hdr->ok[which] = ok ? 1u : 0u;
    }
}

__global__ __launch_bounds__(64) void scan_orders(OrdersArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x, which = blockIdx.x;
    uint32_t *P = reinterpret_cast<uint32_t *>(a.out + sizeof(OrdersHdr)) + (size_t)which * orders_stride(a.tiles);
    if (a.tiles <= kScanMaxTilesLds) {
        OrderLds ord{reinterpret_cast<uint16_t *>(lds)};
        make_order(a, ord, reinterpret_cast<uint32_t *>(lds + 2 * ((a.tiles + 7) & ~(int64_t)7)), P, which, lane, true);
    } else {
        OrderGlobal ord{P};
        make_order(a, ord, reinterpret_cast<uint32_t *>(lds), P, which, lane, false);
    }
}

// jump-ahead tables A_q = a^q, G_q = 1 + a + … + a^(q−1) (mod 2^128), uploaded once per device (freed by mtq_shutdown)
std::mutex g_jump_mu;
U128 *g_jump_tab[64] = {nullptr};
const U128 *jump_tables(int dev)
{
    std::lock_guard<std::mutex> lock(g_jump_mu);
    if (dev < 0 || dev >= 64) return nullptr;
    if (!g_jump_tab[dev]) {
        U128 h[2 * kJump];
        const U128 mult = {kMultLo, kMultHi};
        h[0] = {1ull, 0ull};
        h[kJump] = {0ull, 0ull};
        for (int q = 1; q < kJump; ++q) {
            h[kJump + q] = add128(mul128(h[kJump + q - 1], mult), U128{1ull, 0ull});
            h[q] = mul128(h[q - 1], mult);
        }
        U128 *d = nullptr;
        if (hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
        g_jump_tab[dev] = d;
    }
    return g_jump_tab[dev];
}

std::atomic<bool> g_lds_raised[64][2];   // per device and kernel: the attribute belongs to the function's code object on the current device

int raise_lds(int dev, int which, const void *fn, int bytes)
{
    if (dev < 64 && g_lds_raised[dev][which].load(std::memory_order_acquire)) return MTQ_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return fail(MTQ_ERR_HIP, "could not raise the dynamic LDS limit");
    if (dev < 64) g_lds_raised[dev][which].store(true, std::memory_order_release);
    return MTQ_OK;
}

} // namespace
} // namespace mtq

using namespace mtq;

// mtq_shutdown's part of this file: the device tables (no HIP call is ever made from a static destructor)
void mtq::scan_shutdown()
{
    std::lock_guard<std::mutex> lock(g_jump_mu);
    for (int d = 0; d < 64; ++d) {
        if (g_jump_tab[d]) { (void)hipFree(g_jump_tab[d]); g_jump_tab[d] = nullptr; }
        g_lds_raised[d][0].store(false); g_lds_raised[d][1].store(false);
    }
}

extern "C" int mtq_debug_scan_ticks(uint64_t out[16])
{
    if (!out) return fail(MTQ_ERR_INVALID, "null argument");
    if (int rc = require_device()) return rc;
    unsigned long long h[16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_scan_ticks), sizeof(h)) != hipSuccess) return fail(MTQ_ERR_HIP, "hipMemcpyFromSymbol failed");
    for (int i = 0; i < 16; ++i) out[i] = h[i];
    return MTQ_OK;
}

extern "C" size_t mtq_greedy_scan_scratch_bytes(int64_t count, int64_t tiles)
{
    if (count <= 0 || tiles <= 0) return 0;
    // per tensor: the deltas of two passes (2 x 32 B per tile) and a visiting order in global memory (4 B per tile: tensors of more than
    // kScanMaxTilesLds tiles, and — with shared orders — any tensor that has to shuffle for itself)
    const size_t per = (size_t)tiles * 8 * sizeof(double) + (size_t)tiles * sizeof(uint32_t);
    return (size_t)count * ((per + 255) & ~(size_t)255);
}

extern "C" size_t mtq_scan_orders_bytes(int64_t tiles) { return tiles <= 0 ? 0 : sizeof(OrdersHdr) + 2 * orders_stride(tiles) * sizeof(uint32_t); }
extern "C" size_t mtq_scan_carry_bytes(int64_t count) { return count <= 0 ? 0 : (size_t)count * sizeof(Carry); }

extern "C" int mtq_scan_orders_device(uint64_t seed, int64_t tiles, int n_orders, void *orders, size_t orders_bytes, void *stream)
{
    if (!orders) return fail(MTQ_ERR_INVALID, "null argument");
    if (seed == 0) return fail(MTQ_ERR_INVALID, "seed 0 means 'draw a random seed' in the reference; pass a non-zero seed");
    if (tiles <= 0 || tiles > MTQ_SCAN_DEVICE_MAX_TILES) return fail(MTQ_ERR_INVALID, "tiles out of range");
    if (n_orders < 1 || n_orders > 2) return fail(MTQ_ERR_INVALID, "n_orders must be 1 or 2");
    if (orders_bytes < mtq_scan_orders_bytes(tiles)) return fail(MTQ_ERR_INVALID, "orders is smaller than mtq_scan_orders_bytes()");
    if (int rc = require_device()) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MTQ_ERR_HIP, "hipGetDevice failed");
    const U128 *jt = jump_tables(dev);
    if (!jt) return fail(MTQ_ERR_HIP, "could not upload the jump-ahead tables");
    OrdersArgs a{seed, tiles, n_orders, static_cast<unsigned char *>(orders), jt, jt + kJump};
    hipStream_t st = static_cast<hipStream_t>(stream);
    // ok[1] of a one-order buffer must read 0: the header's flags are cleared first (the kernel's two blocks set their own)
    if (hipMemsetAsync(orders, 0, sizeof(OrdersHdr), st) != hipSuccess) return fail(MTQ_ERR_HIP, "hipMemsetAsync failed");
    size_t lds = kTagSlots;
    if (tiles <= kScanMaxTilesLds) {
        lds += 2 * (size_t)((tiles + 7) & ~(int64_t)7);
        if (int rc = raise_lds(dev, 1, reinterpret_cast<const void *>(scan_orders), 2 * kScanMaxTilesLds + kTagSlots)) return rc;
    }
    hipLaunchKernelGGL(scan_orders, dim3((unsigned)n_orders), dim3(64), lds, st, a);
    return check_launch("mtq_scan_orders_device");
}

extern "C" int mtq_greedy_scan_device_ex(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                                         int metric, double threshold, double elem_count, const uint64_t *seeds, int8_t *maps, int32_t *status,
                                         int32_t *counts, void *scratch, size_t scratch_bytes, const void *orders, int phase, uint32_t *listed,
                                         uint32_t *n_listed, void *carry, void *stream)
{
    if (!stats || !formats || !seeds || !maps || !status || !scratch) return fail(MTQ_ERR_INVALID, "null argument");
    if (count <= 0 || count > (1 << 20) || tiles <= 0) return fail(MTQ_ERR_INVALID, "count and tiles must be positive");
    if (tiles > MTQ_SCAN_DEVICE_MAX_TILES) return fail(MTQ_ERR_UNSUPPORTED, "more tiles than the device scan takes: use the host scan");
    if (metric != MTQ_METRIC_PCC && metric != MTQ_METRIC_MAE && metric != MTQ_METRIC_ATOL) return fail(MTQ_ERR_INVALID, "unknown metric");
    if (n_formats <= 0 || n_formats > MTQ_NUM_TILE_FORMATS) return fail(MTQ_ERR_INVALID, "n_formats must be 1..4");
    if (fmt_mask & MTQ_MASK_SLIM) return fail(MTQ_ERR_INVALID, "the device scan reads full records");
    if (!(elem_count > 0.0)) return fail(MTQ_ERR_INVALID, "elem_count must be positive");
    if (scratch_bytes < mtq_greedy_scan_scratch_bytes(count, tiles)) return fail(MTQ_ERR_INVALID, "scratch is smaller than mtq_greedy_scan_scratch_bytes()");
    if (phase < 0 || phase > 2) return fail(MTQ_ERR_INVALID, "phase must be 0, 1 or 2");
    if (phase != 0) {
        if (metric == MTQ_METRIC_ATOL) return fail(MTQ_ERR_UNSUPPORTED, "the atol walk has no phases");
        if (n_formats < 2) return fail(MTQ_ERR_INVALID, "a split search needs at least two formats");
        if (!carry || (phase == 1 && (!listed || !n_listed))) return fail(MTQ_ERR_INVALID, "a split search needs carry (and listed, n_listed in phase 1)");
        if (count * tiles >= ((int64_t)1 << 32)) return fail(MTQ_ERR_INVALID, "too many tiles for the list of a split search");
    }
    ScanArgs a{};
    a.stats = stats;
    a.tiles = tiles;
    a.rec = 2 + 5 * popcount4(fmt_mask);
    a.n_formats = n_formats;
    for (int f = 0; f < MTQ_NUM_TILE_FORMATS; ++f) {
        const int s = slot_of(fmt_mask, f);
        a.oy[f] = s >= 0 ? 2 + 5 * s : 0;
        a.oy2[f] = s >= 0 ? 3 + 5 * s : 1;
        a.oxy[f] = s >= 0 ? 4 + 5 * s : 1;
        a.oab[f] = s >= 0 ? 5 + 5 * s : kNoOffset;
        a.omx[f] = s >= 0 ? 6 + 5 * s : kNoOffset;
    }
    a.metric = metric;
    for (int i = 0; i < n_formats; ++i) {
        if (!slot_ok(slot_of(fmt_mask, formats[i]))) return fail(MTQ_ERR_INVALID, "a format is not available under fmt_mask");
        for (int j = 0; j < i; ++j) if (formats[j] == formats[i]) return fail(MTQ_ERR_UNSUPPORTED, "the device scan needs distinct formats: use the host scan");
        a.fmt[i] = formats[i];
    }
    a.thr = threshold;
    a.n = elem_count;
    a.seeds = seeds;
    a.maps = maps;
    a.status = status;
    a.counts = counts;
    a.orders = static_cast<const unsigned char *>(orders);
    a.phase = phase;
    a.listed = listed;
    a.n_listed = n_listed;
    a.carry = static_cast<Carry *>(carry);
    if (int rc = require_device()) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(MTQ_ERR_HIP, "hipGetDevice failed");
    const U128 *jt = jump_tables(dev);
    if (!jt) return fail(MTQ_ERR_HIP, "could not upload the jump-ahead tables");
    a.jump_a = jt;
    a.jump_g = jt + kJump;
    // scratch: the deltas of two passes for every tensor first, then (large tensors) all orders
    a.delta = static_cast<double *>(scratch);
    a.delta2 = a.delta + (size_t)count * (size_t)tiles * 4;
    a.order_g = reinterpret_cast<uint32_t *>(static_cast<unsigned char *>(scratch) + (size_t)count * (size_t)tiles * 8 * sizeof(double));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (metric == MTQ_METRIC_ATOL) {
        hipLaunchKernelGGL(greedy_atol, dim3((unsigned)count), dim3(256), 0, st, a);
        return check_launch("mtq_greedy_scan_device");
    }
    const unsigned threads = (orders && phase != 2) ? 128u : 64u;   // a helper wave per tensor when the visiting orders are shared
    // With shared orders a tensor shuffles for itself only when its course leaves the common one (a rejection in pass 1): that rare
    // shuffle may as well run in global scratch, and the block then needs 23 KiB of LDS instead of 23 + 2·tiles — a block that K1's
    // blocks make room for far sooner.  MTQ_SCAN_SHARED_LDS=1 keeps the order in LDS.
    static int shared_lds = -1;
    if (shared_lds < 0) { const char *e = getenv("MTQ_SCAN_SHARED_LDS"); shared_lds = e ? atoi(e) : 0; }
    if (tiles <= kScanMaxTilesLds && !(threads == 128u && !shared_lds)) {
        const size_t lds = 2 * (size_t)((tiles + 7) & ~(int64_t)7) + kFixedLds;
        if (int rc = raise_lds(dev, 0, reinterpret_cast<const void *>(greedy_scan_pcc_lds), 2 * kScanMaxTilesLds + (int)kFixedLds)) return rc;
        hipLaunchKernelGGL(greedy_scan_pcc_lds, dim3((unsigned)count), dim3(threads), lds, st, a);
    } else {
        hipLaunchKernelGGL(greedy_scan_pcc_global, dim3((unsigned)count), dim3(threads), kFixedLds, st, a);
    }
    return check_launch("mtq_greedy_scan_device");
}

extern "C" int mtq_greedy_scan_device(const double *stats, int64_t count, int64_t tiles, uint32_t fmt_mask, const int *formats, int n_formats,
                                      int metric, double threshold, double elem_count, const uint64_t *seeds, int8_t *maps, int32_t *status,
                                      int32_t *counts, void *scratch, size_t scratch_bytes, void *stream)
{
    return mtq_greedy_scan_device_ex(stats, count, tiles, fmt_mask, formats, n_formats, metric, threshold, elem_count, seeds, maps, status, counts,
                                     scratch, scratch_bytes, nullptr, 0, nullptr, nullptr, nullptr, stream);
}
