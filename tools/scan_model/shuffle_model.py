"""Model of the wave-parallel NumPy-compatible shuffle of csrc/mtq_scan.hip (64 lanes emulated with Python loops).

Generator.permutation = Fisher–Yates from the top with random_interval's masked rejection on buffered 32-bit halves of PCG64
outputs (numpy/random/_generator.pyx, distributions.c; restated sequentially in csrc/mtq_host.cpp rng_shuffle_impl).
The device form processes the draw stream 64 positions at a time:
  * lane l evaluates stream position l by LCG jump-ahead (state_q = A_q * state_0 + G_q * inc, q <= 32);
  * acceptance v <= i0 - (accepted before me) is settled for every lane: certain accepts (v <= i0 - l), certain rejects
    (v > i0), the few in between one by one in lane order;
  * the batch is cut where the mask level changes (i crosses a power of two) or i reaches 0; unused positions are handed back;
  * the accepted steps' swaps (i_t = i0 - t, j_t = v) are applied in parallel except those that share a position with
    another step of the batch, which are applied one by one in step order afterwards... wait: order matters only among
    steps that share positions, and a step that shares no position commutes with every other step.
This file checks the model against numpy for many (seed, n) and interleaved calls; run: python tools/scan_model/shuffle_model.py
"""
import numpy as np

M128 = (1 << 128) - 1
PCG_MULT = (0x2360ED051FC65DA4 << 64) | 0x4385DF649FCCF645


def jump_tables(kmax=33):
    """A_q = a^q, G_q = (a^q - 1)/(a - 1) = 1 + a + ... + a^(q-1)  (mod 2^128), q = 0..kmax."""
    A, G = [1], [0]
    for _ in range(kmax):
        G.append((G[-1] * PCG_MULT + 1) & M128)
        A.append((A[-1] * PCG_MULT) & M128)
    return A, G


A_TAB, G_TAB = jump_tables()


def pcg_output(state):
    hi, lo = state >> 64, state & ((1 << 64) - 1)
    x = hi ^ lo
    rot = state >> 122
    return ((x >> rot) | (x << ((64 - rot) & 63))) & ((1 << 64) - 1)


def seed_state(seed):
    """SeedSequence(seed) -> PCG64 (state, inc), as csrc/mtq_host.cpp rng_seed."""
    bg = np.random.PCG64(seed)
    st = bg.state["state"]
    return st["state"], st["inc"]


class Rng:
    def __init__(self, seed):
        self.state, self.inc = seed_state(seed)
        self.has32, self.u32 = False, 0


def wave_shuffle(r: Rng, n: int, arr, swap=True, lanes=64, stats=None):
    if n < 2:
        return
    i0 = n - 1
    while i0 >= 1:
        mask = (1 << i0.bit_length()) - 1
        lo = mask >> 1
        pend = 1 if r.has32 else 0
        # ---- stream positions of this batch
        raw, st_of = [], []
        for l in range(lanes):
            if pend and l == 0:
                raw.append(r.u32); st_of.append(None)
                continue
            q = (l - pend) >> 1          # 0-based output index
            half = (l - pend) & 1
            s = (A_TAB[q + 1] * r.state + G_TAB[q + 1] * r.inc) & M128
            o = pcg_output(s)
            raw.append((o >> 32) if half else (o & 0xFFFFFFFF)); st_of.append((s, o))
        v = [x & mask for x in raw]
        # ---- acceptance
        ok = [v[l] <= i0 - l for l in range(lanes)]
        amb = [(not ok[l]) and v[l] <= i0 for l in range(lanes)]
        for a in range(lanes):
            if amb[a]:
                P = sum(ok[:a])
                if v[a] <= i0 - P:
                    ok[a] = True
        # sanity: sequential definition
        ii, chk = i0, []
        for l in range(lanes):
            c = v[l] <= ii
            chk.append(c); ii -= 1 if c else 0
        assert chk == ok
        kmax = i0 - lo                       # steps allowed at this mask level (ii must stay > lo); lo == 0 at the last level
        total = sum(ok)
        if total >= kmax:                    # cut just after the kmax-th accepted position: what follows belongs to the next mask level
            cnt, c = 0, 0
            for l in range(lanes):
                cnt += ok[l]
                if cnt == kmax:
                    c = l + 1
                    break
            k = kmax
        else:
            c, k = lanes, total
        # ---- swaps of the accepted positions < c
        steps = [(l, v[l]) for l in range(c) if ok[l]]
        assert len(steps) == k
        if swap and k:
            its = [(i0 - t, j) for t, (_l, j) in enumerate(steps)]
            touched = {}
            for t, (i, j) in enumerate(its):
                for p in {i, j}:
                    touched.setdefault(p, []).append(t)
            flagged = sorted({t for ts in touched.values() if len(ts) > 1 for t in ts})
            if stats is not None:
                stats["flagged"] = stats.get("flagged", 0) + len(flagged); stats["batches"] = stats.get("batches", 0) + 1
            fl = set(flagged)
            # parallel part: every unflagged step reads, then writes
            reads = {t: (arr[i], arr[j]) for t, (i, j) in enumerate(its) if t not in fl}
            for t, (ai, aj) in reads.items():
                i, j = its[t]
                arr[i], arr[j] = aj, ai
            for t in flagged:                # serial part, in step order
                i, j = its[t]
                arr[i], arr[j] = arr[j], arr[i]
        i0 -= k
        # ---- hand back: advance the generator by the c consumed positions
        used = c - pend                      # halves taken from fresh outputs
        if used > 0:
            u = (used + 1) >> 1
            r.state = (A_TAB[u] * r.state + G_TAB[u] * r.inc) & M128
            o = pcg_output(r.state)
            r.has32 = bool(used & 1)
            r.u32 = o >> 32
        else:
            r.has32 = False


def check(seed, sizes, stats=None):
    g = np.random.Generator(np.random.PCG64(seed))
    r = Rng(seed)
    for n in sizes:
        want = g.permutation(n)
        arr = list(range(n))
        wave_shuffle(r, n, arr, True, stats=stats)
        assert arr == list(want), (seed, n)
    # generator continuity: the next raw 64-bit output agrees
    nxt = int(g.bit_generator.random_raw())
    st = (A_TAB[1] * r.state + G_TAB[1] * r.inc) & M128
    # numpy's buffered 32-bit half is discarded by random_raw? (it is not: random_raw ignores has_uint32) — compare states instead
    assert pcg_output(st) == nxt, (seed, sizes)


if __name__ == "__main__":
    import sys
    stats = {}
    for seed in (1, 5, 123, 2**31 - 1, 2**40 + 7):
        check(seed, [16384], stats)
        check(seed, [1, 2, 3, 7, 64, 65, 100, 127, 128, 129, 1000, 4096, 2514, 33, 16385])
        check(seed, [17, 16384, 16384, 2514, 5])
    print("ok", stats, "flagged steps per 16384-shuffle:", stats["flagged"] / 5)


def seq_shuffle(r: Rng, n, arr):
    """Sequential restatement (csrc/mtq_host.cpp) on the same Rng model."""
    def next32():
        if r.has32:
            r.has32 = False
            return r.u32
        r.state = (r.state * PCG_MULT + r.inc) & M128
        o = pcg_output(r.state)
        r.has32, r.u32 = True, o >> 32
        return o & 0xFFFFFFFF
    for i in range(n - 1, 0, -1):
        mask = (1 << i.bit_length()) - 1
        while True:
            v = next32() & mask
            if v <= i:
                break
        arr[i], arr[v] = arr[v], arr[i]
