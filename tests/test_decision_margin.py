"""How far the greedy search's decisions sit from flipping under a change of the per-tile sums' summation order.

The records' float64 sums follow this repo's own order (include/mtq.h), the reference forms them with np.sum; golden F3 bounds the difference
at 64 ulp of a tile's largest column and every reference map is reproduced — an argument by cases.  This test adds the argument by margin for
the headline case (BASELINE configs[1], the 4096x4096 tensor whose map the reference pins by SHA-256, F11): every decision of the search is
replayed from the oracle's records with the scan's own float64 operations, the distance of the closest one from the threshold is measured,
and the records are then disturbed by MORE than the bound (every sum of every tile by up to 64 ulp, random signs, five draws): each decision
must come out the same, and the values may move by no more than a small fraction of the closest margin.  CPU only (oracle + NumPy)."""
import numpy as np

from oracle import mtq_oracle as orc
import pytest

from tests.inputs import gen, m1_tensor

ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def replay(stats, slots, orders, final_code, n, thr):
    """→ (values of every visit of passes 1.., decisions agree with final_code?, value of the base pass).  The scan's arithmetic
    (mixed_tile_greedy.py:176-190 with the x-only terms hoisted, csrc/mtq_host.cpp pcc_hoisted): sequential float64 running sums."""
    sum_x = np.cumsum(stats[:, 0])[-1]          # np.cumsum adds one after the other, in tile order, as the scan's initial sums do
    sum_x2 = np.cumsum(stats[:, 1])[-1]
    mean_x = sum_x / n
    am2 = max(sum_x2 - n * mean_x * mean_x, 0.0)

    def value(sy, sy2, sxy):
        mean_y = sy / n
        bm2 = np.maximum(sy2 - n * mean_y * mean_y, 0.0)
        return (sxy - n * mean_x * mean_y) / np.sqrt(am2 * bm2)

    o0 = 2 + 5 * slots[ALL[0]]
    S = [np.cumsum(stats[:, o0 + c])[-1] for c in range(3)]
    base_val = float(value(*S))
    vals, agree = [], True
    for p in range(1, len(ALL)):
        order = orders[p]
        oc, op = 2 + 5 * slots[ALL[p]], 2 + 5 * slots[ALL[p - 1]]
        accepted = final_code[order] >= p
        cand = []
        for c in range(3):
            delta = stats[order, oc + c] - stats[order, op + c]                       # :259-261
            run = np.cumsum(np.concatenate([[S[c]], np.where(accepted, delta, 0.0)]))   # the sums before every visit (a rejected visit adds nothing)
            cand.append(run[:-1] + delta)
            S[c] = run[-1]
        v = value(*cand)
        vals.append(v)
        agree &= bool(np.array_equal(v >= thr, accepted))
    return np.concatenate(vals), agree, base_val


def test_headline_decisions_keep_their_distance_from_the_threshold():
    x = m1_tensor()
    thr, seed = 0.999, 123
    a, counts, st, orders = orc.greedy(x, ALL, "pcc", thr, seed, return_orders=True)
    assert counts == {"bf16": 0, "bfp8": 13870, "bfp4": 2514, "bfp2": 0}             # the reference's result (golden F11)
    stats = st["stats"]
    slots = orc.mask_slots(orc.fmt_mask(ALL))
    code = a.reshape(-1).astype(np.int64)
    n = float(x.size)
    vals, agree, base_val = replay(stats, slots, orders, code, n, thr)
    assert agree and base_val >= thr, "the replay is not the scan"
    assert vals.size == 16384 + 16384 + 2514
    margin = float(np.min(np.abs(vals - thr)))
    rng = np.random.default_rng(7)
    sum_cols = [c for c in range(stats.shape[1]) if c < 2 or (c - 2) % 5 < 4]           # every sum; the maxima feed no pcc decision
    worst = 0.0
    for _ in range(5):
        noisy = stats.copy()
        noisy[:, sum_cols] *= 1.0 + rng.uniform(-1.0, 1.0, size=(stats.shape[0], len(sum_cols))) * 64.0 * 2.0 ** -53
        v2, agree2, base2 = replay(noisy, slots, orders, code, n, thr)
        assert agree2 and base2 >= thr, "a 64-ulp change of the per-tile sums flipped a decision of the headline case"
        worst = max(worst, float(np.max(np.abs(v2 - vals))))
    # the closest of the 35 282 decisions against what 64 ulp on every sum of every tile can move a value
    assert margin > 50.0 * worst, (margin, worst)
    print(f"closest decision {margin:.3e} from the threshold; 64-ulp noise on every tile sum moves a value by at most {worst:.3e}")


@pytest.mark.parametrize("kind,seed,shape,thr", [("heavy_bf16", 11, (1024, 2048), 0.99), ("normal_f32", 12, (1024, 4096), 0.999),
                                                 ("heavy_f32", 13, (768, 1536), 0.995), ("normal_bf16", 14, (2048, 2048), 0.9995)])
def test_other_tensors_keep_their_distance_too(kind, seed, shape, thr):
    """The same argument by margin on tensors of other kinds (heavy tails, float32 storage, other thresholds and shapes — the Llama-3-8B
    k_proj shape among them): no decision flips under 64 ulp of noise on every tile sum, and the closest decision is orders of magnitude
    further from the threshold than that noise moves a value.  (The round-3 review: one tensor is evidence, not proof — so are five;
    the bench adds 24 maps per run against the oracle, tests/fuzz_parity.py thousands against the host scan.)"""
    x = gen(kind, seed, shape)
    a, _counts, st, orders = orc.greedy(x, ALL, "pcc", thr, 123, return_orders=True)
    stats = st["stats"]
    slots = orc.mask_slots(orc.fmt_mask(ALL))
    code = a.reshape(-1).astype(np.int64)
    n = float(x.size)
    vals, agree, base_val = replay(stats, slots, orders, code, n, thr)
    if not (base_val >= thr):
        pytest.skip("the base format already misses the threshold: no pass decides anything")
    assert agree, "the replay is not the scan"
    margin = float(np.min(np.abs(vals - thr)))
    rng = np.random.default_rng(7)
    sum_cols = [c for c in range(stats.shape[1]) if c < 2 or (c - 2) % 5 < 4]
    worst = 0.0
    for _ in range(3):
        noisy = stats.copy()
        noisy[:, sum_cols] *= 1.0 + rng.uniform(-1.0, 1.0, size=(stats.shape[0], len(sum_cols))) * 64.0 * 2.0 ** -53
        v2, agree2, base2 = replay(noisy, slots, orders, code, n, thr)
        assert agree2 and base2 >= thr, "a 64-ulp change of the per-tile sums flipped a decision"
        worst = max(worst, float(np.max(np.abs(v2 - vals))))
    assert margin > 50.0 * worst, (kind, margin, worst)
    print(f"{kind} {shape} thr {thr}: {vals.size} decisions, closest {margin:.3e} from the threshold; noise moves a value by at most {worst:.3e}")
