"""Pins the CPU oracle (oracle/) to vectors produced by the imported reference (tests/golden/).
CPU-only: runs in the build container and on the GPU box alike."""
import json

import numpy as np
import pytest

from oracle import mtq_oracle as orc
from tests.inputs import gen, sha

ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


@pytest.fixture(scope="module")
def meta(golden_dir):
    return json.loads((golden_dir / "golden_meta.json").read_text())


def test_f1_known_answer_bits(golden_dir):
    d = np.load(golden_dir / "f1_quantize_kat.npz")
    x = d["x_bits"].view(np.float32)
    for fmt in ALL + ["fp0"]:
        with np.errstate(all="ignore"):
            y_c = orc.quantize_weight_values(x, fmt).view(np.uint32)
            y_np = orc.quantize_np(x, fmt).view(np.uint32)
        assert np.array_equal(y_c, d[f"y_{fmt}"]), fmt
        assert np.array_equal(y_np, d[f"y_{fmt}"]), fmt


def test_f1_hand_kat():
    # SURVEY §8(c) F1 hand vector (measured on the reference)
    x = np.array([1, .75, .3, -.3, .01, 1.9921875, 1.99609375, -1.5, .5, .25, .125, .0625, 3e-39, 0, -0., 1e-3], dtype=np.float32)
    assert np.array_equal(orc.quantize_weight_values(x, "bfp8"),
                          np.array([1, .75, .296875, -.296875, .015625, 1.984375, 1.984375, -1.5, .5, .25, .125, .0625, 0, 0, 0, 0], dtype=np.float32))
    assert np.array_equal(orc.quantize_weight_values(x, "bfp4"),
                          np.array([1, .75, .25, -.25, 0, 1.75, 1.75, -1.5, .5, .25, 0, 0, 0, 0, 0, 0], dtype=np.float32))
    assert np.array_equal(orc.quantize_weight_values(x, "bfp2"),
                          np.array([1, 1, 0, 0, 0, 1, 1, -1, 0, 0, 0, 0, 0, 0, 0, 0], dtype=np.float32))
    with pytest.raises(ValueError):
        orc.quantize_weight_values(x, "mxfp9")


def test_f2_layouts(golden_dir):
    d = np.load(golden_dir / "f2_layouts.npz")
    names = sorted({k[: -2] for k in d.files if k.endswith("_x")})
    assert len(names) == 10
    for name in names:
        x = d[f"{name}_x"]
        for fmt in ALL:
            want = d[f"{name}_y_{fmt}"]
            got = orc.quantize_weight_values(x, fmt)
            assert got.shape == want.shape
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (name, fmt)
            assert np.array_equal(orc.quantize_np(x, fmt).view(np.uint32), want.view(np.uint32)), (name, fmt)
            # whole-tensor == tile-wise through the 2-D flatten + apply_assignment path
            a = np.full(orc.tiles_hw(*orc.flatten_2d(x)[0].shape), ALL.index(fmt), dtype=np.int8)
            assert np.array_equal(orc.apply_assignment(x, a).view(np.uint32).reshape(-1), want.view(np.uint32).reshape(-1))


def _greedy_names(d):
    return sorted({k[: -len("_assign")] for k in d.files if k.endswith("_assign")})


def test_f4_greedy_maps(golden_dir, meta):
    d = np.load(golden_dir / "f4_greedy.npz")
    names = _greedy_names(d)
    assert len(names) == 14
    for name in names:
        m = meta["f4"][name]
        x = gen(m["kind"], m["seed"], tuple(m["shape"]))
        assert sha(x) == m["x_sha256"], f"input recipe drifted: {name}"
        assign, counts, st = orc.greedy(x, m["formats"], m["metric"], m["threshold"], m["algo_seed"])
        assert np.array_equal(assign, d[f"{name}_assign"]), name
        assert [counts[f] for f in ALL] == list(d[f"{name}_counts"]), name
        y = orc.apply_assignment(x, assign)
        assert sha(y) == m["y_sha256"], name
        cols = d[f"{name}_cols"]
        assert orc.mixed_tile_total_bytes(counts) == cols[3]
        # float columns: float64 moments vs the reference's float32 columns (tolerance from SURVEY §7.3-2:
        # 1e-6 up to 256x256, the reference's own float32 noise above that)
        slots = orc.mask_slots(orc.fmt_mask(m["formats"]))
        pcc, mae, atol = orc.columns_from_stats(st["stats"], slots, assign, x.size)
        tol = 1e-6 if x.size <= 65536 else 2e-5
        assert abs(pcc - cols[0]) <= tol, (name, pcc, cols[0])
        assert abs(mae - cols[1]) <= 1e-6 and abs(atol - cols[2]) <= 1e-6
        assert abs(pcc - orc.pearson_corr_f64(x, y)) <= 1e-7  # float32-rounded products (mixed_tile_greedy.py:159-162)
        assert atol == float(np.max(np.abs(x - y)))


def test_f5_threshold_maps(golden_dir, meta):
    d = np.load(golden_dir / "f5_threshold.npz")
    names = _greedy_names(d)
    assert len(names) == 16
    for name in names:
        m = meta["f5"][name]
        x = d[f"{name}_x"]
        assign, counts, scores = orc.threshold(x, m["formats"], m["metric"], m["threshold"])
        for fmt in m["formats"]:
            assert np.array_equal(scores[fmt], d[f"{name}_score_{fmt}"]), (name, fmt)
        assert np.array_equal(assign, d[f"{name}_assign"]), name
        assert [counts[f] for f in ALL] == list(d[f"{name}_counts"]), name
        y = orc.apply_assignment(x, assign)
        assert np.array_equal(y.view(np.uint32), d[f"{name}_y"].view(np.uint32)), name
        cols = d[f"{name}_cols"]
        assert orc.pearson_corr(x, y) == cols[0]
        assert orc.mixed_tile_total_bytes(counts) == cols[3]


def test_f5_moment_scores_close_to_float32_scores(golden_dir, meta):
    """Per-tile pcc/mae/atol derived from the float64 raw sums (what the HIP path produces) stay within
    the knife-edge band of the reference's float32 scores (<= 3.1e-7 measured)."""
    d = np.load(golden_dir / "f5_threshold.npz")
    for name in _greedy_names(d):
        m = meta["f5"][name]
        x2d, _ = orc.flatten_2d(d[f"{name}_x"])
        fm = [f for f in ALL if f in m["formats"]]
        st = orc.tile_stats(x2d, fm)
        slots = orc.mask_slots(orc.fmt_mask(fm))
        for fmt in fm:
            b = st[:, 2 + 5 * slots[fmt]:]
            want = d[f"{name}_score_{fmt}"].astype(np.float64)
            if m["metric"] == "mae":
                got = b[:, 3] / 1024.0
            elif m["metric"] == "atol":
                got = b[:, 4]
                assert np.array_equal(got, want)
                continue
            else:
                n = 1024.0
                mx, my = st[:, 0] / n, b[:, 0] / n
                am2 = np.maximum(st[:, 1] - n * mx * mx, 0)
                bm2 = np.maximum(b[:, 1] - n * my * my, 0)
                den = np.sqrt(am2 * bm2)
                with np.errstate(all="ignore"):
                    got = np.where(den == 0, np.where(b[:, 3] == 0, 1.0, 0.0), (b[:, 2] - n * mx * my) / den)
            assert np.max(np.abs(got - want)) <= 1e-6, (name, fmt, np.max(np.abs(got - want)))


def test_f10_random_search_maps(golden_dir, meta):
    """mixed_tile_random.py: the oracle's literal restatement reproduces the reference's selected map, counts and
    every sample's float32 pcc/mae/atol exactly (same NumPy Generator, same float32 expressions)."""
    d = np.load(golden_dir / "f10_random.npz")
    assert len(meta["f10"]) == 11
    branches = set()
    for name, m in meta["f10"].items():
        x = d[f"{name}_x"]
        assign, counts, samples = orc.random_search(x, m["formats"], m["metric"], m["threshold"], m["iters"], m["algo_seed"])
        assert np.array_equal(assign, d[f"{name}_assign"]), name
        assert [counts[f] for f in ALL] == list(d[f"{name}_counts"]), name
        rows = np.array([[s["id"], s["total_bytes"], s["pcc"], s["mae"], s["atol"], *[s["counts"][f] for f in ALL]] for s in samples])
        assert np.array_equal(rows, d[f"{name}_samples"]), name
        y = orc.apply_assignment(x, assign)
        assert np.array_equal(y.view(np.uint32), d[f"{name}_y"].view(np.uint32)), name
        col = rows[:, {"pcc": 2, "mae": 3, "atol": 4}[m["metric"]]]
        ok = col >= m["threshold"] if m["metric"] == "pcc" else col <= m["threshold"]
        branches.add("none" if not ok.any() else ("all" if ok.all() else "split"))
    assert branches == {"none", "all", "split"}
    # the two knife-edge thresholds (a sample's float32 score, and that + 1e-9) select different maps
    assert not np.array_equal(d["r_pcc_knife_eq_assign"], d["r_pcc_knife_eps_assign"])


def test_f8_rng_and_bytes(golden_dir):
    d = np.load(golden_dir / "f8_misc.npz")
    assert np.array_equal(np.random.default_rng(123).permutation(16384)[:32], d["perm_123_16384_head"])
    assert np.array_equal(np.random.default_rng(5).permutation(np.array([3, 9, 10, 40, 41, 77, 100], dtype=np.int64)), d["perm_5_subset"])
    assert orc.mixed_tile_total_bytes({"bf16": 0, "bfp8": 0, "bfp4": 128088, "bfp2": 936}) == d["total_bytes_check"][0]
    assert round(d["total_bytes_check"][0]) == 65948829  # notebooks/wq_mixed_tile_walkthrough.ipynb:438
    assert bool(np.float32(0.94) >= 0.94) == bool(d["nep50_f32_ge"][0])
    assert np.array_equal(np.random.default_rng(7).integers(0, 4, size=64, dtype=np.int64), d["integers_7_4_head"])
