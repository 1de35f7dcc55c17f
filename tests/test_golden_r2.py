"""Round-2 reference fixtures (tests/golden/make_golden.py r2), CPU side: the oracle and the package's host backend against
  F3  per-tile float64 sums formed with the reference's own np.sum(float32 expr, dtype=float64) (mixed_tile_greedy.py:147-174),
  F7  the reference's `wq` run end to end on the synthetic:tiny tensors: table.txt, compression_config.used.json, maps,
  F11 the headline size (BASELINE configs[1]: 4096x4096 bf16, greedy pcc>=0.999 seed 123; threshold 0.94 / 0.999) by SHA-256,
  F12 BASELINE configs[2]/[3] tensor shapes (DeepSeek-R1 layer-0 vectors + kv_a_proj_with_mqa, Llama-3-8B k_proj) by SHA-256,
  F13 large-magnitude mae thresholds (knife-edge band must scale with the score),
  F14 (make_golden.py r2b) mixed-tile-greedy under the mae and atol metrics at 1024x768 bf16 / 512x640 float32, two seeds each.
The GPU side of the same fixtures is tests/test_configs_gpu.py."""
import json

import numpy as np
import pytest

from oracle import mtq_oracle as orc
from quantization_analysis_amd import cli, model_source
from tests.inputs import m1_tensor, sha
from tests.test_cli import run_dir, strip_time

ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


@pytest.fixture(scope="module")
def meta2(golden_dir):
    return json.loads((golden_dir / "golden_meta_r2.json").read_text())


def check_summary(want: dict, x: np.ndarray, assign: np.ndarray, counts: dict, stats=None, fmts=ALL):
    """An oracle / backend result against one reference summary (maps and y pinned by SHA-256)."""
    assert list(assign.shape) == want["assign_shape"] and sha(assign.astype(np.int8)) == want["assign_sha256"]
    assert [counts[f] for f in ALL] == want["counts"]
    assert orc.mixed_tile_total_bytes(counts) == want["tile_bytes"]
    y = orc.apply_assignment(x, assign)
    assert sha(y) == want["y_sha256"]
    if stats is not None:
        pcc, mae, atol = orc.columns_from_stats(stats, orc.mask_slots(orc.fmt_mask(fmts)), assign, x.size)
        assert abs(pcc - want["pcc64"]) <= 1e-7 and abs(mae - want["mae64"]) <= 1e-9 * max(1.0, want["mae64"]) and atol == want["atol32"]


def test_f3_tile_sums_against_reference_np_sum(golden_dir):
    """The records' sums (oracle order: groups sequentially, row pairs by a tree) against NumPy's pairwise float64 sums of the
    same float32 terms: equal to a few ulp of the sum of magnitudes, maxima exactly."""
    d = np.load(golden_dir / "f3_tile_sums.npz")
    for tag in ("bf16_256", "f32_96x160"):
        x = d[f"{tag}_x"]
        want = d[f"{tag}_records"]
        x2d, _ = orc.flatten_2d(x)
        got = orc.tile_stats(x2d, ALL)
        assert got.shape == want.shape
        mx = [6 + 5 * k for k in range(4)]
        assert np.array_equal(got[:, mx], want[:, mx]), tag
        sums = [c for c in range(want.shape[1]) if c not in mx]
        # every term of a tile is bounded by the tile's Σx² / Σ|x| scale; 1024 terms: allow 64 ulp of the largest column magnitude
        scale = np.maximum(np.abs(want[:, sums]).max(axis=1, keepdims=True), np.finfo(np.float64).tiny)
        assert np.max(np.abs(got[:, sums] - want[:, sums]) / scale) <= 64 * np.finfo(np.float64).eps, tag


@pytest.mark.parametrize("tag", ["greedy", "threshold"])
def test_f7_wq_table_text_equals_reference(golden_dir, tmp_path, monkeypatch, tag):
    """`wq --backend emulation` on synthetic:tiny prints the table the reference's own wq printed for the same tensors and
    config (everything but the TIME(s) values), writes the same compression_config.used.json and the same maps."""
    monkeypatch.chdir(tmp_path)
    f7 = golden_dir / "f7_wq"
    cfg = tmp_path / "cfg.json"
    cfg.write_text((f7 / f"{tag}_config.json").read_text())
    assert cli.run(["synthetic:tiny", "--compression-config", str(cfg), "--summary", "--results-dir", str(tmp_path / "results"), "--no-plots"]) == 0
    rdir = run_dir(tmp_path / "results")
    assert strip_time((rdir / "table.txt").read_text()) == strip_time((f7 / f"{tag}_table.txt").read_text())
    assert json.loads((rdir / "compression_config.used.json").read_text()) == json.loads((f7 / f"{tag}_compression_config.used.json").read_text())
    maps = np.load(f7 / f"{tag}_assignments.npz")
    assert len(maps.files) == 5
    for name in maps.files:
        got = np.load(next(rdir.rglob(f"{cli._slug(name)}/assignment.npy")))
        assert got.dtype == np.int8 and np.array_equal(got, maps[name]), name


def test_f11_headline_size_oracle(meta2):
    """BASELINE configs[1] at full size: the oracle reproduces the reference's 4096x4096 greedy and threshold results."""
    x = m1_tensor()
    f11 = meta2["f11"]
    assert sha(x) == f11["greedy_pcc999_seed123"]["x_sha256"]
    a, counts, st = orc.greedy(x, ALL, "pcc", 0.999, 123)
    check_summary(f11["greedy_pcc999_seed123"], x, a, counts, st["stats"])
    assert counts == {"bf16": 0, "bfp8": 13870, "bfp4": 2514, "bfp2": 0}
    for thr in (0.94, 0.999):
        a, counts, _scores = orc.threshold(x, ALL, "pcc", thr)
        check_summary(f11[f"threshold_pcc{thr}"], x, a, counts, st["stats"])


def test_f12_config_shapes_oracle(meta2):
    """BASELINE configs[2] (DeepSeek-R1 layer-0 self_attn, threshold) and [3] (Llama-3-8B, greedy) on their own tensor shapes,
    a subset the reference finishes in a minute: 1-D bf16 vectors (partial tile: the per-tile score sees the zero padding,
    the tensor column does not), a 576x7168 fp8-block tensor, a 1024x4096 bf16 projection."""
    for key, want in meta2["f12"].items():
        preset, name, algo, *rest = key.split("|")
        idx = model_source.build_model_index("synthetic:deepseek-r1-layer0" if preset == "deepseek" else "synthetic:llama3-8b")
        x = np.asarray(idx.load(name).float().numpy(), dtype=np.float32)
        assert sha(x) == want["x_sha256"], key
        if algo == "threshold":
            a, counts, _ = orc.threshold(x, ALL, "pcc", float(rest[0]))
            check_summary(want, x, a, counts)
        else:
            a, counts, st = orc.greedy(x, ALL, "pcc", float(rest[0]), int(rest[1]))
            check_summary(want, x, a, counts, st["stats"])


def test_f13_large_magnitude_mae(golden_dir, meta2):
    d = np.load(golden_dir / "f13_large_magnitude.npz")
    x = d["x"]
    assert float(np.abs(x).max()) > 1e4
    for tag, m in meta2["f13"].items():
        if tag.startswith("greedy"):
            a, counts, st = orc.greedy(x, ALL, "mae", m["threshold"], m["algo_seed"])
        else:
            a, counts, _ = orc.threshold(x, ALL, "mae", m["threshold"])
        assert np.array_equal(a, d[f"{tag}_assign"]), tag
        check_summary(m, x, a, counts)
    assert not np.array_equal(d["knife_eq_assign"], d["knife_below_assign"])   # the thresholds really sit on a tile's score


def run_package_algo(name, params, x, backend="emulation"):
    from quantization_analysis_amd.compression_algorithms import create_algorithm
    from quantization_analysis_amd.compression_algorithms.cache import CacheContext
    from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer

    algo = create_algorithm(name, params)
    return algo.run(x, ALL, Quantizer(backend), CacheContext("/tmp/mtq-test-cache", "t", backend, True, "t"))[0]


def check_f13_backend(golden_dir, meta2, backend, to_input=lambda x: x):
    d = np.load(golden_dir / "f13_large_magnitude.npz")
    x = d["x"]
    for tag, m in meta2["f13"].items():
        if tag.startswith("greedy"):
            res = run_package_algo("mixed-tile-greedy", {"metric": "mae", "threshold": m["threshold"], "seed": m["algo_seed"]}, to_input(x), backend)
        else:
            res = run_package_algo("mixed-tile-threshold", {"metric": "mae", "threshold": m["threshold"]}, to_input(x), backend)
        assert np.array_equal(res.meta["assignment"], d[f"{tag}_assign"]), (backend, tag)
        assert [res.tile_counts[f] for f in ALL] == m["counts"] and res.tile_bytes == m["tile_bytes"]
        y = res.y.cpu().numpy() if hasattr(res.y, "cpu") else res.y
        assert sha(np.asarray(y, dtype=np.float32)) == m["y_sha256"], (backend, tag)
        c = res.meta["columns"]
        assert abs(c["mae"] - m["mae64"]) <= 1e-9 * m["mae64"] and c["atol"] == m["atol32"] and abs(c["pcc"] - m["pcc64"]) <= 1e-7


def test_f13_large_magnitude_mae_host_backend(golden_dir, meta2):
    """The package's threshold rule (float64 moments + literal re-score inside a band that scales with the threshold) and
    greedy scan reproduce the reference's maps where mae scores are ~1e3 (an absolute 2e-6 band would miss the knife edge)."""
    check_f13_backend(golden_dir, meta2, "emulation")


def test_f14_greedy_mae_atol_oracle(golden_dir):
    """The reference's mae / atol greedy maps (by SHA-256), counts and columns from the oracle; and what the device-side atol walk
    rests on: the reference's own atol map does not depend on the seed."""
    from tests.inputs import gen

    f14 = json.loads((golden_dir / "golden_meta_r2b.json").read_text())["f14"]
    assert len(f14) == 16
    for key, w in f14.items():
        x = gen(w["kind"], w["seed"], tuple(w["shape"]))
        assert sha(x) == w["x_sha256"], key
        a, counts, st = orc.greedy(x, ALL, w["metric"], w["threshold"], w["algo_seed"])
        check_summary(w, x, a, counts, st["stats"])
    for key, w in f14.items():
        if w["metric"] == "atol" and w["algo_seed"] == 123:
            assert f14[key.rsplit("|", 1)[0] + "|7"]["assign_sha256"] == w["assign_sha256"], key
