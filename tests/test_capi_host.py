"""CPU-only: the C-ABI library loads, exports every symbol include/mtq.h declares, refuses device work
without a GPU, and its HOST decision functions (greedy scan, threshold assign, columns) agree with the
oracle when fed the oracle's stats records."""
import ctypes
import json
import os
import re
from pathlib import Path

import numpy as np
import pytest

from oracle import mtq_oracle as orc
from quantization_analysis_amd import hip_backend as hb
from tests.inputs import gen

ROOT = Path(__file__).resolve().parent.parent
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def test_header_symbols_exported():
    hdr = (ROOT / "include" / "mtq.h").read_text()
    declared = set(re.findall(r"\b(mtq_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(hb.EXPORTS), declared ^ set(hb.EXPORTS)
    L = hb.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.mtq_version() == 143
    assert L.mtq_stats_record_doubles(0xF) == 22 and L.mtq_stats_record_doubles(0b0110) == 12


def test_work_counter_slot_bookkeeping():
    """K1's work-counter slots (csrc/mtq_slot_ring.hpp) against mock events, on the host: round robin, a slot held between
    acquire and release is never shared, every reuse waits for the event its previous user recorded."""
    assert hb.lib().mtq_selftest_slot_ring() == 0


def test_argument_errors_are_reported():
    L = hb.lib()
    assert L.mtq_tile_stats(None, 0, 32, 32, 32, 0xF, None, None) == -1
    assert b"null" in L.mtq_last_error()
    buf = np.zeros(4, dtype=np.float64)
    assert L.mtq_tile_stats(buf.ctypes.data, 7, 32, 32, 32, 0xF, buf.ctypes.data, None) == -1
    assert L.mtq_tile_stats(buf.ctypes.data, 0, 32, 32, 16, 0xF, buf.ctypes.data, None) == -1  # ld < cols
    assert L.mtq_quantize(buf.ctypes.data, 0, 32, 32, 32, 9, buf.ctypes.data, 32, None) == -4
    h = ctypes.c_void_p()
    assert L.mtq_greedy_create(ctypes.byref(h), buf.ctypes.data, 1, 0b0010, 0, 0.9, 1.0, 0) == -1  # base not in mask
    # round 3's device entry points check their arguments before they look for a device
    p = buf.ctypes.data
    fm = (ctypes.c_int * 4)(0, 1, 2, 3)
    assert L.mtq_knife_tiles_device(p, 0, 1, 1024, 32, 32, 32, None, fm, 4, 8, p, p, None) == -1           # no masks
    assert L.mtq_knife_tiles_device(p, 0, 1, 1024, 32, 32, 32, p, fm, 5, 8, p, p, None) == -1              # more formats than exist
    bad = (ctypes.c_int * 4)(0, 1, 2, 7)
    assert L.mtq_knife_tiles_device(p, 0, 1, 1024, 32, 32, 32, p, bad, 4, 8, p, p, None) == -1 and b"format codes" in L.mtq_last_error()
    assert L.mtq_knife_tiles_device(p, 0, 1, 1024, 32, 32, 32, p, fm, 4, 8, p, None, None) == -1           # a list without room for the tiles
    assert L.mtq_scan_orders_device(1, 16, 3, p, 1 << 20, None) < 0 and L.mtq_scan_orders_device(1, 16, 1, None, 0, None) < 0
    assert int(L.mtq_scan_orders_bytes(16384)) >= 128 + 2 * 4 * 16384 and int(L.mtq_scan_carry_bytes(3)) == 3 * int(L.mtq_scan_carry_bytes(1))
    assert int(L.mtq_greedy_scan_scratch_bytes(2, 100)) >= 2 * (100 * 8 * 8 + 100 * 4)
    # round 4: a threshold batch as one call — arguments first, device second
    assert L.mtq_threshold_enqueue(p, 0, 1, 1024, 32, 32, 32, 0xE, 0x1E, fm, 4, 0, 0.999, 2e-6, None, p, p, 8, p, p, p, None, None, None, None) == -1 and b"null" in L.mtq_last_error()
    assert L.mtq_threshold_enqueue(p, 0, 0, 1024, 32, 32, 32, 0xE, 0x1E, fm, 4, 0, 0.999, 2e-6, p, p, p, 8, p, p, p, None, None, None, None) == -1     # count 0
    assert L.mtq_threshold_enqueue(p, 0, 1, 1024, 32, 32, 32, 0xE, 0x1E, fm, 4, 0, 0.999, 2e-6, p, p, p, -1, p, p, p, None, None, None, None) == -1    # negative cap
    assert L.mtq_threshold_columns(p, 1, 1, 0xE, p, p, None, None) == -1 and b"null" in L.mtq_last_error()
    # ragged batches: the table's limits and every matrix checked before the device is asked for
    M = (hb.MtqMatrix * 25)(*[hb.MtqMatrix(p, 32, 32, 32) for _ in range(25)])
    assert L.mtq_tile_stats_ragged(None, 1, 1, 0xF, p, None) == -1 and b"null" in L.mtq_last_error()
    assert L.mtq_tile_stats_ragged(M, 0, 1, 0xF, p, None) == -1 and L.mtq_tile_stats_ragged(M, 25, 1, 0xF, p, None) == -1 and b"MTQ_RAGGED_MAX" in L.mtq_last_error()
    assert L.mtq_tile_stats_ragged(M, 2, 7, 0xF, p, None) == -1                                             # unknown storage type
    assert L.mtq_tile_stats_ragged(M, 2, 1, 0xF, None, None) == -1 and L.mtq_tile_stats_ragged(M, 2, 1, 0x1F, p, None) == -1
    M[1].ld = 16
    assert L.mtq_tile_stats_ragged(M, 2, 1, 0xF, p, None) == -1                                             # a leading dimension below cols
    M[1].ld, M[1].rows = 32, 0
    assert L.mtq_knife_tiles_ragged(M, 2, 1, p, fm, 4, 8, p, p, None) == -1
    assert L.mtq_threshold_enqueue_ragged(M, 2, 1, 0xF, 0xF, fm, 4, 0, 0.999, 2e-6, p, p, p, 8, p, p, p, None, None, None, None) == -1
    M[1].rows = 32
    assert L.mtq_threshold_enqueue_ragged(M, 2, 1, 0xF, 0xF, fm, 4, 0, 0.999, 2e-6, None, p, p, 8, p, p, p, None, None, None, None) == -1 and b"null" in L.mtq_last_error()
    per = (ctypes.c_int64 * 3)(4, 0, 9)
    assert L.mtq_column_sums_device_ragged(p, per, 3, 0xF, p, p, None) == -1 and b"tiles" in L.mtq_last_error()   # a tensor without tiles
    assert L.mtq_column_sums_device_ragged(p, per, 30, 0xF, p, p, None) == -1
    assert L.mtq_threshold_columns_ragged(p, per, 1, 0xF, p, p, None, None) == -1 and b"null" in L.mtq_last_error()


def test_settings_are_read_once_and_refreshable(monkeypatch):
    """quantization_analysis_amd/settings.py: the documented defaults, one read per process, refresh on request."""
    from quantization_analysis_amd.settings import settings

    for k in ("MTQ_PIPE_SLOTS", "MTQ_LAZY", "MTQ_SCAN_STREAMS", "MTQ_KNIFE_CAP"):
        monkeypatch.delenv(k, raising=False)
    s = settings(refresh=True)
    assert (s.pipe_slots, s.lazy, s.scan_streams, s.knife_cap, s.shared_orders, s.lazy_max_listed) == (4, True, None, 128, True, 0.35)
    monkeypatch.setenv("MTQ_PIPE_SLOTS", "6")
    monkeypatch.setenv("MTQ_LAZY", "0")
    assert settings().pipe_slots == 4 and settings().lazy is True            # not re-read behind the process's back
    s = settings(refresh=True)
    assert s.pipe_slots == 6 and s.lazy is False


def test_no_gpu_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(hb.MtqError):
        hb.require_gpu()
    buf = np.zeros(64, dtype=np.float64)
    rc = hb.lib().mtq_tile_stats(buf.ctypes.data, 1, 32, 32, 32, 0xF, buf.ctypes.data, None)
    assert rc == -3 and b"no CPU fallback" in hb.lib().mtq_last_error()


def _host_greedy(x, formats, metric, thr, seed):
    x2d, _ = orc.flatten_2d(x)
    fm = [f for f in ALL if f in formats]
    mask = hb.fmt_mask(fm)
    stats = orc.tile_stats(x2d, fm)
    scan = hb.GreedyScan(stats, mask, metric, thr, float(x.size), formats[0])
    rng = np.random.default_rng(seed)
    for fmt in formats:
        cand = np.where(scan.fixed() == 0)[0]
        if cand.size == 0:
            break
        scan.run_pass(fmt, rng.permutation(cand))
    return scan.assignment(), scan.counts(), scan.value(), stats, mask


def test_host_greedy_scan_matches_golden_maps(golden_dir):
    d = np.load(golden_dir / "f4_greedy.npz")
    meta = json.loads((golden_dir / "golden_meta.json").read_text())["f4"]
    for name, m in meta.items():
        x = gen(m["kind"], m["seed"], tuple(m["shape"]))
        a, counts, value, stats, mask = _host_greedy(x, m["formats"], m["metric"], m["threshold"], m["algo_seed"])
        want = d[f"{name}_assign"]
        assert np.array_equal(a.reshape(want.shape), want), name
        assert [counts[f] for f in ALL] == list(d[f"{name}_counts"]), name
        oa, _oc, ost = orc.greedy(x, m["formats"], m["metric"], m["threshold"], m["algo_seed"])
        assert value == ost["value"], name  # same doubles, same order of operations
        cols = hb.columns_from_stats(stats, mask, a, x.size)
        slots = orc.mask_slots(mask)
        pcc, mae, atol = orc.columns_from_stats(stats, slots, oa, x.size)
        assert (cols["pcc"], cols["mae"], cols["atol"]) == (pcc, mae, atol), name


def test_host_threshold_assign_matches_golden_maps(golden_dir):
    d = np.load(golden_dir / "f5_threshold.npz")
    meta = json.loads((golden_dir / "golden_meta.json").read_text())["f5"]
    total_knife = 0
    for name, m in meta.items():
        x = d[f"{name}_x"]
        x2d, _ = orc.flatten_2d(x)
        fm = [f for f in ALL if f in m["formats"]]
        mask = hb.fmt_mask(fm)
        stats = orc.tile_stats(x2d, fm)
        amap, knife = hb.threshold_assign(stats, mask, m["formats"], m["metric"], m["threshold"], band=2e-6)
        want = d[f"{name}_assign"].reshape(-1)
        # outside the knife-edge band the float64-moment decision IS the reference's float32 decision
        off = np.setdiff1d(np.where(amap != want)[0], knife)
        assert off.size == 0, (name, off)
        total_knife += knife.size
        # scores agree with the reference's float32 scores within the band
        sc = hb.tile_scores(stats, mask, m["metric"])
        for i, f in enumerate(fm):
            assert np.max(np.abs(sc[i] - d[f"{name}_score_{f}"].astype(np.float64))) <= 1e-6, (name, f)
    assert total_knife > 0  # the knife-edge fixtures must exercise the band


def test_knife_tiles_are_decided_lazily_and_like_the_reference(golden_dir):
    """decide_knife_tiles evaluates the literal float32 score only for the (tile, format) pairs inside the band — with a very
    wide band (many knife tiles, partial masks) the patched map must still be the reference's map, and fewer literal
    evaluations than tiles x formats must have been needed."""
    from quantization_analysis_amd.compression_algorithms.mixed_tile_threshold import decide_knife_tiles
    from quantization_analysis_amd.compression_algorithms.tile_utils import reshape_to_2d_with_padding, tile_metrics, to_tiles

    d = np.load(golden_dir / "f5_threshold.npz")
    meta = json.loads((golden_dir / "golden_meta.json").read_text())["f5"]
    evaluated = possible = 0
    for name, m in meta.items():
        x = d[f"{name}_x"]
        x2d, _ = orc.flatten_2d(x)
        fm = [f for f in ALL if f in m["formats"]]
        mask = hb.fmt_mask(fm)
        stats = orc.tile_stats(x2d, fm)
        band = 5e-3 if m["metric"] == "pcc" else 0.2 * float(m["threshold"])
        amap, knife, near = hb.threshold_assign(stats, mask, m["formats"], m["metric"], m["threshold"], band=band, with_near=True)
        assert np.all(near != 0) and knife.size == near.size
        x_tiles = to_tiles(reshape_to_2d_with_padding(x)[0])
        y_tiles = {f: to_tiles(reshape_to_2d_with_padding(orc.quantize_weight_values(x, f))[0]) for f in m["formats"]}
        calls = []

        def literal(fmt, sel):
            calls.append(sel.size)
            return tile_metrics(x_tiles[knife[sel]], y_tiles[fmt][knife[sel]], m["metric"])

        amap[knife] = decide_knife_tiles(amap[knife], near, m["formats"], m["metric"], m["threshold"], literal)
        assert np.array_equal(amap, d[f"{name}_assign"].reshape(-1)), name
        evaluated += sum(calls)
        possible += knife.size * len(m["formats"])
    assert 0 < evaluated < possible


def test_threshold_best_precision_and_order():
    # all tiles fail → highest-bytes format among the requested ones (mixed_tile_threshold.py:115-117)
    x = gen("heavy_f32", 3, (64, 64))
    x2d, _ = orc.flatten_2d(x)
    stats = orc.tile_stats(x2d, ["bfp8", "bfp4", "bfp2"])
    amap, _ = hb.threshold_assign(stats, 0b1110, ["bfp2", "bfp8", "bfp4"], "pcc", 1.5)
    assert set(amap.tolist()) == {1}
    amap, _ = hb.threshold_assign(stats, 0b1110, ["bfp2", "bfp4"], "pcc", -1.0)
    assert set(amap.tolist()) == {3}


def test_numpy_compatible_rng():
    """mtq_rng must reproduce np.random.default_rng(seed).permutation(n) call for call (the greedy visiting order)."""
    meta_rng = np.random.default_rng(2026)
    seeds = [1, 5, 123, 2**31 - 1, 2**32 - 1, 2**32, 2**40 + 3, 2**63 + 11] + [int(s) for s in meta_rng.integers(1, 2**62, size=8)]
    for seed in seeds:
        mine, ref = hb.NumpyCompatRng(seed), np.random.default_rng(seed)
        for n in (16384, 1, 0, 2, 3, 1000, 65537, 7, 129024 // 3):
            assert np.array_equal(mine.permutation(n), ref.permutation(n)), (seed, n)
    c = np.sort(meta_rng.choice(100000, 777, replace=False))
    assert np.array_equal(np.random.default_rng(9).permutation(c), c[hb.NumpyCompatRng(9).permutation(c.size)])


def test_numpy_compatible_rng_integers(golden_dir):
    """mtq_rng_integers ≡ Generator.integers(0, high, size=n, dtype=int64) (the random search's draws,
    mixed_tile_random.py:135), also interleaved with permutation calls on the same generator."""
    d = np.load(golden_dir / "f8_misc.npz")
    assert np.array_equal(hb.NumpyCompatRng(7).integers(4, 64), d["integers_7_4_head"])
    r = hb.NumpyCompatRng(7)
    r.permutation(10)
    assert np.array_equal(r.integers(3, 40), d["integers_7_3_after_perm"])
    for seed in (0, 1, 123, 2**31 - 1, 2**40 + 7):
        for high in (1, 2, 3, 4, 5, 7, 100, 2**31 + 3, 2**32 - 2):
            mine, ref = hb.NumpyCompatRng(seed), np.random.default_rng(seed)
            for n in (0, 1, 7, 4097):
                assert np.array_equal(mine.integers(high, n), ref.integers(0, high, size=n, dtype=np.int64)), (seed, high, n)
                assert np.array_equal(mine.permutation(33), ref.permutation(33)), (seed, high, n)
    with pytest.raises(hb.MtqError):
        hb.NumpyCompatRng(1).integers(0, 4)
    with pytest.raises(hb.MtqError):
        hb.NumpyCompatRng(1).integers(2**32, 4)


def test_greedy_run_equals_numpy_driven_scan(golden_dir):
    d = np.load(golden_dir / "f4_greedy.npz")
    meta = json.loads((golden_dir / "golden_meta.json").read_text())["f4"]
    for name, m in meta.items():
        x = gen(m["kind"], m["seed"], tuple(m["shape"]))
        a_np, counts_np, value_np, stats, mask = _host_greedy(x, m["formats"], m["metric"], m["threshold"], m["algo_seed"])
        a_c, counts_c, cols = hb.greedy_run(stats, mask, m["formats"], m["metric"], m["threshold"], float(x.size), m["algo_seed"])
        assert np.array_equal(a_c, a_np) and counts_c == counts_np, name
        assert np.array_equal(a_c.reshape(d[f"{name}_assign"].shape), d[f"{name}_assign"]), name
    with pytest.raises(hb.MtqError):
        hb.greedy_run(stats, mask, ["bf16"], "pcc", 0.9, 1.0, 0)  # seed 0 must be resolved by the caller


def test_numa_binding_helpers():
    assert hb.parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and hb.parse_cpulist("") == set()
    before = os.sched_getaffinity(0)
    msg = hb.bind_to_gpu_numa_node(0)          # no GPU here: must say so and leave the affinity alone
    assert isinstance(msg, str) and (os.sched_getaffinity(0) == before or "NUMA node" in msg)


def test_identity_bf16_mask_equals_full_records():
    """MTQ_MASK_BF16_IDENTITY: for bf16-valued data the bf16 slot of a record is [Σx, Σx², Σx², 0, 0]; host functions given
    the 17-double records (BFP slots only) and the identity flag must decide exactly like the 22-double records."""
    for kind, shape, metric, thr in (("normal_bf16", (256, 256), "pcc", 0.999), ("heavy_bf16", (192, 160), "pcc", 0.995),
                                     ("heavy_bf16", (96, 160), "mae", 5e-4), ("heavy_bf16", (96, 160), "atol", 2e-2)):
        x = gen(kind, 3, shape)
        full = orc.tile_stats(x, ALL)
        z = np.zeros(full.shape[0])
        assert np.array_equal(full[:, 2:7], np.stack([full[:, 0], full[:, 1], full[:, 1], z, z], axis=1))
        slim = np.ascontiguousarray(np.delete(full, np.s_[2:7], axis=1))
        assert slim.shape[1] == hb.record_doubles(0xE) == 17
        ident = 0xE | hb.MASK_BF16_IDENTITY
        for fmts in (ALL, ["bf16", "bfp4"], ["bfp8", "bf16", "bfp2"]):
            a1, c1, o1 = hb.greedy_run(full, 0xF, fmts, metric, thr, float(x.size), 77)
            a2, c2, o2 = hb.greedy_run(slim, ident, fmts, metric, thr, float(x.size), 77)
            assert np.array_equal(a1, a2) and c1 == c2 and o1 == o2, (kind, metric, fmts)
        m1, k1 = hb.threshold_assign(full, 0xF, ALL, metric, thr, 2e-6)
        m2, k2 = hb.threshold_assign(slim, ident, ALL, metric, thr, 2e-6)
        assert np.array_equal(m1, m2) and np.array_equal(k1, k2)
        assert np.array_equal(hb.tile_scores(full, 0xF, metric).view(np.uint64), hb.tile_scores(slim, ident, metric).view(np.uint64))
        amap = np.random.default_rng(1).integers(0, 4, full.shape[0]).astype(np.int8)
        assert hb.columns_from_stats(full, 0xF, amap, float(x.size)) == hb.columns_from_stats(slim, ident, amap, float(x.size))
    with pytest.raises(hb.MtqError):  # without the flag format 0 is simply absent
        hb.greedy_run(slim, 0xE, ALL, "pcc", 0.999, 1024.0, 5)


def test_slim_records_equal_full_records_for_pcc():
    """MTQ_MASK_SLIM: 3 doubles per format (Σy, Σy², Σxy) are all the pcc scan reads, except in the zero-variance case."""
    for kind, shape, thr in (("normal_bf16", (256, 256), 0.999), ("heavy_f32", (160, 224), 0.99)):
        x = gen(kind, 4, shape)
        full = orc.tile_stats(x, ALL)
        keep = [0, 1] + [2 + 5 * s + k for s in range(4) for k in range(3)]
        slim = np.ascontiguousarray(full[:, keep])
        assert slim.shape[1] == hb.record_doubles(0xF | hb.MASK_SLIM) == 14
        for fmts in (ALL, ["bfp8", "bfp4"]):
            a1, c1, _o1 = hb.greedy_run(full, 0xF, fmts, "pcc", thr, float(x.size), 31)
            a2, c2, o2 = hb.greedy_run(slim, 0xF | hb.MASK_SLIM, fmts, "pcc", thr, float(x.size), 31)
            assert np.array_equal(a1, a2) and c1 == c2 and np.isnan(o2["pcc"])
    # identity + slim together (what the streamed driver sends for bf16 storage): 2 + 3*3 doubles
    x = gen("normal_bf16", 5, (128, 256))
    full = orc.tile_stats(x, ALL)
    both = np.ascontiguousarray(full[:, [0, 1] + [2 + 5 * s + k for s in (1, 2, 3) for k in range(3)]])
    a1, c1, _ = hb.greedy_run(full, 0xF, ALL, "pcc", 0.999, float(x.size), 9)
    a2, c2, _ = hb.greedy_run(both, 0xE | hb.MASK_BF16_IDENTITY | hb.MASK_SLIM, ALL, "pcc", 0.999, float(x.size), 9)
    assert np.array_equal(a1, a2) and c1 == c2 and both.shape[1] == 11
    const = orc.tile_stats(np.full((64, 64), 0.5, dtype=np.float32), ALL)
    with pytest.raises(hb.MtqError, match="zero-variance"):
        hb.greedy_run(np.ascontiguousarray(const[:, keep]), 0xF | hb.MASK_SLIM, ALL, "pcc", 0.999, 4096.0, 3)
    with pytest.raises(hb.MtqError):
        hb.greedy_run(slim, 0xF | hb.MASK_SLIM, ALL, "mae", 1e-3, float(x.size), 3)


def test_greedy_run_batch_equals_single_runs():
    xs = [gen("normal_bf16", s, (256, 256)) for s in range(5)]
    st = np.stack([orc.tile_stats(x, ALL) for x in xs])
    seeds = [11, 12, 13, 14, 15]
    maps, counts, outs = hb.greedy_run_batch(st, 0xF, ALL, "pcc", 0.999, 256 * 256.0, seeds, n_threads=3)
    for i, x in enumerate(xs):
        a, c, cols = hb.greedy_run(st[i], 0xF, ALL, "pcc", 0.999, 256 * 256.0, seeds[i])
        assert np.array_equal(maps[i], a) and [c[f] for f in ALL] == list(counts[i])
        assert (outs[i, 0], outs[i, 1], outs[i, 2]) == (cols["pcc"], cols["mae"], cols["atol"])
        assert np.array_equal(maps[i].reshape(8, 8), orc.greedy(x, ALL, "pcc", 0.999, seeds[i])[0])
    with pytest.raises(hb.MtqError):
        hb.greedy_run_batch(st, 0xF, ALL, "pcc", 0.999, 1.0, [1, 2, 0, 4, 5], n_threads=2)  # a zero seed is reported, not ignored
