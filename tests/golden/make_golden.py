#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference (johanna-rock/quantization_analysis).

Run in the build container only:  python tests/golden/make_golden.py
The reference lives at /root/reference and never travels; only the arrays written here are
committed.  Fixture list follows SURVEY.md §8(c) F1–F8.  Inputs ≤ 64 K elements are stored;
larger inputs are stored as a (generator, seed, shape) recipe plus a SHA-256 of their bytes.
"""
from __future__ import annotations

import hashlib
import json
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import quantization_formats as qf  # noqa: E402  (reference)
from compression_algorithms import create_algorithm  # noqa: E402  (reference)
from compression_algorithms.cache import CacheContext  # noqa: E402
from compression_algorithms.metrics import pearson_corr  # noqa: E402
from compression_algorithms.quantizer import Quantizer  # noqa: E402
from compression_algorithms.tile_utils import mixed_tile_total_bytes, tile_metrics, reshape_to_2d_with_padding  # noqa: E402

OUT = Path(__file__).resolve().parent
ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def to_bf16_valued(x: np.ndarray) -> np.ndarray:
    """Round fp32 to bf16 values (RNE) with plain integer arithmetic — input preparation only."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def gen(kind: str, seed: int, shape) -> np.ndarray:
    """Input recipes shared with tests/inputs.py (kept byte-identical there)."""
    rng = np.random.default_rng(seed)
    if kind == "normal_bf16":
        return to_bf16_valued((rng.standard_normal(shape) * 0.02).astype(np.float32))
    if kind == "normal_f32":
        return (rng.standard_normal(shape) * 0.02).astype(np.float32)
    if kind == "heavy_bf16":
        a = rng.standard_normal(shape) * 0.02
        return to_bf16_valued((a * np.exp(1.5 * rng.standard_normal(shape))).astype(np.float32))
    if kind == "heavy_f32":
        a = rng.standard_normal(shape) * 0.02
        return (a * np.exp(1.5 * rng.standard_normal(shape))).astype(np.float32)
    raise ValueError(kind)


def sha(x: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def cache_ctx(tmp: str) -> CacheContext:
    return CacheContext(root=Path(tmp), tensor_name="t", backend="emulation", recompute=True, run_tag="golden")


def specials_matrix() -> np.ndarray:
    """F1 input: 64×64 fp32 with wide exponent spread, specials and exact rounding ties."""
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((64, 64)) * np.exp2(rng.integers(-40, 40, size=(64, 64)))).astype(np.float32)
    kat = np.array([1, .75, .3, -.3, .01, 1.9921875, 1.99609375, -1.5, .5, .25, .125, .0625, 3e-39, 0, -0., 1e-3], dtype=np.float32)
    x[0, :16] = kat
    x[1, :16] = np.array([np.inf, 1, 2, -3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], dtype=np.float32)
    x[1, 16:32] = np.array([np.nan, 1, 2, -3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], dtype=np.float32)
    x[2, :16] = np.array([-np.inf, np.nan, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1e-45], dtype=np.float32)
    x[3, :16] = 0.0
    x[3, 16:32] = np.float32(1e-40)  # all-denormal group
    x[4, :16] = np.array([2.0 ** -126 * (1 + i / 16) for i in range(16)], dtype=np.float32)  # shared exp 1
    x[4, 16:32] = np.array([2.0 ** -124 * (1 + i / 16) * (-1) ** i for i in range(16)], dtype=np.float32)  # shared exp 3 (< shift_cnt wrap)
    x[5, :16] = np.array([3.0e38, -3.3e38, 1e38, 2e38, 1.7e38, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11], dtype=np.float32)
    # exact ties for every format: mantissa patterns x.1000… against a 1.0 group maximum
    ties = []
    for m in (7, 3, 1):
        step = 2.0 ** (1 - m)
        ties += [k * step + step / 2 for k in range(0, 4)]
    x[6, :16] = np.array(([1.0] + ties + [1.99999988, -1.99999988, 0.0])[:16], dtype=np.float32)
    x[7, :16] = np.array([1.0, 255 / 256, 127 / 128, 2 ** -8, 2 ** -7, 3 * 2 ** -9, 2 ** -9 * 1.0000001, 2 ** -24, 2 ** -30,
                          -(2 ** -8), 0.4999999, 0.5, 0.5000001, 0.25, 0.75, 0.125], dtype=np.float32)
    return x


def main() -> None:
    meta = {"numpy": np.__version__, "reference": "johanna-rock/quantization_analysis @ /root/reference (2026-03-13 snapshot)"}
    q = Quantizer("emulation")

    # ---------------------------------------------------------------- F1: known-answer bits
    x = specials_matrix()
    f1 = {"x_bits": x.view(np.uint32)}
    with np.errstate(all="ignore"):
        for fmt in ALL + ["fp0"]:
            f1[f"y_{fmt}"] = qf.quantize_weight_values(x, fmt).view(np.uint32)
    np.savez_compressed(OUT / "f1_quantize_kat.npz", **f1)

    # ---------------------------------------------------------------- F2: layouts / ranks
    f2 = {}
    shapes = {"s64x64": (64, 64), "s50x70": (50, 70), "s33": (33,), "s1000": (1000,), "s3x40x48": (3, 40, 48),
              "s2x3x17x19": (2, 3, 17, 19), "s0d": (), "s1x1": (1, 1), "s5x16": (5, 16), "s31x15": (31, 15)}
    for i, (name, shp) in enumerate(shapes.items()):
        xi = gen("heavy_f32", 100 + i, shp if shp != () else (1,))
        if shp == ():
            xi = np.float32(xi[0])
        xi = np.asarray(xi, dtype=np.float32)
        f2[f"{name}_x"] = xi
        for fmt in ALL:
            f2[f"{name}_y_{fmt}"] = np.asarray(qf.quantize_weight_values(xi, fmt), dtype=np.float32)
    np.savez_compressed(OUT / "f2_layouts.npz", **f2)

    # ---------------------------------------------------------------- F4: greedy maps
    greedy_cases = [
        # name, kind, seed, shape, formats(order), metric, thr, algo seed
        ("g_pcc_256_bf16", "normal_bf16", 1, (256, 256), ALL, "pcc", 0.999, 123),
        ("g_pcc_heavy_192x160_bf16", "heavy_bf16", 2, (192, 160), ALL, "pcc", 0.995, 5),
        ("g_pcc_1024x768_f32", "normal_f32", 3, (1024, 768), ALL, "pcc", 0.9995, 11),
        ("g_pcc_50x70", "heavy_f32", 4, (50, 70), ALL, "pcc", 0.999, 9),
        ("g_pcc_vec1000", "normal_f32", 5, (1000,), ALL, "pcc", 0.999, 77),
        ("g_mae_130x200", "normal_f32", 6, (130, 200), ALL, "mae", 2e-4, 31),
        ("g_atol_130x200", "normal_f32", 7, (130, 200), ALL, "atol", 4e-3, 32),
        ("g_mae_vec1003", "heavy_f32", 8, (1003,), ALL, "mae", 5e-4, 33),
        ("g_atol_3x40x48", "heavy_bf16", 9, (3, 40, 48), ALL, "atol", 2e-2, 34),
        ("g_pcc_257x95", "heavy_bf16", 10, (257, 95), ALL, "pcc", 0.998, 35),
        ("g_pcc_order_bfp8_bfp4", "normal_bf16", 11, (128, 256), ["bfp8", "bfp4"], "pcc", 0.995, 36),
        ("g_pcc_order_rev", "normal_bf16", 12, (128, 128), ["bfp4", "bfp8", "bf16"], "pcc", 0.99, 37),
        ("g_pcc_2048x1024_bf16", "normal_bf16", 13, (2048, 1024), ALL, "pcc", 0.998, 38),
        ("g_atol_heavy_96x96", "heavy_f32", 14, (96, 96), ALL, "atol", 5e-2, 39),
    ]
    f4 = {}
    f4_meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, kind, seed, shape, fmts, metric, thr, aseed in greedy_cases:
            xi = gen(kind, seed, shape)
            algo = create_algorithm("mixed-tile-greedy", {"metric": metric, "threshold": thr, "seed": aseed, "formats": fmts})
            res = algo.run(xi, fmts, q, cache_ctx(tmp))[0]
            y = res.y
            diff = np.abs(xi - y)
            f4[f"{name}_assign"] = res.meta["assignment"].astype(np.int8)
            f4[f"{name}_counts"] = np.array([res.tile_counts[f] for f in ALL], dtype=np.int64)
            f4[f"{name}_cols"] = np.array([pearson_corr(xi, y), float(np.mean(diff)), float(np.max(diff)), res.tile_bytes], dtype=np.float64)
            if xi.size <= 65536:
                f4[f"{name}_x"] = xi
                f4[f"{name}_y"] = y.astype(np.float32)
            f4_meta[name] = {"kind": kind, "seed": seed, "shape": list(shape), "formats": fmts, "metric": metric,
                             "threshold": thr, "algo_seed": aseed, "x_sha256": sha(xi), "y_sha256": sha(y.astype(np.float32))}
    np.savez_compressed(OUT / "f4_greedy.npz", **f4)
    meta["f4"] = f4_meta

    # ---------------------------------------------------------------- F5: threshold maps + float32 scores
    const_tiles = np.zeros((64, 96), dtype=np.float32)
    const_tiles[:32, 32:64] = 1.0          # constant tile → denom == 0 branch (metrics.py:14-15)
    const_tiles[32:, :32] = gen("normal_f32", 40, (32, 32))
    const_tiles[32:, 64:] = 0.3            # constant, not exactly representable in bfp4/bfp2
    thr_cases = [
        ("t_pcc94_256_bf16", gen("normal_bf16", 21, (256, 256)), ALL, "pcc", 0.94),
        ("t_pcc999_256_bf16", gen("normal_bf16", 21, (256, 256)), ALL, "pcc", 0.999),
        ("t_pcc99_heavy_160x224", gen("heavy_f32", 22, (160, 224)), ALL, "pcc", 0.99),
        ("t_mae_130x200", gen("normal_f32", 23, (130, 200)), ALL, "mae", 3e-4),
        ("t_atol_130x200", gen("normal_f32", 24, (130, 200)), ALL, "atol", 3e-3),
        ("t_pcc_50x70", gen("heavy_f32", 25, (50, 70)), ALL, "pcc", 0.99),
        ("t_pcc_vec1003", gen("normal_f32", 26, (1003,)), ALL, "pcc", 0.99),
        ("t_pcc_const", const_tiles, ALL, "pcc", 0.99),
        ("t_pcc_subset", gen("normal_bf16", 27, (96, 128)), ["bfp8", "bfp2"], "pcc", 0.95),
        ("t_pcc_3x40x48", gen("heavy_bf16", 28, (3, 40, 48)), ALL, "pcc", 0.98),
    ]
    # thresholds placed INSIDE the per-tile score distribution so the maps are mixed, plus knife-edge
    # thresholds: exactly a tile's float32 score, and that score + 1e-9 (still equal after the float32
    # rounding NumPy >= 2 applies to the Python-float threshold).
    def _scores(xi, fmt, metric):
        padded, _si, pad = reshape_to_2d_with_padding(xi)
        th_, tw_ = pad[2] // 32, pad[3] // 32
        tr = padded.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
        pq, _, _ = reshape_to_2d_with_padding(q.quantize(xi, fmt))
        tq = pq.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
        return np.asarray(tile_metrics(tr, tq, metric), dtype=np.float32)

    xs = gen("normal_bf16", 51, (256, 320))
    s4 = _scores(xs, "bfp4", "pcc")
    thr_cases.append(("t_pcc_split_bfp4_median", xs, ALL, "pcc", round(float(np.median(s4)), 6)))
    thr_cases.append(("t_pcc_knife_eq", xs, ALL, "pcc", float(np.sort(s4)[len(s4) // 3])))
    thr_cases.append(("t_pcc_knife_eps", xs, ALL, "pcc", float(np.sort(s4)[len(s4) // 3]) + 1e-9))
    xh = gen("heavy_f32", 52, (224, 192))
    s8 = _scores(xh, "bfp8", "pcc")
    thr_cases.append(("t_pcc_split_bfp8_heavy", xh, ALL, "pcc", round(float(np.median(s8)), 6)))
    sm = _scores(xh, "bfp4", "mae")
    thr_cases.append(("t_mae_split_heavy", xh, ALL, "mae", float(np.median(sm))))
    sa = _scores(xh, "bfp4", "atol")
    thr_cases.append(("t_atol_split_heavy", xh, ALL, "atol", float(np.median(sa))))

    f5 = {}
    f5_meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, xi, fmts, metric, thr in thr_cases:
            algo = create_algorithm("mixed-tile-threshold", {"metric": metric, "threshold": thr, "formats": fmts})
            res = algo.run(xi, fmts, q, cache_ctx(tmp))[0]
            y = res.y
            diff = np.abs(xi - y)
            f5[f"{name}_x"] = xi
            f5[f"{name}_assign"] = res.meta["assignment"].astype(np.int8)
            f5[f"{name}_counts"] = np.array([res.tile_counts[f] for f in ALL], dtype=np.int64)
            f5[f"{name}_cols"] = np.array([pearson_corr(xi, y), float(np.mean(diff)), float(np.max(diff)), res.tile_bytes], dtype=np.float64)
            f5[f"{name}_y"] = y.astype(np.float32)
            # the literal float32 per-tile scores the reference compared (mixed_tile_threshold.py:97-109)
            padded, _si, pad = reshape_to_2d_with_padding(xi)
            th_, tw_ = pad[2] // 32, pad[3] // 32
            tiles_ref = padded.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
            for fmt in fmts:
                pq, _, _ = reshape_to_2d_with_padding(q.quantize(xi, fmt))
                tq = pq.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
                f5[f"{name}_score_{fmt}"] = np.asarray(tile_metrics(tiles_ref, tq, metric), dtype=np.float32)
            f5_meta[name] = {"formats": fmts, "metric": metric, "threshold": thr}
    np.savez_compressed(OUT / "f5_threshold.npz", **f5)
    meta["f5"] = f5_meta


    # ---------------------------------------------------------------- F6: threshold sweep rows
    # the reference's sweep is a script (scripts/sweep_mixed_tile_threshold.py:581-821); its selection and pareto
    # helpers are loaded from the file and driven with the data flow of its main loop (:626-790)
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_sweep", REF + "/scripts/sweep_mixed_tile_threshold.py")
    ref_sweep = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_sweep)
    from compression_algorithms.tile_utils import MIXED_TILE_BYTES_PER_ELEM, MIXED_TILE_FORMATS, reconstruct_from_tiles

    f6 = {}
    for tag, kind, shape, metric, lowest, steps in (("pcc", "normal_bf16", (256, 256), "pcc", 0.9, 10),
                                                    ("mae", "heavy_f32", (160, 224), "mae", 0.02, 8),
                                                    ("atol", "heavy_f32", (96, 128), "atol", 0.3, 6)):
        xf = gen(kind, 61, shape)
        padded_ref, shape_info, pad_info = reshape_to_2d_with_padding(xf)
        th_, tw_ = pad_info[2] // 32, pad_info[3] // 32
        tiles_ref = padded_ref.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
        tiles_by_fmt, scores_by_fmt = {}, {}
        for fmt in ALL:
            pq, _, _ = reshape_to_2d_with_padding(q.quantize(xf, fmt))
            tq = pq.reshape(th_, 32, tw_, 32).transpose(0, 2, 1, 3).reshape(-1, 32, 32)
            tiles_by_fmt[fmt] = tq
            scores_by_fmt[fmt] = tile_metrics(tiles_ref, tq, metric)
        by_prec = sorted(ALL, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))
        hi = max(by_prec, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))
        fmt_order = {f: i for i, f in enumerate(by_prec)}
        ss = np.stack([scores_by_fmt[f] for f in by_prec])
        tstack = np.stack([tiles_by_fmt[f] for f in by_prec])
        start = float(np.max(scores_by_fmt[hi])) if metric == "pcc" else float(np.min(scores_by_fmt[hi]))
        rows = []
        for k, t in enumerate(np.linspace(start, lowest, max(1, steps))):
            a = ref_sweep._compute_assignment(ss, metric, float(t))
            y = reconstruct_from_tiles(tstack[a, np.arange(a.size)], shape_info, pad_info)
            diff = np.abs(xf - y)
            raw = np.bincount(a, minlength=4)
            counts = {f: 0 for f in MIXED_TILE_FORMATS}
            for f, i in fmt_order.items():
                counts[f] = int(raw[i])
            rows.append([k, float(t), mixed_tile_total_bytes(counts), pearson_corr(xf, y), float(np.mean(diff)), float(np.max(diff)),
                         *[counts[f] for f in ALL]])
        f6[f"{tag}_rows"] = np.asarray(rows, dtype=np.float64)
        col = 3 if metric == "pcc" else (4 if metric == "mae" else 5)
        f6[f"{tag}_pareto"] = np.asarray(ref_sweep._pareto_mask([{"size": r[2], "metric": r[col]} for r in rows], metric))
    np.savez_compressed(OUT / "f6_sweep.npz", **f6)


    # ---------------------------------------------------------------- F10: random search (mixed_tile_random.py)
    random_cases = [
        # name, kind, seed, shape, formats, metric, thr, iters, algo seed.  Thresholds sit inside the spread of the
        # samples' scores so that both branches (:160-167 smallest passing map, :168-172 best failing map) are taken.
        ("r_pcc_256_bf16", "normal_bf16", 71, (256, 256), ALL, "pcc", 0.9705, 12, 0),
        # knife edge: the threshold IS sample score #5 of the case above (float32 value as a Python float), and that + 1e-9
        ("r_pcc_knife_eq", "normal_bf16", 71, (256, 256), ALL, "pcc", 0.9704277515411377, 12, 0),
        ("r_pcc_knife_eps", "normal_bf16", 71, (256, 256), ALL, "pcc", 0.9704277515411377 + 1e-9, 12, 0),
        ("r_pcc_none_pass", "normal_bf16", 72, (128, 192), ALL, "pcc", 0.9999, 10, 3),
        ("r_pcc_heavy_160x96", "heavy_f32", 73, (160, 96), ALL, "pcc", 0.9885, 16, 5),
        ("r_mae_130x200", "normal_f32", 74, (130, 200), ALL, "mae", 2.2e-3, 12, 7),
        ("r_atol_50x70", "heavy_f32", 75, (50, 70), ALL, "atol", 0.2, 10, 11),
        ("r_pcc_subset", "normal_bf16", 76, (96, 128), ["bfp4", "bfp8"], "pcc", 0.9972, 14, 13),
        ("r_pcc_single", "normal_bf16", 77, (64, 64), ["bfp4"], "pcc", 0.99, 3, 17),
        ("r_pcc_vec1003", "normal_f32", 78, (1003,), ALL, "pcc", 0.96, 9, 19),
        ("r_mae_3x40x48", "heavy_bf16", 79, (3, 40, 48), ["bf16", "bfp8", "bfp2"], "mae", 6e-3, 8, 23),
    ]
    f10 = {}
    f10_meta = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, kind, seed, shape, fmts, metric, thr, iters, aseed in random_cases:
            xi = gen(kind, seed, shape)
            algo = create_algorithm("mixed-tile-random", {"metric": metric, "threshold": thr, "seed": aseed, "iters": iters, "formats": fmts})
            res = algo.run(xi, fmts, q, cache_ctx(tmp))[0]
            y = res.y
            diff = np.abs(xi - y)
            smp = res.meta["samples"]
            f10[f"{name}_x"] = xi
            f10[f"{name}_assign"] = res.meta["assignment"].astype(np.int8)
            f10[f"{name}_counts"] = np.array([res.tile_counts[f] for f in ALL], dtype=np.int64)
            f10[f"{name}_cols"] = np.array([pearson_corr(xi, y), float(np.mean(diff)), float(np.max(diff)), res.tile_bytes], dtype=np.float64)
            f10[f"{name}_y"] = y.astype(np.float32)
            f10[f"{name}_samples"] = np.array([[s_["id"], s_["total_bytes"], s_["pcc"], s_["mae"], s_["atol"], *[s_["counts"][f] for f in ALL]]
                                               for s_ in smp], dtype=np.float64)
            f10_meta[name] = {"formats": fmts, "metric": metric, "threshold": thr, "iters": iters, "algo_seed": aseed}
    np.savez_compressed(OUT / "f10_random.npz", **f10)
    meta["f10"] = f10_meta

    # ---------------------------------------------------------------- F9: fp8 x scale_inv block dequantisation (loader)
    import torch
    from hf_model_utils import _dequantize_tensor_with_scale_inv

    f9 = {}
    rng9 = np.random.default_rng(99)
    for tag, shape, sshape in (("all_codes", (16, 256), (1, 2)), ("blocks128", (300, 260), (3, 3)), ("ragged", (130, 47), (2, 1)), ("rowscale", (8, 64), (8, 1))):
        if tag == "all_codes":
            wb = np.tile(np.arange(256, dtype=np.uint8), (16, 1))
        else:
            wb = rng9.integers(0, 256, size=shape, dtype=np.uint8)
        sc = np.exp(rng9.standard_normal(sshape)).astype(np.float32) * np.float32(0.01)
        w = torch.from_numpy(wb).view(torch.float8_e4m3fn)
        out = _dequantize_tensor_with_scale_inv(w, torch.from_numpy(sc)).to(dtype=torch.float32).numpy()
        f9[f"{tag}_w"] = wb
        f9[f"{tag}_scale"] = sc
        f9[f"{tag}_out_bits"] = out.view(np.uint32)
    np.savez_compressed(OUT / "f9_fp8_dequant.npz", **f9)

    # ---------------------------------------------------------------- F8: RNG drift + misc scalars
    f8 = {
        "perm_123_16384_head": np.random.default_rng(123).permutation(16384)[:32].astype(np.int64),
        "perm_5_subset": np.random.default_rng(5).permutation(np.array([3, 9, 10, 40, 41, 77, 100], dtype=np.int64)),
        "total_bytes_check": np.array([mixed_tile_total_bytes({"bf16": 0, "bfp8": 0, "bfp4": 128088, "bfp2": 936})]),
        "nep50_f32_ge": np.array([bool(np.float32(0.94) >= 0.94), bool(np.float32(0.999) >= 0.999)]),
        "integers_7_4_head": np.random.default_rng(7).integers(0, 4, size=64, dtype=np.int64),
        "integers_7_3_after_perm": (lambda g: (g.permutation(10), g.integers(0, 3, size=40, dtype=np.int64))[1])(np.random.default_rng(7)),
    }
    np.savez_compressed(OUT / "f8_misc.npz", **f8)

    (OUT / "golden_meta.json").write_text(json.dumps(meta, indent=1, sort_keys=True))
    print("wrote", sorted(p.name for p in OUT.glob("*.npz")))


if __name__ == "__main__":
    main()
