"""oracle/mtq_oracle.py — Python face of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
nothing under quantization_analysis_amd/ does.  Parity status: PINNED by tests/golden/*.npz
(generated from the imported reference by tests/golden/make_golden.py).

Citations are file:line in the reference repository (johanna-rock/quantization_analysis).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libmtq_oracle.so"

MIXED_TILE_FORMATS = ["bf16", "bfp8", "bfp4", "bfp2"]  # tile_utils.py:8
MIXED_TILE_BYTES_PER_ELEM = {"bf16": 2.0, "bfp8": 1.088, "bfp4": 0.50097, "bfp2": 0.25097}  # tile_utils.py:9-14
FMT_CODE = {"bf16": 0, "bfp8": 1, "bfp4": 2, "bfp2": 3, "fp0": 4}
METRIC_CODE = {"pcc": 0, "mae": 1, "atol": 2}
TILE = 32


def build(force: bool = False) -> Path:
    """Compile libmtq_oracle.so with gcc (oracle/Makefile)."""
    src = _HERE / "mtq_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "libmtq_oracle.so"])
    return _LIB_PATH


class _GreedyState(ctypes.Structure):
    _fields_ = [
        ("T", ctypes.c_int64),
        ("metric", ctypes.c_int),
        ("threshold", ctypes.c_double),
        ("n", ctypes.c_double),
        ("sum_x", ctypes.c_double), ("sum_x2", ctypes.c_double), ("sum_y", ctypes.c_double),
        ("sum_y2", ctypes.c_double), ("sum_xy", ctypes.c_double), ("sum_abs", ctypes.c_double),
        ("max_abs", ctypes.c_double),
        ("max_abs_count", ctypes.c_int64),
        ("t_sy", ctypes.c_void_p), ("t_sy2", ctypes.c_void_p), ("t_sxy", ctypes.c_void_p),
        ("t_sab", ctypes.c_void_p), ("t_max", ctypes.c_void_p),
        ("assign", ctypes.c_void_p), ("fixed", ctypes.c_void_p),
        ("counts", ctypes.c_int64 * 4),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_LIB_PATH))
        L.orc_quantize.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]
        L.orc_quantize.restype = None
        L.orc_tile_stats.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint32, ctypes.c_void_p]
        L.orc_tile_stats.restype = None
        L.orc_greedy_init.argtypes = [ctypes.POINTER(_GreedyState), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.orc_greedy_init.restype = None
        L.orc_greedy_pass.argtypes = [ctypes.POINTER(_GreedyState), ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
        L.orc_greedy_pass.restype = None
        L.orc_greedy_value.argtypes = [ctypes.POINTER(_GreedyState)]
        L.orc_greedy_value.restype = ctypes.c_double
        L.orc_greedy_state_size.restype = ctypes.c_size_t
        assert L.orc_greedy_state_size() == ctypes.sizeof(_GreedyState)
        _lib = L
    return _lib


# ----------------------------------------------------------------------------- formats (L0)

def _as_matrix(x: np.ndarray) -> tuple[np.ndarray, tuple]:
    """quantization_formats.py:89-99: groups run along the last axis; rank only batches rows."""
    x = np.asarray(x, dtype=np.float32)
    shape = x.shape
    if x.ndim == 0:
        return x.reshape(1, 1).copy(), shape
    x = np.ascontiguousarray(x)
    if x.ndim == 1:
        return x.reshape(1, -1), shape
    return x.reshape(-1, shape[-1]), shape


def quantize_weight_values(x: np.ndarray, fmt: str) -> np.ndarray:
    """C restatement of quantization_formats.py:171-194 for bf16/bfp8/bfp4/bfp2/fp0."""
    fmt = fmt.lower()
    if fmt not in FMT_CODE:
        raise ValueError(f"Unsupported weight format: {fmt}")  # :194
    m, shape = _as_matrix(x)
    if m.size == 0:
        return np.asarray(x, dtype=np.float32).copy()  # :86-87
    y = np.empty_like(m)
    lib().orc_quantize(m.ctypes.data, m.shape[0], m.shape[1], FMT_CODE[fmt], y.ctypes.data)
    return y.reshape(shape)


def quantize_np(x: np.ndarray, fmt: str) -> np.ndarray:
    """Independent vectorised numpy restatement of the same functions (cross-checks the C code)."""
    fmt = fmt.lower()
    m, shape = _as_matrix(x)
    if fmt == "fp0":
        return np.zeros(shape, dtype=np.float32)
    u = m.view(np.uint32)
    if fmt == "bf16":  # :29-45
        r = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)
        return (r << np.uint32(16)).view(np.float32).reshape(shape)
    mb = {"bfp8": 7, "bfp4": 3, "bfp2": 1}[fmt]
    rows, cols = m.shape
    cpad = -(-cols // 16) * 16
    up = np.zeros((rows, cpad), dtype=np.uint32)
    up[:, :cols] = u
    g = up.reshape(rows, cpad // 16, 16)
    e = (g >> np.uint32(23)) & np.uint32(0xFF)
    shared = e.max(axis=-1, keepdims=True)
    man = ((g & np.uint32(0x7FFFFF)) | np.uint32(1 << 23)).astype(np.uint64)
    d = (shared - e).astype(np.uint64)
    man = np.where(d > 31, 0, man >> np.minimum(d, 31)).astype(np.uint32)
    sh = np.uint32(24 - mb)
    rv = man & np.uint32((1 << int(sh)) - 1)
    tie = np.uint32(1 << (int(sh) - 1))
    man = man >> sh
    upb = (rv > tie) | ((rv == tie) & ((man & np.uint32(1)) == 1))
    man = np.minimum(man + upb.astype(np.uint32), np.uint32((1 << mb) - 1))
    sign = np.where(man == 0, np.uint32(0), g >> np.uint32(31))
    zero = e == 0
    man = np.where(zero, np.uint32(0), man)
    sign = np.where(zero, np.uint32(0), sign)
    msb = np.zeros_like(man)
    for b in range(mb):
        msb = np.where((man >> np.uint32(b)) & np.uint32(1) == 1, np.uint32(b), msb)
    sc = np.uint32(mb - 1) - msb
    ms = (man << (sc + np.uint32(1))) & np.uint32((1 << mb) - 1)
    exp_out = np.where(man == 0, np.uint32(0), shared - sc)
    bits = (sign << np.uint32(31)) | (exp_out << np.uint32(23)) | (ms << np.uint32(23 - mb))
    bits = np.where(man == 0, sign << np.uint32(31), bits).astype(np.uint32)
    return bits.reshape(rows, cpad)[:, :cols].copy().view(np.float32).reshape(shape)


# ----------------------------------------------------------------------------- tiling (L2)

def flatten_2d(xf: np.ndarray) -> tuple[np.ndarray, tuple]:
    """tile_utils.py:91-107 without the padding copy: (h, w) row-major view + shape_info.
    A 1-D vector of n elements becomes ceil(n/32) rows of 32 with a zero-filled last row (:96-102)."""
    xf = np.asarray(xf, dtype=np.float32)
    if xf.ndim == 0:
        return xf.reshape(1, 1), ("scalar", xf.shape)
    if xf.ndim == 1:
        n = xf.shape[0]
        h = int(np.ceil(n / 32.0))
        d = np.zeros((h, 32), dtype=np.float32)
        d.reshape(-1)[:n] = xf
        return d, ("vector", n)
    return np.ascontiguousarray(xf.reshape(-1, xf.shape[-1])), ("nd", xf.shape)


def unflatten(y2d: np.ndarray, shape_info: tuple) -> np.ndarray:
    """tile_utils.py:123-131."""
    if shape_info[0] == "scalar":
        return np.array(y2d[0, 0], dtype=np.float32)
    if shape_info[0] == "vector":
        return y2d.reshape(-1)[: shape_info[1]].astype(np.float32)
    return y2d.reshape(shape_info[1]).astype(np.float32)


def tiles_hw(h: int, w: int) -> tuple[int, int]:
    return -(-h // TILE), -(-w // TILE)


def fmt_mask(formats) -> int:
    m = 0
    for f in formats:
        m |= 1 << MIXED_TILE_FORMATS.index(f)
    return m


def mask_slots(mask: int) -> dict[str, int]:
    slots, k = {}, 0
    for i, f in enumerate(MIXED_TILE_FORMATS):
        if mask & (1 << i):
            slots[f] = k
            k += 1
    return slots


def tile_stats(x2d: np.ndarray, formats) -> np.ndarray:
    """[T, 2+5F] float64 raw sums per 32×32 tile (see orc_tile_stats for layout and order)."""
    x2d = np.ascontiguousarray(x2d, dtype=np.float32)
    h, w = x2d.shape
    th, tw = tiles_hw(h, w)
    mask = fmt_mask(formats)
    rec = 2 + 5 * bin(mask).count("1")
    out = np.empty((th * tw, rec), dtype=np.float64)
    lib().orc_tile_stats(x2d.ctypes.data, h, w, mask, out.ctypes.data)
    return out


def mixed_tile_total_bytes(counts: dict) -> float:
    """tile_utils.py:32-37 (iteration order of the dict, Python float arithmetic)."""
    total = 0.0
    for fmt, count in counts.items():
        total += float(count) * 1024.0 * MIXED_TILE_BYTES_PER_ELEM.get(fmt, 0.0)
    return total


# ----------------------------------------------------------------------------- metrics (L2)

def pearson_corr(a: np.ndarray, b: np.ndarray) -> float:
    """metrics.py:6-16, literal float32 expression (numpy mean / BLAS norm / dot)."""
    a = np.asarray(a, dtype=np.float32).reshape(-1)
    b = np.asarray(b, dtype=np.float32).reshape(-1)
    if a.size == 0:
        return 1.0
    am = a - np.mean(a)
    bm = b - np.mean(b)
    denom = float(np.linalg.norm(am) * np.linalg.norm(bm))
    if denom == 0.0:
        return 1.0 if np.max(np.abs(a - b)) == 0.0 else 0.0
    return float(np.dot(am, bm) / denom)


def pearson_corr_f64(a: np.ndarray, b: np.ndarray) -> float:
    """Two-pass float64 Pearson of the same data: the size-independent reference value."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    if a.size == 0:
        return 1.0
    am = a - a.mean()
    bm = b - b.mean()
    denom = float(np.sqrt(np.dot(am, am) * np.dot(bm, bm)))
    if denom == 0.0:
        return 1.0 if np.max(np.abs(a - b)) == 0.0 else 0.0
    return float(np.dot(am, bm) / denom)


def metric_value(a, b, metric: str) -> float:
    """metrics.py:19-27."""
    if metric == "pcc":
        return pearson_corr(a, b)
    diff = np.abs(np.asarray(a, dtype=np.float32) - np.asarray(b, dtype=np.float32))
    if metric == "mae":
        return float(np.mean(diff))
    if metric == "atol":
        return float(np.max(diff))
    raise ValueError(f"Unsupported metric: {metric}")


def metric_is_good(value, metric: str, threshold) -> bool:
    """metrics.py:30-33. NOTE: with a np.float32 `value` and a Python-float threshold NumPy ≥ 2
    compares in float32 (NEP 50) — callers pass the types the reference passes."""
    return value >= threshold if metric == "pcc" else value <= threshold


def tile_metrics(ref_tiles: np.ndarray, q_tiles: np.ndarray, metric: str) -> np.ndarray:
    """tile_utils.py:46-57 (float32 scores over the zero-padded 1024 elements)."""
    if metric == "pcc":
        return np.asarray([pearson_corr(ref_tiles[i], q_tiles[i]) for i in range(ref_tiles.shape[0])], dtype=np.float32)
    diff = np.abs(ref_tiles - q_tiles)
    if metric == "mae":
        return diff.reshape(diff.shape[0], -1).mean(axis=1)
    if metric == "atol":
        return diff.reshape(diff.shape[0], -1).max(axis=1)
    raise ValueError(f"Unsupported metric: {metric}")


def to_tiles(x2d: np.ndarray) -> np.ndarray:
    """Pad to 32-multiples with zeros and gather (T,32,32), tile id = tr*tiles_w+tc
    (tile_utils.py:109-113, mixed_tile_greedy.py:89-93)."""
    h, w = x2d.shape
    th, tw = tiles_hw(h, w)
    p = np.zeros((th * TILE, tw * TILE), dtype=np.float32)
    p[:h, :w] = x2d
    return p.reshape(th, TILE, tw, TILE).transpose(0, 2, 1, 3).reshape(-1, TILE, TILE)


# ----------------------------------------------------------------------------- algorithms (L3)

def apply_assignment(xf: np.ndarray, assignment: np.ndarray) -> np.ndarray:
    """Reconstruct y from an int8 (tiles_h, tiles_w) map: each tile takes its format's y
    (mixed_tile_threshold.py:125-132, mixed_tile_greedy.py:273,348-352,
    scripts/reconstruct_mixed_tile_assignment.py:82-94)."""
    x2d, info = flatten_2d(xf)
    h, w = x2d.shape
    th, tw = tiles_hw(h, w)
    a = np.asarray(assignment).reshape(th, tw)
    y2d = np.empty_like(x2d)
    sel = np.repeat(np.repeat(a, TILE, axis=0), TILE, axis=1)[:h, :w]
    for idx, fmt in enumerate(MIXED_TILE_FORMATS):
        if np.any(a == idx):
            yq = quantize_weight_values(x2d, fmt)
            y2d = np.where(sel == idx, yq, y2d)
    return unflatten(y2d, info)


def greedy(xf: np.ndarray, tile_formats, metric: str = "pcc", threshold: float = 0.999, seed: int = 123,
           return_orders: bool = False):
    """mixed_tile_greedy.py:72-352 restated as: per-tile raw sums (orc_tile_stats) + the literal
    sequential scan (orc_greedy_*) + numpy's own Generator for the visiting order (:222-231).
    Returns (assignment int8 (tiles_h,tiles_w), counts dict, state dict[, orders])."""
    if seed == 0:
        raise ValueError("seed 0 means 'random' in the reference (mixed_tile_greedy.py:223-224); use a non-zero seed")
    xf = np.asarray(xf, dtype=np.float32)
    if xf.size == 0:  # :78-83
        return np.zeros((1, 1), dtype=np.int8), {f: 0 for f in MIXED_TILE_FORMATS}, {}
    x2d, _info = flatten_2d(xf)
    th, tw = tiles_hw(*x2d.shape)
    T = th * tw
    tile_formats = list(tile_formats)
    mask = fmt_mask(tile_formats)
    slots = mask_slots(mask)
    stats = tile_stats(x2d, [f for f in MIXED_TILE_FORMATS if f in slots])
    rec = stats.shape[1]
    L = lib()
    bufs = {k: np.zeros(T, dtype=np.float64) for k in ("t_sy", "t_sy2", "t_sxy", "t_sab", "t_max")}
    assign = np.zeros(T, dtype=np.int8)
    fixed = np.zeros(T, dtype=np.uint8)
    st = _GreedyState()
    st.T, st.metric, st.threshold, st.n = T, METRIC_CODE[metric], float(threshold), float(xf.size)  # :134
    for k, v in bufs.items():
        setattr(st, k, v.ctypes.data)
    st.assign, st.fixed = assign.ctypes.data, fixed.ctypes.data
    base = tile_formats[0]  # :96
    L.orc_greedy_init(ctypes.byref(st), stats.ctypes.data, rec, slots[base], MIXED_TILE_FORMATS.index(base))
    rng = np.random.default_rng(seed)  # :225
    orders = []
    for fmt in tile_formats:  # :227
        candidates = np.where(fixed == 0)[0]  # :228
        if candidates.size == 0:
            break
        order = np.ascontiguousarray(rng.permutation(candidates), dtype=np.int64)  # :231
        orders.append(order)
        L.orc_greedy_pass(ctypes.byref(st), stats.ctypes.data, rec, slots[fmt], MIXED_TILE_FORMATS.index(fmt),
                          order.ctypes.data, order.size)
    counts = {f: int(st.counts[i]) for i, f in enumerate(MIXED_TILE_FORMATS)}
    state = {
        "value": L.orc_greedy_value(ctypes.byref(st)),
        "sum_x": st.sum_x, "sum_x2": st.sum_x2, "sum_y": st.sum_y, "sum_y2": st.sum_y2,
        "sum_xy": st.sum_xy, "sum_abs": st.sum_abs, "max_abs": st.max_abs,
        "stats": stats,
    }
    out = (assign.reshape(th, tw).copy(), counts, state)
    return out + (orders,) if return_orders else out


def threshold_scores(xf: np.ndarray, tile_formats, metric: str) -> dict[str, np.ndarray]:
    """mixed_tile_threshold.py:97-109: literal float32 per-tile scores (whole-tensor quantize,
    pad, tile, tile_metrics)."""
    x2d, _ = flatten_2d(xf)
    tiles_ref = to_tiles(x2d)
    scores = {}
    for fmt in tile_formats:
        y2d = quantize_weight_values(x2d, fmt)
        scores[fmt] = tile_metrics(tiles_ref, to_tiles(y2d), metric)
    return scores


def threshold_assign(scores: dict[str, np.ndarray], tile_formats, metric: str, threshold: float) -> np.ndarray:
    """mixed_tile_threshold.py:111-123 (flat int8[T])."""
    tile_formats = list(tile_formats)
    by_prec = sorted(tile_formats, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))  # :112-114
    best = max(by_prec, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))  # :115
    T = len(next(iter(scores.values())))
    out = np.full((T,), MIXED_TILE_FORMATS.index(best), dtype=np.int8)
    for t in range(T):
        for fmt in by_prec:
            if metric_is_good(scores[fmt][t], metric, threshold):  # np.float32 vs Python float (NEP 50)
                out[t] = MIXED_TILE_FORMATS.index(fmt)
                break
    return out


def threshold(xf: np.ndarray, tile_formats, metric: str = "pcc", threshold: float = 0.999):
    """mixed_tile_threshold.py:70-137 → (assignment int8 (tiles_h,tiles_w), counts, scores)."""
    xf = np.asarray(xf, dtype=np.float32)
    if xf.size == 0:
        return np.zeros((1, 1), dtype=np.int8), {f: 0 for f in MIXED_TILE_FORMATS}, {}
    x2d, _ = flatten_2d(xf)
    th, tw = tiles_hw(*x2d.shape)
    scores = threshold_scores(xf, tile_formats, metric)
    a = threshold_assign(scores, tile_formats, metric, threshold)
    counts = {f: 0 for f in MIXED_TILE_FORMATS}
    for f in tile_formats:
        counts[f] = int(np.sum(a == MIXED_TILE_FORMATS.index(f)))  # :133-135
    return a.reshape(th, tw), counts, scores


def metric_better(value: float, best: float, metric: str) -> bool:
    """metrics.py:36-39."""
    return value > best if metric == "pcc" else value < best


def random_search(xf: np.ndarray, tile_formats, metric: str = "pcc", threshold: float = 0.999, iters: int = 50, seed: int = 0):
    """mixed_tile_random.py:88-186 → (assignment int8 (tiles_h,tiles_w), counts, samples).  `iters` random maps
    drawn with default_rng(seed).integers (:135); every map is scored on the whole reconstructed tensor in
    float32 (:139-143); the smallest map that meets the threshold wins, else the best-scoring one (:160-173)."""
    xf = np.asarray(xf, dtype=np.float32)
    if xf.size == 0:  # :94-100
        return np.zeros((1, 1), dtype=np.int8), {f: 0 for f in MIXED_TILE_FORMATS}, []
    x2d, _ = flatten_2d(xf)
    th, tw = tiles_hw(*x2d.shape)
    T = th * tw
    fmt_indices = np.asarray([MIXED_TILE_FORMATS.index(f) for f in tile_formats] or list(range(4)), dtype=np.int8)  # :113-116
    rng = np.random.default_rng(seed)  # :117 (seed 0 is a real seed here, unlike the greedy search)
    bpe = np.asarray([MIXED_TILE_BYTES_PER_ELEM[f] for f in MIXED_TILE_FORMATS], dtype=np.float32)  # :118-126
    best_metric = best_assign = best_bytes = None
    samples = []
    for sample_id in range(max(1, iters)):
        choice = rng.integers(0, len(fmt_indices), size=T, dtype=np.int64)  # :135
        a = fmt_indices[choice].astype(np.int8)
        y = apply_assignment(xf, a.reshape(th, tw))  # :137-138 (tile-wise quantisation == whole-tensor quantisation, F2)
        score = metric_value(xf, y, metric)
        diff = np.abs(xf - y)
        counts_arr = np.bincount(a.astype(np.int64), minlength=4)
        counts = {f: int(counts_arr[i]) for i, f in enumerate(MIXED_TILE_FORMATS)}
        samples.append({"id": sample_id, "counts": counts, "total_bytes": mixed_tile_total_bytes(counts),
                        "pcc": pearson_corr(xf, y), "mae": float(np.mean(diff)), "atol": float(np.max(diff))})
        if metric_is_good(score, metric, threshold):  # :160-167
            total = float(np.sum(counts_arr * bpe) * (TILE * TILE))  # float32 product, as the reference
            if best_bytes is None or total < best_bytes:
                best_bytes, best_metric, best_assign = total, score, a.copy()
        elif best_bytes is None:  # :168-172
            if best_metric is None or metric_better(score, best_metric, metric):
                best_metric, best_assign = score, a.copy()
    counts = {f: int(np.sum(best_assign == i)) for i, f in enumerate(MIXED_TILE_FORMATS)}
    return best_assign.reshape(th, tw), counts, samples


def columns_from_stats(stats: np.ndarray, slots: dict, assignment: np.ndarray, n: int) -> tuple[float, float, float]:
    """Tensor-level (pcc, mae, atol) of the reconstruction implied by `assignment`, from the per-tile
    raw sums in float64 (what the hip backend reports).  Sums run in tile order."""
    a = np.asarray(assignment).reshape(-1)
    sx = sx2 = sy = sy2 = sxy = sab = 0.0
    mx = 0.0
    for t in range(stats.shape[0]):
        r = stats[t]
        b = r[2 + 5 * slots[MIXED_TILE_FORMATS[int(a[t])]]:]
        sx += r[0]; sx2 += r[1]; sy += b[0]; sy2 += b[1]; sxy += b[2]; sab += b[3]
        mx = b[4] if (b[4] > mx or b[4] != b[4]) else mx
    nn = float(n)
    mean_x, mean_y = sx / nn, sy / nn
    am2 = max(sx2 - nn * mean_x * mean_x, 0.0)
    bm2 = max(sy2 - nn * mean_y * mean_y, 0.0)
    denom = float(np.sqrt(am2 * bm2))
    pcc = (1.0 if sab == 0.0 else 0.0) if denom == 0.0 else (sxy - nn * mean_x * mean_y) / denom
    return pcc, sab / nn, mx


# ----------------------------------------------------------------------------- loader (K5)

def e4m3fn_table() -> np.ndarray:
    """float32 value of every float8-e4m3fn code (OCP: bias 7, subnormals m*2^-9, S.1111.111 = NaN, no infinities)."""
    t = np.empty(256, dtype=np.float32)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 0xF, b & 7
        if e == 0:
            v = m * 2.0 ** -9
        elif e == 15 and m == 7:
            v = float("nan")
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        t[b] = -v if s else v
    return t


def dequant_fp8_block(w_bytes: np.ndarray, scale_inv: np.ndarray) -> np.ndarray:
    """hf_model_utils.py:199-215: w.float() * scale_inv.repeat_interleave(ceil(dim/scale_dim)) in float32."""
    w = e4m3fn_table()[np.asarray(w_bytes, dtype=np.uint8)]
    sc = np.asarray(scale_inv, dtype=np.float32)
    for ax in range(w.ndim):
        block = max(1, -(-w.shape[ax] // sc.shape[ax]))
        sc = np.repeat(sc, block, axis=ax)
    sc = sc[tuple(slice(0, n) for n in w.shape)]
    with np.errstate(invalid="ignore"):
        return (w * sc).astype(np.float32)
