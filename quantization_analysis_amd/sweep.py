"""Threshold sweep of mixed-tile-threshold (reference scripts/sweep_mixed_tile_threshold.py:145-180,626-790).

The reference quantizes the tensor once per format, scores every tile (float32), then for each of `steps` thresholds
re-gathers 64 MiB of tiles and recomputes float32 tensor metrics.  Here ONE K1 pass gives the per-tile records; every
threshold step is then an O(tiles) selection plus a sum of already-computed records.  Tiles whose
float64-moment score comes within KNIFE_BAND of any threshold get the literal float32 score, so every step's
assignment is the reference's assignment.
"""
from __future__ import annotations

import numpy as np

from .compression_algorithms.mixed_tile_threshold import KNIFE_BAND, knife_width
from .compression_algorithms.quantizer import Quantizer
from .compression_algorithms.tile_search import columns_from_stats, compute_tile_stats, literal_inputs, tile_scores
from .compression_algorithms.tile_utils import MIXED_TILE_BYTES_PER_ELEM, MIXED_TILE_FORMATS, mixed_tile_total_bytes, tile_metrics


LITERAL_CHUNK_TILES = 4096    # tiles per literal re-scoring chunk (16 MiB of x and 16 MiB of y tiles on the host per chunk in flight)


def _literal_threads() -> int:
    """Host threads for the literal re-scoring of one rank: the rank's share of the CPU budget (cgroup quota), MTQ_SWEEP_THREADS overrides."""
    import os

    from .pipeline import cpu_budget
    from .settings import settings

    if settings().sweep_threads is not None:
        return max(1, settings().sweep_threads)
    local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return max(1, cpu_budget() // max(local, 1))


_POOL = None


def _literal_pool():
    """The process's literal-scoring threads, created once: their pinned staging buffers (tile_search.literal_inputs) live as long as they do."""
    global _POOL
    n = _literal_threads()
    if _POOL is None or _POOL[0] != n:
        import concurrent.futures as cf

        if _POOL is not None:
            _POOL[1].shutdown(wait=True)
        _POOL = (n, cf.ThreadPoolExecutor(max_workers=n, thread_name_prefix="mtq-sweep"))
    return _POOL[1]


def compute_assignment(scores_stack: np.ndarray, metric: str, threshold: float) -> np.ndarray:
    """reference :145-155 — scores_stack float32 [F asc-bytes, T]; NumPy compares in float32 (NEP 50)."""
    good = scores_stack >= threshold if metric == "pcc" else scores_stack <= threshold
    # index of the first True per column with the last row forced True (np.argmax(good, axis=0) of :152-153) = the number of leading False
    # among the rows before the last: three vector passes instead of a strided argmax over a [F, T] array (47 of the sweep leg's 280 ms)
    idx = np.zeros(good.shape[1], dtype=np.int32)
    alive = np.ones(good.shape[1], dtype=bool)
    for f in range(good.shape[0] - 1):
        alive &= ~good[f]
        idx += alive
    return idx


def pareto_mask(points: list[dict], metric: str) -> list[bool]:
    """reference :158-174."""
    is_pcc = metric == "pcc"
    mask = [True for _ in points]
    for i, a in enumerate(points):
        for j, b in enumerate(points):
            if i == j:
                continue
            if is_pcc:
                dominates = b["size"] <= a["size"] and b["metric"] >= a["metric"]
                strictly = b["size"] < a["size"] or b["metric"] > a["metric"]
            else:
                dominates = b["size"] <= a["size"] and b["metric"] <= a["metric"]
                strictly = b["size"] < a["size"] or b["metric"] < a["metric"]
            if dominates and strictly:
                mask[i] = False
                break
    return mask


def pareto_frontier(points: list[dict], metric: str) -> list[dict]:
    """reference :177-180."""
    return sorted([p for p, keep in zip(points, pareto_mask(points, metric)) if keep], key=lambda p: p["size"])


def _literal_scores(ts, ids: np.ndarray, fmt: str, quantizer: Quantizer, metric: str):
    """(tile ids, their literal float32 scores under fmt) chunk by chunk (LITERAL_CHUNK_TILES tiles each).  With several chunks —
    identity-like formats put all 10^5 tiles of a large tensor inside the band — the rank's CPU share of threads (_literal_threads) works
    on different chunks: the score is NumPy reductions and BLAS dot products over a chunk's tiles, which run without the interpreter
    lock, and a chunk's tiles come home (the tensor's own device made current in the worker: tile_search.literal_inputs) while other
    chunks are scored.  Same calls per tile, same bits, whatever the chunking."""
    chunks = [ids[c0:c0 + LITERAL_CHUNK_TILES] for c0 in range(0, ids.size, LITERAL_CHUNK_TILES)]

    def one(part):
        xt, yt, got = literal_inputs(ts, part, fmt, quantizer, pinned=True)   # consumed before this thread fetches again
        return got, tile_metrics(xt, yt, metric)

    if len(chunks) <= 1:
        return [one(c) for c in chunks]
    return list(_literal_pool().map(one, chunks))


def sweep_tensor(xf, formats: list[str], metric: str, lowest_metric_val: float, steps: int, quantizer: Quantizer):
    """→ (rows, baseline_points, thresholds).  rows: dicts with the CSV columns of reference :793
    (step, threshold, size_bytes, pcc, mae, atol, <fmt>_tiles)."""
    ts = compute_tile_stats(xf, formats, quantizer)
    by_prec = sorted(formats, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))   # :652
    highest = max(by_prec, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))      # :653
    mask_order = [f for f in MIXED_TILE_FORMATS if f in formats]
    s64_all = tile_scores(ts, metric)                                                 # [F mask order, T]
    s64 = np.stack([s64_all[mask_order.index(f)] for f in by_prec])                   # asc bytes
    s32 = s64.astype(np.float32)

    hi = by_prec.index(highest)
    if metric == "pcc":                                                               # :659-670
        start = float(np.max(s32[hi]))
        if lowest_metric_val > start:
            raise ValueError("lowest-metric-val must be <= start metric for pcc")
    else:
        start = float(np.min(s32[hi]))
        if lowest_metric_val < start:
            raise ValueError("lowest-metric-val must be >= start metric for mae/atol")
    # the start of the sweep is itself a float32 tile score of the reference: take the literal one for the extreme tile
    literal_hi = np.zeros(s64.shape[1], dtype=bool)   # tiles of the highest-precision format that already carry their literal score
    if metric != "atol":
        t_ext = int(np.argmax(s32[hi]) if metric == "pcc" else np.argmin(s32[hi]))
        cand = np.unique(np.concatenate([[t_ext], np.where(np.abs(s64[hi] - s64[hi][t_ext]) <= knife_width(s64[hi][t_ext]))[0]]))
        # identity-like formats (bf16 of bf16 data, bf16 of fp8-block data) put EVERY tile inside the band: the literal float32 scores
        # are 1 ± a few ulp and the reference's start value is their exact maximum (a 2-ulp outlier among 10^5 tiles decides it, so a
        # sample is not enough: golden-size check tests/test_configs_gpu.py::test_config4_sweep_deepseek_layer0) — all of them are scored,
        # in bounded chunks (4 KB per tile through the host)
        for got, scores in _literal_scores(ts, cand, highest, quantizer, metric):
            s32[hi, got] = scores
        literal_hi[cand] = True
        start = float(np.max(s32[hi])) if metric == "pcc" else float(np.min(s32[hi]))
    thresholds = np.linspace(start, lowest_metric_val, max(1, steps))

    # knife edge: (format, tile) pairs whose moment score is within the band of ANY threshold get the literal float32 score
    if metric != "atol":
        thr32 = thresholds.astype(np.float32).astype(np.float64)
        for fi, f in enumerate(by_prec):
            near = np.zeros(s64.shape[1], dtype=bool)
            srt = np.sort(thr32)
            pos = np.searchsorted(srt, s64[fi])
            for off in (-1, 0):
                idx = np.clip(pos + off, 0, srt.size - 1)
                near |= np.abs(s64[fi] - srt[idx]) <= knife_width(srt[idx])
            if fi == hi:
                near &= ~literal_hi
            ids = np.where(near)[0]
            for got, scores in _literal_scores(ts, ids, f, quantizer, metric):
                s32[fi, got] = scores

    baselines = []
    for f in formats:                                                                 # :688-715
        amap = np.full(ts.tiles, MIXED_TILE_FORMATS.index(f), dtype=np.int8)
        c = columns_from_stats(ts, amap)
        mv = c[metric]
        if (metric == "pcc" and mv < lowest_metric_val) or (metric != "pcc" and mv > lowest_metric_val):
            continue
        baselines.append({"label": f.upper(), "size": float(ts.numel) * MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0), "metric": mv,
                          "kind": "baseline", "pcc": c["pcc"], "mae": c["mae"], "atol": c["atol"], f"{f}_tiles": ts.tiles})

    rows, last, last_metrics = [], None, None
    fmt_code = np.asarray([MIXED_TILE_FORMATS.index(f) for f in by_prec], dtype=np.int8)
    for step_idx, thr in enumerate(thresholds):                                       # :729-790
        a_idx = compute_assignment(s32.copy(), metric, float(thr))
        if last is not None and np.array_equal(a_idx, last):
            m = last_metrics
        else:
            c = columns_from_stats(ts, fmt_code[a_idx])
            raw = np.bincount(a_idx, minlength=len(by_prec))
            counts = {f: 0 for f in MIXED_TILE_FORMATS}
            for i, f in enumerate(by_prec):
                counts[f] = int(raw[i])
            m = {"pcc": c["pcc"], "mae": c["mae"], "atol": c["atol"], "size_bytes": mixed_tile_total_bytes(counts), "counts": counts}
            last, last_metrics = a_idx, m
        rows.append({"step": step_idx, "threshold": float(thr), "size_bytes": m["size_bytes"], "pcc": m["pcc"], "mae": m["mae"],
                     "atol": m["atol"], **{f"{f}_tiles": m["counts"].get(f, 0) for f in formats}})
    return rows, baselines, thresholds


def write_csv(path, rows: list[dict], formats: list[str]) -> None:
    """reference :792-798."""
    headers = ["step", "threshold", "size_bytes", "pcc", "mae", "atol", *[f"{f}_tiles" for f in formats]]
    with open(path, "w", encoding="utf-8") as f:
        f.write(",".join(headers) + "\n")
        for row in rows:
            f.write(",".join(str(row.get(h, "")) for h in headers) + "\n")
