"""mixed-tile-random (reference compression_algorithms/mixed_tile_random.py:18-209).

`iters` random per-tile format maps drawn from default_rng(seed).integers; every map is scored on the whole
tensor; the smallest map that meets the threshold wins, otherwise the best-scoring one.

The reference re-quantises the gathered tiles and reconstructs the tensor for every sample (:137-143).  Here one
stats pass (K1 on the GPU for backend "hip") yields per-tile raw sums for every candidate format, and a sample's
pcc / mae / atol are those sums added up under its map — O(tiles) per sample, no re-quantisation.  The draws come
from mtq_rng_integers, a bit-compatible restatement of Generator.integers (pinned against NumPy in
tests/test_capi_host.py), so the maps are the reference's maps.  The reference scores in float32; a sample whose
float64 score is within `SCORE_BAND` of the threshold, or of the incumbent it is compared with, is re-scored with
the literal float32 expression on its reconstruction (K3) so that the selection is the reference's selection.
"""
from __future__ import annotations

import numpy as np

from .base import CompressionAlgorithm, CompressionResult
from .metrics import metric_better, metric_is_good, metric_value
from .mixed_tile_greedy import parse_tile_formats
from .tile_search import TileStats, columns_from_stats, compute_tile_stats, reconstruct
from .tile_utils import MIXED_TILE_BYTES_PER_ELEM, MIXED_TILE_FORMATS, mixed_tile_total_bytes

SCORE_BAND = 1e-5  # |float64-moment score − float32 whole-tensor score| stays below this (relative to max(1, |thr|))


def _host_f32(x) -> np.ndarray:
    return np.asarray(x.float().cpu().numpy() if hasattr(x, "cpu") else x, dtype=np.float32)


def random_search(ts: TileStats, xf, tile_formats: list[str], metric: str, threshold: float, iters: int, seed: int,
                  quantizer, band: float = SCORE_BAND):
    """reference :112-173 on a TileStats → (int8 (tiles_h, tiles_w) map, samples, number of literal re-scores)."""
    from .. import hip_backend as hb

    fmt_indices = np.asarray([MIXED_TILE_FORMATS.index(f) for f in tile_formats] or list(range(len(MIXED_TILE_FORMATS))),
                             dtype=np.int8)  # :113-116
    rng = hb.NumpyCompatRng(seed)  # :117 — seed 0 is an ordinary seed for this algorithm
    bytes_per_elem = np.asarray([MIXED_TILE_BYTES_PER_ELEM[f] for f in MIXED_TILE_FORMATS], dtype=np.float32)  # :118-126
    width = band * max(1.0, abs(threshold))
    maps: list[np.ndarray] = []
    literal: dict[int, float] = {}
    x_host = None

    def literal_score(i: int) -> float:
        nonlocal x_host
        if i not in literal:
            if x_host is None:
                x_host = _host_f32(xf)
            literal[i] = metric_value(x_host, _host_f32(reconstruct(ts, maps[i], quantizer)), metric)  # :138-139
        return literal[i]

    def near(a: float, b: float) -> bool:
        return not (abs(a - b) > width)  # NaN counts as near: let the literal expression decide

    best_id = best_score = best_bytes = None
    samples: list[dict] = []
    for sample_id in range(max(1, iters)):
        choice = rng.integers(len(fmt_indices), ts.tiles)  # :135
        amap = fmt_indices[choice].astype(np.int8)
        maps.append(amap)
        cols = columns_from_stats(ts, amap)
        score = cols[metric]
        counts_arr = np.bincount(amap.astype(np.int64), minlength=len(MIXED_TILE_FORMATS))
        counts = {fmt: int(counts_arr[i]) for i, fmt in enumerate(MIXED_TILE_FORMATS)}
        samples.append({"id": sample_id, "counts": counts, "total_bytes": mixed_tile_total_bytes(counts),
                        "pcc": cols["pcc"], "mae": cols["mae"], "atol": cols["atol"]})  # :147-157
        if near(score, threshold):
            score = literal_score(sample_id)
        if metric_is_good(score, metric, threshold):  # :158-167
            total = float(np.sum(counts_arr * bytes_per_elem) * (32 * 32))
            if best_bytes is None or total < best_bytes:
                best_bytes, best_score, best_id = total, score, sample_id
        elif best_bytes is None:  # :168-172
            if best_score is not None and near(score, best_score):
                score, best_score = literal_score(sample_id), literal_score(best_id)
            if best_score is None or metric_better(score, best_score, metric):
                best_score, best_id = score, sample_id
    return maps[best_id].reshape(ts.tiles_h, ts.tiles_w), samples, len(literal)


class MixedTileRandomCompression(CompressionAlgorithm):
    name = "mixed-tile-random"

    def __init__(self, params: dict | None = None) -> None:
        super().__init__(params=params)
        self.metric = self.params.get("metric", "pcc")
        self.threshold = float(self.params.get("threshold", 0.999))
        self.iters = int(self.params.get("iters", 50))
        self.seed = int(self.params.get("seed", 0))
        self.formats = parse_tile_formats(self.params.get("formats"))
        self.materialize_y = bool(self.params.get("materialize_y", True))
        if self.metric not in {"pcc", "mae", "atol"}:
            raise ValueError(f"Unsupported metric: {self.metric}")
        if self.iters < 1:
            raise ValueError("iters must be >= 1")

    def expected_evals(self, formats) -> int:
        return 1

    _parse_formats = staticmethod(parse_tile_formats)

    @staticmethod
    def _filter_from_formats(formats: list[str]) -> list[str]:
        allowed = [fmt for fmt in formats if fmt in MIXED_TILE_FORMATS]
        if not allowed:
            raise ValueError(
                "mixed-tile-random requires at least one of "
                f"{', '.join(MIXED_TILE_FORMATS)} in quantization_formats"
            )
        return allowed

    def run(self, xf, formats: list[str], quantizer, cache) -> list[CompressionResult]:
        tile_formats = self.formats or self._filter_from_formats(formats)
        size = int(np.asarray(xf).size) if isinstance(xf, np.ndarray) or np.isscalar(xf) else int(xf.numel())
        counts = {fmt: 0 for fmt in MIXED_TILE_FORMATS}
        if size == 0:  # :94-100
            y = np.asarray(xf, dtype=np.float32)
            meta = {"samples": [], "tile_formats": tile_formats, "assignment": np.zeros((1, 1), dtype=np.int8)}
        else:
            ts = compute_tile_stats(xf, tile_formats, quantizer)
            assignment, samples, n_literal = random_search(ts, xf, tile_formats, self.metric, self.threshold, self.iters,
                                                           self.seed, quantizer)
            for idx, fmt in enumerate(MIXED_TILE_FORMATS):  # :179-181
                counts[fmt] = int(np.sum(assignment == idx))
            y = reconstruct(ts, assignment, quantizer) if self.materialize_y else None
            meta = {"samples": samples, "tile_formats": tile_formats, "assignment": assignment,
                    "columns": columns_from_stats(ts, assignment), "literal_rescored_samples": n_literal}
        return [CompressionResult(fmt="MIXED", compression=self.name, y=y, tile_counts=counts,
                                  tile_bytes=mixed_tile_total_bytes(counts), meta=meta)]
