"""mixed-tile-threshold (reference compression_algorithms/mixed_tile_threshold.py:19-162).

Per tile: the lowest-bytes format whose per-tile metric passes, else the highest-bytes one.
The reference scores tiles with a float32 two-pass Pearson / mean / max (tile_utils.py:46-57) and, under
NumPy >= 2, compares that np.float32 with the float32-rounded threshold (metrics.py:30-33, NEP 50).
Here scores come from the float64 stats records (K1 on the GPU for backend "hip", where K4 — the rule below — also
runs on the device and only the map and the knife-edge flags come back); tiles whose score
lies within `knife_band` of the threshold are re-scored with the literal float32 expression so that the
map is the reference's map.
"""
from __future__ import annotations

import numpy as np

from .base import CompressionAlgorithm, CompressionResult
from .cache import CacheContext
from .metrics import metric_is_good
from .mixed_tile_greedy import parse_tile_formats
from .quantizer import Quantizer
from .tile_search import TileStats, columns_from_stats, compute_tile_stats, gather_tiles, reconstruct
from .tile_utils import MIXED_TILE_BYTES_PER_ELEM, MIXED_TILE_FORMATS, mixed_tile_total_bytes, tile_metrics

# float64-moment score vs float32 two-pass score differ by <= 3.1e-7 for scores of order 1 (measured); the
# float32 error is relative, so the band is KNIFE_BAND * max(1, |threshold|) (csrc/mtq_decide.hpp, sweep.knife_width):
# golden F13 holds mae thresholds around 1e3.
KNIFE_BAND = 2e-6


def knife_width(threshold, band: float = KNIFE_BAND):
    """Half-width of the band around float32(threshold) inside which a float64-moment score is re-decided literally."""
    return band * np.maximum(1.0, np.abs(np.asarray(threshold, dtype=np.float64)))


def threshold_assign(ts: TileStats, tile_formats: list[str], metric: str, threshold: float, quantizer: Quantizer,
                     band: float = KNIFE_BAND) -> tuple[np.ndarray, int]:
    """reference :111-123 on a TileStats → (int8 (tiles_h, tiles_w) map, number of re-scored tiles)."""
    from .. import hip_backend as hb

    if ts.on_device and ts.host_stats is None:  # K4 on the device: the records stay where K1 wrote them
        amap, knife, near = hb.threshold_assign_device(ts.stats_dev, ts.mask, tile_formats, metric, threshold, band, with_near=True)
    else:
        amap, knife, near = hb.threshold_assign(ts.stats, ts.mask, tile_formats, metric, threshold, band, with_near=True)
    rescore_knife_tiles(ts, amap, knife, tile_formats, metric, threshold, quantizer, near)
    return amap.reshape(ts.tiles_h, ts.tiles_w), int(knife.size)


def decide_knife_tiles(chosen: np.ndarray, near: np.ndarray, tile_formats: list[str], metric: str, threshold: float, literal_scores) -> np.ndarray:
    """The reference's rule (:111-123) for the knife-edge tiles, evaluating the literal float32 score only where it can
    change the outcome.  `chosen[k]` is K4's format code for knife tile k and `near[k]` the mask of format codes whose float64
    score fell inside the noise band.  K4 walked the formats in ascending bytes up to the chosen one: a looked-at format
    outside the band keeps its float64 decision (failed before the chosen one, passed at it — the same trust every
    non-knife tile gets); a format inside the band, and any format behind the chosen one once that is overturned, is
    decided by `literal_scores(fmt, sel)` → np.float32 scores of knife tiles `sel` (tile_utils.py:46-57 on y quantized
    through the selected backend).  → int8 codes."""
    by_prec = sorted(tile_formats, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))  # :112-114
    best = max(by_prec, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))             # :115
    codes = [MIXED_TILE_FORMATS.index(f) for f in by_prec]
    rank = np.full(len(MIXED_TILE_FORMATS), len(by_prec), dtype=np.int64)
    for i, c in enumerate(codes):
        rank[c] = i
    seen_upto = rank[np.asarray(chosen, dtype=np.int64)]     # index (ascending bytes) of the last format K4 looked at
    out = np.full(len(chosen), MIXED_TILE_FORMATS.index(best), dtype=np.int8)
    open_ = np.ones(len(chosen), dtype=bool)
    for i, (f, c) in enumerate(zip(by_prec, codes)):
        if not open_.any():
            break
        literal = open_ & ((((near >> c) & 1) != 0) | (i > seen_upto))
        passed = open_ & ~literal & (i == seen_upto)
        sel = np.flatnonzero(literal)
        if sel.size:
            scores = literal_scores(f, sel)
            ok = np.fromiter((metric_is_good(scores[k], metric, threshold) for k in range(sel.size)), dtype=bool, count=sel.size)
            passed[sel[ok]] = True                                 # np.float32 vs Python float, as the reference
        out[passed] = c
        open_ &= ~passed
    return out


def rescore_knife_tiles(ts: TileStats, amap: np.ndarray, knife: np.ndarray, tile_formats: list[str], metric: str, threshold: float,
                        quantizer: Quantizer, near: np.ndarray | None = None) -> None:
    """Decide the tiles in `knife` again with the literal float32 per-tile score of the reference (tile_utils.py:46-57 on y
    tiles quantized through the selected backend) and patch the flat int8 map `amap` in place.  Without `near` every format
    of every knife tile is treated as inside the band."""
    if not knife.size:
        return
    x_tiles = gather_tiles(ts, knife)
    if near is None:
        near = np.full(knife.size, 0xF, dtype=np.uint8)

    def literal_scores(fmt: str, sel: np.ndarray) -> np.ndarray:
        xs = x_tiles[sel]
        return tile_metrics(xs, np.asarray(_quantize_tiles(xs, fmt, quantizer), dtype=np.float32), metric)

    amap[knife] = decide_knife_tiles(amap[knife], np.asarray(near, dtype=np.uint8), tile_formats, metric, threshold, literal_scores)


def _quantize_tiles(x_tiles: np.ndarray, fmt: str, quantizer: Quantizer) -> np.ndarray:
    """(k,32,32) host tiles → y tiles through the selected backend (hip: K2 on the device)."""
    k = x_tiles.shape[0]
    return np.asarray(quantizer.quantize(x_tiles.reshape(k * 32, 32), fmt)).reshape(k, 32, 32)


class MixedTileThresholdCompression(CompressionAlgorithm):
    name = "mixed-tile-threshold"

    def __init__(self, params: dict | None = None) -> None:
        super().__init__(params=params)
        self.metric = self.params.get("metric", "pcc")
        self.threshold = float(self.params.get("threshold", 0.999))
        raw_formats = self.params.get("formats", self.params.get("tile_formats"))
        self.tile_formats = parse_tile_formats(raw_formats) if raw_formats is not None else None
        self.materialize_y = bool(self.params.get("materialize_y", True))
        if self.metric not in {"pcc", "mae", "atol"}:
            raise ValueError(f"Unsupported metric: {self.metric}")

    @classmethod
    def from_params(cls, params: dict | None = None) -> "MixedTileThresholdCompression":
        return cls(params=params or {})

    def expected_evals(self, formats: list[str]) -> int:
        return 1

    _parse_formats = staticmethod(parse_tile_formats)

    @staticmethod
    def _filter_from_formats(formats: list[str]) -> list[str]:
        allowed = [fmt for fmt in formats if fmt in MIXED_TILE_FORMATS]
        if not allowed:
            raise ValueError(
                "mixed-tile-threshold requires at least one of "
                f"{', '.join(MIXED_TILE_FORMATS)} in quantization_formats"
            )
        return allowed

    def run(self, xf, formats: list[str], quantizer: Quantizer, cache: CacheContext) -> list[CompressionResult]:
        tile_formats = self.tile_formats or self._filter_from_formats(formats)
        size = int(np.asarray(xf).size) if isinstance(xf, np.ndarray) or np.isscalar(xf) else int(xf.numel())
        if size == 0:  # :76-81
            y = np.asarray(xf, dtype=np.float32)
            counts = {fmt: 0 for fmt in MIXED_TILE_FORMATS}
            meta = {"assignment": np.zeros((1, 1), dtype=np.int8), "tile_formats": tile_formats}
        else:
            ts = compute_tile_stats(xf, tile_formats, quantizer)
            assignment, n_knife = threshold_assign(ts, tile_formats, self.metric, self.threshold, quantizer)
            counts = {fmt: 0 for fmt in MIXED_TILE_FORMATS}
            for fmt in tile_formats:  # :133-135
                counts[fmt] = int(np.sum(assignment == MIXED_TILE_FORMATS.index(fmt)))
            y = reconstruct(ts, assignment, quantizer) if self.materialize_y else None
            meta = {"assignment": assignment, "tile_formats": tile_formats,
                    "columns": columns_from_stats(ts, assignment), "knife_edge_tiles": n_knife}
        return [
            CompressionResult(
                fmt="MIXED",
                compression=self.name,
                y=y,
                tile_counts=counts,
                tile_bytes=mixed_tile_total_bytes(counts),
                meta=meta,
            )
        ]
