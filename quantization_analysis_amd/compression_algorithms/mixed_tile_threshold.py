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

KNIFE_BAND = 2e-6  # float64-moment score vs float32 two-pass score differ by <= 3.1e-7 (measured)


def threshold_assign(ts: TileStats, tile_formats: list[str], metric: str, threshold: float, quantizer: Quantizer,
                     band: float = KNIFE_BAND) -> tuple[np.ndarray, int]:
    """reference :111-123 on a TileStats → (int8 (tiles_h, tiles_w) map, number of re-scored tiles)."""
    from .. import hip_backend as hb

    if ts.on_device and ts.host_stats is None:  # K4 on the device: the records stay where K1 wrote them
        amap, knife = hb.threshold_assign_device(ts.stats_dev, ts.mask, tile_formats, metric, threshold, band)
    else:
        amap, knife = hb.threshold_assign(ts.stats, ts.mask, tile_formats, metric, threshold, band)
    rescore_knife_tiles(ts, amap, knife, tile_formats, metric, threshold, quantizer)
    return amap.reshape(ts.tiles_h, ts.tiles_w), int(knife.size)


def rescore_knife_tiles(ts: TileStats, amap: np.ndarray, knife: np.ndarray, tile_formats: list[str], metric: str, threshold: float,
                        quantizer: Quantizer) -> None:
    """Decide the tiles in `knife` again with the literal float32 per-tile score of the reference (tile_utils.py:46-57 on y
    tiles quantized through the selected backend) and patch the flat int8 map `amap` in place."""
    if not knife.size:
        return
    by_prec = sorted(tile_formats, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))  # :112-114
    best = max(by_prec, key=lambda f: MIXED_TILE_BYTES_PER_ELEM.get(f, 0.0))             # :115
    x_tiles = gather_tiles(ts, knife)
    scores = {f: tile_metrics(x_tiles, np.asarray(_quantize_tiles(x_tiles, f, quantizer), dtype=np.float32), metric)
              for f in by_prec}
    for k, t in enumerate(knife):
        chosen = best
        for f in by_prec:
            if metric_is_good(scores[f][k], metric, threshold):  # np.float32 vs Python float, as the reference
                chosen = f
                break
        amap[t] = MIXED_TILE_FORMATS.index(chosen)


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
