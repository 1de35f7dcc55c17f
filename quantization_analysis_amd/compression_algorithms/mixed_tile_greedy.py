"""mixed-tile-greedy (reference compression_algorithms/mixed_tile_greedy.py:20-378).

Global-metric-constrained greedy demotion of tiles in seeded random order.  Decomposed as:
per-tile stats records for every listed format (tile_search.compute_tile_stats: K1 on the GPU for
backend "hip") + the literal sequential scan on the host (mtq_greedy_run of libmtq_hip.so, C++) with a
NumPy-bit-compatible generator for the visiting order (:222-231).  Bit-identical maps to the reference.
"""
from __future__ import annotations

import secrets

import numpy as np

from .base import CompressionAlgorithm, CompressionResult
from .cache import CacheContext
from .quantizer import Quantizer
from .tile_search import TileStats, columns_from_stats, compute_tile_stats, reconstruct
from .tile_utils import MIXED_TILE_FORMATS, mixed_tile_total_bytes


def parse_tile_formats(value) -> list[str]:
    """reference :41-60 (shared with the threshold algorithm, mixed_tile_threshold.py:39-58)."""
    if value is None or value == "":
        return []
    if isinstance(value, str):
        parts = [p.strip().lower() for p in value.split(",") if p.strip()]
    elif isinstance(value, list):
        parts = [str(p).strip().lower() for p in value if str(p).strip()]
    else:
        raise ValueError("formats must be a comma-separated string or a list of strings")
    formats: list[str] = []
    for part in parts:
        if part not in MIXED_TILE_FORMATS:
            raise ValueError(f"Unsupported mixed-tile format: {part}")
        if part not in formats:
            formats.append(part)
    return formats


def greedy_scan(ts: TileStats, tile_formats: list[str], metric: str, threshold: float, seed: int):
    """reference :95-103 (setup), :133-220 (initial sums), :222-346 (passes) on a TileStats: one call into the
    host C++ scan, whose visiting order is a bit-compatible restatement of np.random.default_rng(seed).permutation
    (mtq_rng, pinned against NumPy in tests/test_capi_host.py)."""
    from .. import hip_backend as hb

    if seed == 0:
        seed = secrets.randbits(31)  # :223-224
    amap, counts, cols = hb.greedy_run(ts.stats, ts.mask, tile_formats, metric, threshold, float(ts.numel), seed)
    value = cols["pcc"] if metric == "pcc" else (cols["mae"] if metric == "mae" else cols["atol"])
    return amap.reshape(ts.tiles_h, ts.tiles_w), counts, value


def greedy_scan_numpy_rng(ts: TileStats, tile_formats: list[str], metric: str, threshold: float, seed: int):
    """Same scan with NumPy's own Generator supplying every pass's order (the literal :225-231)."""
    from .. import hip_backend as hb

    scan = hb.GreedyScan(ts.stats, ts.mask, metric, threshold, float(ts.numel), tile_formats[0])
    try:
        rng = np.random.default_rng(seed)  # :225
        for fmt in tile_formats:  # :227
            candidates = np.where(scan.fixed() == 0)[0]  # :228
            if candidates.size == 0:
                break
            scan.run_pass(fmt, rng.permutation(candidates))  # :231-346
        return scan.assignment().reshape(ts.tiles_h, ts.tiles_w), scan.counts(), scan.value()
    finally:
        scan.close()


class MixedTileGreedyCompression(CompressionAlgorithm):
    name = "mixed-tile-greedy"

    def __init__(self, params: dict | None = None) -> None:
        super().__init__(params=params)
        raw_formats = self.params.get("formats", self.params.get("tile_formats"))
        self.metric = self.params.get("metric", "pcc")
        self.threshold = float(self.params.get("threshold", 0.999))
        self.seed = int(self.params.get("seed", 0))
        self.tile_formats = parse_tile_formats(raw_formats) if raw_formats is not None else None
        self.materialize_y = bool(self.params.get("materialize_y", True))  # ours: skip the y copy in throughput runs
        if self.metric not in {"pcc", "mae", "atol"}:
            raise ValueError(f"Unsupported metric: {self.metric}")

    @classmethod
    def from_params(cls, params: dict | None = None) -> "MixedTileGreedyCompression":
        return cls(params=params or {})

    def expected_evals(self, formats: list[str]) -> int:
        return 1

    _parse_formats = staticmethod(parse_tile_formats)

    @staticmethod
    def _filter_from_formats(formats: list[str]) -> list[str]:
        allowed = [fmt for fmt in formats if fmt in MIXED_TILE_FORMATS]
        if not allowed:
            raise ValueError(
                "mixed-tile-greedy requires at least one of "
                f"{', '.join(MIXED_TILE_FORMATS)} in quantization_formats"
            )
        return allowed

    def run(self, xf, formats: list[str], quantizer: Quantizer, cache: CacheContext) -> list[CompressionResult]:
        tile_formats = self.tile_formats or self._filter_from_formats(formats)
        size = int(np.asarray(xf).size) if isinstance(xf, np.ndarray) or np.isscalar(xf) else int(xf.numel())
        if size == 0:  # :78-83
            y = np.asarray(xf, dtype=np.float32)
            counts = {fmt: 0 for fmt in MIXED_TILE_FORMATS}
            assignment = np.zeros((1, 1), dtype=np.int8)
            meta = {"assignment": assignment, "tile_formats": tile_formats}
        else:
            ts = compute_tile_stats(xf, tile_formats, quantizer)
            assignment, counts, value = greedy_scan(ts, tile_formats, self.metric, self.threshold, self.seed)
            y = reconstruct(ts, assignment, quantizer) if self.materialize_y else None
            meta = {"assignment": assignment, "tile_formats": tile_formats,
                    "columns": columns_from_stats(ts, assignment), "metric_value": value}
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
