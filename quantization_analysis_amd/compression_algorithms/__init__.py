"""Algorithm registry (reference compression_algorithms/__init__.py:11-29).

Built here: none, mixed-tile-greedy (alias mixed-tile), mixed-tile-threshold.  `transpose` and
`mixed-tile-random` of the reference are outside the hot path this package covers and are
reported as unsupported rather than silently mapped to something else.
"""
from __future__ import annotations

from .base import CompressionAlgorithm, CompressionResult
from .config import CompressionConfig, load_compression_config
from .mixed_tile_greedy import MixedTileGreedyCompression
from .mixed_tile_threshold import MixedTileThresholdCompression
from .none import NoneCompression

ALGORITHM_REGISTRY: dict[str, type[CompressionAlgorithm]] = {
    "none": NoneCompression,
    "mixed-tile-greedy": MixedTileGreedyCompression,
    "mixed-tile-threshold": MixedTileThresholdCompression,
    "mixed-tile": MixedTileGreedyCompression,
}


def create_algorithm(name: str, params: dict | None = None) -> CompressionAlgorithm:
    key = name.strip().lower()
    cls = ALGORITHM_REGISTRY.get(key)
    if cls is None:
        raise ValueError(
            f"Unsupported compression algorithm '{name}'. "
            f"Supported: {', '.join(sorted(ALGORITHM_REGISTRY))}"
        )
    return cls.from_params(params or {})
