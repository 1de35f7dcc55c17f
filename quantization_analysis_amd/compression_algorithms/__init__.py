"""Search algorithms selectable from the JSON config (`algorithm` key) — same names as the reference's registry
(compression_algorithms/__init__.py:11-29) for the mixed-tile family and the `none` baseline; `mixed-tile` is the
reference's alias of the greedy search.  The reference's `transpose` experiment (not a mixed-tile search, no
kernel of this package involved) is not provided and is reported as unsupported rather than mapped to
something else.
"""
from __future__ import annotations

from . import base as _base, config as _config
from . import mixed_tile_greedy as _greedy, mixed_tile_random as _random, mixed_tile_threshold as _threshold, none as _none

CompressionAlgorithm, CompressionResult = _base.CompressionAlgorithm, _base.CompressionResult
CompressionConfig, load_compression_config = _config.CompressionConfig, _config.load_compression_config
NoneCompression = _none.NoneCompression
MixedTileGreedyCompression = _greedy.MixedTileGreedyCompression
MixedTileRandomCompression = _random.MixedTileRandomCompression
MixedTileThresholdCompression = _threshold.MixedTileThresholdCompression

ALGORITHM_REGISTRY: dict = {cls.name: cls for cls in (NoneCompression, MixedTileGreedyCompression,
                                                      MixedTileRandomCompression, MixedTileThresholdCompression)}
ALGORITHM_REGISTRY["mixed-tile"] = MixedTileGreedyCompression


def create_algorithm(name: str, params: dict | None = None) -> CompressionAlgorithm:
    """Case-insensitive lookup; an unknown name lists what is available."""
    try:
        algorithm_cls = ALGORITHM_REGISTRY[name.strip().lower()]
    except KeyError:
        known = ", ".join(sorted(ALGORITHM_REGISTRY))
        raise ValueError(f"Unsupported compression algorithm '{name}'. Supported: {known}") from None
    return algorithm_cls.from_params(params if params else {})
