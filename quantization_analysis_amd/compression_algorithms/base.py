"""What every search algorithm returns and the interface `wq` drives.

Mirrors the public surface of the reference's compression_algorithms/base.py:13-44 (field names and order of
`CompressionResult`, `from_params`, `expected_evals`, `run`) so callers written against the reference keep
working.  On the hip backend `y` may be a torch device tensor, or None when the caller asked an algorithm not
to materialise it (`materialize_y=False`); `meta["columns"]` then carries pcc/mae/atol.
"""
from __future__ import annotations

import abc
import dataclasses
from typing import Any, Iterable, Optional


@dataclasses.dataclass
class CompressionResult:
    """One table row of `wq`: a tensor reconstructed under `fmt` by algorithm `compression`."""

    fmt: str                                   # "BF16" … or "MIXED"
    compression: str                           # algorithm name, e.g. "mixed-tile-greedy"
    y: Any                                     # reconstruction, same shape as the input
    tile_counts: Optional[dict] = None         # tiles per mixed-tile format (all four keys), mixed results only
    tile_bytes: Optional[float] = None         # mixed_tile_total_bytes(tile_counts)
    meta: Optional[dict] = None                # "assignment" int8 (tiles_h, tiles_w), "tile_formats", …


class CompressionAlgorithm(abc.ABC):
    """An algorithm is constructed from the `params` object of the JSON config and run once per tensor."""

    name: str = ""

    def __init__(self, params: Optional[dict] = None) -> None:
        self.params = dict(params) if params else {}

    @classmethod
    def from_params(cls, params: Optional[dict] = None):
        return cls(params=params if params else {})

    def expected_evals(self, formats: Iterable[str]) -> int:
        """Progress-bar total (wq:652): one evaluation per requested format unless a subclass says otherwise."""
        return sum(1 for _ in formats)

    @abc.abstractmethod
    def run(self, xf, formats: list, quantizer, cache) -> list:
        """xf: float32 ndarray (or a device tensor on the hip backend) → list[CompressionResult]."""
