"""`none`: every requested format applied to the whole tensor — the baseline rows `wq` prints next to the
selected algorithm (wq:589-590; reference compression_algorithms/none.py:13-31).  Reconstructions are kept in the
per-tensor `.npy` cache and reused when their shape still matches."""
from __future__ import annotations

import numpy as np

from .base import CompressionAlgorithm, CompressionResult


def _to_host(y) -> np.ndarray:
    return np.asarray(y.cpu().numpy() if hasattr(y, "cpu") else y)


class NoneCompression(CompressionAlgorithm):
    name = "none"

    def _reconstruction(self, xf, fmt: str, quantizer, cache):
        cached = cache.load_array(self.name, fmt)
        if cached is not None and cached.shape == xf.shape:
            return cached
        y = quantizer.quantize(xf, fmt)
        cache.save_array(self.name, fmt, _to_host(y))
        return y

    def run(self, xf, formats: list, quantizer, cache) -> list:
        return [CompressionResult(fmt=fmt.upper(), compression=self.name, y=self._reconstruction(xf, fmt, quantizer, cache))
                for fmt in formats]
