"""Backend selector (reference compression_algorithms/quantizer.py:8-34).

backend == "emulation": host NumPy (quantization_formats.py of this package).
backend == "hip":       K2 of libmtq_hip.so on the current HIP device; raises if the library or a
                        GPU is missing — never falls back to the host path.
backend == "ttnn":      recognised for CLI compatibility; this build has no Tenstorrent path.
"""
from __future__ import annotations

import numpy as np

from ..quantization_formats import SUPPORTED_FORMATS, quantize_weight_values

BACKENDS = ("emulation", "hip", "ttnn")


class Quantizer:
    def __init__(self, backend: str, ttnn=None) -> None:
        if backend not in BACKENDS:
            raise ValueError(f"Unsupported backend '{backend}'. Supported: {', '.join(BACKENDS)}")
        self.backend = backend
        self.ttnn = ttnn
        if backend == "hip":
            from .. import hip_backend

            hip_backend.require_gpu()  # fail at construction, like the reference's ttnn import check (wq:603-609)

    def quantize(self, xf, fmt: str):
        """np.ndarray float32 in → np.ndarray float32 out (same shape).  With backend 'hip' a torch
        device tensor (bf16 / fp32) may be passed instead and a device float32 tensor is returned."""
        fmt_l = fmt.lower()
        if self.backend == "ttnn":
            raise RuntimeError("Internal error: TTNN backend selected but ttnn is not initialized.")  # reference :24-25
        if self.backend == "hip":
            if fmt_l not in SUPPORTED_FORMATS:
                raise ValueError(f"Unsupported weight format: {fmt_l}")
            from .. import hip_backend as hb

            is_np = isinstance(xf, np.ndarray) or np.isscalar(xf)
            if is_np and np.asarray(xf).size == 0:
                return np.asarray(xf, dtype=np.float32).copy()
            x2d, info = hb.to_device_2d(xf)
            y = hb.unflatten(hb.quantize(x2d, fmt_l), info)
            return y.cpu().numpy() if is_np else y
        return quantize_weight_values(xf, fmt_l)
