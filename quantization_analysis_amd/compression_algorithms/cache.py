"""On-disk cache of quantized tensors keyed by algorithm / BACKEND / format / tensor
(reference compression_algorithms/cache.py:17-30,68-79; the mixed_* paths there are never called).
The backend path segment keeps `hip` results apart from `emulation` results."""
from __future__ import annotations

import hashlib
import re
from dataclasses import dataclass
from pathlib import Path

import numpy as np


def _safe_tensor_key(tensor_name: str) -> str:
    """reference hf_model_utils.py:121-126."""
    digest = hashlib.sha1(tensor_name.encode("utf-8")).hexdigest()[:12]
    safe = re.sub(r"[^A-Za-z0-9._-]+", "_", tensor_name).strip("_") or "tensor"
    return f"{safe}--{digest}"


@dataclass
class CacheContext:
    root: Path
    tensor_name: str
    backend: str
    recompute: bool
    run_tag: str

    @property
    def safe_tensor(self) -> str:
        return _safe_tensor_key(self.tensor_name)

    def quant_path(self, compression: str, fmt: str) -> Path:
        return self.root / compression / self.backend / fmt / f"{self.safe_tensor}.npy"

    def load_array(self, compression: str, fmt: str) -> np.ndarray | None:
        if self.recompute:
            return None
        path = self.quant_path(compression, fmt)
        if path.exists():
            return np.load(path)
        return None

    def save_array(self, compression: str, fmt: str, y: np.ndarray) -> None:
        path = self.quant_path(compression, fmt)
        path.parent.mkdir(parents=True, exist_ok=True)
        np.save(path, y)
