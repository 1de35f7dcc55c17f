"""The `--compression-config` JSON file (reference compression_algorithms/config.py:8-69).

{"algorithm": "...", "params": {...}, "quantization_formats": [...], "seed": int | 0 | "random", "random_seed": bool}
Seed rules (:52-61): a non-zero int is used as is; 0, "random" or random_seed=true mean "draw a fresh seed per run".
"""
from __future__ import annotations

import dataclasses
import json
import pathlib
from typing import Optional


@dataclasses.dataclass
class CompressionConfig:
    algorithm: str
    params: dict
    quantization_formats: Optional[list]
    seed: Optional[int]
    random_seed: bool


def _formats_field(raw) -> Optional[list]:
    if raw is None:
        return None
    if not isinstance(raw, list):
        raise ValueError("Compression config 'quantization_formats' must be a list of strings")
    cleaned = [str(entry).strip().lower() for entry in raw]
    cleaned = [entry for entry in cleaned if entry]
    return cleaned if cleaned else None


def _seed_fields(raw, want_random: bool) -> tuple:
    """→ (seed or None, random_seed)."""
    if raw is None:
        return None, want_random
    if isinstance(raw, str) and raw.strip().lower() == "random":
        return None, True
    try:
        value = int(raw)
    except (TypeError, ValueError) as exc:
        raise ValueError("Compression config 'seed' must be an int, 0, or 'random'") from exc
    return (None, True) if value == 0 else (value, want_random)


def load_compression_config(path: Optional[str]) -> CompressionConfig:
    if path is None:  # no file: the plain per-format baseline
        return CompressionConfig("none", {}, None, None, False)
    file = pathlib.Path(path)
    if not file.exists():
        raise FileNotFoundError(f"Compression config not found: {path}")
    doc = json.loads(file.read_text(encoding="utf-8"))
    if not isinstance(doc, dict):
        raise ValueError("Compression config must be a JSON object")
    params = doc.get("params")
    if params is None:
        params = {}
    elif not isinstance(params, dict):
        raise ValueError("Compression config 'params' must be an object")
    seed, random_seed = _seed_fields(doc.get("seed"), bool(doc.get("random_seed", False)))
    return CompressionConfig(
        algorithm=str(doc.get("algorithm", "none")).strip().lower(),
        params=params,
        quantization_formats=_formats_field(doc.get("quantization_formats")),
        seed=seed,
        random_seed=random_seed,
    )
