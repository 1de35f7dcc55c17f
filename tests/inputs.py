"""Seeded synthetic inputs shared by the tests, bench.py and the golden generator
(tests/golden/make_golden.py::gen must stay byte-identical to gen() here)."""
from __future__ import annotations

import hashlib

import numpy as np


def to_bf16_valued(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)) << np.uint32(16)
    return r.view(np.float32)


def gen(kind: str, seed: int, shape) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if kind == "normal_bf16":
        return to_bf16_valued((rng.standard_normal(shape) * 0.02).astype(np.float32))
    if kind == "normal_f32":
        return (rng.standard_normal(shape) * 0.02).astype(np.float32)
    if kind == "heavy_bf16":
        a = rng.standard_normal(shape) * 0.02
        return to_bf16_valued((a * np.exp(1.5 * rng.standard_normal(shape))).astype(np.float32))
    if kind == "heavy_f32":
        a = rng.standard_normal(shape) * 0.02
        return (a * np.exp(1.5 * rng.standard_normal(shape))).astype(np.float32)
    raise ValueError(kind)


def sha(x: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


def m1_tensor() -> np.ndarray:
    """SURVEY §8(d) M1 / BASELINE configs[1]: one 4096x4096 bf16 tensor drawn from torch's CPU generator, as float32 values
    (tests/golden/make_golden.py::m1_tensor must stay identical)."""
    import torch

    g = torch.Generator().manual_seed(0)
    return (torch.randn(4096, 4096, generator=g) * 0.02).to(torch.bfloat16).float().numpy()
