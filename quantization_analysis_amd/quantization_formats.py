"""Host (`--backend emulation`) mirror of the reference's quantization_formats.py for the mixed-tile
formats: bf16 RNE round-trip, TTNN-style BFP{8,4,2} with one shared exponent per 16 contiguous
last-axis elements, fp0.  Reference: quantization_formats.py:8,29-45,71-81,84-164,167-194.

This module is plain NumPy and runs without a GPU.  The `hip` backend does NOT route through it:
it calls libmtq_hip.so (see compression_algorithms/quantizer.py).
"""
from __future__ import annotations

import numpy as np

# mxfp4 / nvfp4 (reference :8) are per-element Python proxies outside the mixed-tile path and are not built here.
SUPPORTED_FORMATS = ["bf16", "bfp8", "bfp4", "bfp2", "fp0"]
_MANT_BITS = {"bfp8": 7, "bfp4": 3, "bfp2": 1}


def fp32_to_bf16_round_to_nearest_even(x: np.ndarray) -> np.ndarray:
    """reference :29-35 — RNE on the raw word, uint32 wrap, no NaN special case."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32)
    return ((u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) >> np.uint32(16)).astype(np.uint16)


def bf16_to_fp32(bf16: np.ndarray) -> np.ndarray:
    """reference :38-41."""
    return (np.asarray(bf16, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def quantize_dequantize_bf16(x: np.ndarray) -> np.ndarray:
    return bf16_to_fp32(fp32_to_bf16_round_to_nearest_even(x))


def quantize_dequantize_bfp_ttnn(x: np.ndarray, mant_bits: int) -> np.ndarray:
    """reference :84-164.  Groups = 16 contiguous last-axis elements aligned from index 0; a partial
    last group is completed with +0.0 (the 32x32 tile of the reference only adds zero padding)."""
    x = np.asarray(x, dtype=np.float32)
    if x.size == 0:
        return x.astype(np.float32)
    shape = x.shape
    rows = x.reshape(1, -1) if x.ndim <= 1 else x.reshape(-1, shape[-1])
    n, w = rows.shape
    wp = -(-w // 16) * 16
    u = np.zeros((n, wp), dtype=np.uint32)
    u[:, :w] = rows.view(np.uint32)
    g = u.reshape(n, wp // 16, 16)
    m = mant_bits
    exp = (g >> np.uint32(23)) & np.uint32(0xFF)
    shared = exp.max(axis=-1, keepdims=True)                      # :118-119
    d = shared - exp                                               # :126
    man = (g & np.uint32(0x7FFFFF)) | np.uint32(1 << 23)           # :121,125
    man = np.where(d > 31, np.uint32(0), man >> np.minimum(d, np.uint32(31)))  # :127-131
    shift = np.uint32(24 - m)
    rv = man & np.uint32((1 << (24 - m)) - 1)                      # :136
    tie = np.uint32(1 << (23 - m))
    man = man >> shift                                             # :137
    up = (rv > tie) | ((rv == tie) & ((man & np.uint32(1)) == 1))  # :138-139
    man = np.minimum(man + up.astype(np.uint32), np.uint32((1 << m) - 1))  # :140-141 saturate
    man = np.where(exp == 0, np.uint32(0), man)                    # :145
    sign = np.where(man == 0, np.uint32(0), g >> np.uint32(31))    # :143
    msb = np.zeros_like(man)
    for b in range(m):                                             # decode table :71-81 as msb search
        msb = np.where(((man >> np.uint32(b)) & np.uint32(1)) == 1, np.uint32(b), msb)
    sc = np.uint32(m - 1) - msb
    ms = (man << (sc + np.uint32(1))) & np.uint32((1 << m) - 1)
    exp_out = shared - sc                                          # :154 (uint32 wrap kept)
    bits = (sign << np.uint32(31)) | (exp_out << np.uint32(23)) | (ms << np.uint32(23 - m))  # :158
    bits = np.where(man == 0, np.uint32(0), bits).astype(np.uint32)
    return bits.reshape(n, wp)[:, :w].copy().view(np.float32).reshape(shape)


def quantize_fp0(x: np.ndarray) -> np.ndarray:
    return np.zeros_like(np.asarray(x, dtype=np.float32), dtype=np.float32)


def quantize_weight_values(x: np.ndarray, fmt: str) -> np.ndarray:
    """reference :171-194 for the formats of this path."""
    fmt = fmt.lower()
    x = np.asarray(x, dtype=np.float32)
    if fmt == "bf16":
        return quantize_dequantize_bf16(x)
    if fmt in _MANT_BITS:
        return quantize_dequantize_bfp_ttnn(x, mant_bits=_MANT_BITS[fmt])
    if fmt == "fp0":
        return quantize_fp0(x)
    raise ValueError(f"Unsupported weight format: {fmt}")
