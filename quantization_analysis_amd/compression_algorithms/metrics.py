"""Tensor-level and tile-level quality metrics in the reference's float32 arithmetic
(compression_algorithms/metrics.py:6-39).  These literal forms are what knife-edge decisions are re-scored with;
the bulk of the scoring runs on the float64 stats records instead (tile_search.py).

pcc  : Pearson correlation, two-pass in float32 (mean, centred norms, dot); a zero denominator yields 1.0 for
       identical inputs and 0.0 otherwise.  Higher is better, passes when >= threshold.
mae  : mean |a - b| in float32.  Lower is better, passes when <= threshold.
atol : max |a - b| in float32.   Lower is better, passes when <= threshold.
"""
from __future__ import annotations

import numpy as np

_HIGHER_IS_BETTER = {"pcc"}
_KNOWN = ("pcc", "mae", "atol")


def _flat32(v) -> np.ndarray:
    return np.asarray(v, dtype=np.float32).reshape(-1)


def pearson_corr(a, b) -> float:
    p, q = _flat32(a), _flat32(b)
    if p.size == 0:
        return 1.0
    pc = p - np.mean(p)
    qc = q - np.mean(q)
    scale = float(np.linalg.norm(pc) * np.linalg.norm(qc))
    if scale != 0.0:
        return float(np.dot(pc, qc) / scale)
    identical = np.max(np.abs(p - q)) == 0.0
    return 1.0 if identical else 0.0


def _abs_err(a, b) -> np.ndarray:
    return np.abs(np.asarray(a, dtype=np.float32) - np.asarray(b, dtype=np.float32))


def metric_value(a, b, metric: str) -> float:
    if metric not in _KNOWN:
        raise ValueError(f"Unsupported metric: {metric}")
    if metric == "pcc":
        return pearson_corr(a, b)
    err = _abs_err(a, b)
    return float(np.mean(err)) if metric == "mae" else float(np.max(err))


def metric_is_good(value, metric: str, threshold) -> bool:
    """`value` keeps the type the caller has (np.float32 scores compare in float32 under NumPy >= 2)."""
    return value >= threshold if metric in _HIGHER_IS_BETTER else value <= threshold


def metric_better(a, b, metric: str) -> bool:
    return a > b if metric in _HIGHER_IS_BETTER else a < b
