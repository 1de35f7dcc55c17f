"""compression_algorithms/tile_utils.py of the reference: constants, size model, 2-D flatten/pad,
tile gather/scatter, per-tile metrics (:8-14, :32-37, :46-57, :91-132)."""
from __future__ import annotations

import numpy as np

from .metrics import metric_value, pearson_corr

TILE_HW = 32
# index in this list = int8 code in assignment maps; bytes per element include the shared exponents (1 B per 16 values)
MIXED_TILE_FORMATS = ["bf16", "bfp8", "bfp4", "bfp2"]
MIXED_TILE_BYTES_PER_ELEM = dict(zip(MIXED_TILE_FORMATS, (2.0, 1.088, 0.50097, 0.25097)))


def assignment_to_array(assignment) -> np.ndarray:
    return np.asarray(assignment, dtype=np.int8)


def mixed_tile_total_bytes(counts: dict, tile_hw: int = TILE_HW) -> float:
    """Size model of a mixed map: Σ tiles × elements per tile × bytes per element, accumulated in the iteration
    order of `counts` (Python floats — the BYTES column is compared exactly)."""
    per_tile = float(tile_hw * tile_hw)
    acc = 0.0
    for name, n_tiles in counts.items():
        acc += float(n_tiles) * per_tile * MIXED_TILE_BYTES_PER_ELEM.get(name, 0.0)
    return acc


def format_tag(formats) -> str:
    return "+".join(formats) if formats else "none"


_ROWWISE_DOT_IS_BLAS = None   # does np.vecdot / batched matmul give a row's x.dot(y) bit for bit on this NumPy build?  (checked on first use)


def _rowwise_dot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """a[t].dot(b[t]) for every row t of two C-contiguous float32 [T, n] arrays, as float32 [T], with the bits of the per-row call.
    NumPy's batched matmul of a (1 x n) by an (n x 1) operand runs cblas_sdot per item — the routine ndarray.dot uses — so the Python
    loop is only needed where a build does otherwise: the first call compares the two on a few rows and remembers the answer."""
    global _ROWWISE_DOT_IS_BLAS
    if _ROWWISE_DOT_IS_BLAS is None:
        rng = np.random.default_rng(12345)
        u = rng.standard_normal((16, 1024)).astype(np.float32)
        v = (u * 0.99 + rng.standard_normal((16, 1024)).astype(np.float32) * 0.05).astype(np.float32)
        loop = np.array([u[t].dot(v[t]) for t in range(16)], dtype=np.float32)
        _ROWWISE_DOT_IS_BLAS = bool(np.array_equal(loop.view(np.uint32), np.matmul(u[:, None, :], v[:, :, None])[:, 0, 0].view(np.uint32)))
    if _ROWWISE_DOT_IS_BLAS:
        return np.matmul(a[:, None, :], b[:, :, None])[:, 0, 0]
    return np.array([a[t].dot(b[t]) for t in range(a.shape[0])], dtype=np.float32)


def pearson_corr_tiles(ref_tiles: np.ndarray, q_tiles: np.ndarray) -> np.ndarray:
    """metrics.pearson_corr for every (T, 32, 32) tile pair → float32 [T], bit for bit what the per-tile call returns, without a
    Python-level call per tile (29 µs per tile on the build container; 1.3 µs here): the means and the centring run once over all
    tiles — NumPy reduces a contiguous last axis row by row with the same pairwise routine np.mean uses on one row — the three dot
    products per tile (np.linalg.norm of a real vector IS sqrt(x.dot(x))) come from _rowwise_dot, and the scalar types pearson_corr
    goes through are float32 throughout (np.float32 norms, their float32 product — widened to a Python float and narrowed again
    without loss — and the float32 quotient).  tests/test_algorithms.py pins the equality on 10^4 tiles, degenerate ones included."""
    count = ref_tiles.shape[0]
    if count == 0:
        return np.empty(0, dtype=np.float32)
    p = np.ascontiguousarray(ref_tiles, dtype=np.float32).reshape(count, -1)
    q = np.ascontiguousarray(q_tiles, dtype=np.float32).reshape(count, -1)
    if p.shape[1] == 0:
        return np.ones(count, dtype=np.float32)
    pc = p - np.mean(p, axis=1, keepdims=True)
    qc = q - np.mean(q, axis=1, keepdims=True)
    scale = np.sqrt(_rowwise_dot(pc, pc)) * np.sqrt(_rowwise_dot(qc, qc))         # float32, as np.float32 * np.float32
    ok = scale != 0.0
    with np.errstate(all="ignore"):
        out = (_rowwise_dot(pc, qc) / scale).astype(np.float32)
    for t in np.flatnonzero(~ok):                                                  # zero variance on either side (metrics.py:14-15)
        out[t] = 1.0 if np.max(np.abs(p[t] - q[t])) == 0.0 else 0.0
    return out


def tile_metrics(ref_tiles: np.ndarray, q_tiles: np.ndarray, metric: str) -> np.ndarray:
    """float32 score of every (T, 32, 32) tile pair over all 1024 positions, pads included (:46-57)."""
    count = ref_tiles.shape[0]
    if metric == "pcc":
        return pearson_corr_tiles(ref_tiles, q_tiles)
    if metric not in ("mae", "atol"):
        raise ValueError(f"Unsupported metric: {metric}")
    err = np.abs(ref_tiles - q_tiles).reshape(count, -1)
    return err.mean(axis=1) if metric == "mae" else err.max(axis=1)


def flatten_2d(xf: np.ndarray) -> tuple[np.ndarray, tuple]:
    """The 2-D view of reshape_to_2d_with_padding (:94-107) WITHOUT the padding copy."""
    xf = np.asarray(xf, dtype=np.float32)
    if xf.ndim == 0:
        return xf.reshape(1, 1), ("scalar", xf.shape)
    if xf.ndim == 1:
        n = xf.shape[0]
        h = int(np.ceil(n / 32.0))
        data2d = np.zeros((h, 32), dtype=np.float32)
        data2d.reshape(-1)[:n] = xf.reshape(-1)
        return data2d, ("vector", n)
    return xf.reshape(int(np.prod(xf.shape[:-1])), xf.shape[-1]), ("nd", xf.shape)


def reshape_to_2d_with_padding(xf: np.ndarray) -> tuple[np.ndarray, tuple, tuple]:
    """→ (zero-padded matrix with 32-multiple sides, shape_info for unflatten_2d, (h, w, h_pad, w_pad))."""
    flat, shape_info = flatten_2d(xf)
    h, w = flat.shape
    h_pad, w_pad = -(-h // TILE_HW) * TILE_HW, -(-w // TILE_HW) * TILE_HW
    out = np.zeros((h_pad, w_pad), dtype=np.float32)
    out[:h, :w] = flat
    return out, shape_info, (h, w, h_pad, w_pad)


def to_tiles(padded: np.ndarray, tile_hw: int = 32) -> np.ndarray:
    """(h_pad, w_pad) → (T, 32, 32), tile id = tr*tiles_w + tc (mixed_tile_greedy.py:89-93)."""
    th, tw = padded.shape[0] // tile_hw, padded.shape[1] // tile_hw
    return padded.reshape(th, tile_hw, tw, tile_hw).transpose(0, 2, 1, 3).reshape(-1, tile_hw, tile_hw)


def unflatten_2d(data2d: np.ndarray, shape_info: tuple) -> np.ndarray:
    if shape_info[0] == "scalar":
        return np.array(data2d[0, 0], dtype=np.float32)
    if shape_info[0] == "vector":
        return data2d.reshape(-1)[: shape_info[1]].astype(np.float32)
    if shape_info[0] == "nd":
        return data2d.reshape(shape_info[1]).astype(np.float32)
    raise ValueError("Invalid shape_info")


def reconstruct_from_tiles(tiles: np.ndarray, shape_info: tuple, pad_info: tuple, tile_hw: int = 32) -> np.ndarray:
    h, w, h_pad, w_pad = pad_info
    th, tw = h_pad // tile_hw, w_pad // tile_hw
    padded = tiles.reshape(th, tw, tile_hw, tile_hw).transpose(0, 2, 1, 3).reshape(h_pad, w_pad)
    return unflatten_2d(padded[:h, :w], shape_info)


def global_metric(xf: np.ndarray, tiles: np.ndarray, shape_info: tuple, pad_info: tuple, metric: str) -> float:
    return metric_value(xf, reconstruct_from_tiles(tiles, shape_info, pad_info), metric)
