"""Shared core of the two mixed-tile search algorithms: per-tile raw sums ("stats records") for every
candidate format, and reconstruction of y from an assignment map.

The reference recomputes quantized tiles and per-tile sums inside each algorithm
(mixed_tile_greedy.py:98,133-220,232-254; mixed_tile_threshold.py:97-109).  Here both algorithms
consume one record per tile: [Σx, Σx², {Σy, Σy², Σxy, Σ|x−y|, max|x−y|} per format] in float64 — the
layout of include/mtq.h — produced by

  backend "hip":        K1 of libmtq_hip.so, one pass over HBM, y never materialised;
  backend "emulation":  the NumPy code below on the host (same definition, same summation order).

Sums run over the zero-padded 32x32 tile; pads are exact zeros in x and y, so they equal the reference's
valid-region sums (mixed_tile_greedy.py:122-131) — SURVEY §8(c).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

from .quantizer import Quantizer
from .tile_utils import MIXED_TILE_FORMATS, flatten_2d, unflatten_2d

TILE = 32


def fmt_mask(formats) -> int:
    m = 0
    for f in formats:
        m |= 1 << MIXED_TILE_FORMATS.index(f)
    return m


def mask_formats(mask: int) -> list[str]:
    return [f for i, f in enumerate(MIXED_TILE_FORMATS) if mask & (1 << i)]


def slot_of(mask: int, fmt: str) -> int:
    i = MIXED_TILE_FORMATS.index(fmt)
    return bin(mask & ((1 << i) - 1)).count("1")


@dataclass
class TileStats:
    mask: int
    tiles_h: int
    tiles_w: int
    numel: int                 # xf.size (elem_count of mixed_tile_greedy.py:134)
    shape_info: tuple
    x2d: object                # host ndarray (emulation) or device tensor (hip), the 2-D flatten of xf
    backend: str
    input_was_numpy: bool
    host_stats: Optional[np.ndarray] = None   # float64 [tiles, 2+5F] on the host
    stats_dev: object = None                  # the same records where K1 wrote them (hip backend)

    @property
    def stats(self) -> np.ndarray:
        """Host copy of the records; for the hip backend it is made on first use — the sequential greedy scan needs it,
        the per-tile / reduction decisions (threshold rule, scores, column sums) read the device copy instead."""
        if self.host_stats is None:
            self.host_stats = self.stats_dev.cpu().numpy()
        return self.host_stats

    @property
    def on_device(self) -> bool:
        return self.stats_dev is not None

    @property
    def tiles(self) -> int:
        return self.tiles_h * self.tiles_w

    def block(self, fmt: str) -> np.ndarray:
        """[tiles, 5] view: Σy, Σy², Σxy, Σ|d|, max|d| of one format."""
        s = 2 + 5 * slot_of(self.mask, fmt)
        return self.stats[:, s:s + 5]


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def host_tile_stats(x2d: np.ndarray, formats_in_mask_order: list[str], quantizer: Quantizer) -> np.ndarray:
    """Host definition of the stats record (NumPy).  Terms are float32 expressions summed in float64:
    inside a shared-exponent group the elements within 14 binades of the shared exponent sequentially, the
    rest ("tail", zeros included) sequentially, main + tail; then sequentially over the 4 groups of a row pair
    (rows 2j, 2j+1) and by a balanced binary tree over the 16 row pairs — the order the HIP kernels use."""
    h, w = x2d.shape
    th, tw = -(-h // TILE), -(-w // TILE)

    def lanes(a2d: np.ndarray) -> np.ndarray:  # (h,w) → (T, 64, 16), lane = 2*row + half
        p = np.zeros((th * TILE, tw * TILE), dtype=np.float32)
        p[:h, :w] = a2d
        return p.reshape(th, TILE, tw, 2, 16).transpose(0, 2, 1, 3, 4).reshape(th * tw, 64, 16)

    def reduce_sum(term: np.ndarray) -> np.ndarray:
        main = np.zeros(term.shape[:2], dtype=np.float64)
        tail = np.zeros(term.shape[:2], dtype=np.float64)
        for i in range(16):
            v = term[:, :, i].astype(np.float64)
            main = np.where(is_tail[:, :, i], main, main + v)
            tail = np.where(is_tail[:, :, i], tail + v, tail)
        acc = main + tail
        acc = ((acc[:, 0::4] + acc[:, 1::4]) + acc[:, 2::4]) + acc[:, 3::4]  # the 4 groups of a row pair, sequentially
        while acc.shape[1] > 1:                                               # 16 row pairs: balanced tree
            acc = acc[:, 0::2] + acc[:, 1::2]
        return acc[:, 0]

    xl = lanes(x2d)
    ex = (xl.view(np.uint32) >> np.uint32(23)) & np.uint32(0xFF)
    is_tail = (ex.max(axis=2, keepdims=True) - ex) > 14   # more than 14 binades below the group's shared exponent
    cols = [reduce_sum(xl), reduce_sum(xl * xl)]
    with np.errstate(all="ignore"):
        for fmt in formats_in_mask_order:
            yl = lanes(np.asarray(quantizer.quantize(x2d, fmt), dtype=np.float32))
            d = np.abs(xl - yl)
            cols += [reduce_sum(yl), reduce_sum(yl * yl), reduce_sum(xl * yl), reduce_sum(d),
                     d.reshape(d.shape[0], -1).max(axis=1).astype(np.float64)]
    return np.ascontiguousarray(np.stack(cols, axis=1))


def compute_tile_stats(xf, formats: list[str], quantizer: Quantizer) -> TileStats:
    """One stats record per 32x32 tile of xf for the given mixed-tile formats."""
    mask = fmt_mask(formats)
    fm = mask_formats(mask)
    if quantizer.backend == "hip":
        from .. import hip_backend as hb

        was_np = not _is_torch(xf)
        x2d, info = hb.to_device_2d(xf)
        numel = int(np.asarray(xf).size) if was_np else int(xf.numel())
        th, tw = hb.tiles_hw(*x2d.shape)
        return TileStats(mask, th, tw, numel, info, x2d, "hip", was_np, stats_dev=hb.tile_stats(x2d, mask))
    if _is_torch(xf):
        xf = xf.detach().to("cpu").float().numpy()
    xf = np.asarray(xf, dtype=np.float32)
    x2d, info = flatten_2d(xf)
    th, tw = -(-x2d.shape[0] // TILE), -(-x2d.shape[1] // TILE)
    return TileStats(mask, th, tw, int(xf.size), info, x2d, quantizer.backend, True, host_stats=host_tile_stats(x2d, fm, quantizer))


def reconstruct(ts: TileStats, assignment: np.ndarray, quantizer: Quantizer):
    """y whose tile t uses format MIXED_TILE_FORMATS[assignment[t]] (mixed_tile_threshold.py:125-132,
    mixed_tile_greedy.py:273,348-352).  hip: K3 on the device."""
    a = np.asarray(assignment, dtype=np.int8).reshape(ts.tiles_h, ts.tiles_w)
    if ts.backend == "hip":
        from .. import hip_backend as hb

        y = hb.unflatten(hb.apply_assignment(ts.x2d, a), ts.shape_info)
        return y.cpu().numpy() if ts.input_was_numpy else y
    h, w = ts.x2d.shape
    sel = np.repeat(np.repeat(a, TILE, axis=0), TILE, axis=1)[:h, :w]
    y2d = np.array(ts.x2d, dtype=np.float32, copy=True)
    for idx in np.unique(a):
        yq = quantizer.quantize(ts.x2d, MIXED_TILE_FORMATS[int(idx)])
        y2d = np.where(sel == idx, yq, y2d)
    return unflatten_2d(y2d, ts.shape_info)


def gather_tiles(ts: TileStats, tile_ids: np.ndarray) -> np.ndarray:
    """Zero-padded (k,32,32) float32 host copies of the named tiles of x (for knife-edge re-scoring).  On the hip backend
    the tiles are gathered on the device with one indexed read and come over in one copy."""
    ids = np.asarray(tile_ids, dtype=np.int64).reshape(-1)
    h, w = ts.x2d.shape
    if ts.backend == "hip" and ids.size:
        torch = __import__("torch")
        dev = ts.x2d.device
        t = torch.from_numpy(ids).to(dev)
        ar = torch.arange(TILE, device=dev)
        rows = (t // ts.tiles_w)[:, None] * TILE + ar            # (k, 32)
        cols = (t % ts.tiles_w)[:, None] * TILE + ar
        vals = ts.x2d[rows.clamp(max=h - 1)[:, :, None], cols.clamp(max=w - 1)[:, None, :]].float()
        inside = (rows < h)[:, :, None] & (cols < w)[:, None, :]
        return torch.where(inside, vals, torch.zeros((), dtype=torch.float32, device=dev)).cpu().numpy()
    out = np.zeros((ids.size, TILE, TILE), dtype=np.float32)
    for k, t in enumerate(ids):
        tr, tc = divmod(int(t), ts.tiles_w)
        r0, c0 = tr * TILE, tc * TILE
        blk = ts.x2d[r0:min(r0 + TILE, h), c0:min(c0 + TILE, w)]
        out[k, : blk.shape[0], : blk.shape[1]] = blk
    return out


_PINNED = {}   # thread id -> flat pinned float32 staging buffer of literal_inputs(pinned=True); grows, never shrinks


def _pinned_f32(torch, numel: int):
    import threading

    key = threading.get_ident()
    buf = _PINNED.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty((int(numel * 1.25) + 1024,), dtype=torch.float32, pin_memory=True)
        _PINNED[key] = buf
    return buf[:numel]


def literal_inputs(ts: TileStats, tile_ids: np.ndarray, fmt: str, quantizer, pinned: bool = False) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(x tiles, y tiles, ids) of the named tiles as zero-padded (k,32,32) float32 host arrays, `ids` naming the tile of every row
    (a permutation of tile_ids): what the reference's literal float32 tile score takes (tile_utils.py:46-57 on y = Quantizer.quantize
    of the tiles).  On the hip backend both come from one device call (mtq_knife_tiles_device: the tiles fetched and quantised where the
    tensor lives, 8 KB per tile home, in the order the device listed them) instead of tiles down, tiles up, K2, y down."""
    ids = np.asarray(tile_ids, dtype=np.int64).reshape(-1)
    if ts.backend == "hip" and ids.size and fmt in MIXED_TILE_FORMATS and ts.x2d.stride(-1) == 1:
        from .. import hip_backend as hb

        torch = __import__("torch")
        dev = ts.x2d.device
        # torch's current device / stream and HIP's current device are per THREAD and a new thread starts on device 0: the sweep calls
        # this from pool threads (sweep._literal_scores), so the tensor's own device is made current here — the launches then go to a
        # stream of the device the pointers live on, whatever rank / thread this is
        with torch.cuda.device(dev):
            flags = torch.zeros((ts.tiles,), dtype=torch.int8, device=dev)
            flags[torch.from_numpy(ids).to(dev)] = 1
            k = int(ids.size)
            lst = torch.empty((k + 1,), dtype=torch.int64, device=dev)
            out = torch.empty((2, k, TILE, TILE), dtype=torch.float32, device=dev)
            hb.knife_tiles_device(ts.x2d[None], flags, [fmt], k, lst, out)
            if pinned:
                # the calling thread's own pinned staging buffer (a pageable .cpu() of the sweep's 1.5 GB of tiles ran at a sixth of the link's
                # rate: 190 of the leg's 280 ms); the arrays alias it: valid until this thread's next call
                stage = _pinned_f32(torch, out.numel()).view(out.shape)
                stage.copy_(out, non_blocking=True)
                ids_home = lst[:k].cpu().numpy()        # synchronises the stream behind the staged copy
                torch.cuda.current_stream().synchronize()
                host = stage.numpy()
                return host[0], host[1], ids_home
            host = out.cpu().numpy()
            return host[0], host[1], lst[:k].cpu().numpy()
    xt = gather_tiles(ts, ids)
    yt = np.asarray(quantizer.quantize(xt.reshape(xt.shape[0] * TILE, TILE), fmt), dtype=np.float32).reshape(xt.shape[0], TILE, TILE)
    return xt, yt, ids


def columns_from_stats(ts: TileStats, assignment: np.ndarray) -> dict:
    """Tensor-level pcc / mae / atol of the reconstruction, from the float64 raw sums — summed on the device when the records
    live there and have not been brought to the host anyway (1 B/tile up, seven doubles back)."""
    from .. import hip_backend as hb

    amap = np.asarray(assignment, dtype=np.int8).reshape(-1)
    if ts.on_device and ts.host_stats is None:
        return hb.columns_from_stats_device(ts.stats_dev, ts.mask, amap, float(ts.numel))
    return hb.columns_from_stats(ts.stats, ts.mask, amap, float(ts.numel))


def tile_scores(ts: TileStats, metric: str) -> np.ndarray:
    """float64 [formats in mask order, tiles] per-tile scores (tile_utils.py:46-57 on the raw sums); computed where the
    records are."""
    from .. import hip_backend as hb

    if ts.on_device and ts.host_stats is None:
        return hb.tile_scores_device(ts.stats_dev, ts.mask, metric).cpu().numpy()
    return hb.tile_scores(ts.stats, ts.mask, metric)
