"""Streamed evaluation of a rank's shard behind `wq --backend hip` (replaces the per-tensor loop wq:655-706 for the two
search algorithms the north star names).

The shard's tensors are grouped by (2-D shape, storage type); a group goes through `pipeline.GreedyPipeline` /
`pipeline.ThresholdPipeline` in batches: batched K1 launches, records D2H overlapped with the next chunk's K1, the scans of
a chunk fanned out over host threads (greedy) or K4 on the device (threshold).  The `none` rows (wq:589-590) come out of the
same records (pure-format column sums): one K1 pass per tensor serves every row of its table.  y is not materialised
(K3 runs only for --literal-metrics, which takes the per-tensor path), PNGs are written after the GPU work.
Vectors and scalars (their 2-D image is zero-filled: tile_utils.py:96-102) keep the per-tensor path.
"""
from __future__ import annotations

import time

import numpy as np

from .compression_algorithms.tile_utils import MIXED_TILE_FORMATS
from .pipeline import GreedyPipeline, ThresholdPipeline, default_workers
from .quantization_formats import SUPPORTED_FORMATS

STREAMED_ALGOS = {"mixed-tile-greedy", "mixed-tile-threshold"}
MAX_BATCH_TILES = 1 << 21      # tiles per pipeline batch: bounds the record buffers (≈ 0.3 GB device + 0.2 GB pinned per slot)
K1_LAUNCH_TILES = 1 << 19      # tiles per K1 launch, the size bench.py launches (32 x 4096²)


def streamable(algo, formats, args) -> bool:
    """The streamed route serves `--backend hip` for the two searches when every mixed-tile format of the table is among the
    search's formats (so that the `none` rows come out of the search's own records) and the seed is usable."""
    if args.backend != "hip" or algo.name not in STREAMED_ALGOS or getattr(args, "literal_metrics", False):
        return False
    tile_formats = algo.tile_formats or [f for f in formats if f in MIXED_TILE_FORMATS]
    if not tile_formats or any(f in MIXED_TILE_FORMATS and f not in tile_formats for f in formats):
        return False
    return algo.name != "mixed-tile-greedy" or int(algo.seed) != 0


def group_key(index, name):
    """(rows, cols, storage) of a tensor's exact 2-D flatten (tile_utils.py:103-107), or None for vectors / scalars."""
    shape, dtype = index.shape_dtype(name)
    if len(shape) < 2 or int(np.prod(shape)) == 0:
        return None
    return (int(np.prod(shape[:-1])), int(shape[-1]), dtype)


class ShardEvaluator:
    """Rows of wq's tables for the tensors of one rank, group by group."""

    def __init__(self, index, algo, formats, device, row_w: int, bytes_per_elem: dict):
        self.index, self.algo, self.formats, self.device = index, algo, list(formats), device
        self.row_w, self.bytes_per_elem = row_w, bytes_per_elem
        self.tile_formats = algo.tile_formats or [f for f in formats if f in MIXED_TILE_FORMATS]
        self.pure = [f for f in formats if f in MIXED_TILE_FORMATS]
        self.compute_seconds = 0.0
        self.compute_tiles = 0
        self.k1_ms = 0.0
        self.k1_tiles = 0
        self._pipe = None

    def _pipeline(self, chunk: int):
        a = self.algo
        if a.name == "mixed-tile-greedy":
            if self._pipe is None:
                self._pipe = GreedyPipeline(self.tile_formats, a.metric, a.threshold, a.seed, chunk=chunk, workers=default_workers(), pure_formats=self.pure)
            self._pipe.chunk = chunk
        else:
            if self._pipe is None:
                self._pipe = ThresholdPipeline(self.tile_formats, a.metric, a.threshold, chunk=chunk, pure_formats=self.pure)
            self._pipe.chunk = chunk
        return self._pipe

    def run_group(self, key, items):
        """items: [(tensor idx, name)] of one (rows, cols, storage) group → {idx: (rows [R, ROW_W], assignment int8 map)}."""
        import torch

        rows_, cols_, _dtype = key
        tiles = -(-rows_ // 32) * -(-cols_ // 32)
        per_batch = max(1, MAX_BATCH_TILES // tiles)
        chunk = max(1, K1_LAUNCH_TILES // tiles)
        out = {}
        for b0 in range(0, len(items), per_batch):
            part = items[b0:b0 + per_batch]
            xs = [self.index.load(n, device=self.device) for _i, n in part]
            x3d = torch.stack([x.reshape(rows_, cols_) for x in xs])
            metas = []
            for x in xs:
                xf = x.float()
                ax = xf.abs()
                metas.append((float(xf.min()), float(xf.mean()), float(xf.max()), float(ax.mean()), float(ax.max())))
                del xf, ax
            del xs
            torch.cuda.synchronize()
            pipe = self._pipeline(min(chunk, len(part)))
            if getattr(pipe, "device_scan", False):
                pipe.chunk = len(part)   # scan on the device: one K1 launch and one scan launch (a wave per tensor) per batch
            t0 = time.perf_counter()
            results = pipe.run(x3d)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            self.compute_seconds += dt
            self.compute_tiles += tiles * len(part)
            if hasattr(pipe, "timing"):
                pipe.timing.drain()
                self.k1_ms, self.k1_tiles = pipe.timing.kernel_ms, pipe.timing.tiles
            numel = rows_ * cols_
            per = dt / len(part)   # TIME(s): the batch's wall time shared equally (wq:680-682 times one tensor's run())
            for (idx, _name), r, m in zip(part, results, metas):
                rows = []
                for f in self.formats:   # comp 0 = none (wq:589-590)
                    if f in MIXED_TILE_FORMATS:
                        pcc, mae, atol = r.pure[f]
                    else:                # fp0: y = 0 (metrics.py:14-15)
                        pcc, mae, atol = (1.0 if m[4] == 0.0 else 0.0), m[3], m[4]
                    rows.append([idx, 0, SUPPORTED_FORMATS.index(f), pcc, mae, atol, per, numel * self.bytes_per_elem[f] / 1e9, np.nan, -1, -1, -1, -1, *m[:3]])
                counts = [r.counts.get(k, 0) for k in MIXED_TILE_FORMATS]
                rows.append([idx, 1, -1, r.pcc, r.mae, r.atol, per, float(r.tile_bytes) / 1e9, r.tile_bytes, *counts, *m[:3]])
                out[idx] = (np.asarray(rows, dtype=np.float64).reshape(-1, self.row_w), r.assignment)
            del x3d
        return out

    def close(self):
        if self._pipe is not None and hasattr(self._pipe, "close"):
            self._pipe.close()
