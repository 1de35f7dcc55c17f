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

import concurrent.futures as cf
import time

import numpy as np

from .compression_algorithms.tile_utils import MIXED_TILE_FORMATS
from .pipeline import GreedyPipeline, ThresholdPipeline, cpu_budget, default_workers
from .quantization_formats import SUPPORTED_FORMATS
from .settings import settings

STREAMED_ALGOS = {"mixed-tile-greedy", "mixed-tile-threshold"}
MAX_BATCH_TILES = 1 << 21      # tiles per pipeline batch: bounds the record buffers (≈ 0.3 GB device + 0.2 GB pinned per slot)
K1_LAUNCH_TILES = 1 << 19      # tiles per K1 launch with the host scan (32 x 4096²: a chunk's records cross PCIe beside the next chunk's K1)
MAX_SLOTS = settings().wq_max_slots   # record slots of the streamed pipeline (K1 records + scan scratch of a batch: a few hundred MB each)
# Batches of tensors with more tiles than this take the pipeline's host route (records over PCIe, scans on host threads) beside the
# device-scanned rest of the window.  A device scan is one wave per tensor (≈ 10 ms at 57 344 tiles inside a window, a host core
# needs ≈ 2 ms), but the host route's launches and copies stalled the driver thread for 8 ms per window when tried on the
# Llama-3-8B shapes (24–29 ms per window against 19 ms all on the device): off by default.
DEVICE_SCAN_MAX_TILES = settings().wq_device_scan_max_tiles
MAX_WINDOW_BYTES = 48 << 30    # inputs resident in HBM at once (288 GB per MI355X): the loader fills a window, the pipeline then streams it


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
        self.load_seconds = 0.0      # tensors to HBM (synthetic presets: drawn on the CPU; safetensors: read + H2D) and their min / mean / max
        self.k1_ms = 0.0
        self.k1_tiles = 0
        self._pipe = None

    def _pipeline(self, chunk: int):   # one pipeline per evaluator: its record slots grow to the largest batch and are reused
        a = self.algo
        if a.name == "mixed-tile-greedy":
            if self._pipe is None:
                self._pipe = GreedyPipeline(self.tile_formats, a.metric, a.threshold, a.seed, chunk=chunk, workers=default_workers(), pure_formats=self.pure)   # three search streams (the pipeline's default): 8, one per batch in flight, measured 27 % slower in round 3
            self._pipe.chunk = chunk
        else:
            if self._pipe is None:
                self._pipe = ThresholdPipeline(self.tile_formats, a.metric, a.threshold, chunk=chunk, pure_formats=self.pure)
            self._pipe.chunk = chunk
        return self._pipe

    def _load_batch(self, rows_, cols_, part):
        """→ (x3d on the device, per-tensor (min, mean, max, mean|x|, max|x|)) — the loader; not part of the timed pipeline."""
        import torch

        # tensors are independent (a synthetic preset draws each from its own seeded CPU generator, a checkpoint reads each from its
        # file): loaded by a few threads — the draw and the file read release the interpreter lock — in the order of `part`
        names = [n for _i, n in part]
        workers = max(1, min(8, cpu_budget(), len(names)))
        if workers > 1:
            with cf.ThreadPoolExecutor(max_workers=workers) as pool:
                xs = list(pool.map(lambda n: self.index.load(n, device=self.device), names))
        else:
            xs = [self.index.load(n, device=self.device) for n in names]
        metas = []
        for x in xs:
            xf = x.float()
            ax = xf.abs()
            metas.append((float(xf.min()), float(xf.mean()), float(xf.max()), float(ax.mean()), float(ax.max())))
            del xf, ax
        return torch.stack([x.reshape(rows_, cols_) for x in xs]), metas

    def _rows(self, part, results, metas, numel, per):
        out = {}
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
        return out

    def run_groups(self, groups: dict):
        """groups: {(rows, cols, storage): [(tensor idx, name)]} → {idx: (rows [R, ROW_W], assignment int8 map)}.
        The tensors of a window (at most MAX_WINDOW_BYTES of inputs) are loaded to HBM first; the window's batches then go through
        the pipeline back to back — with the scan on the device, a batch's scans and downloads run beside the next batches' K1
        (GreedyPipeline.run_batches), across shape groups."""
        import torch

        batches = []   # (key, part, tiles)
        for key, items in groups.items():
            rows_, cols_, _dtype = key
            tiles = -(-rows_ // 32) * -(-cols_ // 32)
            per_batch = max(1, MAX_BATCH_TILES // tiles)
            for b0 in range(0, len(items), per_batch):
                batches.append((key, items[b0:b0 + per_batch], tiles))
        # the scan of a tensor is one wave whose time grows with the tensor's tiles: the longest scans are launched first so that
        # the shorter batches' K1 and scans run beside them
        batches.sort(key=lambda b: -b[2])
        out = {}
        w0 = 0
        while w0 < len(batches):
            w1, nbytes = w0, 0
            while w1 < len(batches) and (w1 == w0 or nbytes + self._batch_bytes(batches[w1]) <= MAX_WINDOW_BYTES):
                nbytes += self._batch_bytes(batches[w1])
                w1 += 1
            window = batches[w0:w1]
            t_load = time.perf_counter()
            loaded = [self._load_batch(k[0], k[1], part) for k, part, _t in window]
            torch.cuda.synchronize()
            self.load_seconds += time.perf_counter() - t_load
            greedy = self.algo.name == "mixed-tile-greedy"
            pipe = self._pipeline(1)
            # a generation-2 collection of the interpreter's heap (torch's module graph: tens of ms) would land inside the window's few
            # tens of milliseconds: collect now, and keep the collector off while the window runs
            import gc

            gc.collect()
            gc_was_on = gc.isenabled()
            gc.disable()
            if greedy and getattr(pipe, "device_scan", False):
                pipe.device_scan_max_tiles = DEVICE_SCAN_MAX_TILES
                pipe.host_chunk_tiles = K1_LAUNCH_TILES
                pipe.SLOTS = max(pipe.SLOTS, min(len(window), MAX_SLOTS))   # every batch of a short window in flight at once
                pipe.prepare([x for x, _m in loaded])     # record slots grown to the window's largest batch: allocations are not pipeline time
            try:
                t0 = time.perf_counter()
                if greedy and getattr(pipe, "device_scan", False):
                    pipe.chunk = 1 << 30                  # one K1 launch and one scan launch (a wave per tensor) per batch
                    all_results = pipe.run_batches([x for x, _m in loaded])
                elif hasattr(pipe, "run_batches"):          # threshold: every batch of the window enqueued first, decisions and column sums behind
                    pipe.chunk = 1 << 30                  # one K1 launch per batch
                    all_results = pipe.run_batches([x for x, _m in loaded])
                else:
                    all_results = []
                    for (key, part, tiles), (x3d, _m) in zip(window, loaded):
                        pipe.chunk = min(max(1, K1_LAUNCH_TILES // tiles), len(part))
                        all_results.append(pipe.run(x3d))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            finally:
                if gc_was_on:
                    gc.enable()
            n_tiles = sum(t * len(part) for _k, part, t in window)
            self.compute_seconds += dt
            self.compute_tiles += n_tiles
            if hasattr(pipe, "timing"):
                pipe.timing.drain()
                self.k1_ms, self.k1_tiles = pipe.timing.kernel_ms, pipe.timing.tiles
            for (key, part, tiles), (_x, metas), results in zip(window, loaded, all_results):
                per = dt * (tiles * len(part) / n_tiles) / len(part)   # TIME(s): the window's wall time shared by tiles (wq:680-682 times one tensor's run())
                out.update(self._rows(part, results, metas, key[0] * key[1], per))
            del loaded
            w0 = w1
        return out

    @staticmethod
    def _batch_bytes(batch) -> int:
        (rows_, cols_, dtype), part, _tiles = batch
        return rows_ * cols_ * (2 if dtype == "bf16" else 4) * len(part)

    def close(self):
        if self._pipe is not None and hasattr(self._pipe, "close"):
            self._pipe.close()
