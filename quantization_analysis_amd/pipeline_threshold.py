"""ThresholdPipeline — mixed-tile-threshold with the records on the device (see pipeline.py for the overview)."""
from __future__ import annotations

import concurrent.futures as cf
import contextlib
import ctypes
import os
import time
from dataclasses import replace

import numpy as np

from . import hip_backend as hb
from .compression_algorithms.tile_utils import MIXED_TILE_FORMATS, mixed_tile_total_bytes
from .settings import settings
from .pipeline_common import TensorResult, columns_from_sums_batch


class ThresholdPipeline:
    """mixed-tile-threshold over a (count, rows, cols) device tensor of equally shaped matrices, records never leaving
    the GPU: per chunk K1 (batched) → K4 on the device over all of the chunk's tiles at once → map + knife-edge flags to
    the host (2 B/tile) → the few knife-edge tiles re-scored with the literal float32 expression (as
    compression_algorithms.mixed_tile_threshold does for one tensor) → patched maps back up → column sums on the device.
    There is no host scan: the GPU is the pacing resource."""

    def __init__(self, tile_formats=None, metric: str = "pcc", threshold: float = 0.999, chunk: int = 16, band: float = 2e-6, pure_formats=()):
        import torch

        from .compression_algorithms.quantizer import Quantizer

        hb.require_gpu()
        self.torch = torch
        self.tile_formats = list(tile_formats or MIXED_TILE_FORMATS)
        self.pure_formats = [f for f in pure_formats if f in self.tile_formats]
        self.mask = hb.fmt_mask(self.tile_formats)
        self.metric, self.threshold, self.band = metric, float(threshold), float(band)
        self.chunk = int(chunk)
        self.quantizer = Quantizer("hip")
        self.knife_tiles = 0
        self._side = torch.cuda.Stream()      # the knife-edge tiles' fetch and way home, beside the main stream's K1
        self.knife_cap = settings().knife_cap   # knife-edge tiles per chunk fetched without a round trip (more: one extra trip)
        self._scratch_n = int(hb.lib().mtq_columns_scratch_doubles())
        self._pin = {}
        self._dev = {}

    def close(self) -> None:
        """Drains the device and releases the pinned mirrors now (see GreedyPipeline.close)."""
        self.torch.cuda.synchronize()
        self._pin.clear()
        self._dev.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _devbuf(self, name: str, numel: int, dtype, device):
        """Flat device storage that only grows, by name (a batch's maps, lists, records: an allocator call per batch and buffer was a
        sizeable part of the driver's time on small batches)."""
        t = self._dev.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype or t.device != device:
            t = self.torch.empty((int(numel * 1.25) + 16,), dtype=dtype, device=device)
            self._dev[name] = t
        return t[:numel]

    def _pinned(self, name: str, numel: int, dtype):
        """Flat pinned host storage that only grows (a pinned allocation costs milliseconds)."""
        t = self._pin.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = self.torch.zeros((int(numel * 1.25) + 16,), dtype=dtype, pin_memory=True)
            self._pin[name] = t
        return t[:numel]

    def _knife_tiles_device(self, xc, idx, tiles: int, tiles_w: int):
        """The knife-edge tiles' values and every format's reconstruction of them, on the device: `idx` holds flat (tensor·tiles + tile)
        indices (device int64; entries < 0 are padding and read tile 0).  ONE indexed gather, K2 once per format → float32
        (1 + formats, len(idx), 32, 32), plane 0 the tile itself (pads of ragged edge tiles zero, as the reference's padded view has them)."""
        torch = self.torch
        n, h, w = xc.shape
        dev = xc.device
        k = int(idx.numel())
        flat_idx = idx.clamp(min=0)
        j, t = flat_idx // tiles, flat_idx % tiles
        ar = torch.arange(32, device=dev)
        rows = (t // tiles_w)[:, None] * 32 + ar
        cols = (t % tiles_w)[:, None] * 32 + ar
        vals = xc[j[:, None, None], rows.clamp(max=h - 1)[:, :, None], cols.clamp(max=w - 1)[:, None, :]].float()
        if h % 32 or w % 32:
            inside = (rows < h)[:, :, None] & (cols < w)[:, None, :]
            vals = torch.where(inside, vals, torch.zeros((), dtype=torch.float32, device=dev))
        both = torch.empty((1 + len(self.tile_formats), k, 32, 32), dtype=torch.float32, device=dev)
        both[0] = vals
        flat = both[0].reshape(k * 32, 32)
        for i, f in enumerate(self.tile_formats):
            hb.quantize(flat, f, out=both[1 + i].reshape(k * 32, 32))
        return both

    def _decide(self, host_tiles: np.ndarray, old: np.ndarray, near: np.ndarray) -> np.ndarray:
        """The reference's literal float32 tile score (tile_utils.py:46-57, here tile_utils.pearson_corr_tiles: the per-tile call's bits
        at a third of its cost) for exactly the (tile, format) pairs inside the band, and the map values that follow
        (mixed_tile_threshold.decide_knife_tiles, reference :117-123)."""
        from .compression_algorithms.mixed_tile_threshold import decide_knife_tiles
        from .compression_algorithms.tile_utils import tile_metrics

        fmts = list(self.tile_formats)
        x_tiles = host_tiles[0]

        def literal_scores(fmt: str, sel: np.ndarray) -> np.ndarray:
            return tile_metrics(x_tiles[sel], host_tiles[1 + fmts.index(fmt)][sel], self.metric)

        return decide_knife_tiles(old, near.astype(np.uint8), self.tile_formats, self.metric, self.threshold, literal_scores)

    def run(self, x3d, numel: int | None = None) -> list[TensorResult]:
        """One batch (count, rows, cols): enqueue, decide, wrap."""
        st = self.enqueue(x3d, numel)
        self.decide(st)
        self.torch.cuda.current_stream().synchronize()                                         # one wait for all chunks' sums
        return self.wrap(st)

    def run_batches(self, batches) -> list[list[TensorResult]]:
        """Batches of ANY shapes and storage types — a model's shape groups, vectors as (ceil(n/32), 32) matrices with their element
        count — as `(x3d, numel | None)` pairs or bare tensors, in three sweeps instead of a blocking run() per batch (the reference's
        loop is per tensor, wq:655-706): every batch's K1, K4 and knife-edge listing are ENQUEUED first, back to back on the main
        stream with the listings and their tiles' way home on the side stream; then, batch by batch, the host takes the maps and the
        listed tiles (events: batch k's literal float32 decisions run while the GPU works on the batches behind it) and launches the
        batch's column sums; one wait at the end, then the results are wrapped.  Every batch has its own pinned mirrors (slot = its
        position).  Same results as run() per batch (tests/test_hip_kernels.py::test_threshold_run_batches_equals_run)."""
        items = [(b, None) if not isinstance(b, (tuple, list)) else (b[0], b[1]) for b in batches]
        # Batches the direct kernel serves anyway (float32 storage; bf16 the LDS-staged kernel does not take) travel as RAGGED groups: one
        # launch per stage for up to hb.RAGGED_MAX matrices of any shapes (mtq_threshold_enqueue_ragged) instead of a launch chain per
        # batch — the chains, not the arithmetic, were a model's small tensors' time.  The largest group first: its K1 runs while the host
        # enqueues the others (about 0.1 ms of driver time each).
        plan, ragged = [], {}
        for i, (x, n) in enumerate(items):
            if settings().threshold_ragged and self._raggable(x):
                ragged.setdefault((x.dtype, x.device), []).append(i)
            else:
                plan.append(("uniform", [i]))
        for members in ragged.values():
            group, used = [], 0
            for i in members:
                if group and used + items[i][0].shape[0] > hb.RAGGED_MAX:
                    plan.append(("ragged", group))
                    group, used = [], 0
                group.append(i)
                used += items[i][0].shape[0]
            plan.append(("ragged", group))
        weight = lambda g: sum(items[i][0].numel() for i in g[1])
        plan.sort(key=weight, reverse=True)
        states = []
        for slot, (kind, members) in enumerate(plan):
            if kind == "uniform":
                states.append(self.enqueue(items[members[0]][0], items[members[0]][1], slot=slot, overlap=True))
            else:
                states.append(self.enqueue_ragged([items[i] for i in members], slot=slot, overlap=len(plan) > 1))
        for st in states:
            self.decide(st)
        self.torch.cuda.current_stream().synchronize()
        out: list = [None] * len(items)
        for (kind, members), st in zip(plan, states):
            res = self.wrap(st)
            if kind == "uniform":
                out[members[0]] = res
            else:
                at = 0
                for i in members:
                    c = items[i][0].shape[0]
                    out[i] = [replace(r, index=j) for j, r in enumerate(res[at:at + c])]
                    at += c
        return out

    def _raggable(self, x3d) -> bool:
        """A batch of few matrices that K1's direct kernel serves anyway (hb.RAGGED_MAX of them fit a launch; the LDS-staged bf16 kernel
        is the faster one where it applies: csrc/mtq_kernels.hip tile_stats_launch)."""
        count, rows, cols = x3d.shape
        if count > 4 or x3d.stride(2) != 1:
            return False
        staged = (x3d.dtype == self.torch.bfloat16 and rows % 32 == 0 and cols % 128 == 0 and x3d.data_ptr() % 16 == 0
                  and (x3d.stride(1) * 2) % 16 == 0 and (x3d.stride(0) * 2) % 16 == 0 and (self.mask & 0xE))
        return not staged

    def enqueue_ragged(self, group, slot: int = 0, overlap: bool = False) -> dict:
        """GPU half of a RAGGED group — `group` = (x3d, numel | None) batches of one storage type, every matrix of them an entry of the
        launch's table: K1 → K4 → maps and masks home → knife-edge listing, one launch each for the whole group (mtq_threshold_enqueue_ragged).
        The state reads like enqueue()'s with one chunk whose tiles are numbered through the group's matrices."""
        torch = self.torch
        mats, numels = [], []
        for x3d, numel in group:
            for j in range(x3d.shape[0]):
                mats.append(x3d[j])
                numels.append(x3d.shape[1] * x3d.shape[2] if numel is None else int(numel))
        arr, code, tiles_per = hb.ragged_matrices(mats)
        dev = mats[0].device
        T, n = sum(tiles_per), len(mats)
        identity = mats[0].dtype == torch.bfloat16 and (self.mask & 1) and (self.mask & 0xE)
        k1_mask = self.mask & 0xE if identity else self.mask
        dec_mask = k1_mask | hb.MASK_BF16_IDENTITY if identity else self.mask
        P = 1 + len(self.pure_formats)
        planes = 1 + len(self.tile_formats)
        single = not overlap
        cap = min(self.knife_cap, T)
        both_host = self._pinned(f"both{slot}", 2 * T, torch.int8).view(2, T)
        idx_host = self._pinned(f"idx{slot}", cap + 1, torch.int64).view(1, cap + 1)
        knife_host = self._pinned(f"knife{slot}", planes * cap * 1024, torch.float32).view(1, planes, cap, 32, 32)
        sums_host = self._pinned(f"sums{slot}", P * n * 11, torch.float64).view(P, n, 11)
        both_dev = self._devbuf(f"both{slot}", 2 * T, torch.int8, dev).view(2, T)
        idx_dev = self._devbuf(f"idx{slot}", cap + 1, torch.int64, dev).view(1, cap + 1)
        knife_dev = self._devbuf(f"knife{slot}", planes * cap * 1024, torch.float32, dev).view(1, planes, cap, 32, 32)
        rec = hb.record_doubles(k1_mask)
        recs = self._devbuf(f"recs{slot}", T * rec, torch.float64, dev).view(T, rec)
        fm = (ctypes.c_int * len(self.tile_formats))(*[MIXED_TILE_FORMATS.index(f) for f in self.tile_formats])
        scratch = self._devbuf(f"colscr{slot}_0", P * n * self._scratch_n, torch.float64, dev)
        hb.check(hb.lib().mtq_threshold_enqueue_ragged(
            arr, n, code, k1_mask, dec_mask, fm, len(self.tile_formats), hb.METRIC_CODE[self.metric], self.threshold, self.band, recs.data_ptr(),
            both_dev.data_ptr(), both_host.data_ptr(), cap, idx_dev[0].data_ptr(), knife_dev[0].data_ptr(), idx_host[0].data_ptr(),
            scratch.data_ptr(), sums_host.data_ptr(), torch.cuda.current_stream().cuda_stream, None if single else self._side.cuda_stream))
        landed = torch.cuda.Event()
        landed.record(torch.cuda.current_stream() if single else self._side)
        first = np.concatenate([[0], np.cumsum(tiles_per)]).astype(np.int64)
        return {"x": None, "ragged": {"mats": mats, "arr": arr, "tiles_per": tiles_per, "first": first, "numels": numels}, "tiles_sent": False, "slot": slot, "speculated": True,
                "numel": None, "hw": None, "tiles": T, "dec_mask": dec_mask, "cap": cap, "single": single, "planes": planes,
                "launched": [(0, n, recs, slice(0, T), landed, landed)], "both_host": both_host, "idx_host": idx_host, "knife_host": knife_host,
                "sums_host": sums_host, "both_dev": both_dev, "idx_dev": idx_dev, "knife_dev": knife_dev, "maps": np.empty((T,), dtype=np.int8)}

    def enqueue(self, x3d, numel: int | None = None, slot: int = 0, overlap: bool = False) -> dict:
        """GPU half of a batch, nothing waited for: per chunk K1 → K4 → map and knife-edge mask home → (side stream) the chunk's knife-edge
        tiles listed, fetched, quantised in every format and sent home.  `slot` names the batch's pinned mirrors (batches in flight at
        the same time need different slots); overlap: other batches follow before this one is decided (the listing then always takes the
        side stream)."""
        torch = self.torch
        count, rows, cols = x3d.shape
        dev = x3d.device
        th, tw = hb.tiles_hw(rows, cols)
        tiles, numel = th * tw, (rows * cols if numel is None else int(numel))
        identity = x3d.dtype == torch.bfloat16 and (self.mask & 1) and (self.mask & 0xE)
        k1_mask = self.mask & 0xE if identity else self.mask
        dec_mask = k1_mask | hb.MASK_BF16_IDENTITY if identity else self.mask
        P = 1 + len(self.pure_formats)
        planes = 1 + len(self.tile_formats)
        chunks = [(first, min(self.chunk, count - first)) for first in range(0, count, self.chunk)]
        single = len(chunks) == 1 and not overlap
        cap = min(self.knife_cap, self.chunk * tiles)
        # pinned mirrors of what comes back (2 B per tile, each chunk's knife-edge list and tiles, then 7 sums per tensor):
        # kernels store into them (hb.device_copy) and the driver waits on events, it never blocks in a pageable copy with the GPU idle
        both_host = self._pinned(f"both{slot}", 2 * count * tiles, torch.int8).view(2, count * tiles)
        idx_host = self._pinned(f"idx{slot}", len(chunks) * (cap + 1), torch.int64).view(len(chunks), cap + 1)
        knife_host = self._pinned(f"knife{slot}", len(chunks) * planes * cap * 1024, torch.float32).view(len(chunks), planes, cap, 32, 32)
        sums_host = self._pinned(f"sums{slot}", P * count * 11, torch.float64).view(P, count, 11)
        both_dev = self._devbuf(f"both{slot}", 2 * count * tiles, torch.int8, dev).view(2, count * tiles)   # row 0 the maps, row 1 the knife-edge masks
        idx_dev = self._devbuf(f"idx{slot}", len(chunks) * (cap + 1), torch.int64, dev).view(len(chunks), cap + 1)
        knife_dev = self._devbuf(f"knife{slot}", len(chunks) * planes * cap * 1024, torch.float32, dev).view(len(chunks), planes, cap, 32, 32)
        rec = hb.record_doubles(k1_mask)
        recs_all = self._devbuf(f"recs{slot}", count * tiles * rec, torch.float64, dev).view(count, tiles, rec)
        fm = (ctypes.c_int * len(self.tile_formats))(*[MIXED_TILE_FORMATS.index(f) for f in self.tile_formats])
        main_ptr = torch.cuda.current_stream().cuda_stream
        speculated = len(chunks) == 1   # one chunk: the call sums the columns under K4's maps at once (final unless a knife-edge tile is listed)
        launched = []  # (first, n, records, chunk's tile range, map-landed event, knife-tiles-landed event)
        for c, (first, n) in enumerate(chunks):
            # ONE call per chunk (mtq_threshold_enqueue): K1 → K4 → map and knife-edge masks into the pinned mirror on the main stream; then,
            # behind an event, the chunk's knife-edge tiles found, fetched and quantised in every format (mtq_knife_tiles_device) and their list
            # sent home — on the side stream, beside the next chunk's K1, unless the batch is a single chunk with nothing behind it (a
            # cross-stream wait is a barrier packet).  As a dozen Python-level launches per chunk this cost the driver 10–20 us each: the
            # seven DeepSeek layer-0 tensors were launch-bound at 1.7 ms for 0.45 ms of K1.  The listed tiles themselves follow on demand.
            xs = x3d[first:first + n]
            recs = recs_all[first:first + n]
            part = slice(first * tiles, (first + n) * tiles)
            if len(chunks) > 1:
                # several chunks: a chunk's maps and masks are not adjacent in the batch's [2, T] arrays — the calls one by one
                hb.tile_stats_batched(xs, k1_mask, out=recs)
                hb.threshold_assign_device_raw(recs.view(n * tiles, -1), dec_mask, self.tile_formats, self.metric, self.threshold, self.band,
                                               out=(both_dev[0, part], both_dev[1, part]))
                hb.device_copy(both_host[:, part], both_dev[:, part])
                decided = torch.cuda.Event()
                decided.record()
                with (contextlib.nullcontext() if single else torch.cuda.stream(self._side)):
                    if not single:
                        self._side.wait_event(decided)
                    hb.knife_tiles_device(xs, both_dev[1, part], self.tile_formats, cap, idx_dev[c], knife_dev[c])
                    hb.device_copy(idx_host[c], idx_dev[c])
                    landed = torch.cuda.Event()
                    landed.record()
            else:
                scratch = self._devbuf(f"colscr{slot}_{c}", P * n * self._scratch_n, torch.float64, dev)
                hb.check(hb.lib().mtq_threshold_enqueue(
                    xs.data_ptr(), hb._dtype_code(xs), n, xs.stride(0) if n > 1 else rows * xs.stride(1), rows, cols, xs.stride(1), k1_mask, dec_mask, fm,
                    len(self.tile_formats), hb.METRIC_CODE[self.metric], self.threshold, self.band, recs.data_ptr(), both_dev.data_ptr(), both_host.data_ptr(),
                    cap, idx_dev[c].data_ptr(), knife_dev[c].data_ptr(), idx_host[c].data_ptr(), scratch.data_ptr(), sums_host.data_ptr(), main_ptr,
                    None if single else self._side.cuda_stream))
                landed = torch.cuda.Event()
                landed.record(torch.cuda.current_stream() if single else self._side)
                decided = landed      # the side stream runs behind the masks' copy: one event covers both
            launched.append((first, n, recs, part, decided, landed))
        return {"x": x3d, "tiles_sent": False, "slot": slot, "speculated": speculated, "numel": numel, "hw": (th, tw), "tiles": tiles, "dec_mask": dec_mask, "cap": cap, "single": single, "planes": planes,
                "launched": launched, "both_host": both_host, "idx_host": idx_host, "knife_host": knife_host, "sums_host": sums_host,
                "both_dev": both_dev, "idx_dev": idx_dev, "knife_dev": knife_dev, "maps": np.empty((count, tiles), dtype=np.int8)}

    def decide(self, st: dict) -> None:
        """Host half, part 1: per chunk the map and the listed tiles (events), the literal float32 decisions of the knife-edge tiles
        (mixed_tile_threshold.py:117-123 on the reference's own score), the patched map back up, the column sums launched and sent home."""
        torch = self.torch
        x3d, tiles, cap, planes, dec_mask = st["x"], st["tiles"], st["cap"], st["planes"], st["dec_mask"]
        rg = st.get("ragged")
        dev = st["both_dev"].device
        tw = None if rg else st["hw"][1]
        both_host, idx_host, knife_host, sums_host = st["both_host"], st["idx_host"], st["knife_host"], st["sums_host"]
        both_dev, idx_dev, knife_dev, maps_all = st["both_dev"], st["idx_dev"], st["knife_dev"], st["maps"]
        scratch_n = self._scratch_n
        P = 1 + len(self.pure_formats)
        for c, (first, n, recs, part, decided, landed) in enumerate(st["launched"]):
            landed.synchronize()      # the chunk's list is home ...
            decided.synchronize()     # ... and its map (recorded earlier: passed by now unless the list took the side stream)
            if rg:
                maps_all[:] = both_host[0, part].numpy()
            else:
                maps_all[first:first + n] = both_host[0, part].numpy().reshape(n, tiles)       # the mirror is reused by the slot's next batch
            k = int(idx_host[c, cap])
            final = st["speculated"] and k == 0    # the sums the enqueue call launched under K4's maps stand
            if k:
                near = both_host[1, part].numpy()
                if k <= cap:
                    flat = idx_host[c, :k].numpy().copy()                                      # the list is in no particular order: ids travel with their tiles
                    if not st["tiles_sent"]:                                                   # the tiles were not sent with the list
                        home = knife_host[c].reshape(-1)[:planes * k * 1024].view(planes, k, 32, 32)
                        hb.device_copy(home, knife_dev[c, :, :k].contiguous())
                        torch.cuda.current_stream().synchronize()
                        host_tiles = home.numpy()
                    else:
                        host_tiles = knife_host[c, :, :k].numpy()
                    where = idx_dev[c, :k]
                elif rg:                                                                       # more than the list holds, ragged: matrix by matrix
                    flat = np.flatnonzero(near).astype(np.int64)
                    where = torch.from_numpy(flat).to(dev)
                    owner = np.searchsorted(rg["first"], flat, side="right") - 1
                    parts = []
                    for j in np.unique(owner):
                        m = rg["mats"][j]
                        local = torch.from_numpy(flat[owner == j] - rg["first"][j]).to(dev)
                        parts.append(self._knife_tiles_device(m[None], local, rg["tiles_per"][j], hb.tiles_hw(*m.shape)[1]).cpu().numpy())
                    host_tiles = np.concatenate(parts, axis=1)
                else:                                                                          # more than the list holds: the same steps, one more trip
                    flat = np.flatnonzero(near).astype(np.int64)
                    where = torch.from_numpy(flat).to(dev)
                    host_tiles = self._knife_tiles_device(x3d[first:first + n], where, tiles, tw).cpu().numpy()
                mchunk = maps_all if rg else maps_all[first:first + n].reshape(-1)
                new = self._decide(host_tiles, mchunk[flat], near[flat])
                mchunk[flat] = new
                both_dev[0, part].index_put_((where,), torch.from_numpy(np.ascontiguousarray(new, dtype=np.int8)).to(dev, non_blocking=True))
                self.knife_tiles += k
            dmaps = both_dev[0, part]
            scratch = self._devbuf(f"colscr{st['slot']}_{c}", P * n * scratch_n, torch.float64, dev).view(P, n, scratch_n)
            sp = hb._stream_ptr()
            if rg:
                per = (ctypes.c_int64 * n)(*rg["tiles_per"])

                def columns(maps_dev, q):
                    hb.check(hb.lib().mtq_threshold_columns_ragged(recs.data_ptr(), per, n, dec_mask, maps_dev.data_ptr(), scratch[q].data_ptr(), sums_host[q].data_ptr(), sp))
            else:
                def columns(maps_dev, q):
                    hb.check(hb.lib().mtq_threshold_columns(recs.data_ptr(), n, tiles, dec_mask, maps_dev.data_ptr(), scratch[q].data_ptr(), sums_host[q, first:first + n].data_ptr(), sp))
            if not final:
                columns(dmaps, 0)
            for q, f in enumerate(self.pure_formats):   # wq's `none` rows from the same records
                columns(torch.full((dmaps.numel(),), MIXED_TILE_FORMATS.index(f), dtype=torch.int8, device=dev), 1 + q)

    def wrap(self, st: dict) -> list[TensorResult]:
        """Host half, part 2 (behind a wait for the main stream): counts = the column kernel's histogram of the final device map (the
        reference's np.bincount, mixed_tile_threshold.py:133-135 — on the host it cost more than the whole K4), the columns from the seven sums."""
        rg = st.get("ragged")
        maps_all = st["maps"]
        k = {"pcc": 0, "mae": 1, "atol": 2}[self.metric]
        sums = st["sums_host"].numpy()
        if rg:   # one tensor per matrix of the group, each with its own element count and tile grid
            count = len(rg["mats"])
            numel = np.asarray(rg["numels"], dtype=np.float64)
            grids = [hb.tiles_hw(*m.shape) for m in rg["mats"]]
            maps_of = [maps_all[rg["first"][j]:rg["first"][j + 1]] for j in range(count)]
        else:
            count = maps_all.shape[0]
            numel = float(st["numel"])
            grids = [st["hw"]] * count
            maps_of = [maps_all[j] for j in range(count)]
        cols = columns_from_sums_batch(sums[0], numel)
        pure_cols = [columns_from_sums_batch(sums[1 + i], numel) for i in range(len(self.pure_formats))]
        results: list[TensorResult] = []
        for j in range(count):
            counts = {f: int(sums[0][j, 7 + i]) for i, f in enumerate(MIXED_TILE_FORMATS)}   # the column kernel's histogram of the final map
            pure = {f: tuple(float(v) for v in pure_cols[i][j]) for i, f in enumerate(self.pure_formats)} or None
            results.append(TensorResult(j, maps_of[j].reshape(grids[j]), counts, mixed_tile_total_bytes(counts), float(cols[j, 0]),
                                        float(cols[j, 1]), float(cols[j, 2]), float(cols[j, k]), pure))
        st["x"] = None
        if rg:
            rg["mats"] = None
        return results
