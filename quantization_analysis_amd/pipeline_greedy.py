"""GreedyPipeline — mixed-tile-greedy over batches of equally shaped tensors, one rank = one GPU (see pipeline.py for the overview)."""
from __future__ import annotations

import concurrent.futures as cf
import contextlib
import os
import time
from dataclasses import dataclass, field

import numpy as np

from . import hip_backend as hb
from .compression_algorithms.tile_utils import MIXED_TILE_FORMATS, mixed_tile_total_bytes
from .settings import settings
from .pipeline_common import KernelTiming, TensorResult, _scan_chunk, _sleep_until, _when_landed, columns_from_sums_batch  # noqa: F401


class GreedyPipeline:
    """mixed-tile-greedy over a (count, rows, cols) device tensor of equally shaped bf16/fp32 matrices."""

    SLOTS = None   # record slots = batches in flight: one on the GPU, one queued behind it, the others being searched (see run_steps).
                                                         # Round 3: a batch's search is a chain of launches (orders, phase 1, listed K1, phase 2, column sums) that takes 5–7 ms
                                                         # beside K1 launches of ~2 ms, so three slots made the K1 stream wait for a slot (2.65 against 2.47 ms per step at four)

    def __init__(self, tile_formats=None, metric: str = "pcc", threshold: float = 0.999, seed: int = 123,
                 chunk: int = 8, workers: int = 8, pure_formats=(), scan: str = "auto", scan_streams: int | None = None):
        """scan: "device" — the sequential scan runs on the GPU where K1 wrote the records (csrc/mtq_scan.hip: only maps, counts and
        seven sums per tensor cross PCIe, the host does not scan); "host" — records over PCIe, scans on host threads; "auto" —
        "device" where mtq_greedy_scan_device serves the search (distinct formats), else "host".  MTQ_DEVICE_SCAN=0
        forces "host"."""
        import torch

        hb.require_gpu()
        self.torch = torch
        self.tile_formats = list(tile_formats or MIXED_TILE_FORMATS)
        self.mask = hb.fmt_mask(self.tile_formats)
        # whole-tensor columns of these formats on their own (wq's `none` rows) come out of the same records: no second K1 pass
        self.pure_formats = [f for f in pure_formats if f in self.tile_formats]
        self.metric, self.threshold, self.seed = metric, float(threshold), int(seed)
        if self.seed == 0:
            raise ValueError("seed 0 means 'draw a random seed' in the reference; pass a non-zero seed")
        self.chunk = int(chunk)
        self.workers = int(workers)
        self.pool = cf.ThreadPoolExecutor(max_workers=settings().chunk_tasks)  # chunk-level tasks; the fan-out over tensors happens inside the C call (shared scan pool)
        self.stream = torch.cuda.Stream()        # K1 launches
        self.copy_stream = torch.cuda.Stream()   # records D2H, overlapped with the next chunk's K1
        self.col_stream = torch.cuda.Stream()    # maps up / column sums / sums down of a finished batch (must not queue behind the next batch's copies)
        self.timing = KernelTiming()
        self.SLOTS = type(self).SLOTS or settings().pipe_slots   # per instance: callers that stream many shape groups raise it (streamed.py)
        if scan not in ("auto", "host", "device"):
            raise ValueError("scan must be 'auto', 'host' or 'device'")
        can = hb.device_scan_supported(self.tile_formats, self.metric, 1)
        if scan == "device" and not can:
            raise ValueError("the device scan needs distinct formats")
        self.device_scan = can and scan != "host" and settings().device_scan
        # A device scan is ONE wave per tensor (≈ 0.14–0.19 µs per tile, three passes): right when a batch holds many tensors, a long
        # pole when a batch is a few very large ones — a host core scans a tile in ≈ 0.025 µs.  Callers whose batches are latency-bound
        # (streamed.py: a model's shape groups) lower this limit; batches above it take the host route, the others the device route.
        self.device_scan_max_tiles = settings().device_scan_max_tiles or hb.SCAN_DEVICE_MAX_TILES
        self.host_chunk_tiles = None             # host route: tiles per K1 launch / records copy / scan task (None: self.chunk tensors)
        # device scans + column sums: behind their chunk's K1, beside the next chunks' K1.  A scan is one wave per tensor for a few
        # milliseconds (latency-bound), so consecutive chunks' scans must overlap each other: a ring of streams
        # How many: as many as scans of consecutive batches overlap.  Steps of one shape (bench.py: a 2.0 ms scan launch per 2.5 ms step)
        # overlap two — three streams; with eight, K1's launch measured 2.5 % longer and the step 3 % (806 against 829 M tiles/s:
        # more hardware queues in play).  A model's shape groups (streamed.py) keep every batch of a window in flight and asked for eight in round 2.
        # Round 3 (a step's chain is three search launches and the listed K1 now, four record slots): three streams again — with four,
        # K1's launch is 3 % longer and the 20-step figure 3 % lower (916–927 against 947–956 M tiles/s, tools/r3_env_ab.sh); two starve.
        n_scan = settings().scan_streams if settings().scan_streams is not None else (3 if scan_streams is None else scan_streams)
        self.scan_streams = [torch.cuda.Stream(priority=settings().scan_priority)
                             for _ in range(max(1, n_scan))]   # priority -1: ahead of K1's blocks when a slot opens
        self._scan_rr = 0
        # Round 3.  shared orders: the tensors of a batch are searched with one seed, so the base pass's draws and the permutations of
        # passes 1 and 2 are computed once per launch (mtq_scan_orders_device, beside K1) and a helper wave per tensor gathers the deltas ahead
        # of the visiting wave.  lazy: K1 leaves out the last format of the list (and Σ|x−y|, max|x−y| of the one before it); the search
        # stops before its last pass, the left-out statistics are evaluated for that pass's candidates only (mtq_tile_stats_listed: the
        # tiles that accepted every earlier format — 15 % at pcc >= 0.999), and the last pass follows.  Same maps, same columns.
        self.shared_orders = settings().shared_orders
        self.lazy = (settings().lazy and self.metric == "pcc" and len(self.tile_formats) >= 3 and self.tile_formats[0] == "bf16"
                     and len(set(self.tile_formats)) == len(self.tile_formats) and not self.pure_formats)
        self.listed_tiles = 0                    # tiles the lazy route evaluated late (diagnostics)
        # The lazy route pays when few tiles reach the last pass: K1 <3,1> costs 469 instructions per tile and the listed kernel 432 per
        # listed tile, against 652 for the whole record — break-even at 42 % listed, less the phases' launches.  A batch that listed more
        # switches the route off for tensors of its tile count (the batches already in flight finish as they were enqueued).
        self.lazy_max_listed = settings().lazy_max_listed
        self.lazy_off = {}                       # tiles per tensor -> the listed fraction that switched the lazy route off
        self.host_fallbacks = 0                  # tensors the device scan handed back (zero denominator)
        self.host_seconds = {"enqueue": 0.0, "wait": 0.0, "wrap": 0.0}   # driver-thread time: launching, waiting for results, wrapping them
        self._devbufs = {}
        self._orders_cache = {}   # (device, seed, tiles, orders) -> [device buffer of mtq_scan_orders_device, event until it is complete]
        self._orders_retired = []
        self._bufs = {}
        self._colbufs = {}
        self._unresolved = []  # batches whose device-side columns are in flight (oldest first)
        self._open = []        # enqueued, not yet finished (oldest first)
        self._next_slot = 0

    def _layout(self, x3d):
        """→ (K1 mask, host mask, slim?) for a batch.  bf16 storage: the bf16 candidate is the identity, its record slot would be
        [Σx, Σx², Σx², 0, 0]; K1 then writes the BFP slots only and the host scan synthesises format 0
        (MTQ_MASK_BF16_IDENTITY).  pcc metric: Σ|d| and max feed no decision (bar the zero-variance case), so a slim copy of the
        records (3 doubles per format, MTQ_MASK_SLIM) crosses PCIe and the result's mae / atol come from the device."""
        torch = self.torch
        identity = x3d.dtype == torch.bfloat16 and (self.mask & 1) and (self.mask & 0xE) and settings().identity_records
        k1_mask = self.mask & 0xE if identity else self.mask
        host_mask = k1_mask | hb.MASK_BF16_IDENTITY if identity else self.mask
        slim = self.metric == "pcc" and settings().slim_records
        return k1_mask, host_mask | (hb.MASK_SLIM if slim else 0), slim

    def lazy_plan(self, x3d):
        """→ None, or (layout mask, formats K1 evaluates in full, format K1 evaluates without Σ|x−y| / max|x−y|, format left to the listed
        kernel) when a batch shaped like x3d takes the lazy route (see __init__): bf16 storage in whole 32x128 units — what the exact-integer
        kernel serves — and the search on the device."""
        torch = self.torch
        count, rows, cols = x3d.shape
        th, tw = hb.tiles_hw(rows, cols)
        k1_mask = self._layout(x3d)[0]
        if not (self.lazy and self._use_device_scan(th * tw) and x3d.dtype == torch.bfloat16 and (k1_mask & 1) == 0 and rows % 32 == 0 and cols % 128 == 0):
            return None
        if th * tw in self.lazy_off:
            return None
        bit = lambda f: 1 << MIXED_TILE_FORMATS.index(f)
        last_bit, prev_bit = bit(self.tile_formats[-1]), bit(self.tile_formats[-2])
        return k1_mask, k1_mask & ~last_bit & ~prev_bit, prev_bit, last_bit

    def launch_k1(self, x3d, out=None):
        """The K1 launch a batch's route issues (the whole record, or the lazy route's partial one) on the current stream."""
        plan = self.lazy_plan(x3d)
        if plan is None:
            return hb.tile_stats_batched(x3d, self._layout(x3d)[0], out=out)
        return hb.tile_stats_partial(x3d, plan[0], plan[1], plan[2], out=out)

    def _buffers(self, slot: int, count: int, tiles: int, rec: int, rec_host: int, device):
        """Records of a whole batch: device buffer (full records, what K1 writes), device staging buffer of what crosses PCIe
        (the same buffer unless the records are slimmed) and its pinned host mirror (scans read the pinned memory in place).
        SLOTS of them rotate so that a batch can be queued and another on the GPU / the PCIe link while an earlier one is still
        being scanned."""
        key = (count, tiles, rec, rec_host, str(device))
        if self._bufs.get(slot, (None,))[0] != key:
            torch = self.torch
            dev = torch.empty((count, tiles, rec), dtype=torch.float64, device=device)
            stage = dev if rec_host == rec else torch.empty((count, tiles, rec_host), dtype=torch.float64, device=device)
            host = torch.empty((count, tiles, rec_host), dtype=torch.float64, pin_memory=True)
            self._bufs[slot] = (key, dev, stage, host, host.numpy())
        return self._bufs[slot][1:]

    def reserve(self, x3d) -> None:
        """Allocate both record slots (device + pinned host) for batches shaped like x3d and start the scan threads, so that
        no allocation or thread creation lands in a timed region."""
        torch = self.torch
        count, rows, cols = x3d.shape
        th, tw = hb.tiles_hw(rows, cols)
        k1_mask, host_mask, slim = self._layout(x3d)
        if self._use_device_scan(th * tw):
            self._warm_device_scan(x3d.device)
            for slot in range(self.SLOTS):
                b = self._device_buffers(slot, count, th * tw, hb.record_doubles(k1_mask), x3d.device)
                b["dev"].zero_()
                b["maps_host"].copy_(b["maps_dev"], non_blocking=True)
                b["sums_host"].copy_(b["sums_dev"][:, :, :7], non_blocking=True)
            torch.cuda.synchronize()
            return
        rec_host = hb.record_doubles(host_mask)
        for slot in range(self.SLOTS):
            dev, stage, host, _np = self._buffers(slot, count, th * tw, hb.record_doubles(k1_mask), rec_host, x3d.device)
            dev.zero_()
            stage.zero_()
            host.copy_(stage, non_blocking=True)   # first DMA into the pinned pages (mappings are set up lazily)
        torch.cuda.synchronize()
        hb.greedy_run_batch(np.zeros((self.workers, 1, hb.record_doubles(0xF))), 0xF, ["bf16"], self.metric, self.threshold, 1024.0,
                            [1] * self.workers, self.workers)

    def enqueue(self, x3d, seeds=None, numel: int | None = None) -> dict:
        """GPU half of a batch, non-blocking: per chunk K1 on the launch stream (plus the slim copy of its records) and the
        records' D2H on the copy stream.  At most SLOTS batches may be enqueued and not yet finished."""
        torch = self.torch
        if len(self._open) >= self.SLOTS:
            raise RuntimeError("finish() an enqueued batch before enqueuing another one: every record slot is in use")
        count, rows, cols = x3d.shape
        th, tw = hb.tiles_hw(rows, cols)
        k1_mask, host_mask, slim = self._layout(x3d)
        tiles = th * tw
        slot = self._next_slot
        self._next_slot = (slot + 1) % self.SLOTS
        if self._use_device_scan(tiles):
            return self._enqueue_device(x3d, seeds, numel, slot, k1_mask, host_mask & ~hb.MASK_SLIM, th, tw)
        # the batch SLOTS back read this slot's records in its device-side column sums: those run on the K1 stream (see
        # _launch_columns), ahead of the K1 launches below in stream order — no host-side wait is needed here
        trace = settings().pipe_trace
        t_enq = time.perf_counter()
        rec_host = hb.record_doubles(host_mask)
        dev, stage, host, host_np = self._buffers(slot, count, tiles, hb.record_doubles(k1_mask), rec_host, x3d.device)
        t_buf = time.perf_counter()
        dec_mask = host_mask & ~hb.MASK_SLIM             # what the full records on the device hold (identity bf16 included)
        pending = []  # (event, first_index, n)
        self.stream.wait_stream(torch.cuda.current_stream())
        chunk = self._host_chunk(count, tiles)
        with torch.cuda.stream(self.stream):
            for first in range(0, count, chunk):
                n = min(chunk, count - first)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                hb.tile_stats_batched(x3d[first:first + n], k1_mask, out=dev[first:first + n])
                e1.record(self.stream)
                self.timing.events.append((e0, e1, n * tiles))
                ready = e1
                # Packing stays on the K1 stream: K1 is a persistent grid over every CU, so a kernel on another stream only gets
                # waves once K1 drains — packing there delayed each chunk's copy by a whole K1 launch (measured: 450–517 M tiles/s
                # against 582–603 M).
                if slim:
                    hb.pack_slim_records(dev[first:first + n], k1_mask, out=stage[first:first + n])
                    ready = torch.cuda.Event()
                    ready.record(self.stream)
                self.copy_stream.wait_event(ready)
                with torch.cuda.stream(self.copy_stream):
                    host[first:first + n].copy_(stage[first:first + n], non_blocking=True)
                    done = torch.cuda.Event(blocking=True)   # the driver thread sleeps while it waits: spinning would burn a core of the scan budget
                    done.record(self.copy_stream)
                pending.append((done, first, n))
        enq = {"host_np": host_np, "host_mask": host_mask, "tiles_hw": (th, tw), "numel": rows * cols if numel is None else int(numel),
               "seeds": seeds, "x": x3d, "dev": dev, "k1_mask": k1_mask, "slim": slim}
        # The scans are handed to the chunk-task pool HERE, each behind its records' event: they start the moment the records land,
        # whatever the driver thread is doing (with the submission in finish() the scans of a batch only started once the driver
        # got there — all four chunks at once, 3.3 ms of scans with the GPU idle behind them).
        futures = []
        for done, first, n in pending:
            futures.append((first, n, self.pool.submit(_when_landed, done, _scan_chunk, *self._scan_args(enq, first, n, host_np[first:first + n], host_mask))))
        enq["futures"] = futures
        self._open.append(enq)
        if trace:
            print(f"[pipe] enqueue (host route) {count} x {tiles} tiles in chunks of {chunk}: buffers {1e3 * (t_buf - t_enq):.2f}, launches {1e3 * (time.perf_counter() - t_buf):.2f} ms", flush=True)
        return enq

    # ---------------------------------------------------------------------------------------------------------------------
    # device-resident scan (csrc/mtq_scan.hip)
    # ---------------------------------------------------------------------------------------------------------------------
    def _use_device_scan(self, tiles: int) -> bool:
        return self.device_scan and tiles <= self.device_scan_max_tiles and hb.device_scan_supported(self.tile_formats, self.metric, tiles)

    def _host_chunk(self, count: int, tiles: int) -> int:
        if self.host_chunk_tiles is None:
            return self.chunk
        return min(max(1, self.host_chunk_tiles // tiles), count)

    def _device_buffers(self, slot: int, count: int, tiles: int, rec: int, device) -> dict:
        """Per record slot: K1's full records, the scan's maps / status / scratch, per-format tile counts and the column sums
        (searched map + pure formats), with pinned host mirrors of what comes back: 1 B/tile + a few numbers per tensor.
        The storage is flat and only ever grows: batches of another shape (a model's next shape group) get views of it, so that no
        device or pinned allocation (milliseconds each) lands between two batches."""
        torch = self.torch
        b = self._devbufs.get(slot)
        if b is None or b["device"] != str(device):
            b = {"device": str(device), "flat": {}, "key": None}
            self._devbufs[slot] = b
        key = (count, tiles, rec)
        if b["key"] == key:
            return b
        n_scratch = int(hb.lib().mtq_columns_scratch_doubles())
        P = 1 + len(self.pure_formats)
        nf = len(MIXED_TILE_FORMATS)

        def flat(name, numel, dtype, pinned=False):
            t = b["flat"].get(name)
            if t is None or t.numel() < numel:
                t = torch.zeros((int(numel * 1.25) + 16,), dtype=dtype, pin_memory=True) if pinned else \
                    torch.zeros((int(numel * 1.25) + 16,), dtype=dtype, device=device)
                b["flat"][name] = t
            return t[:numel]

        b.update({
            "key": key,
            "dev": flat("dev", count * tiles * rec, torch.float64).view(count, tiles, rec),
            "maps_dev": flat("maps_dev", count * tiles, torch.int8).view(count, tiles),
            "status_dev": flat("status_dev", count, torch.int32),
            "counts_dev": flat("counts_dev", count * nf, torch.int32).view(count, nf),
            "seeds_dev": flat("seeds_dev", count, torch.int64),
            "seeds_host": flat("seeds_host", count, torch.int64, pinned=True),
            "scratch": flat("scratch", int(hb.lib().mtq_greedy_scan_scratch_bytes(count, tiles)), torch.uint8),
            "sums_dev": flat("sums_dev", P * count * n_scratch, torch.float64).view(P, count, n_scratch),
            "pure_maps": [flat(f"pure_{f}", count * tiles, torch.int8).view(count, tiles) for f in self.pure_formats],
            "maps_host": flat("maps_host", count * tiles, torch.int8, pinned=True).view(count, tiles),
            "status_host": flat("status_host", count, torch.int32, pinned=True),
            "counts_host": flat("counts_host", count * nf, torch.int32, pinned=True).view(count, nf),
            "sums_host": flat("sums_host", P * count * 7, torch.float64, pinned=True).view(P, count, 7),
            # round 3: shared visiting orders of the batch's seed; the split search's candidate list, its length per chunk, the state
            # it carries from phase 1 to phase 2, and the listed kernel's hand-back list
            "listed": flat("listed", count * tiles, torch.int32),
            "n_listed": flat("n_listed", count, torch.int32),
            "carry": flat("carry", int(hb.lib().mtq_scan_carry_bytes(count)), torch.uint8),
            "lscr": flat("lscr", count * tiles + count, torch.int32),
            "n_listed_host": flat("n_listed_host", count, torch.int32, pinned=True),
            "mark": flat("mark", count, torch.int32),   # per chunk: id of the K1 launch that met a tile for the literal fix-up (mtq_tile_stats_partial_begin / _end)
        })
        for f, pm in zip(self.pure_formats, b["pure_maps"]):
            pm.fill_(MIXED_TILE_FORMATS.index(f))
        return b

    def _enqueue_device(self, x3d, seeds, numel, slot: int, k1_mask: int, dec_mask: int, th: int, tw: int) -> dict:
        """GPU work of a batch with the scan on the device: per chunk K1 (launch stream) → scan + tile counts + column sums (scan
        stream, behind the chunk's K1, beside the next chunk's K1) → maps, status, counts and sums D2H (copy stream)."""
        import time

        t_enq = time.perf_counter()
        trace = settings().pipe_trace
        marks = []
        torch = self.torch
        count, rows, cols = x3d.shape
        tiles = th * tw
        b = self._device_buffers(slot, count, tiles, hb.record_doubles(k1_mask), x3d.device)
        if trace:
            marks.append(("buffers", time.perf_counter()))
        n_el = rows * cols if numel is None else int(numel)
        sh = b["seeds_host"].numpy().view(np.uint64)     # the kernel reads uint64 seeds: 2^63 and above are seeds too
        sh[:] = np.uint64(self.seed) if seeds is None else np.asarray([int(v) for v in seeds], dtype=np.uint64)
        if (sh == 0).any():
            raise ValueError("seed 0 means 'draw a random seed' in the reference; pass non-zero seeds")
        per = int(hb.lib().mtq_greedy_scan_scratch_bytes(1, tiles))
        per_carry = int(hb.lib().mtq_scan_carry_bytes(1))
        uniform = seeds is None or len(set(int(v) for v in seeds)) == 1
        shared = self.shared_orders and uniform and self.metric in ("pcc", "mae")
        plan = self.lazy_plan(x3d)   # what K1 writes now, and what the listed kernel writes later (masks by format code)
        lazy = plan is not None
        if lazy:
            _lay, full_now, prev_bit, last_bit = plan
        pending = []
        cur = torch.cuda.current_stream()
        if not cur.query():                             # x3d's producer may still be running there; an idle stream needs no event + barrier packet
            self.stream.wait_stream(cur)
        # No stream-side wait for the slot's previous user: a slot is handed out again only after that batch was finished on the host
        # (enqueue() refuses when every slot is open), and _finish_device returns behind the batch's last event — its scans have read the
        # records, its results are home.  (An event wait here put a barrier packet in front of every K1 launch: 35 µs of idle stream and 0.9 ms of host CPU per step; the step itself measured the same with and without.)
        for first in range(0, count, self.chunk):
            n = min(self.chunk, count - first)
            scan_stream = self.scan_streams[self._scan_rr]
            self._scan_rr = (self._scan_rr + 1) % len(self.scan_streams)
            ci = first // self.chunk
            if shared and first == 0:
                # One set of orders serves every chunk of the batch — and every later batch of the same (seed, tile count): the draws depend
                # on nothing else (mixed_tile_greedy.py:222-231), a model run uses one seed, the bench one shape.  Round 3 drew them again
                # for every batch: 0.5–0.9 ms at the head of every search chain.  Cached per (device, seed, tiles, orders), read-only afterwards.
                n_ord = 2 if len(self.tile_formats) > 2 else 1
                okey = (str(x3d.device), int(sh[0]), tiles, n_ord)
                hit = self._orders_cache.get(okey)
                if hit is None:
                    with torch.cuda.stream(scan_stream):    # needs nothing of K1: runs beside it
                        buf = hb.scan_orders_device(int(sh[0]), tiles, n_ord)
                        orders_ready = torch.cuda.Event()
                        orders_ready.record(scan_stream)
                    if len(self._orders_cache) >= 32:       # a model's shape groups x seeds: a handful; bounded all the same (oldest out of the
                        self._orders_retired.append(self._orders_cache.pop(next(iter(self._orders_cache))))   # table, kept alive until close(): a search in flight may still read it)
                    self._orders_cache[okey] = hit = [buf, orders_ready]
                orders_buf, orders_ready = hit
                if orders_ready is not None and orders_ready.query():
                    hit[1] = orders_ready = None            # drawn and complete: no stream needs to wait for it any more
            with torch.cuda.stream(self.stream):
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record(self.stream)
                # bf16 storage in whole 32x128 units: K1 as the exact-integer kernel alone (it resets its own unit counters), its literal
                # fix-up — usually nothing to do — on the search stream: nothing sits between two K1 launches on this stream
                # (what mtq_tile_stats_partial_begin takes: 16-byte aligned rows of every tensor of the chunk, and no MTQ_FORCE_GENERIC — anything
                # else goes through the one-call forms below, which pick the generic / direct kernel themselves)
                two_launch = (x3d.dtype == torch.bfloat16 and (k1_mask & 1) == 0 and rows % 32 == 0 and cols % 128 == 0 and settings().k1_two_launch
                              and x3d.data_ptr() % 16 == 0 and (x3d.stride(0) * 2) % 16 == 0 and (x3d.stride(1) * 2) % 16 == 0
                              and os.environ.get("MTQ_FORCE_GENERIC", "0") != "1")
                k1_id = None
                if two_launch:
                    k1_id = hb.tile_stats_partial_begin(x3d[first:first + n], k1_mask, full_now if lazy else k1_mask, prev_bit if lazy else 0,
                                                        b["dev"][first:first + n], b["mark"][ci:ci + 1])
                elif lazy:
                    hb.tile_stats_partial(x3d[first:first + n], k1_mask, full_now, prev_bit, out=b["dev"][first:first + n])
                else:
                    hb.tile_stats_batched(x3d[first:first + n], k1_mask, out=b["dev"][first:first + n])
                e1.record(self.stream)
            self.timing.events.append((e0, e1, n * tiles))
            if trace:
                marks.append(("k1", time.perf_counter()))
            scan_stream.wait_event(e1)
            if trace:
                if getattr(self, "_trace_ref", None) is None:
                    self._trace_ref = e0
                    self._trace_t0 = time.perf_counter()
            with torch.cuda.stream(scan_stream):
                # the chunk's seeds go up on its own scan stream (on the copy stream they would queue behind the previous batch's
                # downloads, i.e. behind the previous batch's scans: the scans of consecutive batches would run one after the other)
                b["seeds_dev"][first:first + n].copy_(b["seeds_host"][first:first + n], non_blocking=True)
                recs = b["dev"][first:first + n]
                maps = b["maps_dev"][first:first + n]
                if k1_id is not None:
                    hb.tile_stats_partial_end(x3d[first:first + n], k1_mask, recs, b["mark"][ci:ci + 1], k1_id)
                if shared and orders_ready is not None:
                    scan_stream.wait_event(orders_ready)
                args = (recs, dec_mask, self.tile_formats, self.metric, self.threshold, float(n_el), b["seeds_dev"][first:first + n], maps,
                        b["status_dev"][first:first + n], b["scratch"][first * per:(first + n) * per])
                orders = orders_buf if shared else None
                if lazy:
                    xs = x3d[first:first + n]
                    listed, nl = b["listed"][first * tiles:(first + n) * tiles], b["n_listed"][ci:ci + 1]
                    carry = b["carry"][first * per_carry:(first + n) * per_carry]
                    nl.zero_()
                    hb.greedy_scan_device_ex(*args, counts_out=b["counts_dev"][first:first + n], orders=orders, phase=1, listed=listed, n_listed=nl, carry=carry)
                    hb.tile_stats_listed(xs, k1_mask, last_bit, prev_bit, listed, nl, recs, scratch=b["lscr"][first * tiles + ci:(first + n) * tiles + ci + 1])
                    hb.greedy_scan_device_ex(*args, counts_out=b["counts_dev"][first:first + n], phase=2, carry=carry)
                else:
                    hb.greedy_scan_device_ex(*args, counts_out=b["counts_dev"][first:first + n], orders=orders)
                hb.check(hb.lib().mtq_column_sums_device_batched(recs.data_ptr(), n, tiles, dec_mask, maps.data_ptr(),
                                                                 b["sums_dev"][0, first:first + n].data_ptr(), scan_stream.cuda_stream))
                for k, pm in enumerate(b["pure_maps"]):
                    hb.check(hb.lib().mtq_column_sums_device_batched(recs.data_ptr(), n, tiles, dec_mask, pm[first:first + n].data_ptr(),
                                                                     b["sums_dev"][1 + k, first:first + n].data_ptr(), scan_stream.cuda_stream))
                scanned = torch.cuda.Event(enable_timing=trace)
                scanned.record(scan_stream)
            if trace:
                marks.append(("scan+sums", time.perf_counter()))
            # results home by a kernel on the scan stream itself (stores into the pinned mirrors: hb.device_copy).  An asynchronous
            # device-to-host memcpy of the batch's 2 MB of maps held this thread until its stream had drained — 7 ms, a couple of times
            # in the first steps of a process — and needed a copy stream with two more cross-stream events per chunk
            with torch.cuda.stream(scan_stream):
                hb.device_copy(b["maps_host"][first:first + n], maps)
                hb.device_copy(b["status_host"][first:first + n], b["status_dev"][first:first + n])
                hb.device_copy(b["counts_host"][first:first + n], b["counts_dev"][first:first + n])
                if lazy:
                    hb.device_copy(b["n_listed_host"][ci:ci + 1], b["n_listed"][ci:ci + 1])
                for q in range(b["sums_dev"].shape[0]):
                    hb.device_copy(b["sums_host"][q, first:first + n], b["sums_dev"][q, first:first + n, :7])
                done = torch.cuda.Event(blocking=True, enable_timing=trace)
                done.record(scan_stream)
            pending.append((done, first, n))
            if trace:
                self._trace_rows = getattr(self, "_trace_rows", []) + [(count, tiles, e0, e1, scanned, done, time.perf_counter())]
        enq = {"device": True, "buf": b, "pending": pending, "tiles_hw": (th, tw), "numel": n_el, "x": x3d, "dec_mask": dec_mask,
               "seeds": sh.copy(), "lazy": lazy, "k1_mask": k1_mask}
        self._open.append(enq)
        self.host_seconds["enqueue"] += time.perf_counter() - t_enq
        if trace:
            marks.append(("copies", time.perf_counter()))
            prev = t_enq
            parts = []
            for name, t in marks:
                parts.append(f"{name} {1e3 * (t - prev):.2f}")
                prev = t
            print(f"[pipe] enqueue {count} x {tiles} tiles: " + ", ".join(parts) + " ms", flush=True)
        return enq

    def _finish_device(self, enq: dict) -> list[TensorResult]:
        """Collects a device-scanned batch: waits for the chunks' copies and wraps maps / counts / columns; a tensor the device scan
        handed back (status != 0: a zero denominator, the decision needs Σ|x−y|) is searched by the host scan on its records."""
        import time

        b = enq["buf"]
        th, tw = enq["tiles_hw"]
        n_el = float(enq["numel"])
        k = {"pcc": 0, "mae": 1, "atol": 2}[self.metric]
        results: list[TensorResult] = []
        names = MIXED_TILE_FORMATS
        for done, first, n in enq["pending"]:
            t0 = time.perf_counter()
            _sleep_until(done)
            t1 = time.perf_counter()
            maps = b["maps_host"][first:first + n].numpy().reshape(n, th, tw).copy()   # one copy per chunk: the pinned buffer is reused
            status = b["status_host"][first:first + n].numpy()
            counts = b["counts_host"][first:first + n].numpy().tolist()
            sums = b["sums_host"][:, first:first + n].numpy()
            cols = columns_from_sums_batch(sums[0], n_el).tolist()
            pure_cols = [columns_from_sums_batch(sums[1 + i], n_el).tolist() for i in range(len(self.pure_formats))]
            bad = np.flatnonzero(status)
            for j in range(n):
                pure = {f: tuple(pure_cols[i][j]) for i, f in enumerate(self.pure_formats)} if self.pure_formats else None
                c = dict(zip(names, counts[j]))
                cj = cols[j]
                results.append(TensorResult(first + j, maps[j], c, mixed_tile_total_bytes(c), cj[0], cj[1], cj[2], cj[k], pure))
            if enq.get("lazy"):
                listed = int(b["n_listed_host"][first // self.chunk])
                self.listed_tiles += listed
                if listed > self.lazy_max_listed * n * th * tw:
                    self.lazy_off.setdefault(th * tw, listed / float(n * th * tw))
            for j in bad:   # handed back by the device scan: the host scan on this tensor's records
                self.host_fallbacks += 1
                if enq.get("lazy"):   # the records hold only what the search had asked for so far: the whole record of this one tensor now
                    full = hb.tile_stats_batched(enq["x"][first + j:first + j + 1], enq["k1_mask"])[0].cpu().numpy()
                else:
                    full = b["dev"][first + j].cpu().numpy()
                amap, c, out = hb.greedy_run(full, enq["dec_mask"], self.tile_formats, self.metric, self.threshold, n_el, int(enq["seeds"][first + j]))
                r = results[len(results) - n + j]
                r.assignment, r.counts, r.tile_bytes = amap.reshape(th, tw), c, mixed_tile_total_bytes(c)
                r.pcc, r.mae, r.atol, r.metric_value = out["pcc"], out["mae"], out["atol"], out[self.metric]
            self.host_seconds["wait"] += t1 - t0
            self.host_seconds["wrap"] += time.perf_counter() - t1
            if settings().pipe_trace:
                print(f"[pipe] finish {n} x {th * tw}: waited {1e3 * (t1 - t0):.2f} ms, wrapped in {1e3 * (time.perf_counter() - t1):.2f} ms, handed back {bad.size}", flush=True)
        self._open.pop(0)
        enq["x"] = None
        return results

    def _scan_args(self, enq: dict, first: int, n: int, stats, mask):
        seeds = enq["seeds"]
        sd = [self.seed] * n if seeds is None else [int(v) for v in seeds[first:first + n]]
        return (first, stats, mask, enq["tiles_hw"], enq["numel"], self.tile_formats, self.metric, self.threshold, sd, self.workers)

    def finish(self, enq: dict, defer_columns: bool = False) -> list[TensorResult]:
        """Host half: collects the chunk scans that enqueue() queued behind the records' events; returns when all of them are done.
        The GPU meanwhile works on whatever was enqueued after this batch.  With defer_columns the device-side columns of a
        slim batch are only LAUNCHED here (maps up, batched sums, seven doubles per tensor down, all asynchronous on their
        own stream, all from this thread); resolve(enq) waits for them and fills pcc / mae / atol in."""
        torch = self.torch
        if not self._open or self._open[0] is not enq:
            raise RuntimeError("batches finish in the order they were enqueued")
        if enq.get("device"):
            return self._finish_device(enq)
        futures = enq.pop("futures")
        results: list[TensorResult] = []
        t_fin = time.perf_counter()
        for first, n, fut in futures:
            try:
                results.extend(fut.result())
            except hb.MtqError:
                if not enq["slim"]:
                    raise
                # a zero-variance tensor in this chunk: its decision needs Σ|x−y|, which the slim records do not carry
                full = enq["dev"][first:first + n].cpu().numpy()
                results.extend(_scan_chunk(*self._scan_args(enq, first, n, full, enq["host_mask"] & ~hb.MASK_SLIM)))
        self._open.pop(0)
        enq["x"] = None
        t_scanned = time.perf_counter()
        if enq["slim"] or self.pure_formats:
            self._launch_columns(enq, results)
            if not defer_columns:
                self.resolve(enq)
        if settings().pipe_trace:
            print(f"[pipe] finish (host route) {len(results)} tensors: scans collected after {1e3 * (t_scanned - t_fin):.2f} ms, columns after {1e3 * (time.perf_counter() - t_scanned):.2f} ms", flush=True)
        return results

    def _launch_columns(self, enq: dict, results: list) -> None:
        """pcc / mae / atol of every tensor of the batch from the full records on the device under the maps the scans
        produced: maps up (1 B/tile), one batched reduction, seven doubles per tensor back — launched, not waited for."""
        torch = self.torch
        dev = enq["dev"]
        count, tiles = dev.shape[0], dev.shape[1]
        self._column_buffers(count, tiles, dev.device)
        ring = self._colbufs["ring"]
        flat = ring[self._colbufs["next"] % len(ring)]
        self._colbufs["next"] = (self._colbufs["next"] + 1) % len(ring)
        user = flat.get("user")                          # the batch a ring length back read its sums out of this ring entry
        if user is not None and "col_pending" in user:
            self.resolve(user)
            self._unresolved = [e for e in self._unresolved if e is not user]
        flat["user"] = enq
        n_sets, n_scratch = 1 + len(self.pure_formats), self._colbufs["n_scratch"]
        cb = {"maps_host": flat["maps_host"][:count * tiles].view(count, tiles), "maps_dev": flat["maps_dev"][:count * tiles].view(count, tiles),
              "scratch": flat["scratch"][:n_sets * count * n_scratch].view(n_sets, count, n_scratch),
              "sums_host": flat["sums_host"][:n_sets * count * 7].view(n_sets, count, 7)}
        mh = cb["maps_host"].numpy()
        for i, r in enumerate(results):
            mh[i] = r.assignment.reshape(-1)
        # The reduction runs on the K1 STREAM, between K1 launches: K1 is a persistent grid over every CU, and on a stream of its
        # own this kernel only got waves in the gaps of the K1 stream — once the scans stopped leaving such gaps, the driver
        # thread waited 3 ms per step for it.  Stream order also makes it read the records before a later batch's K1 overwrites
        # them.  Only the two copies stay on the side stream.
        with torch.cuda.stream(self.col_stream):
            cb["maps_dev"].copy_(cb["maps_host"], non_blocking=True)
            maps_up = torch.cuda.Event()
            maps_up.record(self.col_stream)
        self.stream.wait_event(maps_up)
        dec_mask = enq["host_mask"] & ~hb.MASK_SLIM
        if enq["slim"]:   # the searched maps' columns (other metrics: the host scan already produced them)
            hb.check(hb.lib().mtq_column_sums_device_batched(dev.data_ptr(), count, tiles, dec_mask, cb["maps_dev"].data_ptr(),
                                                             cb["scratch"][0].data_ptr(), self.stream.cuda_stream))
        for k, pm in enumerate(self._colbufs["pure_maps"]):   # one constant map per pure format (flat: any (count, tiles) view of it is constant)
            hb.check(hb.lib().mtq_column_sums_device_batched(dev.data_ptr(), count, tiles, dec_mask, pm.data_ptr(), cb["scratch"][1 + k].data_ptr(),
                                                             self.stream.cuda_stream))
        summed = torch.cuda.Event()
        summed.record(self.stream)
        with torch.cuda.stream(self.col_stream):
            self.col_stream.wait_event(summed)
            cb["sums_host"].copy_(cb["scratch"][:, :, :7], non_blocking=True)
            done = torch.cuda.Event(blocking=True)
            done.record(self.col_stream)
        enq["col_pending"] = (done, cb["sums_host"], results)

    def _column_buffers(self, count: int, tiles: int, device) -> None:
        """Ring of flat, only-growing buffers for _launch_columns (maps up, reduction scratch, sums down): batches of another shape get
        views, so that no device or pinned allocation lands between two batches of a run that prepare() has seen."""
        torch = self.torch
        cb = self._colbufs
        n_sets = 1 + len(self.pure_formats)
        if (cb.get("device") == str(device) and cb["cap_tiles"] >= count * tiles and cb["cap_count"] >= count and len(cb["ring"]) >= self.SLOTS):
            return
        for e in cb.get("ring", []):                     # sums still in flight live in the old buffers: collect them first
            user = e.get("user")
            if user is not None and "col_pending" in user:
                self.resolve(user)
                self._unresolved = [u for u in self._unresolved if u is not user]
        cap_tiles = max(count * tiles, cb.get("cap_tiles", 0) if cb.get("device") == str(device) else 0)
        cap_count = max(count, cb.get("cap_count", 0) if cb.get("device") == str(device) else 0)
        n_scratch = int(hb.lib().mtq_columns_scratch_doubles())
        self._colbufs = {"device": str(device), "cap_tiles": cap_tiles, "cap_count": cap_count, "n_scratch": n_scratch, "next": 0, "ring": [
            {"maps_host": torch.empty((cap_tiles,), dtype=torch.int8, pin_memory=True),
             "maps_dev": torch.empty((cap_tiles,), dtype=torch.int8, device=device),
             "scratch": torch.empty((n_sets * cap_count * n_scratch,), dtype=torch.float64, device=device),
             "sums_host": torch.empty((n_sets * cap_count * 7,), dtype=torch.float64, pin_memory=True)} for _ in range(self.SLOTS)],
            "pure_maps": [torch.full((cap_tiles,), MIXED_TILE_FORMATS.index(f), dtype=torch.int8, device=device) for f in self.pure_formats]}

    def resolve(self, enq: dict) -> None:
        """Wait for the columns launched by finish(..., defer_columns=True) and fill them into the batch's results."""
        pend = enq.pop("col_pending", None)
        if pend is None:
            return
        done, sums_host, results = pend
        done.synchronize()
        sums = sums_host.numpy()
        if enq["slim"]:
            cols = columns_from_sums_batch(sums[0], float(enq["numel"]))
            k = {"pcc": 0, "mae": 1, "atol": 2}[self.metric]
            for r, c in zip(results, cols):
                r.pcc, r.mae, r.atol, r.metric_value = float(c[0]), float(c[1]), float(c[2]), float(c[k])
        for j, f in enumerate(self.pure_formats):
            cols = columns_from_sums_batch(sums[1 + j], float(enq["numel"]))
            for r, c in zip(results, cols):
                if r.pure is None:
                    r.pure = {}
                r.pure[f] = (float(c[0]), float(c[1]), float(c[2]))
        enq["dev"] = None

    def _warm_device_scan(self, device) -> None:
        """One-time costs of the device-scan route, paid here instead of inside the first batches: every scan stream's hardware queue
        (created on first use, milliseconds each) and the first launch of both scan kernel variants (code load, LDS limit)."""
        if getattr(self, "_warmed", None) == str(device):
            return
        torch = self.torch
        rec = hb.record_doubles(0xF)
        seeds = torch.ones((1,), dtype=torch.int64, device=device)
        for i, st in enumerate(self.scan_streams):
            with torch.cuda.stream(st):
                tiles = 1 if i % 2 == 0 else hb.SCAN_LDS_MAX_TILES + 1      # visiting order in LDS / in global scratch
                hb.greedy_scan_device(torch.zeros((1, tiles, rec), dtype=torch.float64, device=device), 0xF, ["bf16", "bfp8"], "pcc", 0.5, float(tiles * 1024), seeds)
        pinned = torch.zeros((16,), dtype=torch.int8, pin_memory=True)
        for st in (self.stream, self.copy_stream):
            with torch.cuda.stream(st):
                pinned.copy_(torch.zeros((16,), dtype=torch.int8, device=device), non_blocking=True)
        torch.cuda.synchronize()
        self._warmed = str(device)
        # one tiny batch end to end: the host side's first-use costs too (the first Tensor.numpy() of a process took ~55 ms on the
        # GPU boxes — inside the first batch's wrap-up otherwise)
        g = torch.Generator(device=device)
        g.manual_seed(1)
        tiny = (torch.randn((2, 64, 128), generator=g, device=device) * 0.02).to(torch.bfloat16)
        if self._use_device_scan(8):
            self.finish(self.enqueue(tiny))
            torch.cuda.synchronize()

    def prepare(self, batches) -> None:
        """Grow every record slot's storage to the largest of `batches` (device and pinned allocations cost milliseconds: not between
        two batches of a timed run) and pay the routes' one-time costs.  `batches` in the order run_batches() will get them."""
        first = True
        host_route = []
        for i, x3d in enumerate(batches):
            if first and self.device_scan:
                self._warm_device_scan(x3d.device)
                with self.torch.cuda.stream(self.stream):
                    hb.tile_stats_batched(x3d[:1], self._layout(x3d)[0])  # K1's code object, the work-counter ring, the launch stream's queue
                first = False
            count, rows, cols = x3d.shape
            th, tw = hb.tiles_hw(rows, cols)
            k1_mask, host_mask, slim = self._layout(x3d)
            if not self._use_device_scan(th * tw):
                host_route.append((i, count, th * tw, hb.record_doubles(k1_mask), hb.record_doubles(host_mask), x3d.device, slim))
                continue
            for slot in range(self.SLOTS):
                self._device_buffers(slot, count, th * tw, hb.record_doubles(k1_mask), x3d.device)
        # host-route batches: the record slot each one will land on (slots rotate per enqueue), its pinned mirror touched once, the
        # column buffers at the largest shape, the scan threads started
        for i, count, tiles, rec, rec_host, device, slim in host_route:
            slot = (self._next_slot + i) % self.SLOTS
            dev, stage, host, _np = self._buffers(slot, count, tiles, rec, rec_host, device)
            host.copy_(stage, non_blocking=True)   # first DMA into the pinned pages (mappings are set up lazily)
            if slim or self.pure_formats:
                self._column_buffers(count, tiles, device)
        if host_route:
            hb.greedy_run_batch(np.zeros((self.workers, 1, hb.record_doubles(0xF))), 0xF, ["bf16"], self.metric, self.threshold, 1024.0,
                                [1] * self.workers, self.workers)
        self.torch.cuda.synchronize()

    def run_batches(self, batches, seeds=None) -> list[list[TensorResult]]:
        """Batches of possibly different shapes (a model's shape groups), SLOTS of them in flight: a batch's scans and downloads
        run beside the next batches' K1.  → the results of every batch, in order."""
        torch = self.torch
        out, open_ = [], []
        for i, x3d in enumerate(batches):
            open_.append(self.enqueue(x3d, None if seeds is None else seeds[i]))
            if len(open_) == self.SLOTS:
                out.append(self.finish(open_.pop(0)))
        while open_:
            out.append(self.finish(open_.pop(0)))
        for st in (self.stream, self.copy_stream, *self.scan_streams):
            torch.cuda.current_stream().wait_stream(st)
        if settings().pipe_trace and getattr(self, "_trace_rows", None):
            torch.cuda.synchronize()
            ref = self._trace_ref
            for count, tiles, e0, e1, scanned, done, t_host in self._trace_rows:
                print(f"[pipe] batch {count} x {tiles}: host launch {1e3 * (t_host - self._trace_t0):7.2f} | K1 {ref.elapsed_time(e0):7.2f} .. {ref.elapsed_time(e1):7.2f} | "
                      f"scan+sums done {ref.elapsed_time(scanned):7.2f} | copies done {ref.elapsed_time(done):7.2f} ms", flush=True)
            self._trace_rows, self._trace_ref = [], None
        return out

    def run(self, x3d, seeds=None, numel: int | None = None) -> list[TensorResult]:
        """One batch, start to end (enqueue + finish).  numel: the tensors' element count when the matrices are zero-filled
        2-D images of shorter vectors (tile_utils.py:96-102); the greedy metric divides by it (mixed_tile_greedy.py:134)."""
        torch = self.torch
        results = self.finish(self.enqueue(x3d, seeds, numel))
        for st in (self.stream, self.copy_stream, *self.scan_streams):
            torch.cuda.current_stream().wait_stream(st)
        return results

    def run_steps(self, batches) -> list[TensorResult]:
        """A sequence of batches, SLOTS of them in flight: a batch's GPU work is enqueued two batches before its scans are
        collected, so the K1 stream always holds the next batch's launches while the driver thread waits for the scans of an
        earlier one (with one batch of look-ahead the stream ran dry for ≈ 1 ms per step: the last chunk's copy and scans of
        batch i−1 end about when K1 of batch i does).  Returns the LAST batch's results; every batch is fully processed."""
        torch = self.torch
        results, open_ = [], []
        for x3d in batches:
            open_.append(self.enqueue(x3d))
            if len(open_) == self.SLOTS:
                done = open_.pop(0)
                results = self.finish(done, defer_columns=True)   # columns launched; collected while later batches are scanned
                self._unresolved.append(done)
        while open_:
            done = open_.pop(0)
            results = self.finish(done, defer_columns=bool(open_))
            if open_:
                self._unresolved.append(done)
        while self._unresolved:
            self.resolve(self._unresolved.pop(0))
        for st in (self.stream, self.copy_stream, *self.scan_streams):
            torch.cuda.current_stream().wait_stream(st)
        return results

    def close(self) -> None:
        """Ends the pipeline in an orderly way: its streams drained, its chunk-task threads joined, its device and pinned buffers
        released now — while the HIP runtime is certainly still there — rather than whenever the interpreter gets to them at exit
        (round 2 saw a SIGSEGV inside __cxa_finalize after a profiled `wq` run; hip_backend.shutdown() is the library's half).
        Idempotent; the pipeline must not be used afterwards."""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        try:
            for st in (self.stream, self.copy_stream, self.col_stream, *self.scan_streams):
                st.synchronize()
        finally:
            self.pool.shutdown(wait=True)
            self._open.clear()
            self._unresolved.clear()
            for d in (self._devbufs, self._bufs, self._colbufs, self._orders_cache):
                d.clear()
            self._orders_retired.clear()
            self.timing.events.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
