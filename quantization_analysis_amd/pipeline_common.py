"""What the streamed drivers share: the result type, the K1 timing record, the CPU budget of a rank, the host-scan task and the columns
from seven sums (pipeline.py re-exports these; GreedyPipeline is in pipeline_greedy.py, ThresholdPipeline in pipeline_threshold.py)."""
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


@dataclass
class TensorResult:
    index: int
    assignment: np.ndarray  # int8 (tiles_h, tiles_w)
    counts: dict
    tile_bytes: float
    pcc: float
    mae: float
    atol: float
    metric_value: float
    pure: dict | None = None    # format name → (pcc, mae, atol) of the whole tensor in that one format (the `none` rows of wq), on request


@dataclass
class KernelTiming:
    launches: int = 0
    kernel_ms: float = 0.0      # Σ of HIP-event durations around the K1 launches
    tiles: int = 0              # Σ tiles processed by those launches
    events: list = field(default_factory=list)

    def drain(self) -> None:
        for e0, e1, tiles in self.events:
            e1.synchronize()
            self.kernel_ms += e0.elapsed_time(e1)
            self.tiles += tiles
            self.launches += 1
        self.events.clear()


def cpu_budget() -> int:
    """Hardware threads this process may really use: the cgroup CPU quota when there is one (a gpurun box shows 256
    hardware threads but runs under a 16-CPU quota; exceeding a CFS quota stalls every thread of the job for the rest of
    the 100 ms period), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def default_workers() -> int:
    """Scan threads per rank: the rank's share of the CPU budget (the driver's own threads mostly sleep), at most 32;
    MTQ_SCAN_WORKERS overrides."""
    if settings().scan_workers is not None:
        return settings().scan_workers
    local = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    return max(4, min(32, cpu_budget() // max(local, 1)))


def _scan_chunk(first, stats, mask, tiles_hw, numel, tile_formats, metric, threshold, seeds, n_threads) -> list[TensorResult]:
    """The searches of one chunk of tensors: a single GIL-free C call fanning out over n_threads host threads."""
    maps, counts, outs = hb.greedy_run_batch(stats, mask, tile_formats, metric, threshold, float(numel), seeds, n_threads)
    k = {"pcc": 0, "mae": 1, "atol": 2}[metric]
    res = []
    for j in range(maps.shape[0]):
        c = {f: int(counts[j, i]) for i, f in enumerate(MIXED_TILE_FORMATS)}
        res.append(TensorResult(first + j, maps[j].reshape(tiles_hw), c, mixed_tile_total_bytes(c), float(outs[j, 0]), float(outs[j, 1]),
                                float(outs[j, 2]), float(outs[j, k])))
    return res


def _sleep_until(event, tick: float = 1e-4) -> None:
    """Wait for a HIP event without burning the core: hipEventSynchronize spins on this runtime even for events created with the
    blocking flag (the driver thread showed 100 % CPU while 'waiting'), so the event is polled between short sleeps.  The pipeline
    has a whole step of slack on this wait (several record slots)."""
    import time

    while not event.query():
        time.sleep(tick)


def _when_landed(event, fn, *args):
    """Chunk task of the streamed driver: sleep until the chunk's records are on the host (a blocking HIP event), then scan."""
    event.synchronize()
    return fn(*args)


def columns_from_sums_batch(sums: np.ndarray, n) -> np.ndarray:
    """mtq_columns_from_sums for many tensors at once: sums [count, 7] (Σx, Σx², Σy, Σy², Σxy, Σ|d|, max|d|) → [count, 3]
    pcc, mae, atol — the same double operations in the same order (metrics.py:6-16 as moments), element-wise in NumPy.  n: the tensors'
    element count, one number or one per tensor."""
    n = np.asarray(n, dtype=np.float64)
    sx, sx2, sy, sy2, sxy, sab, mx = (sums[:, i] for i in range(7))
    mean_x, mean_y = sx / n, sy / n
    am2 = np.maximum(sx2 - n * mean_x * mean_x, 0.0)
    bm2 = np.maximum(sy2 - n * mean_y * mean_y, 0.0)
    denom = np.sqrt(am2 * bm2)
    with np.errstate(all="ignore"):
        pcc = np.where(denom == 0.0, np.where(sab == 0.0, 1.0, 0.0), (sxy - n * mean_x * mean_y) / denom)
        mae = np.where(n != 0.0, sab / n, 0.0)
    return np.stack([pcc, mae, mx], axis=1)
