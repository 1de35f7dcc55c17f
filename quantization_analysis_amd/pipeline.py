"""Streamed multi-tensor drivers of the mixed-tile search on one GPU.

A model's matched tensors are independent (reference wq:655 loop), so a rank streams its share.  GreedyPipeline, device route (the
default): K1 over a batch of equally shaped tensors on the K1 stream (the lazy route evaluates what the search reads
first) → on one of three search streams the launch's visiting orders, K1's fix-up, the search (in two phases around the listed K1 on the
lazy route), the column sums and a copy kernel that stores maps, counts and seven sums per tensor into pinned host memory — while the
next batches' K1 is already running (`run_steps` / `run_batches`: four record slots).  Host route (`scan="host"`, repeated formats):
K1 per chunk → records D2H into pinned memory on a copy stream → per-tensor sequential scans on host worker threads (C++ scan pool in
libmtq_hip.so, GIL released).  ThresholdPipeline: K1 → the threshold rule, the knife-edge tiles' list / fetch / quantisation and the
column sums on the device; the host scores the knife-edge tiles literally.  y is not materialised here (assignment maps + pcc / mae /
atol are the outputs the north star names); use compression_algorithms.* for the drop-in run() that returns y.
"""
from __future__ import annotations

from .pipeline_common import KernelTiming, TensorResult, columns_from_sums_batch, cpu_budget, default_workers  # noqa: F401
from .pipeline_greedy import GreedyPipeline  # noqa: F401
from .pipeline_threshold import ThresholdPipeline  # noqa: F401

__all__ = ["GreedyPipeline", "ThresholdPipeline", "TensorResult", "KernelTiming", "columns_from_sums_batch", "cpu_budget", "default_workers"]
