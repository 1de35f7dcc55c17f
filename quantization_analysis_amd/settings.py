"""Every switch of the hip backend, in one place.

The product has no configuration beyond the reference's own (CLI flags, compression config JSON).  What is listed here are
MEASUREMENT switches: A/B levers of experiments recorded in DESIGN.md / profiles/, read ONCE per process by `settings()`
(Python side) or on first use by the library (C side).  Defaults are what ships; nothing here changes results — every
combination is covered by the same parity tests.

Python side (this module)
| variable | default | meaning |
|---|---|---|
| MTQ_LIB | package's libmtq_hip.so | path of the C-ABI library (A/B between builds: tools/r3_env_ab.sh) |
| MTQ_NUMA_BIND | 1 | bind a rank's threads to the NUMA node of its GPU (hip_backend.bind_to_gpu_numa) |
| MTQ_PIPE_SLOTS | 4 | GreedyPipeline record slots = batches in flight |
| MTQ_SCAN_STREAMS | 3 | search streams of the device route |
| MTQ_SCAN_PRIORITY | -1 | HIP priority of the search streams |
| MTQ_CHUNK_TASKS | 8 | chunk-level host tasks of the host-scan route |
| MTQ_SCAN_WORKERS | CPU share, 4..32 | host scan threads per rank |
| MTQ_DEVICE_SCAN | 1 | 0: the search on host threads (records over PCIe) |
| MTQ_DEVICE_SCAN_MAX_TILES | 2^22 | tensors above this take the host scan |
| MTQ_SHARED_ORDERS | 1 | one set of visiting orders per (seed, tile count), cached across batches |
| MTQ_LAZY | 1 | lazy route: K1 <3,1>, listed completion before the last pass |
| MTQ_LAZY_MAX_LISTED | 0.35 | listed fraction above which a tile count leaves the lazy route |
| MTQ_IDENTITY_RECORDS | 1 | bf16 storage: no record slot for the identity bf16 candidate |
| MTQ_SLIM_RECORDS | 1 | host-scan route, pcc: 3 doubles per format cross PCIe |
| MTQ_K1_TWO_LAUNCH | 1 | K1 and its literal fix-up as two launches on two streams |
| MTQ_KNIFE_CAP | 128 | knife-edge tiles per chunk listed without a second round trip (ThresholdPipeline) |
| MTQ_THRESHOLD_RAGGED | 1 | run_batches sends batches the direct K1 serves as ragged groups (one launch per stage for many shapes) |
| MTQ_SWEEP_THREADS | CPU share | literal-scoring threads of the sweep |
| MTQ_WQ_MAX_SLOTS | 8 | record slots of `wq --backend hip`'s windows |
| MTQ_WQ_DEVICE_SCAN_MAX_TILES | 2^22 | device-scan limit behind `wq` |
| MTQ_PIPE_TRACE | 0 | 1: print per-batch driver timings |

C side (getenv on first use; csrc/)
| variable | default | meaning |
|---|---|---|
| MTQ_FORCE_GENERIC | 0 | 1: every K1 through the literal kernel (mtq_kernels.hip) |
| MTQ_K1_UNITS_PER_WAVE | 8 | units a K1 wave takes before it retires (0: persistent grid) |
| MTQ_K1_WAVES | per instantiation | waves per SIMD the K1 grid is sized for |
| MTQ_K1_LDS_PAD | 0 | extra LDS bytes per K1 block |
| MTQ_LISTED_WAVES | 4 | waves per SIMD the listed launch is sized for |
| MTQ_LISTED_DIRECT | unset | set: the listed launch through the direct kernel only |
| MTQ_SCAN_SHARED_LDS | 0 | 1: the helper-wave search keeps its visiting order in LDS |
| MTQ_SCAN_SCALAR | unset | host scan without the eight-wide pass |
| MTQ_SCAN_BUDGET | — | round budget of the device search before it hands a tensor back |

Compile time (-D, builds under build/ for A/B): MTQ_K1_INTDOM (round 3's packed-integer group arithmetic in the bf16 K1),
MTQ_FAST_WAVES, MTQ_ROLLED_WAVES_PER_SIMD_FORCE, MTQ_DIRECT_WAVES_PER_SIMD, MTQ_SCAN_WAVES_PER_EU, MTQ_SCAN_SETPRIO,
MTQ_SCAN_PROFILE / MTQ_SHUF_TICK (shader-clock stamps for tools/scan_ticks.py; never in the shipped kernel).
"""
from __future__ import annotations

import os
from dataclasses import dataclass


def _flag(name: str, default: bool) -> bool:
    v = os.environ.get(name)
    return default if v is None else v != "0"


def _int(name: str, default: int | None) -> int | None:
    v = os.environ.get(name)
    return default if v is None else int(v)


@dataclass(frozen=True)
class Settings:
    pipe_slots: int
    scan_streams: int | None
    scan_priority: int
    chunk_tasks: int
    scan_workers: int | None
    device_scan: bool
    device_scan_max_tiles: int | None
    shared_orders: bool
    lazy: bool
    lazy_max_listed: float
    identity_records: bool
    slim_records: bool
    k1_two_launch: bool
    knife_cap: int
    threshold_ragged: bool
    sweep_threads: int | None
    wq_max_slots: int
    wq_device_scan_max_tiles: int
    pipe_trace: bool
    numa_bind: bool


_CACHED: Settings | None = None


def settings(refresh: bool = False) -> Settings:
    """The process's switches, read from the environment once (refresh=True: again — tests that monkeypatch the environment)."""
    global _CACHED
    if _CACHED is None or refresh:
        _CACHED = Settings(
            pipe_slots=_int("MTQ_PIPE_SLOTS", 4), scan_streams=_int("MTQ_SCAN_STREAMS", None), scan_priority=_int("MTQ_SCAN_PRIORITY", -1),
            chunk_tasks=_int("MTQ_CHUNK_TASKS", 8), scan_workers=_int("MTQ_SCAN_WORKERS", None), device_scan=_flag("MTQ_DEVICE_SCAN", True),
            device_scan_max_tiles=_int("MTQ_DEVICE_SCAN_MAX_TILES", None), shared_orders=_flag("MTQ_SHARED_ORDERS", True), lazy=_flag("MTQ_LAZY", True),
            lazy_max_listed=float(os.environ.get("MTQ_LAZY_MAX_LISTED", "0.35")), identity_records=_flag("MTQ_IDENTITY_RECORDS", True),
            slim_records=_flag("MTQ_SLIM_RECORDS", True), k1_two_launch=_flag("MTQ_K1_TWO_LAUNCH", True), knife_cap=_int("MTQ_KNIFE_CAP", 128), threshold_ragged=_flag("MTQ_THRESHOLD_RAGGED", True),
            sweep_threads=_int("MTQ_SWEEP_THREADS", None), wq_max_slots=_int("MTQ_WQ_MAX_SLOTS", 8),
            wq_device_scan_max_tiles=_int("MTQ_WQ_DEVICE_SCAN_MAX_TILES", 1 << 22), pipe_trace=os.environ.get("MTQ_PIPE_TRACE") == "1",
            numa_bind=_flag("MTQ_NUMA_BIND", True))
    return _CACHED
