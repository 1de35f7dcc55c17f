"""ctypes binding of libmtq_hip.so (include/mtq.h) for PyTorch-ROCm tensors.

This module is the ONLY place the package talks to the GPU.  There is no CPU fallback: if the
shared library is missing, or no HIP device is visible, device entry points raise MtqError.
PyTorch is used for device memory and streams only (tensor.data_ptr(), current stream).
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
# MTQ_LIB selects another build of the same library (kernel A/B experiments); default is the in-tree build.
LIB_PATH = Path(os.environ["MTQ_LIB"]).resolve() if os.environ.get("MTQ_LIB") else _PKG / "libmtq_hip.so"

MIXED_TILE_FORMATS = ["bf16", "bfp8", "bfp4", "bfp2"]
FMT_CODE = {"bf16": 0, "bfp8": 1, "bfp4": 2, "bfp2": 3, "fp0": 4}
METRIC_CODE = {"pcc": 0, "mae": 1, "atol": 2}
DTYPE_BF16, DTYPE_F32 = 0, 1
TILE = 32

# every symbol include/mtq.h declares (tests check the library exports exactly these)
EXPORTS = [
    "mtq_version", "mtq_last_error", "mtq_device_count", "mtq_stats_record_doubles",
    "mtq_shutdown", "mtq_tile_stats", "mtq_tile_stats_batched", "mtq_tile_stats_partial", "mtq_tile_stats_partial_begin", "mtq_tile_stats_partial_end", "mtq_tile_stats_listed", "mtq_quantize", "mtq_apply_assignment", "mtq_dequant_fp8_block", "mtq_pack_slim_records",
    "mtq_greedy_create", "mtq_greedy_pass", "mtq_greedy_assignment", "mtq_greedy_fixed",
    "mtq_greedy_counts", "mtq_greedy_value", "mtq_greedy_destroy",
    "mtq_tile_scores", "mtq_threshold_assign", "mtq_columns_from_stats", "mtq_columns_from_sums", "mtq_tile_scores_device",
    "mtq_threshold_assign_device", "mtq_columns_scratch_doubles", "mtq_column_sums_device", "mtq_column_sums_device_batched",
    "mtq_rng_create", "mtq_rng_permutation", "mtq_rng_integers", "mtq_rng_destroy", "mtq_greedy_run", "mtq_greedy_run_batch",
    "mtq_selftest_slot_ring", "mtq_device_copy_2d", "mtq_knife_tiles_device", "mtq_greedy_scan_scratch_bytes", "mtq_greedy_scan_device", "mtq_greedy_scan_device_ex", "mtq_scan_carry_bytes",
    "mtq_scan_orders_bytes", "mtq_scan_orders_device", "mtq_debug_scan_ticks", "mtq_threshold_enqueue", "mtq_threshold_columns",
    "mtq_tile_stats_ragged", "mtq_knife_tiles_ragged", "mtq_column_sums_device_ragged", "mtq_threshold_enqueue_ragged", "mtq_threshold_columns_ragged",
]


class MtqError(RuntimeError):
    pass


_lib = None


def build(force: bool = False) -> Path:
    """Compile libmtq_hip.so for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU."""
    import subprocess

    srcs = list((_PKG / "csrc").glob("*")) + [_PKG.parent / "include" / "mtq.h"]
    stale = not LIB_PATH.exists() or any(p.stat().st_mtime > LIB_PATH.stat().st_mtime for p in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", str(_PKG / "csrc"), "-s"])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise MtqError(
            f"{LIB_PATH} is missing: build it with `make -C {_PKG / 'csrc'}` (hipcc --offload-arch=gfx950). "
            "The hip backend has no CPU fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64; it must be the ONE HIP runtime of the process, so torch is
    # imported before libmtq_hip.so resolves its libamdhip64.so dependency (two runtimes → "no usable device").
    import torch  # noqa: F401

    L = ctypes.CDLL(str(LIB_PATH))
    vp, i64, u32, ci, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint32, ctypes.c_int, ctypes.c_double
    L.mtq_version.restype = ci
    L.mtq_last_error.restype = ctypes.c_char_p
    L.mtq_device_count.argtypes = [ctypes.POINTER(ci)]
    L.mtq_stats_record_doubles.argtypes = [u32]
    L.mtq_stats_record_doubles.restype = ctypes.c_size_t
    L.mtq_tile_stats.argtypes = [vp, ci, i64, i64, i64, u32, vp, vp]
    L.mtq_tile_stats_batched.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, vp, vp]
    L.mtq_tile_stats_partial.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, u32, u32, vp, vp]
    L.mtq_tile_stats_partial_begin.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, u32, u32, vp, vp, ctypes.POINTER(u32), vp]
    L.mtq_tile_stats_partial_end.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, vp, vp, u32, vp]
    L.mtq_quantize.argtypes = [vp, ci, i64, i64, i64, ci, vp, i64, vp]
    L.mtq_apply_assignment.argtypes = [vp, ci, i64, i64, i64, vp, vp, i64, vp]
    L.mtq_dequant_fp8_block.argtypes = [vp, vp, i64, i64, i64, i64, i64, vp, i64, vp]
    L.mtq_greedy_create.argtypes = [ctypes.POINTER(vp), vp, i64, u32, ci, dbl, dbl, ci]
    L.mtq_greedy_pass.argtypes = [vp, ci, vp, i64]
    L.mtq_greedy_assignment.argtypes = [vp, vp]
    L.mtq_greedy_fixed.argtypes = [vp, vp]
    L.mtq_greedy_counts.argtypes = [vp, vp]
    L.mtq_greedy_value.argtypes = [vp, ctypes.POINTER(dbl)]
    L.mtq_greedy_destroy.argtypes = [vp]
    L.mtq_greedy_destroy.restype = None
    L.mtq_tile_scores.argtypes = [vp, i64, u32, ci, vp]
    L.mtq_threshold_assign.argtypes = [vp, i64, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, i64, ctypes.POINTER(i64)]
    L.mtq_columns_from_stats.argtypes = [vp, i64, u32, vp, dbl, vp]
    L.mtq_columns_from_sums.argtypes = [vp, dbl, vp]
    L.mtq_pack_slim_records.argtypes = [vp, i64, u32, vp, vp]
    L.mtq_tile_scores_device.argtypes = [vp, i64, u32, ci, vp, vp]
    L.mtq_threshold_assign_device.argtypes = [vp, i64, u32, vp, ci, ci, dbl, dbl, vp, vp, vp]
    L.mtq_columns_scratch_doubles.argtypes = []
    L.mtq_columns_scratch_doubles.restype = ctypes.c_size_t
    L.mtq_column_sums_device.argtypes = [vp, i64, u32, vp, vp, vp]
    L.mtq_column_sums_device_batched.argtypes = [vp, i64, i64, u32, vp, vp, vp]
    L.mtq_rng_create.argtypes = [ctypes.POINTER(vp), ctypes.c_uint64]
    L.mtq_rng_permutation.argtypes = [vp, i64, vp]
    L.mtq_rng_integers.argtypes = [vp, i64, i64, vp]
    L.mtq_rng_destroy.argtypes = [vp]
    L.mtq_rng_destroy.restype = None
    L.mtq_greedy_run.argtypes = [vp, i64, u32, vp, ci, ci, dbl, dbl, ctypes.c_uint64, vp, vp, vp]
    L.mtq_greedy_run_batch.argtypes = [vp, i64, i64, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, vp, ci]
    L.mtq_greedy_scan_scratch_bytes.argtypes = [i64, i64]
    L.mtq_greedy_scan_scratch_bytes.restype = ctypes.c_size_t
    L.mtq_greedy_scan_device.argtypes = [vp, i64, i64, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, vp, vp, ctypes.c_size_t, vp]
    L.mtq_greedy_scan_device_ex.argtypes = [vp, i64, i64, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, vp, vp, ctypes.c_size_t, vp, ci, vp, vp, vp, vp]
    L.mtq_scan_carry_bytes.argtypes = [i64]
    L.mtq_scan_carry_bytes.restype = ctypes.c_size_t
    L.mtq_scan_orders_bytes.argtypes = [i64]
    L.mtq_scan_orders_bytes.restype = ctypes.c_size_t
    L.mtq_scan_orders_device.argtypes = [ctypes.c_uint64, i64, ci, vp, ctypes.c_size_t, vp]
    L.mtq_tile_stats_listed.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, u32, u32, vp, vp, i64, vp, vp, vp]
    L.mtq_shutdown.restype = ci
    L.mtq_selftest_slot_ring.restype = ci
    L.mtq_device_copy_2d.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, vp]
    L.mtq_knife_tiles_device.argtypes = [vp, ci, i64, i64, i64, i64, i64, vp, vp, ci, i64, vp, vp, vp]
    L.mtq_threshold_enqueue.argtypes = [vp, ci, i64, i64, i64, i64, i64, u32, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.mtq_threshold_columns.argtypes = [vp, i64, i64, u32, vp, vp, vp, vp]
    L.mtq_tile_stats_ragged.argtypes = [vp, ci, ci, u32, vp, vp]
    L.mtq_knife_tiles_ragged.argtypes = [vp, ci, ci, vp, vp, ci, i64, vp, vp, vp]
    L.mtq_column_sums_device_ragged.argtypes = [vp, vp, ci, u32, vp, vp, vp]
    L.mtq_threshold_enqueue_ragged.argtypes = [vp, ci, ci, u32, u32, vp, ci, ci, dbl, dbl, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    L.mtq_threshold_columns_ragged.argtypes = [vp, vp, ci, u32, vp, vp, vp, vp]
    if L.mtq_version() < 141:
        raise MtqError("libmtq_hip.so is older than this package")
    _lib = L
    # torch registered its exit hooks when it was imported above; a hook registered now runs BEFORE them: the library's threads,
    # events and device tables are released while the HIP runtime is still there (mtq_shutdown; nothing is left to static destructors)
    import atexit

    atexit.register(shutdown)
    return L


def shutdown() -> None:
    """mtq_shutdown: joins the scan threads, drains the devices the library used and frees its device tables and events.  Idempotent;
    registered with atexit by lib().  The library sets itself up again if it is used afterwards."""
    if _lib is not None:
        _lib.mtq_shutdown()


def check(rc: int) -> None:
    if rc != 0:
        raise MtqError(f"libmtq_hip error {rc}: {lib().mtq_last_error().decode()}")


def fmt_mask(formats) -> int:
    m = 0
    for f in formats:
        m |= 1 << MIXED_TILE_FORMATS.index(f)
    return m


def mask_formats(mask: int) -> list[str]:
    return [f for i, f in enumerate(MIXED_TILE_FORMATS) if mask & (1 << i)]


MASK_BF16_IDENTITY = 0x10  # include/mtq.h MTQ_MASK_BF16_IDENTITY (host functions only)
MASK_SLIM = 0x20           # include/mtq.h MTQ_MASK_SLIM (pcc greedy scan on 3-double slots)


def parse_cpulist(text: str) -> set:
    """'0-63,128-191' (sysfs cpulist) → set of CPU numbers."""
    cpus = set()
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa_node(device_index: int) -> str:
    """Restrict this process (and the threads it creates later: scan pool, pinned-memory allocation) to the CPUs of the NUMA
    node its GPU hangs off, read from sysfs through the device's PCI address.  The records are DMA-written into pinned host
    memory and then read by the scan threads: on a two-socket host both want that memory on the GPU's socket.  Returns a
    short description; does nothing (and says so) when the topology cannot be read.  MTQ_NUMA_BIND=0 disables it."""
    from .settings import settings

    if not settings().numa_bind or not hasattr(os, "sched_setaffinity"):
        return "off"
    try:
        p = _torch().cuda.get_device_properties(device_index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
            node = int(f.read())
        if node < 0:
            return f"{bdf}: no NUMA node reported"
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            allowed = parse_cpulist(f.read()) & os.sched_getaffinity(0)
        if not allowed:
            return f"{bdf}: node {node} has no allowed CPU"
        os.sched_setaffinity(0, allowed)
        return f"{bdf} -> NUMA node {node} ({len(allowed)} CPUs)"
    except (OSError, ValueError, AttributeError, RuntimeError) as exc:
        return f"unavailable ({type(exc).__name__})"


def record_doubles(mask: int) -> int:
    return 2 + (3 if mask & MASK_SLIM else 5) * bin(mask & 0xF).count("1")


def pack_slim_records(stats_dev, mask: int, out=None):
    """Device copy of K1's records [..., tiles, 2+5F] without Σ|d| and max: [..., tiles, 2+3F] (mtq_pack_slim_records)."""
    torch = _torch()
    require_gpu()
    lead = tuple(stats_dev.shape[:-1])
    T = int(np.prod(lead))
    if out is None:
        out = torch.empty(lead + (record_doubles(mask | MASK_SLIM),), dtype=torch.float64, device=stats_dev.device)
    check(lib().mtq_pack_slim_records(stats_dev.data_ptr(), T, mask & 0xF, out.data_ptr(), _stream_ptr()))
    return out


def tiles_hw(rows: int, cols: int) -> tuple[int, int]:
    return -(-rows // TILE), -(-cols // TILE)


# ----------------------------------------------------------------------------- device helpers

def _torch():
    import torch

    return torch


def require_gpu() -> None:
    torch = _torch()
    if not torch.cuda.is_available():
        raise MtqError("backend 'hip' needs a visible MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    n = ctypes.c_int(0)
    check(lib().mtq_device_count(ctypes.byref(n)))


def _dtype_code(t) -> int:
    torch = _torch()
    if t.dtype == torch.bfloat16:
        return DTYPE_BF16
    if t.dtype == torch.float32:
        return DTYPE_F32
    raise MtqError(f"hip backend takes bfloat16 or float32 tensors, got {t.dtype}")


def _stream_ptr() -> int:
    return _torch().cuda.current_stream().cuda_stream


def _as_device_matrix(t):
    """2-D, last-dim-contiguous device tensor → (tensor, rows, cols, ld)."""
    if t.dim() != 2 or not t.is_cuda:
        raise MtqError("expected a 2-D device tensor (use to_device_2d for host / N-D input)")
    if t.stride(1) != 1:
        t = t.contiguous()
    return t, t.shape[0], t.shape[1], t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def to_device_2d(x, device=None):
    """Host ndarray / torch tensor of any rank → (2-D device tensor, shape_info) following the 2-D
    flatten of compression_algorithms/tile_utils.py:91-107.  bf16 stays bf16, everything else → fp32."""
    torch = _torch()
    device = device or torch.device("cuda", torch.cuda.current_device())
    if isinstance(x, np.ndarray) or np.isscalar(x):
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32)))
        if np.ndim(x) == 0:
            t = t.reshape(())
    else:
        t = x
        if t.dtype not in (torch.bfloat16, torch.float32):
            t = t.to(torch.float32)
    t = t.to(device, non_blocking=True)
    if t.dim() == 0:
        return t.reshape(1, 1), ("scalar", tuple(t.shape))
    if t.dim() == 1:
        n = t.shape[0]
        h = -(-n // TILE)
        d = torch.zeros((h, TILE), dtype=t.dtype, device=device)
        d.view(-1)[:n] = t
        return d, ("vector", n)
    shape = tuple(t.shape)
    return t.reshape(-1, shape[-1]).contiguous(), ("nd", shape)


def unflatten(y2d, shape_info):
    """Inverse of to_device_2d's flatten (tile_utils.py:123-131) for torch or numpy 2-D arrays."""
    kind, v = shape_info
    if kind == "scalar":
        return y2d.reshape(())
    if kind == "vector":
        return y2d.reshape(-1)[:v]
    return y2d.reshape(v)


def tile_stats(x2d, mask: int, out=None):
    """K1 on a 2-D device tensor → device float64 [tiles, 2+5F] (async on the current stream)."""
    torch = _torch()
    require_gpu()
    x2d, rows, cols, ld = _as_device_matrix(x2d)
    th, tw = tiles_hw(rows, cols)
    rec = record_doubles(mask)
    if out is None:
        out = torch.empty((th * tw, rec), dtype=torch.float64, device=x2d.device)
    check(lib().mtq_tile_stats(x2d.data_ptr(), _dtype_code(x2d), rows, cols, ld, mask, out.data_ptr(), _stream_ptr()))
    return out


def tile_stats_batched(x3d, mask: int, out=None):
    """K1 over a (count, rows, cols) contiguous device tensor in one launch → [count, tiles, rec]."""
    torch = _torch()
    require_gpu()
    if x3d.dim() != 3 or not x3d.is_cuda or not x3d.is_contiguous():
        raise MtqError("expected a contiguous (count, rows, cols) device tensor")
    count, rows, cols = x3d.shape
    th, tw = tiles_hw(rows, cols)
    rec = record_doubles(mask)
    if out is None:
        out = torch.empty((count, th * tw, rec), dtype=torch.float64, device=x3d.device)
    check(lib().mtq_tile_stats_batched(x3d.data_ptr(), _dtype_code(x3d), count, rows * cols, rows, cols, cols, mask,
                                       out.data_ptr(), _stream_ptr()))
    return out


RAGGED_MAX = 24   # include/mtq.h MTQ_RAGGED_MAX


class MtqMatrix(ctypes.Structure):
    """include/mtq.h MtqMatrix: one matrix of a ragged batch (device pointer, leading dimension in elements)."""
    _fields_ = [("x", ctypes.c_void_p), ("rows", ctypes.c_int64), ("cols", ctypes.c_int64), ("ld", ctypes.c_int64)]


def ragged_matrices(mats):
    """(MtqMatrix array, storage code, tiles per matrix) of 2-D device tensors of one storage type with contiguous rows."""
    require_gpu()
    if not 0 < len(mats) <= RAGGED_MAX:
        raise MtqError(f"a ragged batch holds 1..{RAGGED_MAX} matrices")
    code = _dtype_code(mats[0])
    arr = (MtqMatrix * len(mats))()
    tiles = []
    for j, m in enumerate(mats):
        if m.dim() != 2 or not m.is_cuda or m.stride(1) != 1 or _dtype_code(m) != code or m.device != mats[0].device:
            raise MtqError("a ragged batch is 2-D tensors of one storage type on one device with contiguous rows")
        arr[j] = MtqMatrix(m.data_ptr(), m.shape[0], m.shape[1], m.stride(0))
        th, tw = tiles_hw(m.shape[0], m.shape[1])
        tiles.append(th * tw)
    return arr, code, tiles


def tile_stats_ragged(mats, mask: int, out=None):
    """K1 over matrices of ANY shapes (one storage type) in one launch (mtq_tile_stats_ragged) → [sum of their tiles, rec], matrix j's
    tiles row-major behind matrix j-1's."""
    torch = _torch()
    arr, code, tiles = ragged_matrices(mats)
    if out is None:
        out = torch.empty((sum(tiles), record_doubles(mask)), dtype=torch.float64, device=mats[0].device)
    check(lib().mtq_tile_stats_ragged(arr, len(mats), code, mask, out.data_ptr(), _stream_ptr()))
    return out


def tile_stats_partial(x3d, layout_mask: int, full_mask: int, sums_mask: int, out=None):
    """K1 over a (count, rows, cols) device tensor with only part of every record promised (mtq_tile_stats_partial): the five statistics of
    the formats in full_mask, Σy, Σy², Σxy of those in sums_mask; the rest of the layout is unspecified → [count, tiles, rec(layout)]."""
    torch = _torch()
    require_gpu()
    if x3d.dim() != 3 or not x3d.is_cuda or not x3d.is_contiguous():
        raise MtqError("expected a contiguous (count, rows, cols) device tensor")
    count, rows, cols = x3d.shape
    th, tw = tiles_hw(rows, cols)
    if out is None:
        out = torch.empty((count, th * tw, record_doubles(layout_mask)), dtype=torch.float64, device=x3d.device)
    check(lib().mtq_tile_stats_partial(x3d.data_ptr(), _dtype_code(x3d), count, rows * cols, rows, cols, cols, layout_mask, full_mask, sums_mask,
                                       out.data_ptr(), _stream_ptr()))
    return out


def tile_stats_partial_begin(x3d, layout_mask: int, full_mask: int, sums_mask: int, out, mark) -> int:
    """mtq_tile_stats_partial_begin on the current stream (the exact-integer kernel alone) → the launch id tile_stats_partial_end wants.
    mark: int32 device tensor of one element that stays the caller's until _end has run."""
    count, rows, cols = x3d.shape
    lid = ctypes.c_uint32(0)
    check(lib().mtq_tile_stats_partial_begin(x3d.data_ptr(), _dtype_code(x3d), count, rows * cols, rows, cols, cols, layout_mask, full_mask, sums_mask,
                                             out.data_ptr(), mark.data_ptr(), ctypes.byref(lid), _stream_ptr()))
    return int(lid.value)


def tile_stats_partial_end(x3d, layout_mask: int, stats, mark, launch_id: int) -> None:
    """mtq_tile_stats_partial_end on the current stream: the literal fix-up of the tiles that launch could not take (usually none)."""
    count, rows, cols = x3d.shape
    check(lib().mtq_tile_stats_partial_end(x3d.data_ptr(), _dtype_code(x3d), count, rows * cols, rows, cols, cols, layout_mask, stats.data_ptr(),
                                           mark.data_ptr(), int(launch_id), _stream_ptr()))


def quantize(x2d, fmt: str, out=None):
    """K2 on a 2-D device tensor → device float32 (rows, cols)."""
    torch = _torch()
    require_gpu()
    if fmt not in FMT_CODE:
        raise ValueError(f"Unsupported weight format: {fmt}")
    x2d, rows, cols, ld = _as_device_matrix(x2d)
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x2d.device)
    check(lib().mtq_quantize(x2d.data_ptr(), _dtype_code(x2d), rows, cols, ld, FMT_CODE[fmt], out.data_ptr(), out.stride(0), _stream_ptr()))
    return out


def apply_assignment(x2d, assignment, out=None):
    """K3: assignment is an int8 (tiles_h, tiles_w) numpy array or device tensor."""
    torch = _torch()
    require_gpu()
    x2d, rows, cols, ld = _as_device_matrix(x2d)
    th, tw = tiles_hw(rows, cols)
    if isinstance(assignment, np.ndarray):
        assignment = torch.from_numpy(np.ascontiguousarray(assignment, dtype=np.int8)).to(x2d.device)
    a = assignment.to(torch.int8).contiguous()
    if a.numel() != th * tw:
        raise MtqError(f"assignment has {a.numel()} entries, tensor has {th}x{tw} tiles")
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.float32, device=x2d.device)
    check(lib().mtq_apply_assignment(x2d.data_ptr(), _dtype_code(x2d), rows, cols, ld, a.data_ptr(), out.data_ptr(), out.stride(0), _stream_ptr()))
    return out


def dequant_fp8_block(w, scale_inv):
    """K5: w — 2-D device tensor of float8_e4m3fn (or its uint8 bytes); scale_inv — 2-D float32 block scales →
    float32 device tensor w.float() * scale_inv.repeat_interleave(block) (hf_model_utils.py:209-215)."""
    torch = _torch()
    require_gpu()
    if w.dim() != 2 or scale_inv.dim() != 2 or not w.is_cuda:
        raise MtqError("expected 2-D device tensors")
    wb = w.view(torch.uint8).contiguous()
    sc = scale_inv.to(device=w.device, dtype=torch.float32).contiguous()
    out = torch.empty(wb.shape, dtype=torch.float32, device=w.device)
    check(lib().mtq_dequant_fp8_block(wb.data_ptr(), sc.data_ptr(), wb.shape[0], wb.shape[1], wb.stride(0), sc.shape[0], sc.shape[1],
                                      out.data_ptr(), out.stride(0), _stream_ptr()))
    return out


# ----------------------------------------------------------------------------- host decisions

class GreedyScan:
    """H1: the sequential scan of mixed_tile_greedy.py:133-346 on a host copy of the stats."""

    def __init__(self, stats: np.ndarray, mask: int, metric: str, threshold: float, elem_count: float, base_fmt: str):
        self.stats = np.ascontiguousarray(stats, dtype=np.float64)  # kept alive: the handle reads it
        self.T = self.stats.shape[0]
        self._h = ctypes.c_void_p()
        check(lib().mtq_greedy_create(ctypes.byref(self._h), self.stats.ctypes.data, self.T, mask, METRIC_CODE[metric],
                                      float(threshold), float(elem_count), MIXED_TILE_FORMATS.index(base_fmt)))

    def run_pass(self, fmt: str, order: np.ndarray) -> None:
        order = np.ascontiguousarray(order, dtype=np.int64)
        check(lib().mtq_greedy_pass(self._h, MIXED_TILE_FORMATS.index(fmt), order.ctypes.data, order.size))

    def fixed(self) -> np.ndarray:
        out = np.empty(self.T, dtype=np.uint8)
        check(lib().mtq_greedy_fixed(self._h, out.ctypes.data))
        return out

    def assignment(self) -> np.ndarray:
        out = np.empty(self.T, dtype=np.int8)
        check(lib().mtq_greedy_assignment(self._h, out.ctypes.data))
        return out

    def counts(self) -> dict[str, int]:
        c = (ctypes.c_int64 * 4)()
        check(lib().mtq_greedy_counts(self._h, c))
        return {f: int(c[i]) for i, f in enumerate(MIXED_TILE_FORMATS)}

    def value(self) -> float:
        v = ctypes.c_double()
        check(lib().mtq_greedy_value(self._h, ctypes.byref(v)))
        return v.value

    def close(self) -> None:
        if self._h:
            lib().mtq_greedy_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NumpyCompatRng:
    """mtq_rng: bit-compatible with np.random.default_rng(seed).permutation(n) (tests pin it against NumPy)."""

    def __init__(self, seed: int):
        self._h = ctypes.c_void_p()
        check(lib().mtq_rng_create(ctypes.byref(self._h), int(seed)))

    def permutation(self, n: int) -> np.ndarray:
        out = np.empty(int(n), dtype=np.int64)
        check(lib().mtq_rng_permutation(self._h, int(n), out.ctypes.data))
        return out

    def integers(self, high: int, n: int) -> np.ndarray:
        """≡ rng.integers(0, high, size=n, dtype=np.int64)."""
        out = np.empty(int(n), dtype=np.int64)
        check(lib().mtq_rng_integers(self._h, int(high), int(n), out.ctypes.data))
        return out

    def __del__(self):
        try:
            if self._h:
                lib().mtq_rng_destroy(self._h)
        except Exception:
            pass


def greedy_run(stats: np.ndarray, mask: int, formats, metric: str, threshold: float, elem_count: float, seed: int):
    """H1 end to end in one GIL-free call → (int8[T] map, counts dict, columns dict)."""
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    T = stats.shape[0]
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    amap = np.empty(T, dtype=np.int8)
    counts = (ctypes.c_int64 * 4)()
    out = (ctypes.c_double * 9)()
    check(lib().mtq_greedy_run(stats.ctypes.data, T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold),
                               float(elem_count), int(seed), amap.ctypes.data, counts, out))
    return amap, {f: int(counts[i]) for i, f in enumerate(MIXED_TILE_FORMATS)}, {"pcc": out[0], "mae": out[1], "atol": out[2], "sums": tuple(out[3:9])}


SCAN_DEVICE_MAX_TILES = 1 << 22
SCAN_LDS_MAX_TILES = 32768      # csrc/mtq_scan.hip kScanMaxTilesLds: visiting order in LDS up to here, in global scratch above


def device_scan_supported(formats, metric: str, tiles: int) -> bool:
    """What mtq_greedy_scan_device serves (include/mtq.h): the three metrics, distinct formats, tiles up to SCAN_DEVICE_MAX_TILES."""
    return metric in ("pcc", "mae", "atol") and len(set(formats)) == len(formats) and 0 < tiles <= SCAN_DEVICE_MAX_TILES


def device_copy(dst, src) -> None:
    """dst ← src by a kernel on the current stream (mtq_device_copy_2d); dst may be a pinned host tensor, src a device tensor.  Both
    either contiguous with the same number of bytes, or 2-D / 3-D views whose rows (last dimension) are contiguous and whose leading
    dimensions collapse to one pitch."""
    import torch

    if dst.dtype != src.dtype or dst.shape != src.shape:
        raise ValueError("device_copy needs equal shapes and dtypes")
    if dst.numel() == 0:
        return
    esz = dst.element_size()

    def rows_of(t):
        if t.is_contiguous():
            return t.numel() * esz, 1, t.numel() * esz
        if t.dim() < 2 or t.stride(-1) != 1:
            raise ValueError("device_copy needs contiguous rows")
        lead = t.reshape(-1, t.shape[-1]) if all(t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2)) else None
        if lead is None or lead.data_ptr() != t.data_ptr():
            raise ValueError("device_copy needs one pitch over the leading dimensions")
        return t.shape[-1] * esz, lead.shape[0], lead.stride(0) * esz

    w_d, r_d, p_d = rows_of(dst)
    w_s, r_s, p_s = rows_of(src)
    if r_d == 1 and r_s > 1:      # contiguous on one side: its rows are the other side's rows
        w_d, r_d, p_d = w_s, r_s, w_s
    if r_s == 1 and r_d > 1:
        w_s, r_s, p_s = w_d, r_d, w_d
    check(lib().mtq_device_copy_2d(dst.data_ptr(), p_d, src.data_ptr(), p_s, w_d, r_d, _stream_ptr()))


def greedy_scan_device(stats_dev, mask: int, formats, metric: str, threshold: float, elem_count: float, seeds_dev, maps_out=None,
                       status_out=None, scratch=None, counts_out=None):
    """H1 on the device over FULL records [count, tiles, rec] where K1 wrote them → (int8 [count, tiles] maps, int32 [count]
    status) device tensors, asynchronous on the current stream.  seeds_dev: uint64/int64 device tensor [count]."""
    torch = _torch()
    count, T = int(stats_dev.shape[0]), int(stats_dev.shape[1])
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    maps = maps_out if maps_out is not None else torch.empty((count, T), dtype=torch.int8, device=stats_dev.device)
    status = status_out if status_out is not None else torch.empty((count,), dtype=torch.int32, device=stats_dev.device)
    need = int(lib().mtq_greedy_scan_scratch_bytes(count, T))
    if scratch is None:
        scratch = torch.empty((need,), dtype=torch.uint8, device=stats_dev.device)
    elif scratch.numel() < need:   # a caller's slicing bug must not turn into a device allocation per call
        raise ValueError(f"scratch holds {scratch.numel()} bytes, mtq_greedy_scan_scratch_bytes() asks for {need}")
    check(lib().mtq_greedy_scan_device(stats_dev.data_ptr(), count, T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold),
                                       float(elem_count), seeds_dev.data_ptr(), maps.data_ptr(), status.data_ptr(),
                                       counts_out.data_ptr() if counts_out is not None else None, scratch.data_ptr(), int(scratch.numel()), _stream_ptr()))
    return maps, status


def scan_orders_device(seed: int, tiles: int, n_orders: int = 2, out=None):
    """The visiting orders every tensor of a launch shares (mtq_scan_orders_device): generator states and the permutations of
    range(tiles) of passes 1 (and 2) for `seed` → uint8 device buffer, asynchronous on the current stream."""
    torch = _torch()
    require_gpu()
    need = int(lib().mtq_scan_orders_bytes(int(tiles)))
    if out is None:
        out = torch.empty((need,), dtype=torch.uint8, device="cuda")
    elif out.numel() < need:
        raise ValueError("orders buffer is smaller than mtq_scan_orders_bytes()")
    check(lib().mtq_scan_orders_device(int(seed), int(tiles), int(n_orders), out.data_ptr(), int(out.numel()), _stream_ptr()))
    return out


def greedy_scan_device_ex(stats_dev, mask: int, formats, metric: str, threshold: float, elem_count: float, seeds_dev, maps_out, status_out,
                          scratch, counts_out=None, orders=None, phase: int = 0, listed=None, n_listed=None, carry=None) -> None:
    """mtq_greedy_scan_device_ex: the device search with shared visiting orders (orders: scan_orders_device's buffer for the seed all
    tensors share) and / or in phases (1: every pass but the last + the last pass's candidates → listed / n_listed, state → carry;
    2: the last pass).  Caller-owned buffers throughout; scratch must hold mtq_greedy_scan_scratch_bytes(count, tiles) bytes."""
    count, T = int(stats_dev.shape[0]), int(stats_dev.shape[1])
    need = int(lib().mtq_greedy_scan_scratch_bytes(count, T))
    if scratch.numel() < need:
        raise ValueError("scratch is smaller than mtq_greedy_scan_scratch_bytes()")
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    check(lib().mtq_greedy_scan_device_ex(stats_dev.data_ptr(), count, T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold), float(elem_count),
                                          seeds_dev.data_ptr(), maps_out.data_ptr(), status_out.data_ptr(),
                                          counts_out.data_ptr() if counts_out is not None else None, scratch.data_ptr(), int(scratch.numel()),
                                          orders.data_ptr() if orders is not None else None, int(phase),
                                          listed.data_ptr() if listed is not None else None, n_listed.data_ptr() if n_listed is not None else None,
                                          carry.data_ptr() if carry is not None else None, _stream_ptr()))


def tile_stats_listed(x3d, layout_mask: int, full_mask: int, err_mask: int, listed, n_listed, stats, scratch=None) -> None:
    """mtq_tile_stats_listed: for the tiles listed[0 .. n_listed[0]) (device uint32 / int32 tensors; entries tensor * tiles + tile) the five
    statistics of full_mask's formats and Σ|x−y|, max|x−y| of err_mask's, into stats [count, tiles, rec(layout)] in place.  scratch:
    int32 device tensor of listed.numel() + 1 entries (lets bf16 input take the exact-integer kernel), or None."""
    require_gpu()
    if x3d.dim() != 3 or not x3d.is_cuda or not x3d.is_contiguous():
        raise MtqError("expected a contiguous (count, rows, cols) device tensor")
    count, rows, cols = x3d.shape
    check(lib().mtq_tile_stats_listed(x3d.data_ptr(), _dtype_code(x3d), count, rows * cols, rows, cols, cols, layout_mask, full_mask, err_mask,
                                      listed.data_ptr(), n_listed.data_ptr(), int(listed.numel()),
                                      scratch.data_ptr() if scratch is not None else None, stats.data_ptr(), _stream_ptr()))


def greedy_run_batch(stats: np.ndarray, mask: int, formats, metric: str, threshold: float, elem_count: float, seeds, n_threads: int):
    """mtq_greedy_run over a [count, tiles, rec] record array on n_threads host threads (one GIL-free call)
    → (int8 [count, tiles] maps, int64 [count, 4] counts, float64 [count, 9] columns+sums)."""
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    count, T = stats.shape[0], stats.shape[1]
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    sd = np.ascontiguousarray(seeds, dtype=np.uint64)
    maps = np.empty((count, T), dtype=np.int8)
    counts = np.empty((count, 4), dtype=np.int64)
    outs = np.empty((count, 9), dtype=np.float64)
    check(lib().mtq_greedy_run_batch(stats.ctypes.data, count, T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold),
                                     float(elem_count), sd.ctypes.data, maps.ctypes.data, counts.ctypes.data, outs.ctypes.data, int(n_threads)))
    return maps, counts, outs


def tile_scores(stats: np.ndarray, mask: int, metric: str) -> np.ndarray:
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    T = stats.shape[0]
    rows = bin(mask & 0xF).count("1") + (1 if (mask & MASK_BF16_IDENTITY) and not (mask & 1) else 0)  # identity bf16 comes first
    out = np.empty((rows, T), dtype=np.float64)
    check(lib().mtq_tile_scores(stats.ctypes.data, T, mask, METRIC_CODE[metric], out.ctypes.data))
    return out


def threshold_assign(stats: np.ndarray, mask: int, formats, metric: str, threshold: float, band: float = 2e-6, with_near: bool = False):
    """K4 on host stats → (int8[T] map, knife-edge tile ids[, uint8 masks of the format codes inside the band per id])."""
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    T = stats.shape[0]
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    amap = np.empty(T, dtype=np.int8)
    knife = np.empty(T, dtype=np.int64)
    near = np.empty(T, dtype=np.uint8)
    nk = ctypes.c_int64(0)
    check(lib().mtq_threshold_assign(stats.ctypes.data, T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold), float(band),
                                     amap.ctypes.data, knife.ctypes.data, near.ctypes.data, T, ctypes.byref(nk)))
    k = min(nk.value, T)
    return (amap, knife[:k].copy(), near[:k].copy()) if with_near else (amap, knife[:k].copy())


def _score_rows(mask: int) -> int:
    return bin(mask & 0xF).count("1") + (1 if (mask & MASK_BF16_IDENTITY) and not (mask & 1) else 0)


def tile_scores_device(stats_dev, mask: int, metric: str):
    """mtq_tile_scores on device-resident records [tiles, rec] → device float64 [formats, tiles] (async on the current stream)."""
    torch = _torch()
    require_gpu()
    T = stats_dev.shape[0]
    out = torch.empty((_score_rows(mask), T), dtype=torch.float64, device=stats_dev.device)
    check(lib().mtq_tile_scores_device(stats_dev.data_ptr(), T, mask, METRIC_CODE[metric], out.data_ptr(), _stream_ptr()))
    return out


def threshold_assign_device_raw(stats_dev, mask: int, formats, metric: str, threshold: float, band: float = 2e-6, out=None):
    """K4 on device-resident records [T, rec] (any number of tensors' tiles back to back) → device int8 [2, T]: row 0 the
    map, row 1 the knife-edge masks (bit c: format code c scored inside the band; 0 for most tiles); asynchronous on the
    current stream.  `out`: a pair of contiguous int8 device vectors of T entries each (map, masks) to write instead."""
    torch = _torch()
    require_gpu()
    T = stats_dev.shape[0]
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    both = torch.empty((2, T), dtype=torch.int8, device=stats_dev.device) if out is None else out
    for row in (both[0], both[1]):
        if row.dtype != torch.int8 or row.numel() != T or not row.is_contiguous() or not row.is_cuda:
            raise ValueError("threshold_assign_device_raw: out must be two contiguous int8 device vectors of one entry per tile")
    check(lib().mtq_threshold_assign_device(stats_dev.data_ptr(), T, mask, fm, len(formats), METRIC_CODE[metric], float(threshold), float(band),
                                            both[0].data_ptr(), both[1].data_ptr(), _stream_ptr()))
    return both


def knife_tiles_device(x3d, near, formats, cap: int, list_out, tiles_out) -> None:
    """The threshold rule's knife-edge tiles, prepared on the device (mtq_knife_tiles_device): `near` = the int8 masks of
    threshold_assign_device_raw for the (count, rows, cols) batch x3d; list_out int64 [cap + 1] ← flat tile ids (any order) and,
    last, how many were flagged; tiles_out float32 [1 + len(formats), cap, 32, 32] ← their values and every format's reconstruction.
    Asynchronous on the current stream."""
    torch = _torch()
    require_gpu()
    count, rows, cols = x3d.shape
    if x3d.stride(2) != 1 or not x3d.is_cuda:
        raise ValueError("knife_tiles_device needs a device tensor with contiguous rows")
    th, tw = tiles_hw(rows, cols)
    if near.dtype != torch.int8 or near.numel() != count * th * tw or not near.is_contiguous():
        raise ValueError("near must be a contiguous int8 vector of one entry per tile")
    if list_out.dtype != torch.int64 or list_out.numel() != cap + 1 or not list_out.is_contiguous():
        raise ValueError("list_out must be a contiguous int64 vector of cap + 1 entries")
    if cap and (tiles_out.dtype != torch.float32 or tiles_out.numel() != (1 + len(formats)) * cap * 1024 or not tiles_out.is_contiguous()):
        raise ValueError("tiles_out must be a contiguous float32 tensor of (1 + formats) x cap x 32 x 32")
    fm = (ctypes.c_int * len(formats))(*[MIXED_TILE_FORMATS.index(f) for f in formats])
    check(lib().mtq_knife_tiles_device(x3d.data_ptr(), _dtype_code(x3d), count, x3d.stride(0) if count > 1 else rows * x3d.stride(1), rows, cols, x3d.stride(1),
                                       near.data_ptr(), fm, len(formats), int(cap), list_out.data_ptr(), tiles_out.data_ptr() if cap else None, _stream_ptr()))


def threshold_assign_device(stats_dev, mask: int, formats, metric: str, threshold: float, band: float = 2e-6, with_near: bool = False):
    """K4 on device-resident records → (int8[T] map on the host, knife-edge tile ids[, their near masks]): only T + T bytes
    cross PCIe."""
    host = threshold_assign_device_raw(stats_dev, mask, formats, metric, threshold, band).cpu().numpy()
    ids = np.flatnonzero(host[1]).astype(np.int64)
    return (host[0].copy(), ids, host[1][ids].astype(np.uint8)) if with_near else (host[0].copy(), ids)


def columns_from_sums(sums7: np.ndarray, elem_count: float) -> dict:
    """pcc / mae / atol from Σx, Σx², Σy, Σy², Σxy, Σ|d|, max|d| (mtq_columns_from_sums)."""
    sums = np.ascontiguousarray(sums7, dtype=np.float64)
    out = (ctypes.c_double * 9)()
    check(lib().mtq_columns_from_sums(sums.ctypes.data, float(elem_count), out))
    return {"pcc": out[0], "mae": out[1], "atol": out[2], "sums": tuple(out[3:9])}


def columns_from_stats_device(stats_dev, mask: int, assignment, elem_count: float) -> dict:
    """Tensor-level pcc / mae / atol of the reconstruction `assignment` implies, summed on the device from device-resident
    records (fixed tree order); the map goes up (1 B/tile), seven doubles come back."""
    torch = _torch()
    require_gpu()
    T = stats_dev.shape[0]
    if isinstance(assignment, np.ndarray):
        amap = torch.from_numpy(np.ascontiguousarray(assignment, dtype=np.int8).reshape(-1)).to(stats_dev.device)
    else:
        amap = assignment.reshape(-1).to(device=stats_dev.device, dtype=torch.int8).contiguous()
    if amap.numel() != T:
        raise MtqError("assignment has the wrong number of tiles")
    scratch = torch.empty(int(lib().mtq_columns_scratch_doubles()), dtype=torch.float64, device=stats_dev.device)
    check(lib().mtq_column_sums_device(stats_dev.data_ptr(), T, mask, amap.data_ptr(), scratch.data_ptr(), _stream_ptr()))
    sums = np.ascontiguousarray(scratch[:7].cpu().numpy())
    if np.isnan(sums[0]) and not np.isnan(sums[1]):
        raise MtqError("map names a format that is not in fmt_mask")
    out = (ctypes.c_double * 9)()
    check(lib().mtq_columns_from_sums(sums.ctypes.data, float(elem_count), out))
    return {"pcc": out[0], "mae": out[1], "atol": out[2], "sums": tuple(out[3:9])}


def columns_from_stats(stats: np.ndarray, mask: int, assignment: np.ndarray, elem_count: float) -> dict:
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    a = np.ascontiguousarray(assignment, dtype=np.int8).reshape(-1)
    out = (ctypes.c_double * 9)()
    check(lib().mtq_columns_from_stats(stats.ctypes.data, stats.shape[0], mask, a.ctypes.data, float(elem_count), out))
    return {"pcc": out[0], "mae": out[1], "atol": out[2], "sums": tuple(out[3:9])}
