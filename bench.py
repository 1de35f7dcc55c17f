#!/usr/bin/env python3
"""bench.py — mixed-tile-greedy (bf16 → BFP{8,4,2}) throughput on MI355X.

Step  = one pass of the hot path (K1 tile_stats + the sequential greedy scan, both on the GPU by default [--scan host: stats
        D2H + host scan threads] → per-tile assignment maps + pcc/mae/atol on the host) over a batch of `--tensors` (128) synthetic 4096x4096 bf16 tensors
        (BASELINE.json configs[1], streamed; the batch is > 256 MiB so the Infinity Cache cannot hold it).
value = tiles/s, whole job, inputs resident in HBM when the timed region starts.
roofline = the K1 kernel alone: algorithmic 2048 B read per tile / HIP-event launch duration vs 8 TB/s.
cpu_baseline = the C oracle (oracle/, a port of the reference's CPU path) on rank 0's host, 1 thread, on a bounded sample of
        the same tensors; cpu_baseline_threads = the same port on every core of the rank's CPU quota; cpu_baseline_emulation =
        the NumPy host backend (how the reference executes) on one tensor.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python bench.py --gpus N ...            (N > 1, not under a launcher: starts its own N ranks before any GPU call)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch

ROWS = COLS = 4096
FORMATS = ["bf16", "bfp8", "bfp4", "bfp2"]
METRIC, THRESHOLD, SEED = "pcc", 0.999, 123
HBM_PEAK_GBS = 8000.0          # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_TILE_READ = 2048     # 32*32 bf16, SURVEY §8(d)


def make_batch(n: int, rank: int, device) -> torch.Tensor:
    """N(0, 0.02²) rounded to bf16, one seed per tensor (generated on the device)."""
    g = torch.Generator(device=device)
    out = torch.empty((n, ROWS, COLS), dtype=torch.bfloat16, device=device)
    for i in range(n):
        g.manual_seed(1000 * rank + i)
        out[i] = (torch.randn((ROWS, COLS), generator=g, device=device, dtype=torch.float32) * 0.02).to(torch.bfloat16)
    return out


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sample: torch.Tensor, threads: int) -> dict:
    """Three CPU figures on rank 0's host, each timed like wq:680-682 times algo.run (perf_counter around the whole search):
      cpu_baseline            the C port (oracle/mtq_oracle.c), 1 thread, `len(sample)` tensors            kind "port"
      cpu_baseline_threads    the same port, one tensor per thread on `threads` threads (the rank's CPU quota) kind "port"
      cpu_baseline_emulation  the package's NumPy host backend (`--backend emulation`: how the reference executes — single-threaded
                              NumPy, quantize per format + per-tile sums + sequential scan, wq:680-682), one tensor  kind "emulation"
    All three produce the same maps (asserted)."""
    import concurrent.futures as cf

    import numpy as np

    from oracle import mtq_oracle as orc

    xs = [sample[i].float().cpu().numpy() for i in range(sample.shape[0])]
    orc.lib()
    t0 = time.perf_counter()
    tiles = 0
    maps = []
    for x in xs:
        a, _c, _s = orc.greedy(x, FORMATS, METRIC, THRESHOLD, SEED)
        tiles += a.size
        maps.append(a)
    dt = time.perf_counter() - t0
    out = {"cpu_baseline": {"value": tiles / dt, "unit": "tiles/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                            "sample": f"{len(xs)} of the step's 4096x4096 bf16 tensors, oracle/mtq_oracle.c greedy (1 thread), {dt:.1f} s"}}
    if threads > 1:
        reps = max(1, -(-2 * threads // len(xs)))          # at least two tensors per thread
        work = (xs * reps)[: max(2 * threads, len(xs))]
        t0 = time.perf_counter()
        with cf.ThreadPoolExecutor(max_workers=threads) as pool:   # ctypes releases the GIL inside the C port
            got = list(pool.map(lambda x: orc.greedy(x, FORMATS, METRIC, THRESHOLD, SEED)[0], work))
        dtt = time.perf_counter() - t0
        assert all(np.array_equal(g, maps[i % len(xs)]) for i, g in enumerate(got))
        out["cpu_baseline_threads"] = {"value": sum(g.size for g in got) / dtt, "unit": "tiles/s", "cores": threads, "kind": "port", "cpu": cpu_model(),
                                       "sample": f"{len(work)} tensor searches (the same {len(xs)} tensors, repeated) over {threads} threads, one tensor per thread, {dtt:.1f} s"}
    from quantization_analysis_amd.compression_algorithms import create_algorithm
    from quantization_analysis_amd.compression_algorithms.cache import CacheContext
    from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer

    algo = create_algorithm("mixed-tile-greedy", {"metric": METRIC, "threshold": THRESHOLD, "seed": SEED})
    t0 = time.perf_counter()
    res = algo.run(xs[0], FORMATS, Quantizer("emulation"), CacheContext(ROOT / "gpurun_out" / "bench-cache", "bench", "emulation", True, "bench"))[0]
    dte = time.perf_counter() - t0
    assert np.array_equal(res.meta["assignment"], maps[0])
    out["cpu_baseline_emulation"] = {"value": maps[0].size / dte, "unit": "tiles/s", "cores": 1, "kind": "emulation", "cpu": cpu_model(),
                                     "sample": f"1 of the step's tensors through `--backend emulation` (NumPy, y materialised as wq does), {dte:.1f} s"}
    return out


def gather_summary(rows: torch.Tensor, seconds: float, dist, rank: int, world: int):
    """The job's only data-path collective: every rank's fixed-width float64 summary rows to rank 0, and the
    MAX over ranks of the timed region.  → (all rows as ndarray on rank 0 else None, max seconds).  Works on any backend
    (RCCL for the GPU job, gloo in the CPU test)."""
    t_max = torch.tensor([seconds], dtype=torch.float64, device=rows.device)
    if dist is None:
        return rows.cpu().numpy(), seconds
    gathered = [torch.empty_like(rows) for _ in range(world)] if rank == 0 else None
    dist.gather(rows, gathered, dst=0)
    dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    return (torch.cat(gathered).cpu().numpy() if rank == 0 else None), float(t_max.item())


from quantization_analysis_amd.pipeline import cpu_budget, default_workers  # noqa: E402  (no GPU call at import)


def self_launch(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside a launcher, N > 1: start N ranks (one process per GPU) as children of
    `python -m torch.distributed.run` and return its exit code.  Runs BEFORE this process has made any GPU call, and the
    parent never makes one: a process that has touched the GPU is never re-exec'ed (the ranks are fresh children).
    Rank 0 of the child job prints the one JSON line; stdout/stderr pass through."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + argv
    return subprocess.run(cmd, env=env).returncode


def dry_run(args) -> None:
    """--dry-run: the rank plumbing of this file with no GPU and no measurement (CPU test of the launcher branch and of the
    job's only collective): gloo instead of RCCL, made-up summary rows, `value` null."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if "RANK" in os.environ:
        import torch.distributed as dist

        dist.init_process_group("gloo")
    rows = torch.full((2, 11), float(rank), dtype=torch.float64)
    all_rows, dt = gather_summary(rows, 1.0 + rank, dist, rank, world)
    if rank == 0:
        print(json.dumps({"dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ranks_seen": sorted({int(v) for v in all_rows[:, 1]}), "max_seconds": dt,
                          "config": {"sharding": f"tensors x{world}, RCCL gather of summary rows"}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--tensors", type=int, default=128, help="4096x4096 bf16 tensors per step per GPU (128 = 4 GiB, SURVEY §8(d) M1 stream)")
    ap.add_argument("--chunk", type=int, default=None,
                    help="tensors per K1 launch (default: the whole step's batch with the scan on the device — one K1 launch and one scan launch per "
                         "step; 32 with the host scan, where a chunk's records cross PCIe while the next chunk's K1 runs)")
    ap.add_argument("--workers", type=int, default=default_workers(), help="host scan threads per rank (default: this rank's share of the cgroup CPU quota / affinity mask, at most 32)")
    ap.add_argument("--cpu-sample", type=int, default=24, help="tensors timed on the CPU port, ~0.5 s each (0 = skip)")
    ap.add_argument("--scan", choices=["auto", "host", "device"], default="auto",
                    help="where the sequential greedy scan runs: device (csrc/mtq_scan.hip, records never leave the GPU), host (records over PCIe, "
                         "host scan threads), auto = device where it serves the search")
    ap.add_argument("--dry-run", action="store_true", help="rank plumbing only, on the CPU over gloo (tests); prints no measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))   # nothing above this line touches the GPU
    if args.dry_run:
        return dry_run(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from quantization_analysis_amd.hip_backend import bind_to_gpu_numa_node

    numa = bind_to_gpu_numa_node(local_rank)
    dist = None
    if "RANK" in os.environ:  # launched by torch.distributed.run (any N, also N = 1): one process per GPU over RCCL
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)  # nccl == RCCL on ROCm

    from quantization_analysis_amd import hip_backend as hb
    from quantization_analysis_amd.pipeline import GreedyPipeline

    hb.require_gpu()
    batch = make_batch(args.tensors, rank, device)
    pipe = GreedyPipeline(FORMATS, METRIC, THRESHOLD, SEED, chunk=args.chunk or 32, workers=args.workers, scan=args.scan)
    if args.chunk is None:
        args.chunk = args.tensors if pipe.device_scan else 32
    pipe.chunk = args.chunk
    tiles_per_step = args.tensors * (ROWS // 32) * (COLS // 32)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # K1 alone (no scan, no copies beside it): the same launch the steps issue, HIP events on the current stream
    alone_out = pipe.launch_k1(batch[: args.chunk])     # the launch the route issues (round 3: partial records on the lazy route)
    torch.cuda.synchronize()
    alone = []
    for _ in range(5):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        pipe.launch_k1(batch[: args.chunk], out=alone_out)
        a1.record()
        a1.synchronize()
        alone.append(a0.elapsed_time(a1))
    k1_alone_ms = sorted(alone)[len(alone) // 2]
    del alone_out
    pipe.reserve(batch)  # record buffers of both slots + scan threads: allocations, not steps
    # a generation-2 collection of the interpreter's heap (torch's module graph: ~40 ms here) would land inside a 7 ms
    # step every few dozen steps: collect now and move what exists to the permanent generation.  BEFORE the warm-up steps, not
    # between them and the timed ones: the GPU idles while the collector runs, its clocks fall, and the first six launches of the
    # timed region then ran 5-12 % slower (2.7-2.9 ms against 2.5: visible in a 20-step run)
    import gc

    gc.collect()
    gc.freeze()
    res = pipe.run_steps(batch for _ in range(args.warmup))
    pipe.timing.drain()
    pipe.timing.__init__()
    barrier()
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    res = pipe.run_steps(batch for _ in range(args.steps))  # every step fully processed; step s+1's GPU work overlaps step s's scan tail
    barrier()
    dt = time.perf_counter() - t0
    host_cpu_ms = (time.process_time() - cpu0) / args.steps * 1e3   # CPU time of every thread of this rank, per step
    pipe.timing.drain()

    # the only data-path collective: per-tensor summary rows to rank 0; outside the timed steps
    # the rows of the LAST step are gathered so the multi-GPU path is exercised end to end.
    rows = torch.tensor([[r.index, ROWS * COLS, r.pcc, r.mae, r.atol, r.tile_bytes, r.counts["bf16"], r.counts["bfp8"],
                          r.counts["bfp4"], r.counts["bfp2"], 0.0] for r in res], dtype=torch.float64, device=device)
    all_rows, dt = gather_summary(rows, dt, dist, rank, world)

    if rank == 0:
        kt = pipe.timing
        k_ms = kt.kernel_ms / max(kt.launches, 1)
        tiles_per_launch = kt.tiles / max(kt.launches, 1)
        achieved = BYTES_PER_TILE_READ * tiles_per_launch / (k_ms * 1e-3) / 1e9
        traffic = None
        tfile = ROOT / "profiles" / "k1_traffic.json"  # HBM bytes per launch from a separate rocprofv3 --pmc pass
        if tfile.exists():
            try:
                traffic = json.loads(tfile.read_text()).get("hbm_bytes_per_tile") * tiles_per_launch  # measured B/tile x tiles of this run's launches
            except Exception:
                traffic = None
        valu = None
        vfile = ROOT / "profiles" / "k1_valu.json"   # VALU instructions per tile from a separate rocprofv3 --pmc pass
        if vfile.exists():
            try:
                v = json.loads(vfile.read_text())
                valu = {"valu_insts_per_tile": v["valu_insts_per_tile"], "avg_issue_cycles_per_inst": v["avg_issue_cycles_per_inst"],
                        "valu_frac": v["valu_insts_per_tile"] * tiles_per_launch / v["simds"] * v["avg_issue_cycles_per_inst"] / (k_ms * 1e-3 * v["clock_hz"]),
                        "note": "share of the launch during which every SIMD's VALU issue port is taken (instructions per SIMD x issue cost / launch "
                                "time at 2.4 GHz): the kernel is VALU-issue bound, not HBM bound", "source": "profiles/k1_valu.json"}
            except Exception:
                valu = None
        out = {
            "metric": "32x32 tiles/s for mixed-tile-greedy (bf16->BFP{8,4,2}); achieved HBM GB/s vs peak",
            "value": world * args.steps * tiles_per_step / dt,
            "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32/f64", "data": "synthetic",
            "config": {"workload": f"{args.tensors} x 4096x4096 bf16 N(0,0.02^2) per GPU per step, mixed-tile-greedy "
                                   f"{{bf16,bfp8,bfp4,bfp2}} pcc>=0.999 seed 123 (BASELINE.json configs[1], streamed)",
                       "tensors_per_step_per_gpu": args.tensors, "tiles_per_step_per_gpu": tiles_per_step,
                       "k1_chunk": args.chunk, "scan": "device (csrc/mtq_scan.hip)" if pipe.device_scan else f"host ({args.workers} threads)",
                       "host_cpu_ms_per_step": host_cpu_ms, "host_fallbacks": pipe.host_fallbacks,
                       "driver_thread_ms_per_step": {k: v / (args.steps + args.warmup) * 1e3 for k, v in pipe.host_seconds.items()}, "scan_workers": args.workers, "numa_bind": numa, "sharding": f"tensors x{world}, RCCL gather of summary rows"},
            "roofline": {"bound": "hbm", "kernel": "tile_stats (K1)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/k1_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, B/tile x this run's tiles per launch)" if traffic is not None else None,
                         "launch_ms": k_ms, "tiles_per_launch": tiles_per_launch, "launches": kt.launches,
                         "kernel_tiles_per_s": tiles_per_launch / (k_ms * 1e-3),
                         "note": "launch_ms / achieved / frac: HIP events around the K1 launches of the timed region, i.e. with the previous steps' scan "
                                 "kernels (one wave per tensor) and copies running beside them; kernel_alone: the same launch with nothing beside it",
                         "kernel_alone": {"launch_ms": k1_alone_ms, "achieved": BYTES_PER_TILE_READ * args.chunk * (ROWS // 32) * (COLS // 32) / (k1_alone_ms * 1e-3) / 1e9,
                                          "frac": BYTES_PER_TILE_READ * args.chunk * (ROWS // 32) * (COLS // 32) / (k1_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "valu": valu},
            "summary": {"tensors": int(all_rows.shape[0]), "mean_pcc": float(all_rows[:, 2].mean()),
                        "counts_bf16_bfp8_bfp4_bfp2": [int(all_rows[:, 6 + i].sum()) for i in range(4)]},
        }
        if args.cpu_sample > 0 and world == 1:   # the CPU legs are a N = 1 report: the other ranks would wait in the barrier below
            out.update(cpu_baseline(batch[: min(args.cpu_sample, args.tensors)], cpu_budget()))
        print(json.dumps(out), flush=True)
    pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
