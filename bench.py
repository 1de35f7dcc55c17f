#!/usr/bin/env python3
"""bench.py — mixed-tile-greedy (bf16 → BFP{8,4,2}) throughput on MI355X.

Step  = one pass of the hot path (K1 tile_stats + the sequential greedy scan, both on the GPU by default [--scan host: stats
        D2H + host scan threads] → per-tile assignment maps + pcc/mae/atol on the host) over a batch of `--tensors` (128) synthetic 4096x4096 bf16 tensors
        (BASELINE.json configs[1], streamed; the batch is > 256 MiB so the Infinity Cache cannot hold it).
value = tiles/s, whole job, inputs resident in HBM when the timed region starts; the median of `--regions` (3) timed regions of
        `--steps` steps each, every region bracketed by barrier + synchronize (a 20-step region lasts 50 ms: one scheduling hiccup of
        the host moved a single region by 5 %).
roofline = the K1 kernel alone: algorithmic 2048 B read per tile / HIP-event launch duration vs 8 TB/s.
cpu_baseline = the C oracle (oracle/, a port of the reference's CPU path) on rank 0's host, 1 thread, on a bounded sample of
        the same tensors; cpu_baseline_threads = the same port on every core of the rank's CPU quota; cpu_baseline_emulation =
        the NumPy host backend (how the reference executes) on one tensor.
extra   = one leg per further BASELINE.json config, N = 1 only (--legs none skips them): the single-tensor latency of configs[1],
        mixed-tile-threshold and the 50-step sweep on the DeepSeek-R1 layer-0 shapes (configs[2], [4]), the 224 Llama-3-8B tensors
        through the streamed pipeline (configs[3]).
--workload llama3-8b: configs[3] as the step — 224 synthetic tensors, drawn on the device, LPT-sharded over the N ranks, one RCCL
        gather of the summary rows: strong scaling (the total work is fixed).

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
METRIC_NAME = "32x32 tiles/s for mixed-tile-greedy (bf16->BFP{8,4,2}); achieved HBM GB/s vs peak"


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


def cpu_baseline(sample: torch.Tensor, threads: int, gpu_results=None) -> dict:
    """Three CPU figures on rank 0's host, each timed like wq:680-682 times algo.run (perf_counter around the whole search):
      cpu_baseline            the C port (oracle/mtq_oracle.c), 1 thread, `len(sample)` tensors            kind "port"
      cpu_baseline_threads    the same port, one tensor per thread on `threads` threads (the rank's CPU quota) kind "port"
      cpu_baseline_emulation  the package's NumPy host backend (`--backend emulation`: how the reference executes — single-threaded
                              NumPy, quantize per format + per-tile sums + sequential scan, wq:680-682), one tensor  kind "emulation"
    All three produce the same maps (asserted).  `gpu_results` (the timed steps' own TensorResults of the same tensors): the GPU's maps and
    counts must equal the oracle's and its pcc / mae / atol columns the oracle's columns_from_stats to 1e-12 — "parity_on_bench_inputs" in
    the line; a mismatch raises, so a bench line is only ever printed for results the checker agrees with (wq:679-706)."""
    import concurrent.futures as cf

    import numpy as np

    from oracle import mtq_oracle as orc

    xs = [sample[i].float().cpu().numpy() for i in range(sample.shape[0])]
    orc.lib()
    t0 = time.perf_counter()
    tiles = 0
    maps = []
    states = []
    for x in xs:
        a, c, st = orc.greedy(x, FORMATS, METRIC, THRESHOLD, SEED)
        tiles += a.size
        maps.append(a)
        states.append((c, st))
    dt = time.perf_counter() - t0
    parity = None
    if gpu_results is not None:
        slots = orc.mask_slots(orc.fmt_mask(FORMATS))
        worst = 0.0
        for i, (a, (c, st)) in enumerate(zip(maps, states)):
            r = gpu_results[i]
            if not np.array_equal(r.assignment, a) or any(int(r.counts[f]) != int(c[f]) for f in c):
                raise SystemExit(f"bench: the GPU's map of tensor {i} differs from the oracle's ({int((r.assignment != a).sum())} tiles)")
            cols = orc.columns_from_stats(st["stats"], slots, a, xs[i].size)
            for got, want in zip((r.pcc, r.mae, r.atol), cols):
                worst = max(worst, abs(got - want))
            if worst > 1e-12:
                raise SystemExit(f"bench: the GPU's pcc / mae / atol of tensor {i} are {worst:.3e} off the oracle's columns_from_stats")
        parity = {"maps_equal_oracle": f"{len(maps)}/{len(maps)}", "counts_equal_oracle": True, "columns_max_abs_diff": worst,
                  "checked": "the timed region's last step: assignment maps and counts of the --cpu-sample tensors against oracle/mtq_oracle greedy, "
                             "pcc / mae / atol against its columns_from_stats (tolerance 1e-12)"}
    out = {"cpu_baseline": {"value": tiles / dt, "unit": "tiles/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                            "sample": f"{len(xs)} of the step's 4096x4096 bf16 tensors, oracle/mtq_oracle.c greedy (1 thread), {dt:.1f} s"}}
    if parity is not None:
        out["parity_on_bench_inputs"] = parity
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
    (RCCL for the GPU job, gloo in the CPU test).  Ranks may hold different numbers of rows (a model's shards): the counts go first."""
    t_max = torch.tensor([seconds], dtype=torch.float64, device=rows.device)
    if dist is None:
        return rows.cpu().numpy(), seconds
    counts = [torch.zeros((1,), dtype=torch.int64, device=rows.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device))
    most = max(int(c.item()) for c in counts)
    padded = torch.zeros((most, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    padded[: rows.shape[0]] = rows
    gathered = [torch.empty_like(padded) for _ in range(world)] if rank == 0 else None
    dist.gather(padded, gathered, dst=0)
    dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
    if rank != 0:
        return None, float(t_max.item())
    return torch.cat([g[: int(c.item())] for g, c in zip(gathered, counts)]).cpu().numpy(), float(t_max.item())


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


# ---------------------------------------------------------------------------------------------------------------------------------
# configs[3]: Llama-3-8B, all model.layers.* linear weights (224 tensors), sharded over the ranks
# ---------------------------------------------------------------------------------------------------------------------------------
MAX_BATCH_TILES = 1 << 21      # tiles per pipeline batch (streamed.py)


def llama_shard(rank: int, world: int):
    """→ (index, names of all 224 tensors, this rank's tensor indices, {(rows, cols): [tensor idx]} of the shard).  Pure host work:
    every rank computes the same LPT partition (model_source.lpt_shards) without communication (wq:655)."""
    from quantization_analysis_amd import model_source as ms

    index = ms.build_model_index("synthetic:llama3-8b")
    names = ms.resolve_selected_tensors(index, "model.layers")
    mine = ms.lpt_shards(names, index.numel, world)[rank]
    groups: dict = {}
    for i in mine:
        shape, _dtype = index.shape_dtype(names[i])
        groups.setdefault((int(shape[0]), int(shape[1])), []).append(i)
    return index, names, mine, groups


def llama_batches(index, names, groups, device):
    """The shard's tensors drawn on the device (one seeded generator call per tensor) and stacked per shape into pipeline batches
    of at most MAX_BATCH_TILES tiles, largest tensors first (their scans are the long poles).  → [(tensor indices, x3d)]."""
    out = []
    for (rows, cols), idxs in sorted(groups.items(), key=lambda kv: -(kv[0][0] * kv[0][1])):
        per = max(1, MAX_BATCH_TILES // ((rows // 32) * (cols // 32)))
        for b0 in range(0, len(idxs), per):
            part = idxs[b0:b0 + per]
            out.append((part, torch.stack([index.load(names[i], device=device, draw_on_device=True) for i in part])))
    return out


def tiles_of(x3d) -> int:
    return x3d.shape[0] * -(-x3d.shape[1] // 32) * -(-x3d.shape[2] // 32)


def dry_run(args) -> None:
    """--dry-run: the rank plumbing of this file with no GPU and no measurement (CPU test of the launcher branch, of the model
    workload's sharding and of the job's only collective): gloo instead of RCCL, made-up summary rows, `value` null."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    dist = None
    if "RANK" in os.environ:
        import torch.distributed as dist

        dist.init_process_group("gloo")
    if args.workload == "llama3-8b":
        index, names, mine, groups = llama_shard(rank, world)
        rows = torch.tensor([[i, index.numel(names[i]), 1.0, 0.0, 0.0, 0.0, 0, 0, 0, 0, float(rank)] for i in mine], dtype=torch.float64)
        all_rows, dt = gather_summary(rows, 1.0 + rank, dist, rank, world)
        if rank == 0:
            per_rank = [int((all_rows[:, 10] == r).sum()) for r in range(world)]
            numel = [float(all_rows[all_rows[:, 10] == r, 1].sum()) for r in range(world)]
            print(json.dumps({"dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "strong",
                              "tensors": int(all_rows.shape[0]), "tensors_seen_once": len(set(all_rows[:, 0].tolist())) == len(names) == int(all_rows.shape[0]),
                              "tensors_per_rank": per_rank, "imbalance": max(numel) / (sum(numel) / world), "max_seconds": dt,
                              "config": {"workload": "llama3-8b", "sharding": f"224 tensors LPT over {world} ranks, RCCL gather of summary rows"}}), flush=True)
    else:
        rows = torch.full((2, 11), float(rank), dtype=torch.float64)
        all_rows, dt = gather_summary(rows, 1.0 + rank, dist, rank, world)
        if rank == 0:
            print(json.dumps({"dry_run": True, "value": None, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ranks_seen": sorted({int(v) for v in all_rows[:, 1]}), "max_seconds": dt,
                              "config": {"sharding": f"tensors x{world}, RCCL gather of summary rows"}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def median(v):
    s = sorted(v)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def timed(fn, reps: int = 5):
    """median wall milliseconds of fn() (synchronised), after one untimed call."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return median(ts), ts


# ---------------------------------------------------------------------------------------------------------------------------------
# extra legs (N = 1): one driver-run number per BASELINE.json config beside the headline
# ---------------------------------------------------------------------------------------------------------------------------------
def leg_latency(sample, device) -> dict:
    """configs[1] as written — ONE 4096x4096 bf16 tensor: the plug-in call the reference times (wq:680-682: perf_counter around
    algo.run, y materialised) through Quantizer("hip"), and the same search through the streamed pipeline with a batch of one."""
    from quantization_analysis_amd import hip_backend as hb
    from quantization_analysis_amd.compression_algorithms import create_algorithm
    from quantization_analysis_amd.compression_algorithms.cache import CacheContext
    from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer
    from quantization_analysis_amd.pipeline import GreedyPipeline

    x = sample[0]
    cache = CacheContext(ROOT / "gpurun_out" / "bench-cache", "bench", "hip", True, "bench")
    algo = create_algorithm("mixed-tile-greedy", {"metric": METRIC, "threshold": THRESHOLD, "seed": SEED})
    q = Quantizer("hip")
    run_ms, _ = timed(lambda: algo.run(x, FORMATS, q, cache))
    with GreedyPipeline(FORMATS, METRIC, THRESHOLD, SEED, chunk=1, workers=2) as pipe:
        one = x[None]
        pipe.reserve(one)
        pipe_ms, _ = timed(lambda: pipe.run(one))
        route = "lazy" if pipe.lazy_plan(one) is not None else "whole records"
    k1_ms, _ = timed(lambda: hb.tile_stats_batched(one, 0xE))
    return {"workload": "one 4096x4096 bf16 tensor, mixed-tile-greedy {bf16,bfp8,bfp4,bfp2} pcc>=0.999 seed 123 (BASELINE.json configs[1])",
            "algo_run_ms": run_ms, "pipeline_batch_of_one_ms": pipe_ms, "pipeline_route": route, "k1_whole_records_ms": k1_ms, "tiles": 16384,
            "note": "median of 5 wall-clock calls each, synchronised; algo.run = the plug-in seam (K1, search, K3 for y), what wq:680-682 times"}


def deepseek_tensors(device):
    from quantization_analysis_amd import model_source as ms

    index = ms.build_model_index("synthetic:deepseek-r1-layer0")
    names = ms.resolve_selected_tensors(index, "model.layers.0.self_attn")
    return names, [index.load(n, device=device, draw_on_device=True) for n in names]


def leg_threshold(device) -> dict:
    """configs[2]: mixed-tile-threshold over the seven DeepSeek-R1 model.layers.0.self_attn tensors (five float32 matrices after the
    fp8 x scale dequantisation, two bf16 vectors) — K1 (tile_stats_direct for float32 storage) + K4 on the device + the knife-edge
    re-score, all seven through ThresholdPipeline.run_batches (vectors as (n/32, 32) matrices with their element count)."""
    import numpy as np

    from quantization_analysis_amd import hip_backend as hb
    from quantization_analysis_amd.pipeline import ThresholdPipeline

    names, xs = deepseek_tensors(device)
    mats = [x for x in xs if x.dim() == 2]
    vecs = [x for x in xs if x.dim() != 2]
    tiles = sum(-(-x.shape[0] // 32) * -(-x.shape[1] // 32) for x in mats) + sum(-(-x.numel() // 1024) for x in vecs)
    def as_batch(x):   # a 1-D tensor is the (ceil(n/32), 32) matrix of tile_utils.py:96-102 with its own element count
        if x.dim() == 2:
            return (x[None], None)
        n = x.numel()
        rows = -(-n // 32)
        m = torch.zeros((rows * 32,), dtype=x.dtype, device=x.device)
        m[:n] = x
        return (m.view(1, rows, 32), n)

    batches = [as_batch(x) for x in xs]
    with ThresholdPipeline(FORMATS, "pcc", THRESHOLD, chunk=1) as pipe:
        last = []

        def once():
            last[:] = pipe.run_batches(batches)   # every tensor's K1 / K4 / knife listing enqueued first, decisions and column sums behind them

        ms_, _ = timed(once, reps=5)
        knife = pipe.knife_tiles // 6
    # parity on the leg's own inputs: one matrix and one vector against the oracle (maps bit for bit; columns from the same records' sums)
    from oracle import mtq_oracle as orc

    checked = []
    for i in (min(range(len(xs)), key=lambda k: xs[k].numel() if xs[k].dim() == 2 else 1 << 62), next(k for k, x in enumerate(xs) if x.dim() != 2)):
        a, c = orc.threshold(xs[i].float().cpu().numpy(), FORMATS, "pcc", THRESHOLD)[:2]
        r = last[i][0]
        if not np.array_equal(r.assignment, a) or any(int(r.counts[f]) != int(c[f]) for f in c):
            raise SystemExit(f"bench: threshold leg, tensor {names[i]}: the GPU's map differs from the oracle's")
        checked.append(names[i])
    # the float32 K1 alone on the matrices (HIP events): its share of the 4096 B/tile read roofline
    f32 = [x for x in mats if x.dtype == torch.float32]
    f32_tiles = sum(-(-x.shape[0] // 32) * -(-x.shape[1] // 32) for x in f32)
    out = hb.tile_stats_ragged(f32, 0xF)          # the launch run_batches issues for them: one ragged batch (mtq_tile_stats_ragged)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hb.tile_stats_ragged(f32, 0xF, out=out)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    k1 = median(ts)
    return {"workload": f"DeepSeek-R1 model.layers.0.self_attn, {len(xs)} tensors ({len(mats)} matrices, {len(vecs)} vectors), mixed-tile-threshold pcc>=0.999 (BASELINE.json configs[2])",
            "value": tiles / (ms_ * 1e-3), "unit": "tiles/s", "ms": ms_, "tiles": tiles, "knife_edge_tiles": knife,
            "maps_equal_oracle": checked,
            "roofline": {"bound": "hbm", "kernel": "tile_stats_direct<float, 15> (K1, float32 storage)", "achieved": 4096 * f32_tiles / (k1 * 1e-3) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 4096 * f32_tiles / (k1 * 1e-3) / 1e9 / HBM_PEAK_GBS, "launch_ms": k1, "tiles": f32_tiles,
                         "traffic": None, "note": "4096 B read per float32 tile; the five matrices as one ragged launch (+ its fix-up launch), HIP events"}}


def leg_sweep(device) -> dict:
    """configs[4]: the 50-step pcc-threshold sweep + pareto front (scripts/sweep_mixed_tile_threshold.py's core, sweep.sweep_tensor)
    over the same DeepSeek-R1 layer-0 tensors: ONE K1 pass per tensor, every step a selection + sum of records."""
    from quantization_analysis_amd.compression_algorithms.quantizer import Quantizer
    from quantization_analysis_amd.sweep import pareto_frontier, sweep_tensor

    names, xs = deepseek_tensors(device)
    q = Quantizer("hip")
    steps = 50
    state = {}

    def once():
        state["rows"] = [sweep_tensor(x, FORMATS, "pcc", 0.9, steps, q)[0] for x in xs]

    ms_, _ = timed(once, reps=3)
    front = [len(pareto_frontier([{"size": r["size_bytes"], "metric": r["pcc"]} for r in rows], "pcc")) for rows in state["rows"]]
    tiles = sum((-(-x.shape[0] // 32) * -(-x.shape[1] // 32)) if x.dim() == 2 else -(-x.numel() // 1024) for x in xs)
    return {"workload": f"DeepSeek-R1 model.layers.0.self_attn, {len(xs)} tensors, {steps}-step pcc-threshold sweep + pareto (BASELINE.json configs[4])",
            "value": tiles * steps / (ms_ * 1e-3), "unit": "tile-steps/s", "ms": ms_, "tiles": tiles, "steps": steps, "pareto_points_per_tensor": front}


def leg_llama(device) -> dict:
    """configs[3] on one GPU: the 224 Llama-3-8B linear weights (drawn on the device: the loader is reported, not measured with the
    pipeline) through GreedyPipeline.run_batches, shape group by shape group — the --workload llama3-8b step with N = 1."""
    from quantization_analysis_amd.pipeline import GreedyPipeline

    index, names, mine, groups = llama_shard(0, 1)
    t0 = time.perf_counter()
    batches = llama_batches(index, names, groups, device)
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    xs = [x for _p, x in batches]
    tiles = sum(tiles_of(x) for x in xs)
    with GreedyPipeline(FORMATS, METRIC, THRESHOLD, SEED, chunk=1 << 30, workers=default_workers()) as pipe:
        pipe.SLOTS = max(pipe.SLOTS, min(len(xs), 8))
        pipe.prepare(xs)
        ms_, all_ms = timed(lambda: pipe.run_batches(xs), reps=5)
        pipe.timing.drain()
        k1_ms, k1_tiles = pipe.timing.kernel_ms, pipe.timing.tiles
        res = pipe.run_batches(xs)
        fallbacks = pipe.host_fallbacks
    counts = [sum(r.counts[f] for rs in res for r in rs) for f in FORMATS]
    # parity on the leg's own inputs: the first tensor of the smallest shape group (k_proj, 1024 x 4096) against the oracle's greedy search
    import numpy as np

    from oracle import mtq_oracle as orc

    bi = min(range(len(xs)), key=lambda k: xs[k].shape[1] * xs[k].shape[2])
    a, c, _st = orc.greedy(xs[bi][0].float().cpu().numpy(), FORMATS, METRIC, THRESHOLD, SEED)
    r0 = res[bi][0]
    if not np.array_equal(r0.assignment, a) or any(int(r0.counts[f]) != int(c[f]) for f in c):
        raise SystemExit(f"bench: llama leg, tensor {names[batches[bi][0][0]]}: the GPU's map differs from the oracle's")
    return {"maps_equal_oracle": [names[batches[bi][0][0]]], "workload": "Llama-3-8B model.layers.* linear weights, 224 bf16 tensors (6.8 M tiles, 14 GB), mixed-tile-greedy pcc>=0.999 seed 123 (BASELINE.json configs[3]) on ONE GPU",
            "value": tiles / (ms_ * 1e-3), "unit": "tiles/s", "pipeline_ms": ms_, "pipeline_ms_all": all_ms, "tiles": tiles, "tensors": len(mine), "batches": len(xs),
            "loader_seconds": load_s, "loader": "synthetic tensors drawn on the device, one seeded generator call per tensor",
            "k1_tiles_per_s": k1_tiles / max(k1_ms, 1e-9) * 1e3, "host_fallbacks": fallbacks, "counts_bf16_bfp8_bfp4_bfp2": counts}


def run_legs(sample, device) -> dict:
    out = {}
    for name, fn in (("latency_single_tensor", lambda: leg_latency(sample, device)), ("threshold_deepseek_layer0", lambda: leg_threshold(device)),
                     ("sweep_deepseek_layer0", lambda: leg_sweep(device)), ("llama3_8b", lambda: leg_llama(device))):
        t0 = time.perf_counter()
        try:
            out[name] = fn()
            out[name]["leg_seconds"] = time.perf_counter() - t0
        except Exception as exc:  # noqa: BLE001 — a leg that fails is reported, the headline line still comes out
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out


def k1_profile_blocks(tiles_per_launch: float, k_ms: float, lazy: bool = True):
    """traffic / VALU blocks of the roofline object from the committed PMC passes (profiles/k1_traffic.json, k1_valu.json): the lazy
    route's kernel (partial records) at the top level of either file, the whole-record kernel under `whole_records`."""
    traffic = traffic_source = valu = None
    tfile = ROOT / "profiles" / "k1_traffic.json"  # HBM bytes per launch from a separate rocprofv3 --pmc pass
    if tfile.exists():
        try:
            t = json.loads(tfile.read_text())
            per_tile = t["hbm_bytes_per_tile"] if lazy else t["whole_records"]["hbm_bytes_per_tile"]
            traffic = per_tile * tiles_per_launch  # measured B/tile x tiles of this run's launches
            traffic_source = t.get("source", "profiles/k1_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, B/tile x this run's tiles per launch)")
            if not lazy:
                traffic_source += "; this run: " + t["whole_records"]["kernel"]
        except Exception:
            traffic = traffic_source = None
    vfile = ROOT / "profiles" / "k1_valu.json"   # VALU instructions per tile from a separate rocprofv3 --pmc pass
    if vfile.exists():
        try:
            v = json.loads(vfile.read_text())
            w = v if lazy else v["whole_records"]
            valu = {"valu_insts_per_tile": w["valu_insts_per_tile"], "avg_issue_cycles_per_inst": w["avg_issue_cycles_per_inst"],
                    "valu_frac": w["valu_insts_per_tile"] * tiles_per_launch / v["simds"] * w["avg_issue_cycles_per_inst"] / (k_ms * 1e-3 * v["clock_hz"]),
                    "note": "share of the launch during which every SIMD's VALU issue port is taken (instructions per SIMD x issue cost / launch "
                            "time at 2.4 GHz): the kernel is VALU-issue bound, not HBM bound", "source": "profiles/k1_valu.json"}
        except Exception:
            valu = None
    return traffic, traffic_source, valu


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--regions", type=int, default=3, help="timed regions of --steps steps each; value is their median")
    ap.add_argument("--workload", choices=["m1", "llama3-8b"], default="m1",
                    help="m1: 128 x 4096x4096 bf16 per GPU per step (BASELINE.json configs[1], weak scaling); llama3-8b: the 224 linear weights of "
                         "Llama-3-8B sharded over the ranks (configs[3], strong scaling)")
    ap.add_argument("--tensors", type=int, default=128, help="4096x4096 bf16 tensors per step per GPU (128 = 4 GiB, SURVEY §8(d) M1 stream)")
    ap.add_argument("--chunk", type=int, default=None,
                    help="tensors per K1 launch (default: the whole step's batch with the scan on the device — one K1 launch and one scan launch per "
                         "step; 32 with the host scan, where a chunk's records cross PCIe while the next chunk's K1 runs)")
    ap.add_argument("--workers", type=int, default=default_workers(), help="host scan threads per rank (default: this rank's share of the cgroup CPU quota / affinity mask, at most 32)")
    ap.add_argument("--cpu-sample", type=int, default=24, help="tensors timed on the CPU port, ~0.5 s each (0 = skip)")
    ap.add_argument("--legs", choices=["all", "none"], default="all", help="the extra legs (other BASELINE configs), rank 0 at N = 1 only")
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

    hb.require_gpu()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "llama3-8b":
        run_llama_workload(args, dist, rank, world, device, barrier, numa)
    else:
        run_m1_workload(args, dist, rank, world, device, barrier, numa)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    hb.shutdown()   # the library's threads, events and device tables go while the HIP runtime is still there (also registered with atexit)


def run_llama_workload(args, dist, rank, world, device, barrier, numa) -> None:
    """configs[3] as the step: every rank streams its LPT shard of the 224 tensors (resident in HBM, drawn on the device before the
    timed region), shape group by shape group; one gather of the summary rows.  Strong scaling: the model is the job."""
    import gc

    from quantization_analysis_amd.pipeline import GreedyPipeline

    index, names, mine, groups = llama_shard(rank, world)
    t0 = time.perf_counter()
    batches = llama_batches(index, names, groups, device)
    torch.cuda.synchronize()
    load_s = time.perf_counter() - t0
    xs = [x for _p, x in batches]
    my_tiles = sum(tiles_of(x) for x in xs)
    pipe = GreedyPipeline(FORMATS, METRIC, THRESHOLD, SEED, chunk=1 << 30, workers=args.workers, scan=args.scan)
    pipe.SLOTS = max(pipe.SLOTS, min(len(xs), 8))
    pipe.prepare(xs)
    gc.collect()
    gc.freeze()
    res = None
    for _ in range(max(1, args.warmup)):
        res = pipe.run_batches(xs)
    pipe.timing.drain()
    pipe.timing.__init__()
    region_s, my_region_s = [], []
    for _ in range(max(1, args.regions)):
        barrier()
        t0 = time.perf_counter()
        for _s in range(args.steps):
            res = pipe.run_batches(xs)
        torch.cuda.synchronize()
        mine_dt = time.perf_counter() - t0
        barrier()
        my_region_s.append(mine_dt)
        region_s.append(time.perf_counter() - t0)
    pipe.timing.drain()
    mid = sorted(range(len(region_s)), key=lambda i: region_s[i])[len(region_s) // 2]
    rows = torch.tensor([[i, index.numel(names[i]), r.pcc, r.mae, r.atol, r.tile_bytes, r.counts["bf16"], r.counts["bfp8"], r.counts["bfp4"], r.counts["bfp2"], float(rank)]
                         for (part, _x), rs in zip(batches, res) for i, r in zip(part, rs)], dtype=torch.float64, device=device)
    all_rows, dt = gather_summary(rows, region_s[mid], dist, rank, world)
    per_rank = torch.tensor([my_region_s[mid], float(my_tiles), load_s], dtype=torch.float64, device=device)
    if dist is not None:
        allp = [torch.empty_like(per_rank) for _ in range(world)]
        dist.all_gather(allp, per_rank)
        per_rank_all = torch.stack(allp).cpu().numpy()
    else:
        per_rank_all = per_rank[None].cpu().numpy()
    if rank == 0:
        total_tiles = int(per_rank_all[:, 1].sum())
        kt = pipe.timing
        secs = per_rank_all[:, 0] / args.steps
        frac = BYTES_PER_TILE_READ * kt.tiles / max(kt.kernel_ms, 1e-9) / 1e6 / HBM_PEAK_GBS
        out = {"metric": METRIC_NAME, "value": total_tiles * args.steps / dt, "unit": "tiles/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32/f64", "data": "synthetic",
               "config": {"workload": "Llama-3-8B model.layers.* linear weights: 224 synthetic bf16 tensors (6.8 M tiles, 14 GB) drawn on the device, mixed-tile-greedy "
                                      "{bf16,bfp8,bfp4,bfp2} pcc>=0.999 seed 123 (BASELINE.json configs[3]); a step = the whole model once",
                          "sharding": f"224 tensors LPT by element count over {world} ranks (model_source.lpt_shards), RCCL gather of summary rows",
                          "tensors": int(all_rows.shape[0]), "tiles": total_tiles, "regions_ms_per_step": [s / args.steps * 1e3 for s in region_s],
                          "per_rank_pipeline_ms_per_step": (secs * 1e3).tolist(), "per_rank_tiles": per_rank_all[:, 1].astype(int).tolist(),
                          "imbalance_tiles": float(per_rank_all[:, 1].max() / per_rank_all[:, 1].mean()), "imbalance_time": float(secs.max() / secs.mean()),
                          "loader_seconds_per_rank": per_rank_all[:, 2].tolist(), "scan": "device (csrc/mtq_scan.hip)" if pipe.device_scan else "host",
                          "host_fallbacks": pipe.host_fallbacks, "numa_bind": numa},
               "roofline": {"bound": "hbm", "kernel": "tile_stats (K1)", "achieved": frac * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac, "traffic": None,
                            "launches": kt.launches, "note": "rank 0's K1 launches of the timed regions (HIP events), all shape groups together"},
               "summary": {"tensors": int(all_rows.shape[0]), "mean_pcc": float(all_rows[:, 2].mean()),
                           "counts_bf16_bfp8_bfp4_bfp2": [int(all_rows[:, 6 + i].sum()) for i in range(4)]}}
        print(json.dumps(out), flush=True)
    pipe.close()


def run_m1_workload(args, dist, rank, world, device, barrier, numa) -> None:
    import numpy as np

    from quantization_analysis_amd.pipeline import GreedyPipeline

    batch = make_batch(args.tensors, rank, device)
    pipe = GreedyPipeline(FORMATS, METRIC, THRESHOLD, SEED, chunk=args.chunk or 32, workers=args.workers, scan=args.scan)
    if args.chunk is None:
        args.chunk = args.tensors if pipe.device_scan else 32
    pipe.chunk = args.chunk
    tiles_per_step = args.tensors * (ROWS // 32) * (COLS // 32)

    # K1 alone (no scan, no copies beside it): the same launch the steps issue, HIP events on the current stream
    alone_out = pipe.launch_k1(batch[: args.chunk])     # the launch the route issues (round 3: partial records on the lazy route)
    for _ in range(4):                                   # the process's first launches run on an idle GPU's clocks (1.85 against 1.72 ms)
        pipe.launch_k1(batch[: args.chunk], out=alone_out)
    torch.cuda.synchronize()
    alone = []
    for _ in range(9):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        pipe.launch_k1(batch[: args.chunk], out=alone_out)
        a1.record()
        a1.synchronize()
        alone.append(a0.elapsed_time(a1))
    k1_alone_ms = sorted(alone)[len(alone) // 2]
    del alone_out
    pipe.reserve(batch)  # record buffers of every slot + scan threads: allocations, not steps
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
    regions, cpu_ms = [], []
    for _ in range(max(1, args.regions)):   # every region: exactly --steps steps between two barriers; every step fully processed
        barrier()
        cpu0 = time.process_time()
        t0 = time.perf_counter()
        res = pipe.run_steps(batch for _ in range(args.steps))  # step s+1's GPU work overlaps step s's scan tail
        barrier()
        regions.append(time.perf_counter() - t0)
        cpu_ms.append((time.process_time() - cpu0) / args.steps * 1e3)   # CPU time of every thread of this rank, per step
    pipe.timing.drain()
    mid = sorted(range(len(regions)), key=lambda i: regions[i])[len(regions) // 2]
    host_cpu_ms = cpu_ms[mid]

    # the only data-path collective: per-tensor summary rows to rank 0; outside the timed steps
    # the rows of the LAST step are gathered so the multi-GPU path is exercised end to end.
    rows = torch.tensor([[r.index, ROWS * COLS, r.pcc, r.mae, r.atol, r.tile_bytes, r.counts["bf16"], r.counts["bfp8"],
                          r.counts["bfp4"], r.counts["bfp2"], float(rank)] for r in res], dtype=torch.float64, device=device)   # last column: the rank that searched the tensor
    all_rows, dt = gather_summary(rows, regions[mid], dist, rank, world)
    if dist is not None:   # every region's MAX over ranks, for the record
        rt = torch.tensor(regions, dtype=torch.float64, device=device)
        dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        regions_max = rt.cpu().tolist()
    else:
        regions_max = list(regions)

    out = None
    if rank == 0:
        kt = pipe.timing
        k_ms = kt.kernel_ms / max(kt.launches, 1)
        tiles_per_launch = kt.tiles / max(kt.launches, 1)
        achieved = BYTES_PER_TILE_READ * tiles_per_launch / (k_ms * 1e-3) / 1e9
        lazy = pipe.lazy_plan(batch[: args.chunk]) is not None
        traffic, traffic_source, valu = k1_profile_blocks(tiles_per_launch, k_ms, lazy)
        total_steps = args.steps * len(regions)
        out = {
            "metric": METRIC_NAME,
            "value": world * args.steps * tiles_per_step / dt,
            "unit": "tiles/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32/f64", "data": "synthetic",
            "config": {"workload": f"{args.tensors} x 4096x4096 bf16 N(0,0.02^2) per GPU per step, mixed-tile-greedy "
                                   f"{{bf16,bfp8,bfp4,bfp2}} pcc>=0.999 seed 123 (BASELINE.json configs[1], streamed)",
                       "tensors_per_step_per_gpu": args.tensors, "tiles_per_step_per_gpu": tiles_per_step,
                       "regions": len(regions), "regions_tiles_per_s": [world * args.steps * tiles_per_step / s for s in regions_max],
                       "value_is": "the median region (each region = exactly --steps steps between barrier + synchronize)",
                       "k1_chunk": args.chunk, "scan": "device (csrc/mtq_scan.hip)" if pipe.device_scan else f"host ({args.workers} threads)",
                       "route": ("lazy: K1 evaluates bfp8 (five statistics) and bfp4 (three sums); the search stops before its last pass, the listed kernel evaluates bfp2 and "
                                 "bfp4's error statistics for that pass's candidates, the last pass follows (DESIGN.md §4)") if lazy else "whole records",
                       "listed_tiles_per_step": pipe.listed_tiles / max(total_steps + args.warmup, 1), "shared_visiting_orders": bool(pipe.shared_orders and pipe.device_scan),
                       "ranks_seen_by_the_collective": int(dist.get_world_size()) if dist is not None else 1,
                       "per_rank_tiles_per_step": [int(v) for v in np.bincount(all_rows[:, 10].astype(np.int64), minlength=world) * (ROWS // 32) * (COLS // 32)],
                       "host_cpu_ms_per_step": host_cpu_ms, "host_fallbacks": pipe.host_fallbacks,
                       "driver_thread_ms_per_step": {k: v / (total_steps + args.warmup) * 1e3 for k, v in pipe.host_seconds.items()}, "scan_workers": args.workers, "numa_bind": numa, "sharding": f"tensors x{world}, RCCL gather of summary rows"},
            "roofline": {"bound": "hbm", "kernel": "tile_stats (K1)" + (": tile_stats_bf16_rolled<3, 1> (partial records)" if lazy else ""), "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "launch_ms": k_ms, "tiles_per_launch": tiles_per_launch, "launches": kt.launches,
                         "kernel_tiles_per_s": tiles_per_launch / (k_ms * 1e-3),
                         "note": "launch_ms / achieved / frac: HIP events around the K1 launches of the timed regions, i.e. with the earlier steps' search kernels (two "
                                 "waves per tensor), the listed K1 and the column sums running beside them; kernel_alone: the same launch with nothing beside it",
                         "kernel_alone": {"launch_ms": k1_alone_ms, "achieved": BYTES_PER_TILE_READ * args.chunk * (ROWS // 32) * (COLS // 32) / (k1_alone_ms * 1e-3) / 1e9,
                                          "frac": BYTES_PER_TILE_READ * args.chunk * (ROWS // 32) * (COLS // 32) / (k1_alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "valu": valu},
            "summary": {"tensors": int(all_rows.shape[0]), "mean_pcc": float(all_rows[:, 2].mean()),
                        "counts_bf16_bfp8_bfp4_bfp2": [int(all_rows[:, 6 + i].sum()) for i in range(4)]},
        }
        if args.cpu_sample > 0 and world == 1:   # the CPU legs are a N = 1 report: the other ranks would wait in the barrier below
            k = min(args.cpu_sample, args.tensors)
            by_index = {r.index: r for r in res}
            out.update(cpu_baseline(batch[:k], cpu_budget(), [by_index[i] for i in range(k)]))
    pipe.close()
    if rank == 0:
        if args.legs == "all" and world == 1:
            sample = batch[:1].clone()
            del batch
            torch.cuda.empty_cache()
            out["extra"] = run_legs(sample, device)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
