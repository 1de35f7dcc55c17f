"""wq — weight quantization analyzer CLI (reference: `wq`, 888 lines) for the mixed-tile path.

Same flags, seed rules, tables and result artifacts as the reference (`wq:37-79,553-586,629-647,753-848,881-882`),
plus `--backend hip`.  Differences forced by the environment: the model comes from `model_source` (synthetic presets or a
local safetensors directory, no Hub access), `--no-plots` skips the PNG artifacts, and when launched under
`python -m torch.distributed.run` the matched tensors are sharded over the ranks (one process per GPU) with a single
gather of fixed-width summary rows to rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import secrets
import sys
import time
from datetime import datetime
from pathlib import Path

import numpy as np

from .compression_algorithms import create_algorithm, load_compression_config
from .compression_algorithms.cache import CacheContext
from .compression_algorithms.metrics import pearson_corr
from .compression_algorithms.quantizer import BACKENDS, Quantizer
from .compression_algorithms.tile_utils import MIXED_TILE_FORMATS
from .model_source import build_model_index, lpt_shards, resolve_format_list, resolve_selected_tensors, safe_repo_revision_key
from .quantization_formats import SUPPORTED_FORMATS

# The HIP runtime maps a process's streams onto this many hardware queues (its default is 4), and kernels of streams that share a
# queue run one after the other.  A window of the streamed search (streamed.py) has the scans of several shape groups in flight
# beside K1 and the copies: with 4 queues the scans of a model's third and fourth group waited for the first group's to end
# (the Llama-3-8B shapes: 25 ms per window against 19 ms with 8 queues; bench.py's one-shape steps lose 2 % with 8 and keep 4).
# Read by the runtime at its first call — nothing in this process has touched the GPU yet; a value already in the environment wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FORMAT_BYTES_PER_ELEM = {"bf16": 2.0, "bfp8": 1.088, "bfp4": 0.50097, "bfp2": 0.25097, "fp0": 0.0}  # wq:132-140
MIXED_ALGOS = {"mixed-tile-greedy", "mixed-tile-random", "mixed-tile-threshold"}
ROW_W = 16  # summary row: idx, comp, fmt, pcc, mae, atol, time, gb, tile_bytes, 4 counts, xmin, xmean, xmax


def parse_args(argv=None) -> argparse.Namespace:
    parser = argparse.ArgumentParser(prog="wq", description="Weight quantization analyzer (mixed-tile path, MI355X backend).")
    parser.add_argument("repo_or_url", help="'synthetic:<preset>[:seed]' or a local directory of *.safetensors.")
    parser.add_argument("filter_query", nargs="*", help="Optional filter: substring, or dotted torch-style prefix path.")
    parser.add_argument("--revision", default="main", help="Revision label (default: main).")
    parser.add_argument("--cache-dir", default="data/hf-cache", help="Kept for CLI compatibility (unused offline).")
    parser.add_argument("--limit", type=int, default=None, help="Optional max matched tensors.")
    parser.add_argument("--backend", choices=list(BACKENDS), default="emulation",
                        help="Quantization backend for BFP formats (default: emulation; hip = MI355X kernels).")
    parser.add_argument("--compression-config", type=str, default=None, help="Path to a JSON compression config file (default: none).")
    parser.add_argument("--recompute", action="store_true", help="Recompute and overwrite cached quantized tensors.")
    parser.add_argument("--summary", action="store_true", help="Print the aggregate summary (default: off).")
    parser.add_argument("--results-dir", default="results", help="Root of the result artifacts (default: results).")
    parser.add_argument("--no-plots", action="store_true", help="Skip the PNG artifacts (maps, size-vs-accuracy, random samples).")
    parser.add_argument("--literal-metrics", action="store_true",
                        help="hip backend: materialise y (K2/K3), copy it to the host and print the reference's float32 PCC/MAE/ATOL "
                             "expression (metrics.py:6-16, wq:683-687) instead of the float64-moment columns of the K1 records.")
    parser.add_argument("--no-stream", action="store_true", help="hip backend: evaluate tensor by tensor instead of in streamed groups.")
    return parser.parse_args(argv)


def _slug(s: str) -> str:
    return re.sub(r"[^a-zA-Z0-9._-]+", "_", s).strip("_") or "tensor"  # wq:112-113


def resolve_seed(config, algo_params: dict):
    """wq:553-586 → (used_seed, seed_source); mutates algo_params['seed']."""
    seed_source = None
    used_seed = None
    if config.seed is not None:
        used_seed, seed_source = int(config.seed), "config"
    elif config.random_seed:
        used_seed, seed_source = secrets.randbits(31), "random"
    if used_seed is not None:
        algo_params["seed"] = used_seed
    elif "seed" in algo_params:
        try:
            p = int(algo_params["seed"])
        except (TypeError, ValueError):
            used_seed, seed_source = algo_params["seed"], "params"
        else:
            if p == 0:
                used_seed, seed_source = secrets.randbits(31), "random"
            else:
                used_seed, seed_source = p, "params"
            algo_params["seed"] = used_seed
    return used_seed, seed_source


def _assignment_mapping(assignment: np.ndarray) -> dict:
    return {
        "tile_hw": 32,
        "format_to_int": {fmt: idx for idx, fmt in enumerate(MIXED_TILE_FORMATS)},
        "int_to_format": MIXED_TILE_FORMATS,
        "assignment_shape": list(assignment.shape),
    }


def write_assignment_outputs(out_dir: Path, tensor_name: str, assignment: np.ndarray, algo_dir: str, draw: bool = True) -> None:
    """wq:295-316: <algo_dir>/<slug>/{assignment.npy (int8), assignment_mapping.json, <slug>_assignment.png}."""
    mt_dir = out_dir / algo_dir / _slug(tensor_name)
    mt_dir.mkdir(parents=True, exist_ok=True)
    np.save(mt_dir / "assignment.npy", assignment.astype(np.int8))
    with (mt_dir / "assignment_mapping.json").open("w", encoding="utf-8") as f:
        json.dump(_assignment_mapping(assignment), f, indent=2)
    if draw:
        from . import plots

        plots.write_assignment_map(mt_dir / f"{_slug(tensor_name)}_assignment.png", assignment)


def write_random_outputs(out_dir: Path, tensor_name: str, samples: list, tile_formats: list, assignment, draw: bool = True) -> None:
    """wq:151-218: mixed_tile_random/{<slug>.csv (one row per sample), <slug>_assignment.npy, <slug>_assignment_mapping.json,
    <slug>.png}."""
    import csv

    if not samples:
        return
    mt_dir = out_dir / "mixed_tile_random"
    mt_dir.mkdir(parents=True, exist_ok=True)
    slug = _slug(tensor_name)
    with (mt_dir / f"{slug}.csv").open("w", newline="", encoding="utf-8") as f:
        w = csv.writer(f)
        w.writerow(["sample_id", *[f"{fmt}_tiles" for fmt in tile_formats], "total_gb", "pcc", "mae", "atol"])
        for s in samples:
            w.writerow([s.get("id"), *[s.get("counts", {}).get(fmt, 0) for fmt in tile_formats],
                        float(s.get("total_bytes", 0.0)) / 1e9, s.get("pcc"), s.get("mae"), s.get("atol")])
    if assignment is not None:
        np.save(mt_dir / f"{slug}_assignment.npy", assignment.astype(np.int8))
        with (mt_dir / f"{slug}_assignment_mapping.json").open("w", encoding="utf-8") as f:
            json.dump(_assignment_mapping(assignment), f, indent=2)
    if draw:
        from . import plots

        plots.write_random_samples(mt_dir / f"{slug}.png", samples)


def write_size_plot(out_dir: Path, tensor_name: str, metric_name: str, rows, formats, algo_name: str) -> None:
    """wq:436-497: baselines (the `none` rows) and the MIXED row of one tensor → <algo_dir>/<slug>/size_vs_accuracy.png.
    rows: this tensor's summary rows (ROW_W wide), comp index 0 = none, 1 = the selected algorithm."""
    from . import plots

    col = {"pcc": 3, "mae": 4, "atol": 5}[metric_name]
    mixed_rows = [r for r in rows if int(r[1]) == 1]
    if not mixed_rows:
        return
    total_tiles = next((int(sum(r[9:13])) for r in mixed_rows if r[9] >= 0), None)
    points = []
    for r in rows:
        if int(r[1]) == 0:
            fmt = formats_of_row(r)
            tiles = {f: 0 for f in MIXED_TILE_FORMATS}
            if total_tiles is not None and fmt in tiles:
                tiles[fmt] = total_tiles
            points.append({"label": fmt.upper(), "bytes": r[7] * 1e9, "metric": r[col], "kind": "baseline",
                           **{f"{f}_tiles": tiles[f] for f in MIXED_TILE_FORMATS}})
    for r in mixed_rows:
        points.append({"label": "MIXED", "bytes": r[7] * 1e9, "metric": r[col], "kind": "mixed",
                       **{f"{f}_tiles": max(0, int(r[9 + i])) for i, f in enumerate(MIXED_TILE_FORMATS)}})
    plots.write_size_vs_accuracy(out_dir / algo_name.replace("-", "_") / _slug(tensor_name) / "size_vs_accuracy.png", metric_name,
                                 points, MIXED_TILE_FORMATS)


def formats_of_row(r) -> str:
    return "mixed" if int(r[2]) < 0 else SUPPORTED_FORMATS[int(r[2])]


def _columns_emulation(xf: np.ndarray, y: np.ndarray):
    diff = np.abs(xf - y)  # wq:684-687, literal float32
    return pearson_corr(xf, y), float(np.mean(diff)), float(np.max(diff))


def _none_rows_hip(x, formats, quantizer):
    """`none` baseline on the hip backend: pcc/mae/atol of every pure mixed-tile format come from ONE
    K1 pass (sum the per-tile records); fp0 from three device reductions.  y is not materialised or cached."""
    from .compression_algorithms.tile_search import columns_from_stats, compute_tile_stats

    out = {}
    mixed = [f for f in formats if f in MIXED_TILE_FORMATS]
    if mixed:
        ts = compute_tile_stats(x, mixed, quantizer)
        for f in mixed:
            amap = np.full(ts.tiles, MIXED_TILE_FORMATS.index(f), dtype=np.int8)
            c = columns_from_stats(ts, amap)
            out[f] = (c["pcc"], c["mae"], c["atol"])
    if "fp0" in formats:
        ax = x.float().abs()
        mx = float(ax.max()) if ax.numel() else 0.0
        out["fp0"] = (1.0 if mx == 0.0 else 0.0, float(ax.mean()) if ax.numel() else 0.0, mx)  # metrics.py:14-15 with y = 0
    return out


def _none_rows_literal(x, formats, quantizer):
    """--literal-metrics: every pure format's y through K2, to the host, into the reference's float32 expression (wq:683-687)."""
    xh = x.float().cpu().numpy()
    out = {}
    for f in formats:
        y = quantizer.quantize(x, f)
        out[f] = _columns_emulation(xh, y.cpu().numpy() if hasattr(y, "cpu") else np.asarray(y, dtype=np.float32))
    return out


def _evaluate_tensor(idx, name, index, algorithms, formats, quantizer, args, run_tag, processed_root, results_dir):
    """One tensor through [none, selected] → list of summary rows (np.float64 [R, ROW_W])."""
    hip = args.backend == "hip"
    if hip:
        import torch

        x = index.load(name, device=torch.device("cuda", torch.cuda.current_device()))
        xf32 = x.float()
        meta = (float(xf32.min()), float(xf32.mean()), float(xf32.max())) if x.numel() else (0.0, 0.0, 0.0)
        numel = int(x.numel())
    else:
        x = np.asarray(index.load(name).float().numpy(), dtype=np.float32)
        meta = (float(np.min(x)), float(np.mean(x)), float(np.max(x))) if x.size else (0.0, 0.0, 0.0)
        numel = int(x.size)
    cache_ctx = CacheContext(root=processed_root, tensor_name=name, backend=args.backend, recompute=args.recompute, run_tag=run_tag)
    rows = []
    for ci, algo in enumerate(algorithms):
        t0 = time.perf_counter()
        if hip and algo.name == "none":
            cols = _none_rows_literal(x, formats, quantizer) if args.literal_metrics else _none_rows_hip(x, formats, quantizer)
            elapsed = time.perf_counter() - t0
            for f in formats:
                pcc, mae, atol = cols[f]
                rows.append([idx, ci, SUPPORTED_FORMATS.index(f), pcc, mae, atol, elapsed, numel * FORMAT_BYTES_PER_ELEM[f] / 1e9,
                             np.nan, -1, -1, -1, -1, *meta])
            continue
        results = algo.run(xf=x, formats=formats, quantizer=quantizer, cache=cache_ctx)
        if hip:
            import torch

            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # wall time of run(), wq:680-682
        for res in results:
            if hip and res.meta and "columns" in res.meta and not (args.literal_metrics and res.y is not None):
                c = res.meta["columns"]
                pcc, mae, atol = c["pcc"], c["mae"], c["atol"]
            else:
                y = res.y.cpu().numpy() if hasattr(res.y, "cpu") else res.y
                xh = x.float().cpu().numpy() if hip else x
                pcc, mae, atol = _columns_emulation(xh, y)
            fmt_l = res.fmt.lower()
            gb = float(res.tile_bytes) / 1e9 if res.tile_bytes is not None else numel * FORMAT_BYTES_PER_ELEM.get(fmt_l, 0.0) / 1e9
            counts = [res.tile_counts.get(k, 0) for k in MIXED_TILE_FORMATS] if res.tile_counts else [-1, -1, -1, -1]
            fcode = -1 if res.fmt == "MIXED" else SUPPORTED_FORMATS.index(fmt_l)
            rows.append([idx, ci, fcode, pcc, mae, atol, elapsed, gb, res.tile_bytes if res.tile_bytes is not None else np.nan,
                         *counts, *meta])
            draw = not args.no_plots
            if res.compression == "mixed-tile-random" and res.meta:  # wq:711-722
                if isinstance(res.meta.get("samples"), list) and res.meta.get("tile_formats"):
                    write_random_outputs(results_dir, name, res.meta["samples"], res.meta["tile_formats"], res.meta.get("assignment"), draw)
            elif res.compression in MIXED_ALGOS and res.meta and isinstance(res.meta.get("assignment"), np.ndarray):
                write_assignment_outputs(results_dir, name, res.meta["assignment"], res.compression.replace("-", "_"), draw)
    out = np.asarray(rows, dtype=np.float64).reshape(-1, ROW_W)
    if not args.no_plots and len(algorithms) > 1 and algorithms[1].name in {"mixed-tile-threshold", "mixed-tile-greedy"}:  # wq:743-750
        write_size_plot(results_dir, name, algorithms[1].params.get("metric", "pcc"), out, formats, algorithms[1].name)
    return out


def _print_tables(names, rows, comp_names, shapes, table_lines, summary: bool, formats):
    comp_w = max(len("COMP"), max(len(n) for n in comp_names))

    def emit(line=""):
        print(line)
        table_lines.append(line)

    by_tensor: dict[int, list] = {}
    for r in rows:
        by_tensor.setdefault(int(r[0]), []).append(r)
    aggregate: dict[tuple, list] = {}
    for ti, name in enumerate(names):
        trs = by_tensor.get(ti, [])
        if not trs:
            continue
        emit(name)
        emit(f"  shape={shapes[ti]} min={trs[0][13]:.3e} mean={trs[0][14]:.3e} max={trs[0][15]:.3e}")  # wq:82-84
        for ci, comp in enumerate(comp_names):
            crs = [r for r in trs if int(r[1]) == ci]
            if not crs:
                continue
            fmts = ["MIXED" if int(r[2]) < 0 else SUPPORTED_FORMATS[int(r[2])].upper() for r in crs]
            fmt_w = max(len(f) for f in fmts)
            mixed = comp in MIXED_ALGOS
            pcc_w = max(len("PCC"), max(len(f"{r[3]: .5f}") for r in crs))
            mae_w = max(len("MAE"), max(len(f"{r[4]:.3e}") for r in crs))
            atol_w = max(len("ATOL"), max(len(f"{r[5]:.3e}") for r in crs))
            time_w = max(len("TIME(s)"), max(len(f"{r[6]:.3f}") for r in crs))
            gb_w = max(len("GB"), max(len(f"{r[7]:.3f}") for r in crs))
            header = (f"  {'COMP'.ljust(comp_w)}  {'FORMAT'.ljust(fmt_w)}  {'PCC'.rjust(pcc_w)}  {'MAE'.rjust(mae_w)}  "
                      f"{'ATOL'.rjust(atol_w)}  {'TIME(s)'.rjust(time_w)}  {'GB'.rjust(gb_w)}")
            if mixed:  # wq:765-812
                cw = {k: max(len(k.upper()), max(len(str(int(r[9 + i]))) for r in crs)) for i, k in enumerate(MIXED_TILE_FORMATS)}
                bytes_w = max(len("BYTES"), max(len(f"{(0.0 if np.isnan(r[8]) else r[8]):,.0f}") for r in crs))
                header += "  " + "  ".join(k.upper().rjust(cw[k]) for k in MIXED_TILE_FORMATS) + f"  {'BYTES'.rjust(bytes_w)}"
            emit(header)
            for r, f in zip(crs, fmts):
                line = (f"  {comp.ljust(comp_w)}  {f.ljust(fmt_w)}  {f'{r[3]: .5f}'.rjust(pcc_w)}  {f'{r[4]:.3e}'.rjust(mae_w)}  "
                        f"{f'{r[5]:.3e}'.rjust(atol_w)}  {f'{r[6]:.3f}'.rjust(time_w)}  {f'{r[7]:.3f}'.rjust(gb_w)}")
                if mixed:
                    line += "  " + "  ".join(str(int(r[9 + i])).rjust(cw[k]) for i, k in enumerate(MIXED_TILE_FORMATS))
                    line += f"  {f'{(0.0 if np.isnan(r[8]) else r[8]):,.0f}'.rjust(bytes_w)}"
                emit(line)
                aggregate.setdefault((comp, f), []).append(r)
            emit()
    if summary:  # wq:851-879
        emit("Summary (mean across matched tensors)")
        for comp in comp_names:
            for f in (["MIXED"] if comp in MIXED_ALGOS else [x.upper() for x in formats]):
                rs = aggregate.get((comp, f), [])
                if not rs:
                    continue
                pcc, mae, atol = (float(np.mean([r[k] for r in rs])) for k in (3, 4, 5))
                bv = [r[8] for r in rs if not np.isnan(r[8])]
                btxt = f"  bytes={float(np.mean(bv)):,.0f}" if bv else ""
                emit(f"  {comp.ljust(comp_w)} {f:>5}  pcc={pcc: .5f}  mae={mae:.3e}  atol={atol:.3e}{btxt}")


HIP_COLUMNS_NOTE = ("# columns (--backend hip): PCC / MAE / ATOL from float64 moments of the K1 tile records; the reference prints a float32 "
                    "BLAS Pearson that is itself off by up to 1e-4 at 4096x4096 (use --literal-metrics for that expression)")
HIP_LITERAL_NOTE = "# columns (--backend hip --literal-metrics): the reference's float32 expression (metrics.py:6-16) on y from K2 / K3"


class _JobError(Exception):
    """An error every rank reports through the job's status exchange instead of leaving its peers in a collective."""


def _any_rank_failed(dist, failed: bool, device) -> bool:
    """MAX over ranks of a failure flag: every rank reaches every collective of the job whether or not its own work failed."""
    if dist is None:
        return failed
    import torch

    flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    return bool(flag.item())


def _evaluate_shard(shard, tensor_names, index, algorithms, selected_algo, formats, quantizer, args, run_tag, processed_root, results_dir, rank):
    """This rank's tensors → (summary rows [R, ROW_W], note lines).  hip + a search algorithm: streamed groups (streamed.py)."""
    from . import streamed

    notes = []
    per_tensor = list(shard)
    rows_by_idx = {}
    deferred = []   # (idx, name, assignment): artifacts written after the GPU work
    if not args.no_stream and streamed.streamable(selected_algo, formats, args):
        import torch

        device = torch.device("cuda", torch.cuda.current_device())
        groups: dict = {}
        per_tensor = []
        for i in shard:
            k = streamed.group_key(index, tensor_names[i])
            if k is None:
                per_tensor.append(i)
            else:
                groups.setdefault(k, []).append((i, tensor_names[i]))
        ev = streamed.ShardEvaluator(index, selected_algo, formats, device, ROW_W, FORMAT_BYTES_PER_ELEM)
        try:
            for idx, (rows, assignment) in ev.run_groups(groups).items():
                rows_by_idx[idx] = rows
                deferred.append((idx, tensor_names[idx], assignment))
        finally:
            ev.close()
        streamed_note = None
        if ev.compute_tiles:
            k1 = f", K1 {ev.k1_tiles / max(ev.k1_ms, 1e-9) / 1e3:.1f} M tiles/s" if ev.k1_ms else ""
            streamed_note = (f"[rank {rank}] streamed {len(deferred)} tensors in {len(groups)} shape groups: {ev.compute_tiles} tiles in {ev.compute_seconds:.3f} s "
                             f"of GPU pipeline = {ev.compute_tiles / max(ev.compute_seconds, 1e-9) / 1e6:.1f} M tiles/s (loader and artifacts excluded{k1})")
            load_s = ev.load_seconds
    else:
        streamed_note = None
    for i in per_tensor:
        rows_by_idx[i] = _evaluate_tensor(i, tensor_names[i], index, algorithms, formats, quantizer, args, run_tag, processed_root, results_dir)
    algo_dir = selected_algo.name.replace("-", "_")
    t_art = time.perf_counter()
    for idx, name, assignment in deferred:   # wq:696-750, after the GPU work
        write_assignment_outputs(results_dir, name, assignment, algo_dir, not args.no_plots)
        if not args.no_plots:
            write_size_plot(results_dir, name, selected_algo.params.get("metric", "pcc"), rows_by_idx[idx], formats, selected_algo.name)
    if streamed_note:   # where the wall time of the streamed part went: the path itself is the middle number
        notes.append(streamed_note + f"; wall: loader {load_s:.1f} s, pipeline {ev.compute_seconds:.3f} s, maps and plots to disk {time.perf_counter() - t_art:.1f} s")
    ordered = [rows_by_idx[i] for i in sorted(rows_by_idx)]
    return (np.concatenate(ordered) if ordered else np.zeros((0, ROW_W))), notes


def run(argv=None) -> int:
    args = parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    coll_dev = None
    try:   # torch sizes its CPU thread pool by the machine (256 hardware threads on a GPU box); under a cgroup CPU quota that
        # oversubscribes the quota in the loader and gets every thread of the job — the GPU driver's included — throttled
        import torch

        from .pipeline import cpu_budget

        torch.set_num_threads(max(1, min(torch.get_num_threads(), cpu_budget() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))))
    except Exception:  # noqa: BLE001
        pass
    if args.backend == "hip":
        import torch

        if torch.cuda.is_available():
            torch.cuda.set_device(local_rank)
            from .hip_backend import bind_to_gpu_numa_node

            bind_to_gpu_numa_node(local_rank)  # host threads and pinned memory next to this rank's GPU
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "hip" and torch.cuda.is_available():
            coll_dev = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=coll_dev)  # RCCL
        else:
            coll_dev = torch.device("cpu")
            dist.init_process_group("gloo")

    # Every rank runs the same sequence of collectives — status, broadcast, status, gather, barrier — whatever happens to its
    # own work: a failure is carried through the status exchanges and the whole job exits non-zero together.
    try:
        err = None
        config = algo_params = used_seed = seed_source = None
        try:
            config = load_compression_config(args.compression_config)
            algo_params = dict(config.params)
            used_seed, seed_source = resolve_seed(config, algo_params)
        except Exception as exc:  # noqa: BLE001
            err = exc
        if _any_rank_failed(dist, err is not None, coll_dev):
            if err is not None:
                if dist is None:
                    raise err
                print(f"error [rank {rank}]: {err}", file=sys.stderr)
            return 1
        run_tag = datetime.now().strftime("%Y%m%d-%H%M%S")
        if dist is not None:  # one seed and one run tag for the whole job (rank 0 decides)
            box = [used_seed, seed_source, run_tag]
            dist.broadcast_object_list(box, src=0)
            used_seed, seed_source, run_tag = box
            if used_seed is not None:
                algo_params["seed"] = used_seed

        mine = np.zeros((0, ROW_W))
        notes: list[str] = []
        ctx = {}
        try:
            selected_algo = create_algorithm(config.algorithm, algo_params)
            baseline = create_algorithm("none", {})
            algorithms = [baseline] if selected_algo.name == "none" else [baseline, selected_algo]
            filter_query = " ".join(args.filter_query).strip() or None
            formats = resolve_format_list(config.quantization_formats, SUPPORTED_FORMATS)

            index = build_model_index(args.repo_or_url, revision=args.revision)
            try:
                tensor_names = resolve_selected_tensors(index, filter_query)
            except RuntimeError:
                tensor_names = []
            if args.limit is not None:
                tensor_names = tensor_names[: max(0, args.limit)]
            if not tensor_names:
                raise _JobError("No tensors matched.")
            try:
                quantizer = Quantizer(backend=args.backend)
                if args.backend == "ttnn":
                    raise RuntimeError("TTNN backend requires `ttnn` in the active Python environment.")  # wq:603-609
            except Exception as exc:  # noqa: BLE001
                raise _JobError(f"error: {exc}") from exc

            comp_names = [a.name for a in algorithms]
            if rank == 0:
                print(f"{index.repo_id} @{index.revision} - {len(tensor_names)} tensors")
                print(f"formats: {', '.join(formats)}")
                print(f"compression: {', '.join(comp_names)}")
                print(f"backend: {args.backend}" + (f"  ranks: {world}" if world > 1 else ""))
                if args.compression_config:
                    print(f"config: {args.compression_config}")
                print()

            results_dir = Path(args.results_dir) / index.repo_id.replace("/", "__") / selected_algo.name / run_tag  # wq:629-631
            results_dir.mkdir(parents=True, exist_ok=True)
            if rank == 0:
                used_params = dict(algo_params)
                if used_seed is not None:
                    used_params.pop("seed", None)
                used_config = {"algorithm": config.algorithm, "quantization_formats": formats, "params": used_params}
                if used_seed is not None:
                    used_config["seed"] = used_seed
                    if seed_source:
                        used_config["seed_source"] = seed_source
                with (results_dir / "compression_config.used.json").open("w", encoding="utf-8") as f:  # wq:636-647
                    json.dump(used_config, f, indent=2)
            processed_root = Path("data/processed") / safe_repo_revision_key(index.repo_id, index.revision)

            shards = lpt_shards(tensor_names, index.numel, world)
            per_tensor_rows = len(formats) + (0 if selected_algo.name == "none" else 1)
            ctx = {"shards": shards, "per_tensor_rows": per_tensor_rows, "tensor_names": tensor_names, "index": index, "comp_names": comp_names,
                   "formats": formats, "results_dir": results_dir}
            mine, notes = _evaluate_shard(shards[rank], tensor_names, index, algorithms, selected_algo, formats, quantizer, args, run_tag,
                                          processed_root, results_dir, rank)
        except _JobError as exc:
            err = exc
            print(str(exc), file=sys.stderr)
        except Exception as exc:  # noqa: BLE001
            err = exc
            if dist is None:
                raise
            import traceback

            print(f"error [rank {rank}]: {exc}", file=sys.stderr)
            traceback.print_exc()
        if _any_rank_failed(dist, err is not None, coll_dev):
            return 1
        for line in notes:
            print(line)

        shards, per_tensor_rows, tensor_names, index = ctx["shards"], ctx["per_tensor_rows"], ctx["tensor_names"], ctx["index"]
        comp_names, formats, results_dir = ctx["comp_names"], ctx["formats"], ctx["results_dir"]
        if dist is not None:  # the ONE data-path collective: fixed-width summary rows → rank 0
            import torch

            cap = max(len(s) for s in shards) * per_tensor_rows
            buf = torch.full((cap, ROW_W), float("nan"), dtype=torch.float64, device=coll_dev)
            buf[: mine.shape[0]] = torch.from_numpy(mine).to(coll_dev)
            gathered = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, gathered, dst=0)
            if rank == 0:
                allr = torch.cat(gathered).cpu().numpy()
                allr = allr[~np.isnan(allr[:, 0])]
        else:
            allr = mine

        if rank == 0:
            order = np.lexsort((np.arange(allr.shape[0]), allr[:, 1], allr[:, 0]))
            allr = allr[order]
            shapes = {}
            for i, n in enumerate(tensor_names):
                shapes[i] = tuple(index.specs[n].shape) if n in index.specs else "?"
            table_lines: list[str] = []
            if args.backend == "hip":   # which definition the float columns use (the reference's table has no such line)
                note = HIP_LITERAL_NOTE if args.literal_metrics else HIP_COLUMNS_NOTE
                print(note)
                table_lines.append(note)
            _print_tables(tensor_names, allr, comp_names, shapes, table_lines, args.summary, formats)
            (results_dir / "table.txt").write_text("\n".join(table_lines) + "\n", encoding="utf-8")  # wq:881-882
            print(f"results: {results_dir}")
        return 0
    finally:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()


def main() -> None:
    raise SystemExit(run())


if __name__ == "__main__":
    main()
