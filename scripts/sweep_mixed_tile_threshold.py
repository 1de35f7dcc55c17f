#!/usr/bin/env python3
"""Sweep mixed-tile-threshold over a range of metric thresholds (reference scripts/sweep_mixed_tile_threshold.py).
Same flags; `--backend hip` and `--no-plots` added; tensors come from quantization_analysis_amd.model_source (offline).
Under torch.distributed.run the matched tensors are sharded over the ranks; every rank writes the CSVs and the per-tensor
plot of its tensors, and the Pareto frontiers are gathered to rank 0 (one gather of small Python objects) for the two
overlay figures."""
from __future__ import annotations

import argparse
import fnmatch
import json
import os
import re
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

from quantization_analysis_amd.compression_algorithms.mixed_tile_greedy import parse_tile_formats
from quantization_analysis_amd.compression_algorithms.quantizer import BACKENDS, Quantizer
from quantization_analysis_amd.model_source import build_model_index, filter_tensor_names, lpt_shards
from quantization_analysis_amd import plots
from quantization_analysis_amd.sweep import pareto_frontier, sweep_tensor, write_csv

_LAYER = re.compile(r"(?:^|.*\.)layers\.(\d+)\.(.+)$")
_EXPERT = re.compile(r"^(.*\bexperts)\.(\d+)\.(.+)$")


def split_name(tensor_name: str):
    """'model.layers.3.mlp.experts.7.up_proj.weight' → (layer 3, group 'mlp.experts.up_proj.weight', expert 7);
    names without a `layers.N.` part keep layer None (reference :293-310)."""
    m = _LAYER.match(tensor_name)
    layer, rest = (int(m.group(1)), m.group(2)) if m else (None, tensor_name)
    e = _EXPERT.match(rest)
    return (layer, f"{e.group(1)}.{e.group(3)}", int(e.group(2))) if e else (layer, rest, None)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Sweep mixed-tile-threshold over a range of metric thresholds.")
    p.add_argument("repo_or_url")
    p.add_argument("tensor_name", help="Tensor name or filter (regex by default).")
    p.add_argument("--regex", action="store_true", default=True)
    p.add_argument("--no-regex", dest="regex", action="store_false")
    p.add_argument("--list-matches", action="store_true")
    p.add_argument("--revision", default="main")
    p.add_argument("--cache-dir", default="data/hf-cache")
    p.add_argument("--backend", choices=list(BACKENDS), default="emulation")
    p.add_argument("--formats", default="bf16,bfp8,bfp4,bfp2")
    p.add_argument("--metric", choices=["pcc", "mae", "atol"], default="pcc")
    p.add_argument("--lowest-metric-val", type=float, default=0.9)
    p.add_argument("--steps", type=int, default=50)
    p.add_argument("--out-dir", default=None)
    p.add_argument("--no-plots", action="store_true", help="CSV / JSON artifacts only")
    return p.parse_args(argv)


def select_tensors(index, query: str, use_regex: bool) -> list[str]:
    """reference :313-348."""
    names = index.tensor_names
    weight_like = [n for n in names if "weight" in n.lower() and not n.lower().endswith("_scale_inv")]
    cands = weight_like if weight_like else names
    if use_regex:
        try:
            pat = re.compile(query)
        except re.error as exc:
            raise RuntimeError(f"Invalid regex '{query}': {exc}") from exc
        m = [n for n in cands if pat.search(n)]
        if m:
            return sorted(m)
        raise RuntimeError("No tensors matched the regex query.")
    if query in cands:
        return [query]
    if any(ch in query for ch in "*?[]"):
        m = [n for n in cands if fnmatch.fnmatch(n, query)]
        if m:
            return sorted(m)
    m = [n for n in cands if query.lower() in n.lower()] or filter_tensor_names(cands, query)
    if m:
        return sorted(m)
    raise RuntimeError("No tensors matched the filter query.")


def main(argv=None) -> int:
    args = parse_args(argv)
    formats = parse_tile_formats(args.formats)
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    index = build_model_index(args.repo_or_url, revision=args.revision)
    selected = select_tensors(index, args.tensor_name, args.regex)
    if args.list_matches:
        print(f"Matched {len(selected)} tensor(s):")
        for n in selected:
            print(f"  {n}")
        return 0
    device = None
    if args.backend == "hip":
        import torch

        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    quantizer = Quantizer(args.backend)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if args.backend == "hip" else "gloo", **({"device_id": device} if device is not None else {}))
    stamp = [time.strftime("%Y%m%d-%H%M%S")]
    if dist is not None:   # one output directory for the whole job: rank 0's clock decides (ranks may start across a second boundary)
        dist.broadcast_object_list(stamp, src=0)
    base_out = Path(args.out_dir) if args.out_dir else Path("results") / index.repo_id.replace("/", "__") / "mixed_tile_threshold_sweep" / stamp[0]
    detail = base_out / "details"
    detail.mkdir(parents=True, exist_ok=True)
    lines = []  # (tensor name, frontier points, baseline points) of this rank's tensors
    failed = None
    for i in lpt_shards(selected, index.numel, world)[rank]:
        name = selected[i]
        try:
            x = index.load(name, device=device)
            xf = x if device is not None else x.float().numpy()
            rows, baselines, _thr = sweep_tensor(xf, formats, args.metric, args.lowest_metric_val, args.steps, quantizer)
        except Exception as exc:  # noqa: BLE001 — the rank still takes part in every collective below, then the job exits non-zero
            print(f"error: {exc}" if isinstance(exc, ValueError) else f"error [rank {rank}] {name}: {exc}")
            failed = exc
            break
        out = detail / name.replace("/", "_").replace(".", "_")
        out.mkdir(parents=True, exist_ok=True)
        (out / "sweep_config.json").write_text(json.dumps({
            "repo_or_url": args.repo_or_url, "tensor_name": name, "revision": args.revision, "backend": args.backend,
            "formats": formats, "metric": args.metric, "lowest_metric_val": args.lowest_metric_val, "steps": args.steps}, indent=2))
        write_csv(out / "sweep_results.csv", rows, formats)
        mixed = [{"label": f"t{r['step']}", "size": r["size_bytes"], "metric": r[args.metric], "kind": "mixed",
                  **{f"{f}_tiles": r.get(f"{f}_tiles", 0) for f in formats}} for r in rows]
        front = pareto_frontier(baselines + mixed, args.metric)
        (out / "pareto.json").write_text(json.dumps([{k: p[k] for k in ("label", "size", "metric", "kind")} for p in front], indent=1))
        if not args.no_plots:
            plots.write_sweep_pareto(out / "size_vs_metric.png", args.metric, front, formats, name)
        lines.append((name, front, baselines))
        print(f"[rank {rank}] {name}: {len(rows)} steps, pareto {len(front)} points -> {out}")

    any_failed = failed is not None
    if dist is not None:  # frontiers are a few KB per tensor; the failure flag rides along, so every rank makes the same calls
        box = [None] * world if rank == 0 else None
        dist.gather_object((lines, failed is not None), box, dst=0)
        verdict = [any(f for _l, f in box)] if rank == 0 else [None]
        dist.broadcast_object_list(verdict, src=0)
        any_failed = bool(verdict[0])
        if rank == 0:
            lines = [entry for part, _f in box for entry in part]
        dist.barrier()
        dist.destroy_process_group()
    if any_failed:
        return 1
    if rank == 0 and not args.no_plots:
        by_weight, by_layer, base_weight, base_layer = {}, {}, {}, {}
        for name, front, baselines in sorted(lines, key=lambda e: e[0]):
            layer, group, expert = split_name(name)
            entry = {"layer_id": layer, "expert_id": expert, "weight_name": group}
            if front:
                by_weight.setdefault(group, []).append(dict(entry, points=front))
                if layer is not None:
                    by_layer.setdefault(layer, []).append(dict(entry, points=front))
            if baselines:
                base_weight.setdefault(group, []).append(dict(entry, points=baselines))
                if layer is not None:
                    base_layer.setdefault(layer, []).append(dict(entry, points=baselines))
        floor = args.lowest_metric_val if args.metric == "pcc" else None
        plots.write_overlays(base_out / "weight_overlays.png", args.metric, by_weight, base_weight, "weight", floor)
        plots.write_overlays(base_out / "layer_overlays.png", args.metric, by_layer, base_layer, "layer", floor)
    if rank == 0:
        print(f"Wrote sweep results to {base_out}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
