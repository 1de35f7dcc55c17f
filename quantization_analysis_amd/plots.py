"""PNG artifacts of `wq` and the threshold sweep: the per-tensor tile-assignment map, the
size-vs-accuracy scatter and the random-search sample scatter.  Same file names and the same information as the
reference's writers (`wq:151-218` samples, `wq:221-283` map, `wq:335-412` size plot); drawn with matplotlib's Agg
canvas on the host from data that is already there (maps, table rows) — nothing here touches the GPU.

Every writer returns the path it wrote, or None when matplotlib is unavailable or there is nothing to draw, so a
headless box without matplotlib still produces every non-image artifact.
"""
from __future__ import annotations

import contextlib
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from .compression_algorithms.tile_utils import MIXED_TILE_BYTES_PER_ELEM, MIXED_TILE_FORMATS

DPI = 160
BASELINE_COLOR, MIXED_COLOR = "#1f77b4", "#d62728"
PCC_BUCKET_COLORS = ((0.999, "#2ca02c"), (0.99, "#ffbf00"), (-np.inf, "#d62728"))  # wq:116-131: good / mid / bad


def _pyplot():
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        return None
    return plt


@contextlib.contextmanager
def _figure(plt, path: Path, size):
    fig, ax = plt.subplots(figsize=size)
    try:
        yield ax
        fig.tight_layout()
        path.parent.mkdir(parents=True, exist_ok=True)
        fig.savefig(path, dpi=DPI)
    finally:
        plt.close(fig)


def pcc_color(value: float) -> str:
    for floor, color in PCC_BUCKET_COLORS:
        if value >= floor:
            return color
    return PCC_BUCKET_COLORS[-1][1]  # NaN


def _tick_positions(n: int) -> np.ndarray:
    """Every tile index up to 64 tiles, then about 32 labels per axis."""
    return np.arange(0, n, 1 if n <= 64 else max(1, n // 32))


def write_assignment_map(path: Path, assignment: np.ndarray) -> Optional[Path]:
    """Tile grid coloured by format: the more bytes per element, the darker the blue; unassigned (< 0) tiles grey."""
    plt = _pyplot()
    amap = np.asarray(assignment)
    if plt is None or amap.size == 0:
        return None
    from matplotlib.colors import ListedColormap
    from matplotlib.patches import Patch

    amap = amap.reshape(amap.shape[0], -1).astype(np.int16)
    rows, cols = amap.shape
    heavy_first = sorted(MIXED_TILE_FORMATS, key=MIXED_TILE_BYTES_PER_ELEM.get, reverse=True)
    shades = plt.get_cmap("Blues")(np.linspace(0.95, 0.15, len(heavy_first)))
    shade_of = dict(zip(heavy_first, shades))
    palette = ListedColormap([shade_of[f] for f in MIXED_TILE_FORMATS])
    palette.set_bad("gray")
    edge = lambda n: float(np.clip(0.4 * n, 6.0, 18.0))  # noqa: E731  0.4 inch per tile, 6..18 inch
    with _figure(plt, Path(path), (edge(cols), edge(rows))) as ax:
        ax.imshow(np.ma.masked_less(amap, 0), cmap=palette, vmin=-0.5, vmax=len(MIXED_TILE_FORMATS) - 0.5, interpolation="nearest")
        for n, set_ticks, set_labels in ((cols, ax.set_xticks, ax.set_xticklabels), (rows, ax.set_yticks, ax.set_yticklabels)):
            at = _tick_positions(n)
            set_ticks(at)
            set_labels([str(int(i)) for i in at], fontsize=7)
        ax.set_xticks(np.arange(cols + 1) - 0.5, minor=True)
        ax.set_yticks(np.arange(rows + 1) - 0.5, minor=True)
        ax.grid(which="minor", color="white", linewidth=0.5, alpha=0.6)
        ax.tick_params(which="minor", bottom=False, left=False)
        ax.set(xlabel="Tile X", ylabel="Tile Y", title="Tile format assignment")
        ax.legend(handles=[Patch(color=shade_of[f], label=f.upper()) for f in heavy_first], title="Data format",
                  loc="upper right", fontsize=8)
    return Path(path)


def _size_unit(max_bytes: float) -> tuple:
    for scale, unit in ((1e9, "GB"), (1e6, "MB")):
        if max_bytes >= scale:
            return scale, unit
    return 1e3, "KB"


def write_size_vs_accuracy(path: Path, metric_name: str, points: Sequence[dict], formats: Sequence[str] = MIXED_TILE_FORMATS) -> Optional[Path]:
    """points: {"label", "bytes", "metric", "kind": "baseline"|"mixed", "<fmt>_tiles": int…}.  For pcc, points below
    half of the best value are dropped so that fp0/bfp2 baselines do not flatten the axis (wq:351-356)."""
    plt = _pyplot()
    pts = list(points)
    if metric_name == "pcc" and pts:
        floor = 0.5 * max(p["metric"] for p in pts)
        pts = [p for p in pts if p["metric"] >= floor]
    if plt is None or not pts:
        return None
    from matplotlib.lines import Line2D

    scale, unit = _size_unit(max(p["bytes"] for p in pts))
    with _figure(plt, Path(path), (6.0, 4.5)) as ax:
        for p in pts:
            baseline = p.get("kind") == "baseline"
            x, y = p["bytes"] / scale, p["metric"]
            ax.scatter([x], [y], color=BASELINE_COLOR if baseline else MIXED_COLOR, marker="o" if baseline else "X", s=50)
            tiles = " ".join(f"{f}:{p[f'{f}_tiles']}" for f in formats if p.get(f"{f}_tiles") is not None)
            ax.annotate(f"{p['label']} ({y:.3g}, {p['bytes'] / 1e6:.2f}MB)" + (f" [{tiles}]" if tiles else ""), (x, y),
                        textcoords="offset points", xytext=(4, 4), fontsize=6)
        ax.set(xlabel=f"Size ({unit})", ylabel=metric_name.upper(), title="Size vs accuracy")
        ax.grid(True, alpha=0.3)
        if formats:
            def key(marker, label, color=None):
                return Line2D([0], [0], marker=marker, color="w", label=label, markerfacecolor=color, markersize=7)

            ax.legend(handles=[key("o", "Baseline", BASELINE_COLOR), key("X", "Mixed", MIXED_COLOR),
                               Line2D([0], [0], color="w", label="Annot: label (metric, size) [fmt:tiles]")], loc="best", fontsize=8)
    return Path(path)


def write_random_samples(path: Path, samples: Sequence[dict]) -> Optional[Path]:
    """One dot per random map: pcc against total size, coloured by the pcc bucket, labelled with the sample id."""
    plt = _pyplot()
    if plt is None or not samples:
        return None
    pcc = [float(s.get("pcc", 0.0)) for s in samples]
    gb = [float(s.get("total_bytes", 0.0)) / 1e9 for s in samples]
    with _figure(plt, Path(path), (6.5, 4.5)) as ax:
        ax.scatter(pcc, gb, c=[pcc_color(v) for v in pcc], s=28, alpha=0.9)
        for s, x, y in zip(samples, pcc, gb):
            ax.annotate(str(s.get("id")), (x, y), textcoords="offset points", xytext=(4, 4), fontsize=7)
        ax.set(xlabel="PCC", ylabel="Total size (GB)", title="Mixed-tile random samples")
        ax.grid(True, alpha=0.3)
    return Path(path)


def mix_color(point: dict) -> tuple:
    """Colour of a sweep point by its tile mix: red = bfp2 share, blue = bfp4 share, green = bfp8 + bf16 share,
    square-rooted and normalised so that small shares stay visible (sweep…:183-194)."""
    share = np.asarray([float(point.get(f"{f}_tiles", 0.0)) for f in ("bfp2", "bfp8", "bf16", "bfp4")])
    if share.sum() <= 0.0:
        return (0.2, 0.2, 0.8)
    share = share / share.sum()
    rgb = np.sqrt(np.asarray([share[0], share[1] + share[2], share[3]]))
    return tuple(rgb / max(1e-8, rgb.sum()))


def _padded(lo: float, hi: float, frac: float = 0.03) -> tuple:
    pad = max(hi - lo, 1e-9) * frac
    return lo - pad, hi + pad


def _annotate_baselines(ax, points, scale, unit, offset=(4, 4), **kw) -> None:
    for p in points:
        if p.get("kind") == "baseline":
            x = p["size"] / scale
            ax.annotate(f"{p['label']} ({x:.2f}{unit})", (x, p["metric"]), textcoords="offset points", xytext=offset, fontsize=6, **kw)


def write_sweep_pareto(path: Path, metric_name: str, frontier: Sequence[dict], formats: Sequence[str], tensor_name: str) -> Optional[Path]:
    """`size_vs_metric.png` of one tensor: its Pareto frontier (already sorted by size) as a poly-line whose points and
    segments carry the tile-mix colour; pure-format baselines on the frontier are labelled (sweep…:209-290)."""
    plt = _pyplot()
    pts = list(frontier)
    if plt is None or not pts:
        return None
    from matplotlib.collections import LineCollection
    from matplotlib.lines import Line2D

    scale, unit = _size_unit(max(p["size"] for p in pts))
    xy = np.asarray([[p["size"] / scale, p["metric"]] for p in pts], dtype=np.float64)
    rgb = np.asarray([mix_color(p) for p in pts], dtype=np.float64)
    with _figure(plt, Path(path), (6.5, 4.5)) as ax:
        if len(pts) > 1:
            ax.add_collection(LineCollection(np.stack([xy[:-1], xy[1:]], axis=1), colors=(rgb[:-1] + rgb[1:]) / 2.0, linewidths=1.5))
        ax.scatter(xy[:, 0], xy[:, 1], color=rgb, s=20)
        _annotate_baselines(ax, pts, scale, unit)
        ax.set(xlabel=f"Size ({unit})", ylabel=metric_name.upper(), title=f"Size vs metric sweep — {tensor_name}",
               xlim=_padded(xy[:, 0].min(), xy[:, 0].max()), ylim=_padded(xy[:, 1].min(), xy[:, 1].max()))
        ax.grid(True, alpha=0.3)
        pure = {"bf16": (0.0, 1.0, 0.0), "bfp8": (0.0, 1.0, 0.0), "bfp4": (0.0, 0.0, 1.0), "bfp2": (1.0, 0.0, 0.0)}
        ax.legend(handles=[Line2D([0], [0], marker="o", color=pure.get(f, (0.2, 0.2, 0.8)), label=f.upper(),
                                  markerfacecolor=pure.get(f, (0.2, 0.2, 0.8)), markersize=6) for f in formats], loc="best", fontsize=8)
    return Path(path)


def write_overlays(path: Path, metric_name: str, panels: dict, baselines: dict, by: str, metric_floor: Optional[float] = None) -> Optional[Path]:
    """Aggregate figures of a multi-tensor sweep, one panel per key of `panels` (sweep…:351-578):
      by="weight": key = weight name without its `layers.N.` prefix, one frontier per layer, shaded by layer index
                   (`weight_overlays.png`);
      by="layer" : key = layer index, one frontier per weight, one colour per weight name, experts lightened by expert
                   index, with a figure legend (`layer_overlays.png`).
    panels[key] = [{"points": frontier, "layer_id", "expert_id", "weight_name"}…]; baselines likewise (pure formats)."""
    plt = _pyplot()
    if plt is None or not panels:
        return None
    from matplotlib.lines import Line2D

    keys = sorted(panels)
    every = [p for group in (panels, baselines) for lines in group.values() for line in lines for p in line["points"]]
    if not every:
        return None
    lo = metric_floor if (metric_name == "pcc" and metric_floor is not None) else min(p["metric"] for p in every)
    hi = max(p["metric"] for p in every)
    weight_names = sorted({line.get("weight_name") for lines in panels.values() for line in lines if line.get("weight_name")})
    if by == "layer":
        if not weight_names:
            return None
        pal = plt.get_cmap("tab20" if len(weight_names) <= 20 else "hsv")
        weight_color = {n: (pal(i) if len(weight_names) <= 20 else pal(i / max(1, len(weight_names) - 1))) for i, n in enumerate(weight_names)}
    blues = plt.get_cmap("Blues")
    fig, axes = plt.subplots(1, len(keys), figsize=(max(6.0, 4.0 * len(keys)), 4.5), squeeze=False)
    try:
        for ax, key in zip(axes[0], keys):
            lines, base = panels[key], baselines.get(key, [])
            pts = [p for line in list(lines) + list(base) for p in line["points"]]
            if not pts:
                ax.set_axis_off()
                continue
            scale, unit = _size_unit(max(p["size"] for p in pts))
            ids = [line[("layer_id" if by == "weight" else "expert_id")] for line in lines
                   if line.get("layer_id" if by == "weight" else "expert_id") is not None]
            first, span = (min(ids), max(1, max(ids) - min(ids))) if ids else (0, 1)
            for line in lines:
                if len(line["points"]) < 2 and by == "weight":
                    continue
                xs = [p["size"] / scale for p in line["points"]]
                ys = [p["metric"] for p in line["points"]]
                if by == "weight":
                    lid = line.get("layer_id")
                    color = blues(0.5 if lid is None else 0.9 - 0.8 * (lid - first) / span)
                else:
                    color = tuple(weight_color.get(line.get("weight_name"), (0.2, 0.2, 0.8)))[:3]
                    if line.get("expert_id") is not None:
                        fade = float(np.clip(0.6 * (line["expert_id"] - first) / span, 0.0, 1.0))
                        color = tuple(c + (1.0 - c) * fade for c in color)
                ax.plot(xs, ys, color=color, linewidth=1.5)
            for line in base:
                for p in line["points"]:
                    ax.scatter([p["size"] / scale], [p["metric"]], color=mix_color(p), marker="o", s=30, edgecolors="black", linewidths=0.4)
            if base:
                _annotate_baselines(ax, [dict(p, kind="baseline") for p in base[0]["points"]], scale, unit, offset=(6, 0), ha="left", va="center")
            ax.set(title=(f"Layer {key}" if by == "layer" else str(key)), xlabel=f"Size ({unit})", ylim=_padded(lo, hi),
                   xlim=_padded(min(p["size"] for p in pts) / scale, max(p["size"] for p in pts) / scale))
            ax.grid(True, alpha=0.3)
        axes[0][0].set_ylabel(metric_name.upper())
        if by == "layer":
            fig.legend(handles=[Line2D([0], [0], color=weight_color[n], lw=2, label=n) for n in weight_names], loc="upper center",
                       bbox_to_anchor=(0.5, 1.02), ncol=min(4, len(weight_names)), fontsize=8)
            fig.tight_layout(rect=(0.0, 0.0, 1.0, 0.95))
        else:
            fig.tight_layout()
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        fig.savefig(path, dpi=DPI)
    finally:
        plt.close(fig)
    return Path(path)
