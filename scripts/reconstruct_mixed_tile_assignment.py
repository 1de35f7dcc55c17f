#!/usr/bin/env python3
"""Reconstruct a quantized tensor from a mixed-tile assignment map
(reference scripts/reconstruct_mixed_tile_assignment.py:82-137).  hip backend: K3 (mtq_apply_assignment)."""
from __future__ import annotations

import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

import numpy as np

from quantization_analysis_amd.compression_algorithms.quantizer import BACKENDS, Quantizer
from quantization_analysis_amd.compression_algorithms.tile_search import TileStats, reconstruct
from quantization_analysis_amd.compression_algorithms.tile_utils import MIXED_TILE_FORMATS, flatten_2d
from quantization_analysis_amd.model_source import build_model_index


def main(argv=None) -> int:
    p = argparse.ArgumentParser(description="Reconstruct a quantized tensor using a mixed-tile assignment map.")
    p.add_argument("repo_or_url")
    p.add_argument("tensor_name")
    p.add_argument("assignment", help="Path to assignment .npy file (ints per tile).")
    p.add_argument("--assignment-mapping", default=None)
    p.add_argument("--revision", default="main")
    p.add_argument("--cache-dir", default="data/hf-cache")
    p.add_argument("--backend", choices=list(BACKENDS), default="emulation")
    p.add_argument("--out", default=None)
    args = p.parse_args(argv)

    index = build_model_index(args.repo_or_url, revision=args.revision)
    a = np.load(args.assignment)
    if args.assignment_mapping:
        names = json.loads(Path(args.assignment_mapping).read_text()).get("int_to_format", MIXED_TILE_FORMATS)
        a = np.vectorize(lambda v: MIXED_TILE_FORMATS.index(names[int(v)]))(a)
    quantizer = Quantizer(args.backend)
    x = index.load(args.tensor_name)
    if args.backend == "hip":
        from quantization_analysis_amd import hip_backend as hb

        x2d, info = hb.to_device_2d(x)
        th, tw = hb.tiles_hw(*x2d.shape)
        ts = TileStats(0, th, tw, int(x.numel()), info, x2d, "hip", True)
    else:
        xf = x.float().numpy()
        x2d, info = flatten_2d(xf)
        th, tw = -(-x2d.shape[0] // 32), -(-x2d.shape[1] // 32)
        ts = TileStats(0, th, tw, int(xf.size), info, x2d, args.backend, True)
    if a.size != th * tw:
        print(f"error: assignment has {a.size} entries, tensor has {th}x{tw} tiles")
        return 1
    y = reconstruct(ts, a.astype(np.int8), quantizer)
    out = args.out or str(Path(args.assignment).with_suffix("")) + "_recon.npy"
    np.save(out, np.asarray(y, dtype=np.float32))
    print(f"wrote {out} {np.asarray(y).shape}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
