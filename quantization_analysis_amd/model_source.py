"""Offline tensor sources for the wq CLI.

The reference builds its model index from the Hugging Face Hub (hf_model_utils.py:135-196, network only).  Neither the
build container nor the GPU box has a network, so `repo_or_url` is resolved here to
  * `synthetic:<preset>[:seed]` — random-init tensors with the names, shapes and dtypes of a public model
    (SURVEY §8(d) M2–M4: `gpt2`, `deepseek-r1-layer0`, `llama3-8b`, `tiny`), or
  * a local directory of `*.safetensors` files (same header parsing idea as hf_model_utils.py:80-90; bf16 stays bf16).
Name filtering keeps the reference's semantics (hf_model_utils.py:60-77, 290-301).
"""
from __future__ import annotations

import hashlib
import re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Callable, Optional

import numpy as np


def filter_tensor_names(names: list[str], query: Optional[str]) -> list[str]:
    """reference hf_model_utils.py:60-77: dotted query = case-insensitive prefix match on '.'-split parts,
    otherwise substring match."""
    if not query:
        return sorted(names)
    trimmed = query.strip()
    if not trimmed:
        return sorted(names)
    if "." in trimmed:
        qparts = [p.lower() for p in trimmed.split(".") if p]
        out = []
        for name in names:
            parts = name.lower().split(".")
            if len(parts) >= len(qparts) and parts[: len(qparts)] == qparts:
                out.append(name)
        return sorted(out)
    needle = trimmed.lower()
    return sorted([n for n in names if needle in n.lower()])


def resolve_format_list(values: Optional[list[str]], supported: list[str]) -> list[str]:
    """reference hf_model_utils.py:317-335."""
    if not values:
        return supported
    seen = set()
    out: list[str] = []
    for raw in values:
        v = raw.strip().lower()
        if v == "all":
            for s in supported:
                if s not in seen:
                    seen.add(s)
                    out.append(s)
            continue
        if v not in supported:
            raise ValueError(f"Unsupported format '{raw}'. Supported: {', '.join(supported)}, all")
        if v not in seen:
            seen.add(v)
            out.append(v)
    return out


def safe_repo_revision_key(repo_id: str, revision: str) -> str:
    """reference hf_model_utils.py:114-118."""
    digest = hashlib.sha1(f"{repo_id}@{revision}".encode("utf-8")).hexdigest()[:12]
    safe_repo = repo_id.replace("/", "__")
    safe_rev = re.sub(r"[^A-Za-z0-9._-]+", "_", revision)
    return f"{safe_repo}--{safe_rev}--{digest}"


def dequantize_with_scale_inv(tensor, inv_scale):
    """Host form of the reference's block dequantisation (hf_model_utils.py:199-215)."""
    assert tensor.ndim == inv_scale.ndim
    for i, (ts, ss) in enumerate(zip(tensor.shape, inv_scale.shape)):
        inv_scale = inv_scale.repeat_interleave(max(1, -(-int(ts) // int(ss))) if ss > 0 else 1, dim=i)
    slices = tuple(slice(0, int(s)) for s in tensor.shape)
    return tensor.float() * inv_scale[slices].float()


@dataclass
class TensorSpec:
    shape: tuple
    dtype: str  # "bf16" | "f32"
    seed: int
    kind: str = "normal"  # "normal": N(0, 0.02²); "ones": layernorm-like; "fp8block": fp8-e4m3-like values × 128x128 block scales


@dataclass
class ModelIndex:
    repo_id: str
    revision: str
    specs: dict = field(default_factory=dict)             # name -> TensorSpec (synthetic)
    files: dict = field(default_factory=dict)             # name -> safetensors path (local)

    @property
    def tensor_names(self) -> list[str]:
        return list(self.specs) + list(self.files)

    def numel(self, name: str) -> int:
        if name in self.specs:
            return int(np.prod(self.specs[name].shape)) if self.specs[name].shape else 1
        from safetensors import safe_open

        with safe_open(self.files[name], framework="pt") as f:
            return int(np.prod(f.get_slice(name).get_shape()))

    def shape_dtype(self, name: str) -> tuple:
        """(shape, "bf16" | "f32") of what load() returns, without loading (safetensors: from the header)."""
        if name in self.specs:
            return tuple(self.specs[name].shape), self.specs[name].dtype
        from safetensors import safe_open

        with safe_open(self.files[name], framework="pt") as f:
            sl = f.get_slice(name)
            return tuple(sl.get_shape()), ("bf16" if str(sl.get_dtype()).upper() in ("BF16", "BFLOAT16") and f"{name}_scale_inv" not in self.files else "f32")

    def load(self, name: str, device=None, draw_on_device: bool = False):
        """→ torch tensor (bf16 kept as bf16, everything else float32) on `device` (None = host).  draw_on_device: a synthetic
        tensor is drawn by the device's generator (seeded per tensor) instead of the CPU's — other values of the same distribution,
        three orders of magnitude faster for a whole model (bench.py's model-sized workloads, where the loader must not be what is
        measured); the default keeps one set of values for every backend and rank."""
        import torch

        if name in self.files:
            from safetensors import safe_open

            with safe_open(self.files[name], framework="pt") as f:
                t = f.get_tensor(name)
            scale_name = f"{name}_scale_inv"
            if scale_name in self.files and not name.endswith("_scale_inv"):  # hf_model_utils.py:273-281
                with safe_open(self.files[scale_name], framework="pt") as f:
                    sc = f.get_tensor(scale_name)
                if device is not None and t.dtype == torch.float8_e4m3fn and t.dim() == 2 and sc.dim() == 2:
                    from . import hip_backend as hb

                    return hb.dequant_fp8_block(t.to(device), sc.to(device))  # K5 on the GPU
                t = dequantize_with_scale_inv(t, sc)
                return t.to(device) if device is not None else t
            if t.dtype != torch.bfloat16:
                t = t.to(torch.float32)
            return t.to(device) if device is not None else t
        spec = self.specs[name]
        dev = device if (draw_on_device and device is not None) else "cpu"  # CPU generator by default: the same tensor whatever backend / rank evaluates it
        g = torch.Generator(device=dev)
        g.manual_seed(spec.seed)
        shape = spec.shape if spec.shape else (1,)
        if spec.kind == "ones":
            t = torch.ones(shape, device=dev) + 0.01 * torch.randn(shape, generator=g, device=dev)
        elif spec.kind == "fp8block":
            # DeepSeek-style: fp8-e4m3-like mantissas (3 bits) times a 128x128 block scale (hf_model_utils.py:199-215)
            base = torch.randn(shape, generator=g, device=dev)
            m, e = torch.frexp(base)
            base = torch.ldexp(torch.round(m * 16) / 16, e)
            bh, bw = -(-shape[0] // 128), -(-shape[1] // 128)
            scale = torch.exp(torch.randn((bh, bw), generator=g, device=dev)) * 0.01
            t = base * scale.repeat_interleave(128, 0)[: shape[0]].repeat_interleave(128, 1)[:, : shape[1]]
        else:
            t = torch.randn(shape, generator=g, device=dev) * 0.02
        t = t.reshape(spec.shape)
        t = t.to(torch.bfloat16) if spec.dtype == "bf16" else t.to(torch.float32)
        return t.to(device) if device is not None else t


def _preset(name: str, seed0: int) -> dict:
    specs: dict[str, TensorSpec] = {}

    def add(n, shape, dtype, kind="normal"):
        specs[n] = TensorSpec(tuple(shape), dtype, seed0 + len(specs), kind)

    if name == "gpt2":  # fp32 checkpoint; h.0 block (Conv1D weights are [in, out])
        add("wte.weight", (50257, 768), "f32")
        add("h.0.ln_1.weight", (768,), "f32", "ones")
        add("h.0.attn.c_attn.weight", (768, 2304), "f32")
        add("h.0.attn.c_proj.weight", (768, 768), "f32")
        add("h.0.ln_2.weight", (768,), "f32", "ones")
        add("h.0.mlp.c_fc.weight", (768, 3072), "f32")
        add("h.0.mlp.c_proj.weight", (3072, 768), "f32")
    elif name == "deepseek-r1-layer0":  # model.layers.0.self_attn tensors (public config shapes)
        p = "model.layers.0.self_attn."
        add(p + "q_a_proj.weight", (1536, 7168), "f32", "fp8block")
        add(p + "q_a_layernorm.weight", (1536,), "bf16", "ones")
        add(p + "q_b_proj.weight", (24576, 1536), "f32", "fp8block")
        add(p + "kv_a_proj_with_mqa.weight", (576, 7168), "f32", "fp8block")
        add(p + "kv_a_layernorm.weight", (512,), "bf16", "ones")
        add(p + "kv_b_proj.weight", (32768, 512), "f32", "fp8block")
        add(p + "o_proj.weight", (7168, 16384), "f32", "fp8block")
    elif name == "llama3-8b":  # 32 layers x 7 linear weights, bf16
        for layer in range(32):
            p = f"model.layers.{layer}."
            add(p + "self_attn.q_proj.weight", (4096, 4096), "bf16")
            add(p + "self_attn.k_proj.weight", (1024, 4096), "bf16")
            add(p + "self_attn.v_proj.weight", (1024, 4096), "bf16")
            add(p + "self_attn.o_proj.weight", (4096, 4096), "bf16")
            add(p + "mlp.gate_proj.weight", (14336, 4096), "bf16")
            add(p + "mlp.up_proj.weight", (14336, 4096), "bf16")
            add(p + "mlp.down_proj.weight", (4096, 14336), "bf16")
    elif name == "tiny":  # test preset: mixed dtypes, ragged shapes, a vector
        add("model.layers.0.attn.q.weight", (96, 160), "bf16")
        add("model.layers.0.attn.k.weight", (50, 70), "f32")
        add("model.layers.0.norm.weight", (100,), "bf16", "ones")
        add("model.layers.1.attn.q.weight", (64, 256), "bf16")
        add("model.layers.1.mlp.up.weight", (130, 200), "f32")
        add("model.layers.1.mlp.up.weight_scale_inv", (2, 2), "f32")
        add("lm_head.bias", (33,), "f32")
    else:
        raise ValueError(f"Unknown synthetic preset '{name}'. Known: gpt2, deepseek-r1-layer0, llama3-8b, tiny")
    return specs


def build_model_index(repo_or_url: str, revision: str = "main") -> ModelIndex:
    """Offline replacement of hf_model_utils.build_model_index (:135-196)."""
    if repo_or_url.startswith("synthetic:"):
        parts = repo_or_url.split(":")
        preset = parts[1]
        seed0 = int(parts[2]) if len(parts) > 2 else 0
        return ModelIndex(repo_id=f"synthetic/{preset}", revision=revision, specs=_preset(preset, seed0))
    path = Path(repo_or_url)
    if path.is_dir():
        from safetensors import safe_open

        files = {}
        for fp in sorted(path.glob("*.safetensors")):
            with safe_open(str(fp), framework="pt") as f:
                for k in f.keys():
                    files[k] = str(fp)
        if not files:
            raise RuntimeError(f"{path}: no *.safetensors files")
        return ModelIndex(repo_id=path.resolve().name, revision=revision, files=files)
    raise RuntimeError(
        f"'{repo_or_url}' is neither 'synthetic:<preset>' nor a local directory of *.safetensors. "
        "Downloading from the Hugging Face Hub (reference hf_model_utils.py:145-157) is not available in this build."
    )


def resolve_selected_tensors(index: ModelIndex, filter_query: Optional[str]) -> list[str]:
    """reference hf_model_utils.py:290-301."""
    all_names = index.tensor_names
    weight_like = [n for n in all_names if "weight" in n.lower() and not n.lower().endswith("_scale_inv")]
    selected = filter_tensor_names(weight_like if weight_like else all_names, filter_query)
    if not selected:
        selected = filter_tensor_names(all_names, filter_query)
    if not selected:
        raise RuntimeError("No tensors matched the filter query.")
    return selected


def lpt_shards(names: list[str], numel: Callable[[str], int], world: int) -> list[list[int]]:
    """Longest-processing-time assignment of tensor INDICES to ranks by element count;
    deterministic, so every rank computes the same partition without communication."""
    order = sorted(range(len(names)), key=lambda i: (-numel(names[i]), i))
    load = [0] * world
    shards: list[list[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += numel(names[i])
    return [sorted(s) for s in shards]
