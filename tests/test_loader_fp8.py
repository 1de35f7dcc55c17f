"""K5: float8-e4m3fn x inverse block scale dequantisation of the loader against vectors produced by the
reference's `_dequantize_tensor_with_scale_inv` (tests/golden/f9_fp8_dequant.npz)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import mtq_oracle as orc
from quantization_analysis_amd import model_source

TAGS = ("all_codes", "blocks128", "ragged", "rowscale")


def same_bits(a: np.ndarray, b_bits: np.ndarray) -> bool:
    a_bits = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    nan = np.isnan(a) & np.isnan(b_bits.view(np.float32))
    return np.array_equal(np.where(nan, 0, a_bits), np.where(nan, 0, b_bits))


def test_oracle_and_host_loader_match_reference(golden_dir):
    d = np.load(golden_dir / "f9_fp8_dequant.npz")
    for tag in TAGS:
        w, sc, want = d[f"{tag}_w"], d[f"{tag}_scale"], d[f"{tag}_out_bits"]
        assert same_bits(orc.dequant_fp8_block(w, sc), want), tag
        host = model_source.dequantize_with_scale_inv(torch.from_numpy(w).view(torch.float8_e4m3fn), torch.from_numpy(sc)).numpy()
        assert same_bits(host, want), tag


@pytest.mark.gpu
def test_k5_matches_reference(golden_dir):
    from quantization_analysis_amd import hip_backend as hb

    d = np.load(golden_dir / "f9_fp8_dequant.npz")
    for tag in TAGS:
        w, sc, want = d[f"{tag}_w"], d[f"{tag}_scale"], d[f"{tag}_out_bits"]
        got = hb.dequant_fp8_block(torch.from_numpy(w).cuda().view(torch.float8_e4m3fn), torch.from_numpy(sc).cuda()).cpu().numpy()
        assert same_bits(got, want), tag
    # a strided (non-16-byte-aligned) view takes the scalar path
    w, sc = d["blocks128_w"], d["blocks128_scale"]
    got = hb.dequant_fp8_block(torch.from_numpy(w).cuda()[:, 3:250], torch.from_numpy(sc[:, :2]).cuda()).cpu().numpy()
    assert same_bits(got, orc.dequant_fp8_block(w[:, 3:250], sc[:, :2]).view(np.uint32))


@pytest.mark.gpu
def test_local_safetensors_loader_dequantises_on_device(tmp_path, golden_dir):
    from safetensors.torch import save_file

    d = np.load(golden_dir / "f9_fp8_dequant.npz")
    w, sc = d["blocks128_w"], d["blocks128_scale"]
    try:
        save_file({"layer.w.weight": torch.from_numpy(w).view(torch.float8_e4m3fn), "layer.w.weight_scale_inv": torch.from_numpy(sc),
                   "layer.norm.weight": torch.ones(64, dtype=torch.bfloat16)}, str(tmp_path / "m.safetensors"))
    except Exception as exc:  # an older safetensors without float8 support
        pytest.skip(f"safetensors cannot store float8: {exc}")
    idx = model_source.build_model_index(str(tmp_path))
    assert model_source.resolve_selected_tensors(idx, None) == ["layer.norm.weight", "layer.w.weight"]
    got = idx.load("layer.w.weight", device=torch.device("cuda", 0))
    assert got.is_cuda and got.dtype == torch.float32 and same_bits(got.cpu().numpy(), d["blocks128_out_bits"])
    assert same_bits(idx.load("layer.w.weight").numpy(), d["blocks128_out_bits"])   # host path, same values
    assert idx.load("layer.norm.weight").dtype == torch.bfloat16
