"""Host-side pieces of the streamed drivers that need no GPU: the batched column formulas equal the C ones bit for bit,
and the drivers refuse to exist without a device (no CPU fallback)."""
import numpy as np
import pytest

from oracle import mtq_oracle as orc
from quantization_analysis_amd import hip_backend as hb
from quantization_analysis_amd.pipeline import GreedyPipeline, ThresholdPipeline, columns_from_sums_batch
from tests.inputs import gen

ALL = ["bf16", "bfp8", "bfp4", "bfp2"]


def test_batched_column_formulas_equal_the_c_ones():
    rows = []
    for seed, kind in enumerate(("normal_bf16", "heavy_f32", "heavy_bf16", "normal_f32")):
        x = gen(kind, 200 + seed, (96, 160))
        st = orc.tile_stats(x, ALL)
        amap = np.random.default_rng(seed).integers(0, 4, st.shape[0]).astype(np.int8)
        c = hb.columns_from_stats(st, 0xF, amap, float(x.size))
        rows.append((np.array(list(c["sums"]) + [c["atol"]]), x.size, c))
    rows.append((np.array([3.0, 9.0, 3.0, 9.0, 9.0, 0.0, 0.0]), 1, None))          # zero variance, identical → pcc 1
    rows.append((np.array([3.0, 9.0, 2.0, 4.0, 6.0, 1.0, 1.0]), 1, None))          # zero variance, different → pcc 0
    for sums, n, c in rows:
        got = columns_from_sums_batch(sums[None, :], float(n))[0]
        want = hb.columns_from_sums(sums, float(n))
        assert (got[0], got[1], got[2]) == (want["pcc"], want["mae"], want["atol"])
        if c is not None:
            assert (want["pcc"], want["mae"], want["atol"]) == (c["pcc"], c["mae"], c["atol"])


def test_drivers_need_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hb.MtqError):
        GreedyPipeline(ALL, "pcc", 0.999, 123)
    with pytest.raises(hb.MtqError):
        ThresholdPipeline(ALL, "pcc", 0.999)
