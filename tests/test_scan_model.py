"""The device scan's wave-parallel shuffle procedure (csrc/mtq_scan.hip wave_shuffle), as modelled on the host with 64 emulated
lanes (tools/scan_model/shuffle_model.py), against numpy.random.Generator.permutation: jump-ahead draws, exact acceptance, the cut
at mask-level boundaries, hand-back of unused stream positions, parallel swaps + ordered conflicting swaps.  The HIP kernel follows
this model step for step; its own parity tests are the -m gpu tests of tests/test_hip_kernels.py."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools" / "scan_model"))
import shuffle_model as sm  # noqa: E402


def test_wave_shuffle_model_equals_numpy_permutation():
    stats = {}
    for seed in (1, 123, 2**31 - 1, 2**40 + 7):
        sm.check(seed, [1, 2, 3, 7, 64, 65, 100, 127, 128, 129, 1000, 4096, 2514, 33], stats)   # interleaved calls: generator continuity
        sm.check(seed, [161, 160, 73, 16384], stats)
    assert stats["flagged"] > 0 and stats["batches"] > 100      # the conflicting-step path was exercised


def test_level_boundary_cut_is_needed():
    """The case the model caught: a batch whose accepted draws number exactly the steps left at this mask level must still be cut
    there — the positions after the cut are re-masked at the next level (seed 1, the call sequence below, n = 2514)."""
    g = np.random.Generator(np.random.PCG64(1))
    r = sm.Rng(1)
    for n in (1, 2, 3, 7, 64, 65, 100, 127, 128, 129, 1000, 4096, 2514):
        want = list(g.permutation(n))
        arr = list(range(n))
        sm.wave_shuffle(r, n, arr)
        assert arr == want, n


def test_streamed_grouping_and_streamable():
    from types import SimpleNamespace

    from quantization_analysis_amd import model_source, streamed
    from quantization_analysis_amd.compression_algorithms import create_algorithm

    idx = model_source.build_model_index("synthetic:deepseek-r1-layer0")
    keys = {n: streamed.group_key(idx, n) for n in idx.tensor_names}
    assert keys["model.layers.0.self_attn.q_a_layernorm.weight"] is None                      # vectors keep the per-tensor path
    assert keys["model.layers.0.self_attn.o_proj.weight"] == (7168, 16384, "f32")
    args = SimpleNamespace(backend="hip", literal_metrics=False)
    fmts = ["bf16", "bfp8", "bfp4", "bfp2", "fp0"]
    assert streamed.streamable(create_algorithm("mixed-tile-greedy", {"seed": 5}), fmts, args)
    assert not streamed.streamable(create_algorithm("mixed-tile-greedy", {"seed": 0}), fmts, args)           # seed 0 = random
    assert not streamed.streamable(create_algorithm("mixed-tile-greedy", {"seed": 5, "formats": "bfp8,bfp4"}), fmts, args)  # `none` rows need bfp2 too
    assert not streamed.streamable(create_algorithm("mixed-tile-random", {"seed": 5}), fmts, args)
    assert not streamed.streamable(create_algorithm("mixed-tile-threshold", {}), fmts, SimpleNamespace(backend="emulation", literal_metrics=False))
    assert not streamed.streamable(create_algorithm("mixed-tile-threshold", {}), fmts, SimpleNamespace(backend="hip", literal_metrics=True))
    assert streamed.streamable(create_algorithm("mixed-tile-threshold", {}), fmts, args)
