"""A short run of the randomised GPU-vs-oracle sweep (tests/fuzz_parity.py) in the regular GPU suite."""
import pytest


@pytest.mark.gpu
def test_random_shapes_values_and_masks_match_the_oracle_bit_for_bit():
    from tests.fuzz_parity import run

    assert run(300, 20261004) == 0
