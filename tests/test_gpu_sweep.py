"""GPU: a short fixed-seed run of the randomised parity sweep (tests/sweep_parity.py)."""
import pytest

import sweep_parity

pytestmark = pytest.mark.gpu


def test_random_sizes_kinds_and_parameters_match_the_oracle(hip):
    assert sweep_parity.sweep(12, seed=2026, verbose=False) > 500
