"""GPU: short fixed-seed runs of the randomised sweeps (tests/sweep_parity.py,
tests/sweep_matching.py)."""
import pytest

import sweep_matching
import sweep_parity

pytestmark = pytest.mark.gpu


def test_random_sizes_kinds_and_parameters_match_the_oracle(hip):
    assert sweep_parity.sweep(12, seed=2026, verbose=False) > 500


def test_random_matching_cases_match_the_oracle(hip):
    assert sweep_matching.sweep(25, seed=2026, verbose=False) > 100
