"""Short randomised stress of the search path on the GPU (tools/stress_search.py runs the same loop for minutes):
random index shapes, row-length laws, value stores, batch sizes, k, column skew, every kernel variant, each result
validated against the independent scores-only kernel."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_cases_all_variants():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import stress_search
    assert stress_search.run(budget=20.0, seed0=7) >= 5
