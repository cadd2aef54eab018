"""Randomised parity hunt on the GPU (tools/stress_random.py): random corpus shape / column law / store / batch / k / record-layout
options; the filter search must return the 8-query CSR scan's ids and scores bit for bit in every draw."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_and_options_match_the_csr_scan(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_random.py"), "16", str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "stress_random: ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
