"""A short run of tools/hostile_fuzz.py in the suite: descriptors and states no scenario of the reference produces but the
C ABI accepts (parameters over many decades, zeros, runaway scripted cars) -- HIP path vs CPU oracle, plans and episodes,
bit for bit.  The long runs are in profiles/r04_hostile_fuzz.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_hostile_descriptors_bitwise(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hostile_fuzz.py"), "--cases", "200", "--seed", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-1500:])
    assert "done: 200 cases, 0 mismatches" in r.stdout
