"""A short random sweep of the fused launch sequences against the generic sequence of the same library (tools/fuzz_fused.py holds the loop; profiles/r06_fuzz_fused.txt
its long runs): shapes, ranks, algorithm, precision, iteration counts and downloads between iterations drawn from a seeded generator -- the fixed cases of
tests/test_gpu_fused_*.py cover the corners by hand, this one the combinations nobody thought of.  Tolerances as there: fp32 2e-5 on the factors / 1e-5 on the
error, fp64 1e-11."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_shapes_fused_against_generic():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_fused.py"), "16", "23"], capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k != "NMFAMD_NO_FUSED_MU"})
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "16 of 16 agree" in out.stdout
