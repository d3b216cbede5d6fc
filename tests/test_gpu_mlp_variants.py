"""The symmetric ("lockstep") MLP kernels stay in the library as the comparison point for the specialised-wave kernels and are
selected by environment switches that the library reads once per process: run the MLP operator tests once more in a child process with the
switches set, so that both forms stay checked against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("switches", [{"KASF_MLP_FWD_LOCKSTEP": "1", "KASF_MLP_BWD_LOCKSTEP": "1"}, {"KASF_MLP_BWD_XCHG": "1"}])
def test_symmetric_mlp_kernels_still_match(switches):
    env = dict(os.environ, **switches)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-m", "gpu", "-k", "mlp"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
