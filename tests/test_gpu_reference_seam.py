"""GPU half of tests/test_reference_seam.py: the reference's own LinearEXL3 / RMSNorm / RoPE classes running over this build's `exllamav3_ext`
module on cuda:0, compared with the oracle.  Needs BOTH a GPU and /root/reference (the reference tree does not travel to the driver's GPU box,
where this test is skipped; tests/test_reference_seam.py runs the same script on CPU up to the device check)."""
import glob
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
STUB_DIR = os.path.join(ROOT, "exllamav3_amd", "stub")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "exllamav3")) or not glob.glob(os.path.join(STUB_DIR, "exllamav3_ext*.so")),
                    reason="needs /root/reference next to a GPU")
def test_reference_classes_over_the_stub_match_the_oracle(dev):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([STUB_DIR, ROOT, REF]))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_reference_seam_script.py"), "cuda:0"], capture_output=True, text=True,
                       timeout=900, env=env, cwd="/tmp")
    assert r.returncode == 0 and "REFERENCE_CALLS_OK cuda:0" in r.stdout, (r.stdout + r.stderr)[-3000:]
