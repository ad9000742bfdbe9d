"""Checkpoint interoperability with the real reference -- build container only (needs
/root/reference); skipped elsewhere.  The work happens in tests/helpers/ref_interop.py, in a subprocess, because the
reference import shim patches torch process-wide."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_reference_resume_reads_our_checkpoints_and_we_read_theirs():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "ref_interop.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "INTEROP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_oracle_generator_matches_reference_for_every_activation():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "ref_activations.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "ACTIVATIONS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference tree not present")
def test_reference_callers_accept_our_trainer():
    """utils.write_loss (utils.py:277-305) run over both trainers logs the same tags with the same types; every `trainer.<attr>`
    use of train.py / test_on_folder.py exists on ours and binds against our signatures (tests/helpers/ref_callers.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "ref_callers.py")], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "CALLERS_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
