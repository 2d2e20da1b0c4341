"""Guard of the "from scratch" rule (build container only): no function of
tike_amd/ with >= 8 body lines may resemble its namesake in the reference.
`tools/similarity.py` compares token streams (docstrings stripped, formatting
normalised).  The limit is 0.6, on the whole function (signature included,
although parameter lists are the drop-in contract) and on the body alone.
Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/tike"),
                    reason="needs the reference tree (build container only)")
def test_no_function_resembles_its_reference_namesake():
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tools", "similarity.py"),
         "--threshold", "0.6", "--min-lines", "8"],
        capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:]
    listing = subprocess.run(
        [sys.executable, os.path.join(ROOT, "tools", "similarity.py"),
         "--threshold", "0.0", "--min-lines", "8"],
        capture_output=True, text=True, timeout=600).stdout
    bodies = [float(line.split("(body ")[1].split(")")[0])
              for line in listing.splitlines() if "(body " in line]
    assert bodies and max(bodies) < 0.6, max(bodies)
