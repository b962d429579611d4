"""A short run of tools/scale_sweep.py inside the GPU suite: property checks (encode -> decode == the set, pair
algebra identities, every Get(i) of the loop) over nine (K, N) geometries -- 2-, 4- and 8-byte keys -- at
sizes between the oracle-checked parity cases and the full-size tests, where the 2-byte-key fault of round 3
sat.  The tool runs for minutes with other seeds; here half a minute with a fixed one."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_scale_sweep_half_a_minute(gpu):
    here = os.path.dirname(os.path.abspath(__file__))
    tool = os.path.join(here, "..", "tools", "scale_sweep.py")
    out = subprocess.run([sys.executable, tool, "--seed", "11", "--seconds", "30", "--max-size", "4e7"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    last = out.stdout.strip().splitlines()[-1]
    assert last.startswith("sweep ok") and int(last.split(",")[1].split()[0]) >= 50, last
