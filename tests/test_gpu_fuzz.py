"""The randomised cross-checks of tools/ (random lattices, operators, option sets, meshes, rotations, and one handle driven through random
sequences of changes) as short, seeded runs inside the suite: a few hundred cases each against the CPU oracle.  The long campaigns
(tens of thousands of cases) are run by hand on the GPU box; DESIGN.md section 2 has their record."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,seed", [("fuzz_recursion.py", 1), ("fuzz_stages.py", 1), ("fuzz_misc.py", 1), ("fuzz_stateful.py", 1)])
def test_seeded_fuzz_run(tool, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "8", str(seed)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="8"))
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    assert "0 failures" in tail and not tail.startswith("0 "), tail          # some cases ran, none failed
