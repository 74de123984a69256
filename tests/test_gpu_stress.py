"""The randomised stress scripts as bounded `-m gpu` tests (fixed seeds, a few dozen random cases each): two shipped bugs of rounds 2-4
were found only by these (a last frame run twice; tickets left mid-count after a capacity overflow), so they run wherever the parity
tests run.  Every script compares a form of the product path with another on the same random input (or with the CPU oracle) and exits 1
on the first difference; here that is a failed test with the script's output.  Longer soaks: run the scripts by hand (scripts/README.md)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (script, arguments, environment): sized for ~5-15 s each on an MI355X box
CASES = [
    ("stress_batch.py", ["14", "101"], {}),            # psfm_connect_batch vs one psfm_connect per sequence
    ("stress_batch.py", ["3", "102"], {"PSFM_STRESS_BIG": "1"}),   # ... at DAVIS / Sintel-sized frames
    ("stress_optimize.py", ["14", "103"], {}),         # track_optimize on the device vs the CPU oracle
    ("stress_sharded.py", ["14", "104"], {}),          # connect_sharded at 1 / 2 / 3 (thread-)ranks vs psfm_connect
    ("stress_threads.py", ["3", "105"], {}),           # several host threads, every way of setting the workers up
    ("stress_persist.py", ["30", "106"], {}),          # the persistent loop vs one launch per frame
    ("stress_consumers.py", ["25", "107"], {}),        # traj_to_matches on the device vs the host tables
    ("stress_ingest.py", ["40", "108"], {}),           # .flo stacks through the native reader, damaged files refused by name
]


@pytest.mark.parametrize("script,args,env", CASES, ids=["%s-%s%s" % (c[0][:-3], c[1][1], "-big" if c[2] else "") for c in CASES])
def test_stress_script(script, args, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + args, capture_output=True, text=True, env=e, timeout=420)
    tail = (r.stdout + r.stderr)[-2500:]
    assert r.returncode == 0, tail
    assert "Traceback" not in r.stderr, tail
