import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The oracle's OpenMP team idles between the thousands of short solves the property tests make: let the threads sleep instead of
# spinning (on a shared box the spin competes with the test itself; results do not depend on it).
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: whole BASELINE.json sequences against the one-core CPU oracle (minutes)")
