import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "particle-sfm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


# The oracle's OpenMP team idles between the thousands of short solves the property tests make: on a small shared box let the
# threads sleep instead of spinning (the spin competes with the test itself; results do not depend on it).  The many-core GPU
# boxes keep the default: there the oracle walks whole sequences through thousands of back-to-back parallel regions.
if (os.cpu_count() or 1) <= 32:
    os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: whole BASELINE.json sequences against the one-core CPU oracle (minutes)")
