import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    path = os.path.join(GOLDEN, name)
    if name.endswith(".gz"):
        with gzip.open(path, "rt") as f:
            return json.load(f) if ".json" in name else f.read()
    with open(path) as f:
        return json.load(f) if name.endswith(".json") else f.read()


@pytest.fixture(scope="session")
def O():
    import oracle
    oracle.build(ref=False)
    return oracle.Oracle()


@pytest.fixture(scope="session")
def R():
    """The real reference build; only where oracle/_ref exists (never on a box
    without /root/reference unless the prebuilt .so travelled with the snapshot)."""
    import oracle
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return oracle.Ref()
