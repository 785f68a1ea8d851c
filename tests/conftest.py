import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference oracle/_ref/libse_ref.so")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "golden_digests.json")) as f:
        dig = json.load(f)
    with open(os.path.join(g, "ref_kats.json")) as f:
        kats = json.load(f)
    c1 = dict(np.load(os.path.join(g, "golden_c1.npz")))
    return {"digests": dig, "kats": kats, "c1": c1}
