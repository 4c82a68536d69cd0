import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "local-features_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(ROOT, "local-features_amd", "models", "mkd")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def rel_l2(a, b):
    """Per-vector relative L2 error, the metric BASELINE.json's north_star states (gate 1e-4)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.linalg.norm(b, axis=-1)


@pytest.fixture(scope="session")
def oracle():
    from oracle import MkdOracle
    return MkdOracle(os.path.join(MODELS, "concat-pca-liberty.safetensors"))


@pytest.fixture(scope="session")
def oracles():
    from oracle import MkdOracle
    return {n: MkdOracle(os.path.join(MODELS, f"concat-pca-{n}.safetensors"))
            for n in ("liberty", "notredame", "yosemite")}


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
