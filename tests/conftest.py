"""pytest configuration: markers, import paths, golden loader."""
import importlib
import os
import sys

# before anything touches the HIP runtime (pytorch-deepfepe_amd/__init__.py explains; the package sets it too, on import)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

import numpy as np
import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dfepe():
    """The product package (directory name has a hyphen, so it is imported through importlib)."""
    return importlib.import_module("pytorch-deepfepe_amd")


@pytest.fixture(scope="session")
def oracle():
    return importlib.import_module("oracle.deepf_oracle")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)

    return load
