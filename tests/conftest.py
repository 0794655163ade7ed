import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def a3d():
    """The package (its directory name starts with a digit, hence importlib)."""
    return importlib.import_module("3danimals_amd")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def kuhn(res, scale=7.0):
    tetgrid = importlib.import_module("3danimals_amd").tetgrid
    v, t = tetgrid.kuhn_grid(res)
    return torch.from_numpy(v) * scale, torch.from_numpy(t)


def seeded(shape, seed, low=0.0, high=1.0):
    return importlib.import_module("3danimals_amd").synthetic.seeded(shape, seed, low, high)
