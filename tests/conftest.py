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


def pytest_collection_modifyitems(config, items):
    # `-m gpu` / `-m "not gpu"` do the selection; additionally skip gpu tests when no device is visible
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __getitem__(self, k):
        a = self.z[k]
        if a.dtype.kind in "US":
            return a
        t = torch.from_numpy(np.array(a))
        return t

    def scalar(self, k):
        return self.z[k].item()

    def keys(self):
        return list(self.z.keys())

    def spec(self):
        keys = [str(k) for k in self.z["spec_keys"]]
        shapes = [tuple(int(x) for x in str(s).split(",")) if str(s) else () for s in self.z["spec_shapes"]]
        return list(zip(keys, shapes))


@pytest.fixture
def golden():
    return Golden
