import json
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


def load_golden(name):
    """Returns (arrays: dict[str, torch.Tensor], meta: dict|None, sd: dict[str, Tensor])."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    arrays, sd, meta = {}, {}, None
    for k in z.files:
        if k == "meta":
            meta = json.loads(str(z[k]))
            continue
        t = torch.from_numpy(z[k])
        if k.startswith("sd::"):
            sd[k[4:]] = t
        else:
            arrays[k] = t
    return arrays, meta, sd


@pytest.fixture(scope="session")
def golden():
    return load_golden
