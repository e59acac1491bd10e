import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def gold():
    import numpy as np
    import torch

    def load(name):
        z = np.load(os.path.join(GOLD, name + '.npz'))
        return {k: torch.from_numpy(z[k]) for k in z.files}
    return load


@pytest.fixture(scope='session')
def manifest():
    import json

    def load(name):
        with open(os.path.join(GOLD, name + '.json')) as f:
            return json.load(f)
    return load
