import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def orc():
    """the CPU oracle (test infrastructure)"""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope='session')
def cpe():
    import cpe_amd
    return cpe_amd


@pytest.fixture(scope='session')
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail('test marked gpu but no GPU is visible')
    import cpe_amd
    cpe_amd.lib.load()          # fail loudly if the HIP extension is missing
    return torch.device('cuda:0')


GOLDEN = os.path.join(ROOT, 'tests', 'golden')
