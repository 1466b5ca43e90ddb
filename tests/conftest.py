import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)

PKG = 'arbitrary-hands-3d-reconstruction_amd'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pkg(sub=None):
    return importlib.import_module(PKG + ('.' + sub if sub else ''))


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope='session')
def synth_sd():
    return pkg('synth').make_state_dict(seed=0)


@pytest.fixture(scope='session')
def mano_tables():
    return pkg('synth').make_mano_tables(seed=1)


@pytest.fixture(scope='session')
def frames2():
    return pkg('synth').make_frames(2, seed=0)
