import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


@pytest.fixture(scope='session')
def fep():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module('fem-elastoplasticity_amd')


def dp_materials(n_int):
    """Material constants of the reference demo (DP:910-933)."""
    young, poisson, c0, phi = 1e7, 0.48, 450, np.pi / 9
    shear = young / (2 * (1 + poisson)) * np.ones(n_int)
    bulk = young / (3 * (1 - 2 * poisson)) * np.ones(n_int)
    eta = 3 * np.tan(phi) / np.sqrt(9 + 12 * np.tan(phi) ** 2) * np.ones(n_int)
    c = 3 * c0 / np.sqrt(9 + 12 * np.tan(phi) ** 2) * np.ones(n_int)
    return shear, bulk, eta, c


def relerr(a, b):
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    d = np.abs(a - b).max() if a.size else 0.0
    s = max(np.abs(b).max() if b.size else 0.0, 1e-300)
    return d / s
