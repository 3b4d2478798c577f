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


def needs_ablation_build(fep):
    """Skips unless the loaded library was built with -DFEP_ABLATION (`python fem-elastoplasticity_amd/build.py --ablation`,
    then FEP_LIB_PATH=.../csrc/libfep_hip_abl.so): the tests that compare a kernel variant behind a measurement switch with
    the product's form.  The product library has no such switches."""
    if not fep.lib().fep_build_is_ablation():
        pytest.skip('needs the -DFEP_ABLATION build of the library (FEP_LIB_PATH=.../libfep_hip_abl.so)')


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


def relerr_points(a, b):
    """Worst integration point: max over points k of  max_i |a[i,k] - b[i,k]| / max_i |b[i,k]|  for (rows, n_int)
    arrays (a point whose reference column is all zero - the apex tangent - must be reproduced exactly: its error
    counts as absolute).  `relerr` scales with the array's global maximum and hides errors in small points."""
    a = np.asarray(a, dtype=float).reshape(-1, np.shape(a)[-1])
    b = np.asarray(b, dtype=float).reshape(-1, np.shape(b)[-1])
    if a.size == 0:
        return 0.0
    d = np.abs(a - b).max(axis=0)
    s = np.abs(b).max(axis=0)
    return float(np.where(s > 0, d / np.where(s > 0, s, 1.0), d).max())


def relerr_rows(A, B):
    """Worst matrix row: max over rows i of  max_j |A_ij - B_ij| / max_j |B_ij|  (dense or SciPy sparse, same shape)."""
    import scipy.sparse as ssp
    if ssp.issparse(A) or ssp.issparse(B):
        A, B = ssp.csr_matrix(A), ssp.csr_matrix(B)
        D = abs(A - B).tocsr()
        d = np.asarray(D.max(axis=1).todense()).ravel()
        s = np.asarray(abs(B).max(axis=1).todense()).ravel()
    else:
        d = np.abs(np.asarray(A) - np.asarray(B)).max(axis=1)
        s = np.abs(np.asarray(B)).max(axis=1)
    return float(np.where(s > 0, d / np.where(s > 0, s, 1.0), d).max()) if d.size else 0.0


@pytest.fixture(scope='session')
def tsx_csv_dir(tmp_path_factory):
    """coord.csv / elem.csv of the tsx-tunnel mesh in the reference's on-disk format (2 x n_n coordinates; 3 x n_e
    vertex ids, 1-BASED; comma separated), written from the arrays recorded in tsx.npz — the files themselves are not
    kept in the repository."""
    g = load_golden('tsx')
    d = tmp_path_factory.mktemp('tsx_csv')
    np.savetxt(d / 'coord.csv', g['coord'], delimiter=',', fmt='%.17g')
    np.savetxt(d / 'elem.csv', g['elem'] + 1, delimiter=',', fmt='%d')
    return str(d)
