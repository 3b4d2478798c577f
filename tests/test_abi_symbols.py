"""The C-ABI library builds, loads without a GPU and exports every symbol include/fep.h declares.
No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def built(fep):
    fep.build()
    return fep


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'fep.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    names = re.findall(r'\b(fep_[a-z0-9_]+)\s*\(', txt)
    return sorted(set(names))


def test_header_symbols_are_exported(built):
    names = _declared()
    assert len(names) >= 24
    l = ctypes.CDLL(built.lib_path())
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, missing
    # and the Python binding table covers exactly the header
    from importlib import import_module
    proto = import_module('fem-elastoplasticity_amd._lib').PROTOTYPES
    assert sorted(proto) == names
    # and nothing else is exported under the library's prefix: exports == header
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', built.lib_path()], stdout=subprocess.PIPE, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith('fep_')
                       and ln.split()[-2] in ('T', 't', 'W')})
    assert exported == names, sorted(set(exported) ^ set(names))


def test_library_metadata_calls(built):
    l = built.lib()
    assert l.fep_version() == 1
    assert l.fep_strerror(0) == b'ok' and l.fep_strerror(-5) == b'index out of range'
    for t, (p, q) in {1: (3, 1), 2: (6, 7), 3: (4, 4), 4: (8, 9), 5: (15, 12)}.items():
        a, b = ctypes.c_int(), ctypes.c_int()
        assert l.fep_element_shape(t, ctypes.byref(a), ctypes.byref(b)) == 0 and (a.value, b.value) == (p, q)
    assert l.fep_element_shape(0, None, None) == -1
    assert l.fep_ctx_destroy(None) == 0 and l.fep_ctx_sizes(None, None) == -1


def test_gfx950_code_object_present(built):
    blob = open(built.lib_path(), 'rb').read()
    assert b'gfx950' in blob and b'element_kernel' in blob


def test_missing_library_fails_loudly(built, monkeypatch):
    from importlib import import_module
    m = import_module('fem-elastoplasticity_amd._lib')
    monkeypatch.setattr(m, '_LIB', None)
    monkeypatch.setattr(m, 'lib_path', lambda: '/nonexistent/libfep_hip.so')
    with pytest.raises(ImportError, match='no CPU fallback'):
        m.lib()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'fem-elastoplasticity_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle' not in txt.replace('no oracle', ''), f


def test_aggregate_host_without_gpu(built):
    """fep_aggregate_host is pure host code: every node lands in exactly one aggregate, aggregates are connected
    neighbourhoods of a root (pass 1) plus attached leftovers (pass 2)."""
    import numpy as np
    import scipy.sparse as ssp
    n = 13
    idx = np.arange(n * n).reshape(n, n)
    rows, cols = [], []
    for di, dj in ((0, 0), (0, 1), (1, 0), (0, -1), (-1, 0), (1, 1), (-1, -1)):           # P1 node graph (7-point)
        a = idx[max(0, -di):n - max(0, di), max(0, -dj):n - max(0, dj)]
        b = idx[max(0, di):n - max(0, -di), max(0, dj):n - max(0, -dj)]
        rows.append(a.ravel()); cols.append(b.ravel())
    G = ssp.csr_matrix((np.ones(sum(r.size for r in rows)), (np.concatenate(rows), np.concatenate(cols))), shape=(n * n, n * n))
    ip, ix = G.indptr.astype(np.int32), G.indices.astype(np.int32)
    agg = np.full(n * n, -7, dtype=np.int32)
    na = ctypes.c_int64()
    rc = built.lib().fep_aggregate_host(n * n, ip.ctypes.data_as(ctypes.c_void_p), ix.ctypes.data_as(ctypes.c_void_p),
                                        agg.ctypes.data_as(ctypes.c_void_p), ctypes.byref(na))
    assert rc == 0 and 0 < na.value < n * n // 3
    assert agg.min() == 0 and agg.max() == na.value - 1 and np.unique(agg).size == na.value
    A = ssp.csr_matrix((np.ones(n * n), (np.arange(n * n), agg)), shape=(n * n, na.value))
    sizes = np.asarray(A.sum(axis=0)).ravel()
    assert sizes.min() >= 1 and sizes.max() <= 2 * 7
    # connected: the sub-graph of every aggregate has one component
    for a in range(na.value):
        m = np.flatnonzero(agg == a)
        assert ssp.csgraph.connected_components(G[m][:, m], directed=False)[0] == 1
    assert built.lib().fep_aggregate_host(0, None, None, None, None) == -1
