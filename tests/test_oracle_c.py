"""Pins the oracle's C restatement (oracle/fep_oracle_c.c) against the reference's golden vectors and
the NumPy oracle.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, dp_materials, load_golden, relerr
from oracle import fep_oracle as orc

NQ = {'P1': 1, 'P2': 7, 'Q1': 4, 'Q2': 9}


@pytest.fixture(scope='module')
def clib():
    subprocess.run(['make', '-s', '-C', os.path.join(ROOT, 'oracle')], check=True)
    l = C.CDLL(os.path.join(ROOT, 'oracle', 'liboracle_c.so'))
    l.oracle_return_map.restype = None
    l.oracle_element_matrices.restype = None
    return l


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize('case', ['dp_none', 'dp_ep_accept', 'tsx_ep_accept'])
def test_c_return_map_vs_reference(clib, case):
    g = load_golden('retmap')
    n = g['E'].shape[1]
    E = np.ascontiguousarray(g['E'])
    ep = None if case == 'dp_none' else g['Ep'].copy()
    e0 = np.ascontiguousarray(g['e0'].ravel()) if case.startswith('tsx') else None
    s = np.empty((4, n)); ds = np.empty((9, n)); ind = np.empty(n, dtype=np.uint8); cnt = np.zeros(2, dtype=np.int64)
    clib.oracle_return_map(C.c_int64(n), _p(E), _p(e0), _p(ep), _p(g['shear']), _p(g['bulk']), _p(g['eta']), _p(g['c']),
                           int(case.endswith('accept')), _p(s), _p(ds), _p(ind), _p(cnt))
    assert np.array_equal(ind.astype(bool), g[case + '_ind_p'])
    assert relerr(s, g[case + '_s']) <= 1e-13 and relerr(ds, g[case + '_ds']) <= 1e-13
    if ep is not None:
        assert relerr(ep, g[case + '_ep_prev_after']) <= 1e-13
    assert cnt.sum() == g[case + '_ind_p'].sum()


@pytest.mark.parametrize('t', ['P1', 'P2', 'Q1', 'Q2'])
def test_c_element_matrices_assemble_to_reference_K(clib, t):
    g = load_golden('hotpath_dp')
    tb = load_golden('tables')
    d1, d2, wf = tb[f'dp_{t}_dhatp1'], tb[f'dp_{t}_dhatp2'], tb[f'dp_{t}_wf']
    elem, coord = g[f'{t}_elements'], g[f'{t}_coordinates']
    n_p, n_e = elem.shape
    n_q = NQ[t]
    dphi1, dphi2, w, _ = orc.geometry(elem, coord, d1, d2, wf)
    ds, s = g[f'{t}_acc0_ds'], g[f'{t}_acc0_s']
    nd = 2 * n_p
    Ke = np.empty((n_e, nd, nd)); fe = np.empty((n_e, nd))
    clib.oracle_element_matrices(n_p, n_q, C.c_int64(n_e), _p(np.ascontiguousarray(dphi1)), _p(np.ascontiguousarray(dphi2)),
                                 _p(np.ascontiguousarray(w.ravel())), _p(np.ascontiguousarray(ds)), _p(np.ascontiguousarray(s)),
                                 _p(Ke), _p(fe))
    n_dof = 2 * coord.shape[1]
    K = np.zeros((n_dof, n_dof)); F = np.zeros(n_dof)
    dofs = (2 * elem.T[:, :, None] + np.arange(2)[None, None, :]).reshape(n_e, nd)      # local DOF 2a+comp -> global
    for e in range(n_e):
        K[np.ix_(dofs[e], dofs[e])] += Ke[e]
        F[dofs[e]] += fe[e]
    assert relerr(K, g[f'{t}_acc0_K_t']) <= 1e-13
    assert relerr(F, g[f'{t}_acc0_F']) <= 1e-13
    assert np.abs(Ke - Ke.transpose(0, 2, 1)).max() <= 1e-12 * np.abs(Ke).max()
