"""
Pins the oracle (oracle/fep_oracle.py) against outputs of the reference itself
(fixtures written by tests/golden/make_golden.py).  CPU only.
"""
import numpy as np
import pytest
import scipy.sparse as ssp

from conftest import dp_materials, load_golden, relerr
from oracle import fep_oracle as orc

ELS = ('P1', 'P2', 'Q1', 'Q2')
NQ = {'P1': 1, 'P2': 7, 'Q1': 4, 'Q2': 9, 'P4': 12}


def tables(mod, t):
    g = load_golden('tables')
    k = f'{mod}_{t}_'
    return g[k + 'dhatp1'], g[k + 'dhatp2'], g[k + 'wf']


# ---- a2 return map ---------------------------------------------------------
@pytest.mark.parametrize('case', ['dp_none', 'dp_ep', 'dp_ep_accept', 'tsx_ep', 'tsx_ep_accept'])
def test_return_map_vs_reference(case):
    g = load_golden('retmap')
    tsx = case.startswith('tsx')
    accept = case.endswith('accept')
    ep_in = None if case == 'dp_none' else g['Ep'].copy()
    E = g['E'].copy()
    r = orc.return_map(E, ep_in, g['shear'], g['bulk'], g['eta'], g['c'], accept,
                       e0=g['e0'] if tsx else None, tsx=tsx)
    assert np.array_equal(E, g['E']), 'e must not be modified (DP:663)'
    assert np.array_equal(r['ind_p'], g[case + '_ind_p'])
    # all three branches are present in the fixture
    assert 0 < r['n_smooth'] < E.shape[1] and 0 < r['n_apex'] and r['n_smooth'] + r['n_apex'] < E.shape[1]
    assert relerr(r['s'], g[case + '_s']) <= 1e-14
    assert relerr(r['ds'], g[case + '_ds']) <= 1e-14
    assert relerr(r['ep'], g[case + '_ep']) <= 1e-14 if np.abs(g[case + '_ep']).max() > 0 else not r['ep'].any()
    assert (r['lambda_final'] is None) == bool(g[case + '_lambda_is_none'])
    if ep_in is not None:
        assert relerr(ep_in, g[case + '_ep_prev_after']) <= 1e-14      # C4: in-place mutation on accept
        assert (r['ep'] is ep_in) == bool(g[case + '_ep_is_alias'])


def test_return_map_all_elastic_quirks():
    g = load_golden('retmap')
    sh, bu, eta, c = dp_materials(64)
    r = orc.return_map(g['Eel'], np.zeros((4, 64)), sh, bu, eta, c, True)
    assert relerr(r['s'], g['dp_elastic_s']) <= 1e-15 and relerr(r['ds'], g['dp_elastic_ds']) <= 1e-15
    assert (r['lambda_final'] is None) == bool(g['dp_elastic_lambda_is_none'])
    assert not r['ep'].any() and not g['dp_elastic_ep'].any()
    ep_in = g['tsx_elastic_ep_in'].copy()
    r = orc.return_map(g['Eel'], ep_in, sh, bu, eta, c, True, e0=np.zeros((4, 1)), tsx=True)
    assert relerr(r['s'], g['tsx_elastic_s']) <= 1e-15 and relerr(r['ds'], g['tsx_elastic_ds']) <= 1e-15
    # TSX early-out (TSX:1103): 'ep' is fresh zeros even on accept, lambda is zeros (1,n)
    assert np.array_equal(r['ep'], g['tsx_elastic_ep'])
    assert (r['lambda_final'] is None) == bool(g['tsx_elastic_lambda_is_none'])
    assert np.array_equal(r['lambda_final'], g['tsx_elastic_lambda'])


# ---- a6 elastic setup ------------------------------------------------------
@pytest.mark.parametrize('t', ELS)
def test_elastic_setup_vs_reference(t):
    g = load_golden('setup_dp')
    d1, d2, wf = tables('dp', t)
    K, B, w, iD, jD, D = orc.elastic_setup(g[f'{t}_elements'], g[f'{t}_coordinates'], g[f'{t}_shear'],
                                           g[f'{t}_bulk'], d1, d2, wf)
    assert w.shape == g[f'{t}_weight'].shape and np.array_equal(w, g[f'{t}_weight'])
    assert np.array_equal(iD, g[f'{t}_iD']) and np.array_equal(jD, g[f'{t}_jD'])
    for name, M in (('B', B), ('D', D)):
        M = M.tocsr()
        assert np.array_equal(M.indptr, g[f'{t}_{name}_indptr'])
        assert np.array_equal(M.indices, g[f'{t}_{name}_indices'])     # bit-exact indexing
        assert np.array_equal(M.data, g[f'{t}_{name}_data'])           # and values
    assert relerr(K.toarray(), g[f'{t}_K']) <= 1e-15


# ---- a1..a5 on a mesh ------------------------------------------------------
@pytest.mark.parametrize('t', ELS)
@pytest.mark.parametrize('accept', [False, True])
def test_hot_path_vs_reference(t, accept):
    g = load_golden('hotpath_dp')
    d1, d2, wf = tables('dp', t)
    elem, coord = g[f'{t}_elements'], g[f'{t}_coordinates']
    n_int = elem.shape[1] * NQ[t]
    sh, bu, eta, c = dp_materials(n_int)
    K, B, w, iD, jD, D = orc.elastic_setup(elem, coord, sh, bu, d1, d2, wf)
    assert relerr(K.toarray(), g[f'{t}_K_elast']) <= 1e-15
    ctx = dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=sh, bulk=bu, eta=eta, c=c)
    tag = f'{t}_acc{int(accept)}_'
    E, cp, K_t, F = orc.hot_path(g[f'{t}_U'], g[f'{t}_Ep_old'].copy(), ctx, apply_plastic_strain=accept)
    assert relerr(E, g[tag + 'E']) <= 1e-15
    assert np.array_equal(cp['ind_p'], g[tag + 'ind_p'])
    assert relerr(cp['s'], g[tag + 's']) <= 1e-14 and relerr(cp['ds'], g[tag + 'ds']) <= 1e-14
    assert relerr(cp['ep'], g[tag + 'ep']) <= 1e-14 if accept else not cp['ep'].any()
    assert relerr(K_t.toarray(), g[tag + 'K_t']) <= 1e-14
    assert relerr(F, g[tag + 'F']) <= 1e-14


# ---- the reference's own CSV dumps (tsx-tunnel/*.csv) -----------------------
def _tsx_setup(t, coord, elem):
    d1, d2, wf = tables('tsx', t) if t != 'P1' else tables('dp', 'P1')
    n_int = elem.shape[1] * NQ[t]
    G = 60000 / (2 * (1 + 0.2))
    Kb = 60000 / (3 * (1 - 2 * 0.2))
    out = orc.elastic_setup(elem, coord, G * np.ones(n_int), Kb * np.ones(n_int), d1, d2, wf)
    Q = np.ones(coord.shape, dtype=bool)                   # TSX:1695-1699
    Q[0, np.abs(coord[0]) > 49.99] = 0
    Q[1, np.abs(coord[1]) > 49.99] = 0
    return out, Q


def test_tsx_p1_tangent_vs_csv_dump():
    g = load_golden('tsx')
    (K, B, w, iD, jD, D), Q = _tsx_setup('P1', g['coord'], g['elem'])
    qf = Q.flatten(order='F')
    assert qf.sum() == 908
    Kqq = K.tocsr()[qf][:, qf].tocoo()
    ref = ssp.coo_matrix((g['p1_Kqq_val'], (g['p1_Kqq_row'], g['p1_Kqq_col'])), shape=(908, 908)).toarray()
    assert relerr(Kqq.toarray(), ref) <= 1e-14             # the reference itself
    csv = ssp.coo_matrix((g['kqq_val'], (g['kqq_row'], g['kqq_col'])), shape=(908, 908)).toarray()
    # k_tangent_qq.csv: MATLAB dump from un-rounded coordinates:
    # identical pattern, values to 1e-4 relative
    assert np.array_equal(csv != 0, Kqq.toarray() != 0) and (csv != 0).sum() == 12056
    assert np.abs(Kqq.toarray() - csv).max() <= 1e-4 * np.abs(csv).max()


def test_tsx_p2_initial_stress_load_vs_csv_dump():
    g = load_golden('tsx')
    (K, B, w, iD, jD, D), Q = _tsx_setup('P2', g['p2_coord'], g['p2_elem'])
    s0 = np.array([-45.0, -11.0, 0.0, -60.0]).reshape((-1, 1)) * np.ones((1, w.size))
    F0 = orc.internal_force(B, w, s0).reshape((2, -1), order='F')         # TSX:1737
    assert relerr(F0, g['p2_F0']) <= 1e-14
    assert np.array_equal(Q, g['p2_Q']) and Q.sum() == 3594
    f0q = F0.T[Q.T]
    assert np.abs(f0q - g['f0q']).max() <= 2e-4 * np.abs(g['f0q']).max()   # f0q.csv, rounded coordinates
    assert relerr(K.diagonal(), g['p2_K_diag']) <= 1e-13
    assert np.abs(g['fq']).max() < 1e-13                                   # fq.csv pins nothing numerically


def test_tsx_p4_elastic_K():
    g = load_golden('tsx')
    (K, B, w, iD, jD, D), Q = _tsx_setup('P4', g['p4_coord'], g['p4_elem'])
    assert relerr(K.diagonal(), g['p4_K_diag']) <= 1e-12
    assert abs(np.sqrt((K.tocsr().data ** 2).sum()) - g['p4_K_frob']) <= 1e-12 * g['p4_K_frob']
    assert abs(w.sum() - g['p4_weight_sum']) <= 1e-13 * g['p4_weight_sum']


# ---- config 1: Elasticity2D P1 K (SURVEY 8c pins) ---------------------------
@pytest.mark.parametrize('level,nnz,trace,frob', [(1, 7680, 4.215902547065e+08, 2.054604152327e+07),
                                                 (3, 117120, 6.745444075305e+09, 8.396141008266e+07)])
def test_el_p1_K_pins(level, nnz, trace, frob):
    g = load_golden('el_p1')
    d1, d2, wf = tables('dp', 'P1')
    elem = g[f'l{level}_elements_1based'].astype(np.int64) - 1             # EL:389
    coord = g[f'l{level}_coordinates']
    n_int = elem.shape[1]
    G = 206900 / (2 * (1 + 0.29))
    Kb = 206900 / (3 * (1 - 2 * 0.29))
    K, B, w, *_ = orc.elastic_setup(elem, coord, G * np.ones(n_int), Kb * np.ones(n_int), d1, d2, wf)
    K = K.tocsr()
    assert K.nnz == nnz == int(g[f'l{level}_nnz'])
    assert abs(K.diagonal().sum() - trace) <= 1e-12 * trace
    assert abs(np.sqrt((K.data ** 2).sum()) - frob) <= 1e-12 * frob
    assert abs(w.sum() - 75.0) <= 1e-12 * 75
    assert relerr(K @ np.cos(np.arange(K.shape[0]) * 0.37), g[f'l{level}_Kx']) <= 1e-13


def _tangent_by_differences(fn, e, h=1e-9):
    n = e.shape[1]
    fd = np.zeros((9, n))
    for j in range(3):
        de = np.zeros((3, n))
        de[j] = h
        sp, sm = fn(e + de)['s'], fn(e - de)['s']
        for i in range(3):
            fd[3 * i + j] = (sp[i] - sm[i]) / (2 * h)
    return fd


def test_ds_is_the_consistent_tangent_of_s():
    """Independent of the reference: `ds` (DP:703-741) is d s[0:3] / d e on every branch (elastic: Hooke, smooth:
    the cone's consistent tangent, apex: zero), checked by central differences with the plastic strain held fixed."""
    from conftest import dp_materials
    rng = np.random.default_rng(3)
    n = 3000
    sh, bu, eta, c = dp_materials(n)
    e = rng.normal(0, 3e-4, size=(3, n))
    e[0:2] += rng.normal(1e-4, 2e-4, size=(1, n))
    ep = rng.normal(0, 2e-5, size=(4, n))
    r = orc.return_map(e, ep.copy(), sh, bu, eta, c)
    assert min((~r['ind_p']).sum(), r['n_smooth'], r['n_apex']) > 300
    fd = _tangent_by_differences(lambda x: orc.return_map(x, ep.copy(), sh, bu, eta, c), e)
    assert np.abs(fd - r['ds']).max() <= 1e-7 * np.abs(r['ds']).max()
    D = r['ds'].reshape(3, 3, n)
    assert np.array_equal(D, D.transpose(1, 0, 2))                   # symmetric (associated flow): K_tangent is too
