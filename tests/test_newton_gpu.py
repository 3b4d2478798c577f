"""
End-to-end callers of the hot path (BASELINE configs 2 and 3 at the sizes the reference itself can run):
the re-stated load-stepping / Newton drivers against traces recorded from the reference drivers.

Tolerance: displacements 1e-10 relative to the field's maximum (north_star), load history exact,
footing pressure 1e-9 relative.  The linear solves differ (SciPy SuperLU vs the reference's dense LAPACK),
Newton converges both to criterion < 1e-12.
"""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def test_dp_p1_level1_driver_vs_reference_trace(fep):
    g = load_golden('dp_p1_level1_trace')
    h = fep.solve_strip_footing('P1', level=1)
    assert len(h['zeta']) == 16 == len(g['zeta'])
    assert np.allclose(h['zeta'], g['zeta'], rtol=0, atol=1e-15)                 # same adaptive step sequence
    assert h['n_calls'] == int(g['n_calls']) == 125                             # same number of return-map calls
    # the reference logs the pressure of step k at the start of step k+1 (DP:1034)
    assert np.abs(np.array(h['pressure'][:15]) - g['pressure'][1:16]).max() <= 1e-9 * np.abs(g['pressure']).max()
    assert relerr(h['pressure'][14], 16.83867398886026) <= 1e-9                 # SURVEY 8c pin (zeta = 0.52)
    for k in range(16):
        assert relerr(h['U'][k], g['U_accepted'][k]) <= 1e-10, k
    assert h['counts'][-1] == tuple(g['counts'][-1]) == (599, 171)
    assert relerr(h['Ep'], g['Ep_final']) <= 1e-9


def _accepted(g):
    """Indices of the accepted attempts of a recorded DP run: the reference logs the load factor of every ATTEMPTED
    step (DP:1032) and, with it, the footing pressure of the last accepted one (DP:1034); after a rejected attempt the
    next load factor is smaller (DP:1117)."""
    z = g['zeta']
    n_acc = g['U_accepted'].shape[0]
    acc = [i for i in range(len(z) - 1) if z[i + 1] > z[i]]
    if len(acc) < n_acc:
        acc.append(len(z) - 1)
    assert len(acc) == n_acc
    return acc


@pytest.mark.parametrize('t,n_steps,n_acc,n_calls,last_counts', [('Q1', 25, 24, 199, (700, 607)), ('Q2', 17, 16, 186, (986, 319)),
                                                                  ('P2', 13, 13, 130, (1718, 725))])
def test_dp_driver_other_element_types_vs_reference_trace(fep, t, n_steps, n_acc, n_calls, last_counts):
    """`elasticity_fem(element_type, level=1)` of the reference (DP:901-1131) run unmodified for the element types it
    accepts besides P1 (tests/golden/make_golden.py gen_dp_trace_types): adaptive load history incl. the rejected
    attempts (Q1, Q2), number of return-map calls, every accepted displacement to 1e-10, pressures to 1e-9."""
    g = load_golden(f'dp_{t.lower()}_level1_trace')
    assert len(g['zeta']) == n_steps and g['U_accepted'].shape[0] == n_acc and int(g['n_calls']) == n_calls
    acc = _accepted(g)
    h = fep.solve_strip_footing(t, level=1)
    assert len(h['zeta']) == n_acc
    assert np.allclose(h['zeta'], g['zeta'][acc], rtol=0, atol=1e-15)           # same accepted load factors
    # Newton stops at criterion < 1e-12 (DP:1086): an iterate that lands within rounding of that threshold takes one
    # iteration more or less with another linear solver (SuperLU here, dense LAPACK there); same rejections otherwise
    assert abs(h['n_calls'] - n_calls) <= 2, h['n_calls']
    for k in range(n_acc):
        assert relerr(h['U'][k], g['U_accepted'][k]) <= 1e-10, k
    pmax = np.abs(g['pressure']).max()
    for k, i in enumerate(acc):
        if i + 1 < len(g['pressure']):
            assert abs(h['pressure'][k] - g['pressure'][i + 1]) <= 1e-9 * pmax, k
    assert h['counts'][-1] == tuple(g['counts'][-1]) == last_counts
    assert relerr(h['Ep'], g['Ep_final']) <= 1e-9


@pytest.mark.parametrize('t', ['P2', 'P4'])
def test_tsx_driver_on_the_demo_element_types_vs_reference_replay(fep, tsx_csv_dir, t):
    """The TSX load-step sequence on the element types the reference's demo runs (tsx-tunnel/sandbox.py:3-4: P4; the
    driver only works for P2 / P4, TSX:1629-1633), against the replay recorded with the reference's own functions and
    dense solves (make_golden.py gen_tsx_p2p4): 17 steps, plastic-point counts, call count, displacements 1e-10."""
    g = load_golden(f'tsx_{t.lower()}_trace')
    # the mesh comes from coord.csv / elem.csv + midpoints, as the reference's driver reads it (TSX:1687-1690)
    h = fep.solve_tsx_tunnel(element_type=t, mesh_dir=tsx_csv_dir)
    assert len(h['zeta']) == 17 == len(g['zeta']) and np.allclose(h['zeta'], g['zeta'], rtol=0, atol=1e-15)
    assert h['n_plast'] == g['nplast'].tolist() and h['n_plast'][-1] > 0
    assert h['n_calls'] == int(g['n_calls'])
    assert relerr(h['F0'], g['F0']) <= 1e-12
    for k, step in enumerate(g['steps']):
        assert relerr(h['U'][int(step)], g['U_steps'][k]) <= 1e-10, step
    assert np.abs(np.array(h['displ']) - g['U_mon']).max() <= 1e-10 * np.abs(g['U_mon']).max()


def test_tsx_p1_driver_vs_reference_replay(fep):
    g = load_golden('tsx')
    h = fep.solve_tsx_tunnel(g['coord'], g['elem'], 'P1')
    assert len(h['zeta']) == 17 and np.allclose(h['zeta'], g['p1_zeta'], rtol=0, atol=1e-15)
    assert h['n_plast'] == g['p1_nplast'].tolist() == [0] * 13 + [1, 1, 2, 3]
    assert relerr(h['F0'], g['p1_F0']) <= 1e-12
    assert relerr(h['U'][12], g['p1_U_step13']) <= 1e-10
    assert relerr(h['U'][-1], g['p1_U_final']) <= 1e-10
    assert abs(h['displ'][-1] - (-0.0019794496707526746)) <= 1e-10 * 0.0019794496707526746     # SURVEY 8c pin


@pytest.mark.parametrize('route', ['node', 'patch', 'coo'])
def test_tsx_p1_mesh_one_step_on_every_table_variant(fep, monkeypatch, route):
    """The call that aborted in round 2's first GPU run (gpurun_out/r2a: tsx-tunnel P1 mesh, 476 nodes, unstructured, tiles
    kept as strips, through fep_step_host; profiles/r03_ablation.md): one host-array step on that mesh on every route of
    the P1 path the product ships — the node route (its run / list tables are chosen by the mesh) and the element route in
    both forms — with the plan replayed against the mesh first (FEP_VALIDATE_PLAN), checked against the oracle.  Once per
    route, no repetition.  (The table variants behind FEP_P1_PATH exist in the -DFEP_ABLATION build only.)"""
    from oracle import fep_oracle as orc
    if route != 'node':
        monkeypatch.setenv('FEP_ROUTE', route)
    monkeypatch.setenv('FEP_VALIDATE_PLAN', '1')
    g = load_golden('tsx')
    elem, coord = g['elem'], g['coord']
    n = elem.shape[1]
    rng = np.random.default_rng(3)
    from conftest import dp_materials
    sh, bu, eta, c = dp_materials(n)
    x, y = coord                                                   # the tunnel mesh spans [-50, 50]^2
    U = 1e-5 * np.array([y * (x / 10) + 0.5 * x * (y > 0), -0.6 * y * (x < 0) + 0.8 * y * (x >= 0)])
    U += rng.normal(0, 5e-7, size=U.shape)                         # 148 smooth / 128 apex / 611 elastic points
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(sh, bu, eta, c)
    r = ctx.step(U, np.zeros((4, n)), want=('s', 'ds', 'ind_p', 'K', 'F'))
    r_kf = ctx.step(U, np.zeros((4, n)), want=('K', 'F'))
    d1, d2, wf = fep.element_tables('P1')
    K, B, w, iD, jD, D = orc.elastic_setup(elem, coord, sh, bu, d1, d2, wf)
    E, cp, K_t, F = orc.hot_path(U, np.zeros((4, n)), dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=sh, bulk=bu,
                                                          eta=eta, c=c))
    assert min(cp['n_smooth'], cp['n_apex'], n - cp['n_smooth'] - cp['n_apex']) > 50 and np.array_equal(r['ind_p'], cp['ind_p'])
    assert relerr(r['s'], cp['s']) <= 1e-13 and relerr(r['ds'], cp['ds']) <= 1e-13
    assert np.abs((r['K'] - K_t).data).max() <= 1e-12 * np.abs(K_t.data).max() and relerr(r['F'], F) <= 1e-12
    assert np.array_equal(r_kf['K'].data, r['K'].data) and np.array_equal(r_kf['F'], r['F'])
    ctx.close()


def test_tsx_p2_first_steps(fep):
    """P2 mesh (midpoints from the reference generator, fixture): elastic steps are linear in zeta."""
    g = load_golden('tsx')
    h = fep.solve_tsx_tunnel(g['p2_coord'], g['p2_elem'], 'P2', n_load_steps=17)
    assert len(h['zeta']) == 17
    assert np.abs(h['F0'].T[h['Q'].T] - g['f0q']).max() <= 2e-4 * np.abs(g['f0q']).max()       # f0q.csv
    el = [k for k, n in enumerate(h['n_plast']) if n == 0]
    assert len(el) >= 5
    for k in el[1:]:
        assert relerr(h['U'][k] / h['zeta'][k], h['U'][el[0]] / h['zeta'][el[0]]) <= 1e-9


def test_tsx_p4_full_run_consistent_with_p2_and_p1(fep):
    """Config 3 on the P4 mesh (15-node triangles, 14 288 free DOFs): no reference trace exists for it (the reference's
    own TSX driver does not run under NumPy >= 1.24); the 17 load steps must converge and the monitored displacement
    must agree with the P2 and the pinned P1 result to discretisation accuracy, elastic steps must be linear in zeta."""
    g = load_golden('tsx')
    h4 = fep.solve_tsx_tunnel(g['p4_coord'], g['p4_elem'], 'P4')
    h2 = fep.solve_tsx_tunnel(g['p2_coord'], g['p2_elem'], 'P2')
    assert len(h4['zeta']) == 17 == len(h2['zeta']) and abs(h4['zeta'][-1] - 1.0) < 1e-12
    d1 = -0.0019794496707526746                                     # P1, SURVEY 8c
    e21, e42 = abs(h2['displ'][-1] - d1), abs(h4['displ'][-1] - h2['displ'][-1])
    assert e21 <= 0.2 * abs(d1) and e42 <= 0.03 * abs(d1) and e42 < 0.25 * e21      # P1 -1.98e-3, P2 -2.24e-3, P4 -2.27e-3
    el = [k for k, n in enumerate(h4['n_plast']) if n == 0]
    assert len(el) >= 3
    for k in el[1:]:
        assert relerr(h4['U'][k] / h4['zeta'][k], h4['U'][el[0]] / h4['zeta'][el[0]]) <= 1e-9
    assert h4['n_plast'][-1] > 0


@pytest.mark.parametrize('t', ['P1', 'P2', 'Q1', 'Q2'])
def test_transform_vs_reference_golden(fep, t):
    """`transform` (DP:760-816) against outputs of the reference's own function (make_golden.py gen_transform): the
    device kernel behind fep_transform_host and the host restatement the drivers use."""
    g = load_golden('transform')
    elem, coord, w, q, want = (g[f'{t}_{k}'] for k in ('elements', 'coordinates', 'weight', 'q_int', 'q_node'))
    ctx = fep.MeshContext(elem, coord)
    got = ctx.transform(q)
    ctx.close()
    tol = 1e-14 * np.abs(q).max()                # every node against the size of the averaged values (they cancel)
    assert np.abs(got - want).max() <= tol
    assert np.abs(fep.transform(q, elem, w) - want).max() <= tol


def test_transform_matches_definition(fep):
    rng = np.random.default_rng(0)
    mesh = fep.square_mesh(3, 'P2', 3)
    elem = mesh['elements']
    w = rng.uniform(1, 2, elem.shape[1] * 7)
    q = rng.normal(size=w.size)
    got = fep.transform(q, elem, w)
    n_n = mesh['coordinates'].shape[1]
    f1 = np.zeros(n_n); f2 = np.zeros(n_n)
    for e in range(elem.shape[1]):
        for qq in range(7):
            k = e * 7 + qq
            for a in range(6):
                f1[elem[a, e]] += w[k] * q[k]
                f2[elem[a, e]] += w[k]
    assert np.allclose(got, f1 / f2, rtol=1e-14)


def test_config2_65k_p1_single_load_step(fep):
    """BASELINE configs[1]: ~65k P1 elements (N=181), first load step, Newton to 1e-12; the converged
    iterate's hot-path outputs are checked against the oracle on the same displacement."""
    from oracle import fep_oracle as orc
    from conftest import dp_materials
    h = fep.solve_strip_footing('P1', n_cells=181, max_steps=1)
    assert len(h['zeta']) == 1 and h['zeta'][0] == 1 / 1000 and h['newton_its'][0] <= 25
    mesh = h['mesh']
    elem, coord = mesh['elements'], mesh['coordinates']
    assert elem.shape[1] == 65522
    U = h['U'][0]
    d1, d2, wf = fep.element_tables('P1')
    sh, bu, eta, c = dp_materials(elem.shape[1])
    K, B, w, iD, jD, D = orc.elastic_setup(elem, coord, sh, bu, d1, d2, wf)
    E, cp, K_t, F = orc.hot_path(U, np.zeros((4, elem.shape[1])),
                                 dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=sh, bulk=bu, eta=eta, c=c))
    ctx = fep.MeshContext(elem, coord)
    ctx.set_materials(sh, bu, eta, c)
    r = ctx.step(U, np.zeros((4, elem.shape[1])), want=('s', 'ds', 'ind_p', 'K', 'F'))
    assert cp['n_smooth'] + cp['n_apex'] > 0 and (r['n_smooth'], r['n_apex']) == (cp['n_smooth'], cp['n_apex'])
    assert np.array_equal(r['ind_p'], cp['ind_p'])
    assert relerr(r['s'], cp['s']) <= 1e-13 and relerr(r['ds'], cp['ds']) <= 1e-13
    assert np.abs((r['K'] - K_t).data).max() <= 1e-12 * np.abs(K_t.data).max()
    assert relerr(r['F'], F) <= 1e-12
    # converged: the residual vanishes on the free DOFs
    qf = mesh['Q'].flatten(order='F')
    assert np.abs(r['F'][qf]).max() <= 1e-8 * np.abs(r['F']).max()


class _OracleContext:
    """MeshContext look-alike backed by the CPU oracle (test infrastructure): same driver loop, other hot path."""

    def __init__(self, elem, coord, d1, d2, wf):
        from oracle import fep_oracle as orc
        self.orc, self.elem, self.coord, self.tab = orc, elem, coord, (d1, d2, wf)
        self.n_int = elem.shape[1] * wf.size

    def set_materials(self, sh, bu, eta, c):
        one = np.ones(self.n_int)
        self.m = (sh * one, bu * one, eta * one, c * one)
        K, B, w, iD, jD, D = self.orc.elastic_setup(self.elem, self.coord, self.m[0], self.m[1], *self.tab)
        self.c = dict(K_elast=K, B=B, D_elast=D, weight=w, iD=iD, jD=jD, shear=self.m[0], bulk=self.m[1],
                      eta=self.m[2], c=self.m[3])

    def geometry(self):
        return None, None, self.c['weight'], None

    def step(self, U, ep_prev=None, apply_plastic_strain=False, want=()):
        U2 = np.asarray(U).reshape((2, -1), order='F') if np.ndim(U) == 1 else U
        E, cp, K_t, F = self.orc.hot_path(U2, ep_prev, self.c, apply_plastic_strain=apply_plastic_strain)
        return {'K': K_t.tocsr(), 'F': F, 's': cp['s'], 'ds': cp['ds'], 'ind_p': cp['ind_p'],
                'n_smooth': cp['n_smooth'], 'n_apex': cp['n_apex']}

    def close(self):
        pass


def test_config4_in_miniature_ten_load_steps_gpu_vs_oracle_driver(fep):
    """BASELINE configs[3] scaled down (N=48: 4 608 P1 elements, 10 accepted load steps): the SAME driver loop is
    run with the GPU hot path and with the CPU oracle as hot path; every accepted displacement, the load history,
    the plastic-point counts and the final plastic strain must agree."""
    a = fep.solve_strip_footing('P1', n_cells=48, max_steps=10)
    b = fep.solve_strip_footing('P1', n_cells=48, max_steps=10, context_factory=_OracleContext)
    assert len(a['zeta']) == len(b['zeta']) == 10 and np.allclose(a['zeta'], b['zeta'], rtol=0, atol=1e-15)
    assert a['n_calls'] == b['n_calls'] and a['newton_its'] == b['newton_its']
    assert a['counts'] == b['counts'] and a['counts'][-1][0] > 500          # well into the plastic regime
    for k in range(10):
        assert relerr(a['U'][k], b['U'][k]) <= 1e-10, k
    assert np.abs(np.array(a['pressure']) - np.array(b['pressure'])).max() <= 1e-9 * max(b['pressure'])
    assert relerr(a['Ep'], b['Ep']) <= 1e-9
