"""
Callers of the hot path: the reference's load-stepping / semismooth-Newton drivers
re-stated around the GPU step, with a SciPy sparse solve in place of the reference's dense
`np.linalg.solve` on a (2 n_n)^2 boolean-masked matrix.

  solve_strip_footing   Plasticity2D_DP/pythonFEM.py:986-1131  (adaptive load steps, footing pressure)
  solve_tsx_tunnel      tsx-tunnel/pythonFEM.py:1729-1832      (17 uniform steps of the initial-stress factor;
                        the accepting call leaves `apply_plastic_strain` False)

Per Newton iterate ONE call `MeshContext.step` replaces DP:1043-1058 (strain, return map, tangent, residual).
Step control, stopping norms and the extrapolation of the next iterate follow the reference line by line.
`transform` (DP:760-816, nodal averaging used for the footing pressure that steers the step size) is host
post-processing and is re-stated with `np.bincount`.
"""
import numpy as np
import scipy.sparse.linalg as sspl

from .hotpath import MeshContext
from .mesh import square_mesh
from .tables import ELEMENT_SHAPE, _coerce, element_tables


def transform(q_int, elements, weight):
    """Integration-point values -> nodal values, weighted average over the adjacent points (DP:760-816)."""
    n_p, n_e = elements.shape
    w = np.asarray(weight, dtype=float).ravel()
    n_q = w.size // n_e
    nodes = np.repeat(np.asarray(elements), n_q, axis=1)                 # (n_p, n_int)
    n_n = int(nodes.max()) + 1
    wq = w * np.asarray(q_int, dtype=float).ravel()
    f1 = np.bincount(nodes.ravel(), weights=np.tile(wq, n_p), minlength=n_n)
    f2 = np.bincount(nodes.ravel(), weights=np.tile(w, n_p), minlength=n_n)
    return f1 / f2


def _solve_free(K, rhs, qf):
    Kqq = K[qf][:, qf].tocsc()
    return sspl.spsolve(Kqq, rhs[qf])


def _energy(K, v):
    return np.sqrt(v @ (K @ v))


def solve_strip_footing(element_type='P1', level=1, n_cells=None, size_xy=10, max_steps=None, zeta_max=1.0,
                        device=None, log=None, context_factory=None):
    """Strip-footing benchmark of Plasticity2D_DP (DP:901-1131).  `level` as in the reference
    (N = size_xy * 2**level cells per side) or `n_cells` directly.  Returns a dict with the load history
    ('zeta', 'pressure'), the accepted displacements 'U' (list of (2,n_n)), final 'Ep', counters.
    `context_factory(elements, coordinates, dhatp1, dhatp2, wf)` may supply another object with MeshContext's
    `set_materials / step / geometry / close` (the tests drive the same loop with the CPU oracle that way)."""
    t = _coerce(element_type)
    young, poisson, c0, phi = 1e7, 0.48, 450, np.pi / 9                                   # DP:910-933
    shear0 = young / (2 * (1 + poisson))
    bulk0 = young / (3 * (1 - 2 * poisson))
    eta0 = 3 * np.tan(phi) / np.sqrt(9 + 12 * (np.tan(phi)) ** 2)
    c_0 = 3 * c0 / np.sqrt(9 + 12 * (np.tan(phi)) ** 2)
    mesh = square_mesh(size_xy * 2 ** level if n_cells is None else n_cells, t, size_xy)  # DP:945
    elem, coord, Q = mesh['elements'], mesh['coordinates'], mesh['Q']
    q_nd = mesh['dirichlet_nodes'][1, :] > 0
    n_n = coord.shape[1]
    d1, d2, wf = element_tables(t)
    ctx = (context_factory or (lambda *a: MeshContext(*a, device=device)))(elem, coord, d1, d2, wf)
    n_int = ctx.n_int
    ctx.set_materials(shear0, bulk0, eta0, c_0)
    K_elast = ctx.step(np.zeros(2 * n_n), want=('K',))['K']                               # DP:977
    _, _, weight, _ = ctx.geometry()
    qf = Q.flatten(order='F')

    d_zeta = 1 / 1000                                                                     # DP:989-994
    d_zeta_min = d_zeta / 1300
    d_zeta_old = d_zeta
    zeta_old = 0.0
    Ud = -d_zeta * mesh['dirichlet_nodes']                                                # DP:997-1004
    f = -(K_elast @ Ud.flatten(order='F'))
    U_it = Ud.flatten(order='F')
    U_it[qf] = _solve_free(K_elast, f, qf)
    dU = np.zeros(2 * n_n)
    U = np.zeros(2 * n_n)
    U_old = -U_it
    Ep_old = np.zeros((4, n_int))
    pressure_old = 0.0
    hist = {'zeta': [], 'pressure': [], 'U': [], 'counts': [], 'n_calls': 0, 'newton_its': []}
    criterion = None
    while True:
        zeta = zeta_old + d_zeta                                                          # DP:1031
        its = 0
        for _ in range(25):                                                               # DP:1040
            r = ctx.step(U_it, Ep_old, apply_plastic_strain=False, want=('K', 'F'))       # DP:1043-1058
            hist['n_calls'] += 1
            its += 1
            dU[qf] = _solve_free(r['K'], -r['F'], qf)                                     # DP:1062-1066
            U_new = U_it + dU
            q1, q2, q3 = _energy(K_elast, dU), _energy(K_elast, U_it), _energy(K_elast, U_new)   # DP:1072-1074
            criterion = q1 / (q2 + q3)
            if np.isnan(criterion):                                                       # DP:1076
                break
            U_it = U_new
            if criterion < 1e-12:                                                         # DP:1086
                break
        if criterion < 1e-10:                                                             # DP:1091
            U_old = U
            U = U_it
            r = ctx.step(U, Ep_old, apply_plastic_strain=True, want=('s',))               # DP:1095-1098
            hist['n_calls'] += 1
            zeta_old = zeta
            d_zeta_old = d_zeta
            pressure_arr = transform(r['s'][1, :], elem, weight)                          # DP:1105
            pressure = -np.mean(pressure_arr[q_nd]) / c0
            hist['zeta'].append(zeta)
            hist['pressure'].append(pressure)
            hist['U'].append(U.reshape((2, -1), order='F').copy())
            hist['counts'].append((r['n_smooth'], r['n_apex']))
            hist['newton_its'].append(its)
            if log:
                log(f'zeta={zeta:.6g} pressure={pressure:.10g} its={its} smooth/apex={r["n_smooth"]}/{r["n_apex"]}')
            if pressure - pressure_old < 0.1 and criterion < 1e-12:                       # DP:1109
                d_zeta *= 2
            pressure_old = pressure
        else:
            d_zeta /= 2                                                                   # DP:1117
        U_it = d_zeta * (U - U_old) / d_zeta_old + U                                      # DP:1120
        if zeta_old >= zeta_max:                                                          # DP:1123
            break
        if d_zeta < d_zeta_min:                                                           # DP:1127
            break
        if max_steps is not None and len(hist['zeta']) >= max_steps:
            break
    hist['Ep'] = Ep_old
    hist['mesh'] = mesh
    ctx.close()
    return hist


def solve_tsx_tunnel(coords, elem, element_type='P1', n_load_steps=17, monitor=(0, 40), device=None, log=None):
    """TSX tunnel excavation (TSX:1637-1832) on a given mesh (`coords` (2,n_n), `elem` (n_p,n_e) 0-based; the
    reference reads coord.csv / elem.csv and, for P2/P4, adds midpoints first).  Returns the history of the
    monitored displacement, plastic-point counts and accepted displacements."""
    t = _coerce(element_type)
    young, nu = 60000, 0.2                                                                # TSX:1663-1672
    shear0 = young / (2 * (1 + nu))
    bulk0 = young / (3 * (1 - 2 * nu))
    fr = 49 * np.pi / 180
    eta0 = 3 * np.tan(fr) / np.sqrt(9 + 12 * (np.tan(fr)) ** 2)
    c_0 = 3 * 18.7 / np.sqrt(9 + 12 * (np.tan(fr)) ** 2)
    s0 = np.array([-45.0, -11.0, 0.0, -60.0]).reshape((-1, 1))                             # TSX:1675-1681
    tr0 = s0[0] + s0[1] + s0[3]
    init_strain = np.array([-nu * tr0 + (1 + nu) * s0[0], -nu * tr0 + (1 + nu) * s0[1], [0.0],
                            -nu * tr0 + (1 + nu) * s0[3]], dtype=float).reshape((-1, 1)) / young
    Q = np.ones(coords.shape, dtype=bool)                                                 # TSX:1695-1699
    Q[0, coords[0, :] < -49.99] = 0
    Q[0, coords[0, :] > 49.99] = 0
    Q[1, coords[1, :] < -49.99] = 0
    Q[1, coords[1, :] > 49.99] = 0
    qf = Q.flatten(order='F')
    n_n = coords.shape[1]
    d1, d2, wf = element_tables(t)
    ctx = MeshContext(elem, coords, d1, d2, wf, device=device)
    n_int = ctx.n_int
    assert n_int == elem.shape[1] * ELEMENT_SHAPE[t][1]
    ctx.set_materials(shear0, bulk0, eta0, c_0)
    K = ctx.step(np.zeros(2 * n_n), want=('K',))['K']                                     # TSX:1722
    _, F0 = ctx.assemble(None, s0 * np.ones((1, n_int)))                                   # TSX:1737

    d_zeta = 1 / n_load_steps                                                             # TSX:1730-1735
    d_zeta_min = d_zeta / 10
    d_zeta_old = d_zeta
    zeta_old = 0.0
    U_elast = np.zeros(2 * n_n)
    U_elast[qf] = _solve_free(K, -F0, qf)                                                 # TSX:1748
    U_it = d_zeta * U_elast
    dU = np.zeros(2 * n_n)
    U = np.zeros(2 * n_n)
    U_old = -U_it
    Ep_old = np.zeros((4, n_int))
    hist = {'zeta': [], 'displ': [], 'n_plast': [], 'U': [], 'n_calls': 0}
    criterion = None
    while True:
        zeta = zeta_old + d_zeta
        e0 = zeta * init_strain                                                           # TSX:1765
        for _ in range(25):
            r = ctx.step(U_it, Ep_old, e0=e0, want=('K', 'F'))                            # TSX:1771-1778
            hist['n_calls'] += 1
            dU[qf] = _solve_free(r['K'], -r['F'], qf)                                     # TSX:1781
            U_new = U_it + dU
            criterion = _energy(K, dU) / (_energy(K, U_it) + _energy(K, U_new))           # TSX:1788-1792
            if np.isnan(criterion):
                break
            U_it = U_new
            if criterion < 1e-12:
                break
        if criterion < 1e-10:                                                             # TSX:1804
            U_old = U
            U = U_it
            r = ctx.step(U, Ep_old, e0=e0, want=('ind_p',))          # accept WITHOUT apply_plastic_strain (C7)
            hist['n_calls'] += 1
            Ep_old = np.zeros((4, n_int))                            # 'ep' of a non-accepting call, TSX:1809
            zeta_old = zeta
            d_zeta_old = d_zeta
            Um = U.reshape((2, -1), order='F')
            hist['zeta'].append(zeta)
            hist['displ'].append(Um[monitor])
            hist['n_plast'].append(int(r['ind_p'].sum()))
            hist['U'].append(Um.copy())
            if log:
                log(f'zeta={zeta:.6g} U{monitor}={Um[monitor]:.16g} n_plast={hist["n_plast"][-1]}')
        else:
            d_zeta = d_zeta / 2                                                           # TSX:1818
        U_it = d_zeta * (U - U_old) / d_zeta_old + U                                      # TSX:1821
        if zeta_old >= 1:                                                                 # TSX:1824
            break
        if d_zeta < d_zeta_min:
            break
    hist['F0'] = F0.reshape((2, -1), order='F')
    hist['Q'] = Q
    ctx.close()
    return hist
