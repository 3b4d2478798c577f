#!/usr/bin/env python3
"""
Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE
(read-only checkout at /root/reference) in the build container and recording
inputs / outputs of its own functions.  The reference's source never enters
this repository: only arrays (inputs, expected outputs) are written.

Run (build container only; the GPU box has no /root/reference):
    python tests/golden/make_golden.py

`numba` is not installed; the reference imports `njit` but never applies it
(DP:24), so a 2-line stub module is put on sys.path.  matplotlib runs with the
Agg backend.  Everything is deterministic (fixed seeds).
"""
import hashlib
import importlib.util
import logging
import os
import sys
import tempfile
import time

import numpy as np

REF = os.environ.get('FEP_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))


def _load_reference():
    os.environ['MPLBACKEND'] = 'Agg'
    stub = tempfile.mkdtemp(prefix='fep_stub_')
    with open(os.path.join(stub, 'numba.py'), 'w') as fh:
        fh.write('def njit(*a, **k):\n'
                 '    return a[0] if (len(a) == 1 and callable(a[0]) and not k) else (lambda f: f)\n')
    sys.path.insert(0, stub)
    mods = {}
    for name, sub in (('dp', 'Plasticity2D_DP'), ('tsx', 'tsx-tunnel'), ('el', 'Elasticity2D')):
        spec = importlib.util.spec_from_file_location('ref_' + name, os.path.join(REF, sub, 'pythonFEM.py'))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        mods[name] = m
    logging.disable(logging.CRITICAL)
    return mods


def sha(a):
    a = np.ascontiguousarray(a)
    return np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print(f'{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB  ({len(arrs)} arrays)')


def csr_parts(prefix, M):
    M = M.tocsr()
    return {prefix + '_data': M.data, prefix + '_indices': M.indices.astype(np.int64),
            prefix + '_indptr': M.indptr.astype(np.int64), prefix + '_shape': np.array(M.shape)}


def dp_materials(n_int):
    young, poisson, c0, phi = 1e7, 0.48, 450, np.pi / 9          # DP:910-933
    shear = young / (2 * (1 + poisson)) * np.ones(n_int)
    bulk = young / (3 * (1 - 2 * poisson)) * np.ones(n_int)
    eta = 3 * np.tan(phi) / np.sqrt(9 + 12 * np.tan(phi) ** 2) * np.ones(n_int)
    c = 3 * c0 / np.sqrt(9 + 12 * np.tan(phi) ** 2) * np.ones(n_int)
    return shear, bulk, eta, c


def strain_recipe(rng, n, f_s, f_a):
    """SURVEY 8(d) config-4 prescribed-strain recipe."""
    u = rng.random(n)
    E = rng.normal(0, 1e-7, size=(3, n))
    E[0:2] -= 2e-6
    sm = u < f_s
    ap = np.logical_and(u >= f_s, u < f_s + f_a)
    E[2, sm] += rng.uniform(4e-4, 8e-4, size=sm.sum())
    E[0, sm] -= 1e-5
    E[0:2, ap] += 4e-4
    return E


# --------------------------------------------------------------------------
def gen_tables(R):
    arrs = {}
    for mod, types in (('dp', ('P1', 'P2', 'Q1', 'Q2')), ('tsx', ('P2', 'P4'))):
        m = R[mod]
        for t in types:
            et = m.LagrangeElementType[t]
            xi, wf = m.get_quadrature_volume(et)
            hatp, d1, d2 = m.get_local_basis_volume(et, xi)
            key = f'{mod}_{t}_'
            arrs[key + 'xi'] = xi
            arrs[key + 'wf'] = wf
            arrs[key + 'hatp'] = np.asarray(hatp, dtype=float)
            arrs[key + 'dhatp1'] = np.asarray(d1, dtype=float)
            arrs[key + 'dhatp2'] = np.asarray(d2, dtype=float)
    save('tables', **arrs)


def gen_mesh_dp(R):
    m = R['dp']
    arrs = {}
    for t in ('P1', 'P2', 'Q1', 'Q2'):
        et = m.LagrangeElementType[t]
        mesh = m.assemble_mesh(0, et, 4)                       # N_x = 4 (DP:67)
        for k in ('coordinates', 'elements', 'dirichlet_nodes', 'Q'):
            arrs[f'{t}_n4_{k}'] = np.asarray(mesh[k])
        mesh = m.assemble_mesh(1, et, 10)                      # the demo's own level 1
        for k in ('coordinates', 'elements', 'dirichlet_nodes', 'Q'):
            a = np.asarray(mesh[k])
            if k == 'elements':
                a = a.astype(np.int64)
            arrs[f'{t}_l1_{k}_sha'] = sha(a)
            arrs[f'{t}_l1_{k}_shape'] = np.array(a.shape)
    save('mesh_dp', **arrs)


def _jiggled_mesh(m, t, N, rng, size=10.0):
    et = m.LagrangeElementType[t]
    mesh = m.assemble_mesh(0, et, N)
    coord = mesh['coordinates'] * (size / N)
    # move interior nodes a little so that Jacobians are not all equal
    interior = np.logical_and.reduce([coord[0] > 0, coord[0] < size, coord[1] > 0, coord[1] < size])
    coord = coord.copy()
    coord[:, interior] += rng.uniform(-0.08, 0.08, size=(2, interior.sum())) * (size / N)
    return et, np.asarray(mesh['elements']).astype(np.int64), coord


def gen_setup(R):
    m = R['dp']
    rng = np.random.default_rng(11)
    arrs = {}
    for t in ('P1', 'P2', 'Q1', 'Q2'):
        et, elem, coord = _jiggled_mesh(m, t, 4, rng)
        xi, wf = m.get_quadrature_volume(et)
        _, d1, d2 = m.get_local_basis_volume(et, xi)
        n_int = elem.shape[1] * wf.size
        shear, bulk, _, _ = dp_materials(n_int)
        shear = shear * rng.uniform(0.8, 1.2, n_int)           # heterogeneous moduli
        bulk = bulk * rng.uniform(0.8, 1.2, n_int)
        K, B, w, iD, jD, D = m.get_elastic_stiffness_matrix(elem.copy(), coord, shear, bulk, d1, d2, wf)
        arrs.update({f'{t}_elements': elem, f'{t}_coordinates': coord, f'{t}_shear': shear, f'{t}_bulk': bulk,
                     f'{t}_K': K.toarray(), f'{t}_weight': w, f'{t}_iD': iD, f'{t}_jD': jD})
        arrs.update(csr_parts(f'{t}_B', B))
        arrs.update(csr_parts(f'{t}_D', D))
    save('setup_dp', **arrs)


def gen_retmap(R):
    rng = np.random.default_rng(7)
    arrs = {}
    n = 1500
    shear, bulk, eta, c = dp_materials(n)
    shear = shear * rng.uniform(0.7, 1.3, n)
    bulk = bulk * rng.uniform(0.7, 1.3, n)
    eta = eta * rng.uniform(0.7, 1.3, n)
    c = c * rng.uniform(0.7, 1.3, n)
    E = strain_recipe(rng, n, 0.4, 0.2)
    Ep = rng.normal(0, 2e-5, size=(4, n))
    arrs.update(dict(shear=shear, bulk=bulk, eta=eta, c=c, E=E, Ep=Ep))
    dp, tsx = R['dp'], R['tsx']
    e0 = np.array([[-3e-5], [2e-5], [0.0], [-4e-5]])
    arrs['e0'] = e0
    cases = {
        'dp_none': lambda: dp.construct_constitutive_problem(E.copy(), None, shear, bulk, eta, c),
        'dp_ep': lambda ep: dp.construct_constitutive_problem(E.copy(), ep, shear, bulk, eta, c),
        'dp_ep_accept': lambda ep: dp.construct_constitutive_problem(E.copy(), ep, shear, bulk, eta, c, True),
        'tsx_ep': lambda ep: tsx.construct_constitutive_problem(E.copy(), e0, ep, shear, bulk, eta, c),
        'tsx_ep_accept': lambda ep: tsx.construct_constitutive_problem(E.copy(), e0, ep, shear, bulk, eta, c, True),
    }
    for name, fn in cases.items():
        if name == 'dp_none':
            r = fn()
            ep_after = None
        else:
            ep_in = Ep.copy()
            r = fn(ep_in)
            ep_after = ep_in
        arrs[name + '_s'] = r['s']
        arrs[name + '_ds'] = r['ds']
        arrs[name + '_ind_p'] = np.asarray(r['ind_p'])
        arrs[name + '_ep'] = r['ep']
        arrs[name + '_lambda_is_none'] = np.array(r['lambda_final'] is None)
        if ep_after is not None:
            arrs[name + '_ep_prev_after'] = ep_after
            arrs[name + '_ep_is_alias'] = np.array(r['ep'] is ep_in)
    # all-elastic calls (TSX early-out TSX:1103; DP still returns None for lambda)
    Eel = rng.normal(0, 1e-8, size=(3, 64))
    sh, bu, et_, c_ = dp_materials(64)
    arrs['Eel'] = Eel
    r = dp.construct_constitutive_problem(Eel.copy(), np.zeros((4, 64)), sh, bu, et_, c_, True)
    arrs['dp_elastic_s'] = r['s']; arrs['dp_elastic_ds'] = r['ds']; arrs['dp_elastic_ep'] = r['ep']
    arrs['dp_elastic_lambda_is_none'] = np.array(r['lambda_final'] is None)
    ep_in = rng.normal(0, 1e-9, size=(4, 64))
    arrs['tsx_elastic_ep_in'] = ep_in.copy()
    r = tsx.construct_constitutive_problem(Eel.copy(), 0 * e0, ep_in, sh, bu, et_, c_, True)
    arrs['tsx_elastic_s'] = r['s']; arrs['tsx_elastic_ds'] = r['ds']; arrs['tsx_elastic_ep'] = r['ep']
    arrs['tsx_elastic_lambda'] = np.zeros((1, 64)) if r['lambda_final'] is None else r['lambda_final']
    arrs['tsx_elastic_lambda_is_none'] = np.array(r['lambda_final'] is None)
    save('retmap', **arrs)


def gen_hotpath(R):
    """One Newton-iteration's worth of the hot path (DP:1043-1058) on small
    jiggled meshes of every element type, with a displacement field that puts
    points on all three branches."""
    m = R['dp']
    rng = np.random.default_rng(23)
    arrs = {}
    for t, N in (('P1', 8), ('P2', 5), ('Q1', 6), ('Q2', 4)):
        et, elem, coord = _jiggled_mesh(m, t, N, rng)
        xi, wf = m.get_quadrature_volume(et)
        _, d1, d2 = m.get_local_basis_volume(et, xi)
        n_n = coord.shape[1]
        n_int = elem.shape[1] * wf.size
        shear, bulk, eta, c = dp_materials(n_int)
        K, B, w, iD, jD, D = m.get_elastic_stiffness_matrix(elem.copy(), coord, shear, bulk, d1, d2, wf)
        # smooth shear/compression field + noise: mixes elastic / smooth / apex
        x, y = coord
        U = np.array([2.5e-4 * y * (x / 10) + 1.2e-4 * x * (y > 5), -1.5e-4 * y * (x < 5) + 2.0e-4 * y * (x >= 5)])
        U += rng.normal(0, 2e-5, size=U.shape)
        Ep_old = rng.normal(0, 1e-5, size=(4, n_int))
        for accept in (False, True):
            tag = f'{t}_acc{int(accept)}_'
            Ep_in = Ep_old.copy()
            E = (B @ U.reshape((-1, 1), order='F')).reshape((3, -1), order='F')     # DP:1043
            cp = m.construct_constitutive_problem(E, Ep_in, shear, bulk, eta, c, apply_plastic_strain=accept)
            vD = np.tile(w, (9, 1)) * cp['ds']                                      # DP:1047
            import scipy.sparse as ssp
            D_p = ssp.csr_matrix((m.flatten_row(vD)[0], (m.flatten_row(iD)[0] - 1, m.flatten_row(jD)[0] - 1)),
                                 shape=(3 * n_int, 3 * n_int))
            K_t = K + B.T * (D_p - D) * B                                           # DP:1050
            F = B.T * np.reshape(np.tile(w, (3, 1)) * cp['s'][0:3, :], (3 * n_int, 1), order='F')  # DP:1058
            arrs.update({tag + 'E': np.asarray(E), tag + 's': cp['s'], tag + 'ds': cp['ds'],
                         tag + 'ind_p': np.asarray(cp['ind_p']), tag + 'ep': cp['ep'],
                         tag + 'K_t': K_t.toarray(), tag + 'F': np.asarray(F).ravel()})
        arrs.update({f'{t}_elements': elem, f'{t}_coordinates': coord, f'{t}_U': U, f'{t}_Ep_old': Ep_old,
                     f'{t}_K_elast': K.toarray(), f'{t}_weight': w})
        cnt = np.array([int(np.sum(cp['ind_p'])), n_int])
        arrs[f'{t}_n_plastic'] = cnt
        print('  hotpath', t, 'plastic', cnt)
    save('hotpath_dp', **arrs)


def _dp_trace(R, t):
    """The reference's own demo driver `elasticity_fem(element_type, level=1)` (DP:901-1131), run unmodified
    with recording hooks on module attributes (Plasticity2D_DP/sandbox.py runs it for P1)."""
    m = R['dp']
    et = m.LagrangeElementType[t]
    n_n = m.assemble_mesh(1, et, 10)['coordinates'].shape[1]
    rec = {'U': [], 'counts': [], 'calls': 0, 'accept_s': []}
    orig_ccp = m.construct_constitutive_problem
    orig_fc = m.flatten_col

    def ccp(e, ep_prev, shear, bulk, eta, c, apply_plastic_strain=False):
        rec['calls'] += 1
        r = orig_ccp(e, ep_prev, shear, bulk, eta, c, apply_plastic_strain=apply_plastic_strain)
        if apply_plastic_strain:
            rec['accept_s'].append(r['s'].copy())
            rec['U'].append(rec['last_fc'])      # flatten_col(U) at DP:1095 precedes the accepting call
            rec['accept_ep'] = r['ep'].copy()
        return r

    class H(logging.Handler):
        def emit(self, record):
            msg = record.getMessage()
            if msg.startswith('pressure = '):
                rec.setdefault('pressure', []).append(float(msg.split('=')[1]))
            elif msg.startswith('load factor = '):
                rec.setdefault('zeta', []).append(float(msg.split('=')[1]))
            elif msg.startswith('plastic integration points'):
                import re
                g = re.search(r'smooth portion = (\d+), apex = (\d+)', msg)
                rec['counts'].append((int(g.group(1)), int(g.group(2))))

    def fc(v):
        if isinstance(v, np.ndarray) and v.ndim == 2 and v.shape[0] == 2 and v.shape[1] == n_n:
            rec['last_fc'] = np.array(v, copy=True)
        return orig_fc(v)

    logging.disable(logging.NOTSET)
    h = H()
    logging.getLogger().addHandler(h)
    logging.getLogger().setLevel(logging.INFO)
    for hh in list(logging.getLogger().handlers):
        if hh is not h:
            logging.getLogger().removeHandler(hh)
    m.construct_constitutive_problem = ccp
    m.flatten_col = fc
    t0 = time.time()
    import io, contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        m.elasticity_fem(et, 1, draw=False)
    m.construct_constitutive_problem = orig_ccp
    m.flatten_col = orig_fc
    logging.getLogger().removeHandler(h)
    logging.disable(logging.CRITICAL)
    # flatten_col(dU) etc. are also (2,n_n); accepted U are the ones seen right before an accepting call.
    print(f'  dp trace {t}: {time.time() - t0:.1f}s, calls={rec["calls"]}, zeta steps={len(rec["zeta"])}, '
          f'accepted={len(rec["U"])}, last counts={rec["counts"][-1]}')
    return rec


def gen_dp_trace(R):
    rec = _dp_trace(R, 'P1')
    save('dp_p1_level1_trace',
         zeta=np.array(rec['zeta']), pressure=np.array(rec['pressure']),
         counts=np.array(rec['counts']), n_calls=np.array(rec['calls']),
         accept_s=np.array(rec['accept_s'])[[0, 7, 15]], accept_steps=np.array([0, 7, 15]),
         U_accepted=np.array(rec['U']), Ep_final=rec['accept_ep'])


def gen_dp_trace_types(R, types=('Q1', 'Q2', 'P2')):
    """The same driver for the other element types `elasticity_fem` accepts (DP:901-903, 945), level 1
    (Q1 6 s, Q2 83 s, P2 138 s in this container).  Per type: the load history, the counters the
    reference logs, every accepted displacement, the stress of the first and the last accepted step and the
    final plastic strain."""
    for t in types:
        rec = _dp_trace(R, t)
        last = len(rec['accept_s']) - 1
        save(f'dp_{t.lower()}_level1_trace',
             zeta=np.array(rec['zeta']), pressure=np.array(rec['pressure']),
             counts=np.array(rec['counts']), n_calls=np.array(rec['calls']),
             accept_s=np.array(rec['accept_s'])[[0, last]], accept_steps=np.array([0, last]),
             U_accepted=np.array(rec['U']), Ep_final=rec['accept_ep'])


def gen_transform(R):
    """`transform` (DP:760-816): integration-point field -> nodal field, on small jiggled meshes."""
    m = R['dp']
    rng = np.random.default_rng(31)
    arrs = {}
    for t, N in (('P1', 7), ('P2', 5), ('Q1', 6), ('Q2', 4)):
        et, elem, coord = _jiggled_mesh(m, t, N, rng)
        xi, wf = m.get_quadrature_volume(et)
        _, d1, d2 = m.get_local_basis_volume(et, xi)
        n_int = elem.shape[1] * wf.size
        shear, bulk, _, _ = dp_materials(n_int)
        _, _, w, *_ = m.get_elastic_stiffness_matrix(elem.copy(), coord, shear, bulk, d1, d2, wf)
        q = rng.normal(0, 1, size=n_int) * 1e3 + 250.0
        qn = np.asarray(m.transform(q, elem, w)).ravel()
        arrs.update({f'{t}_elements': elem, f'{t}_coordinates': coord, f'{t}_weight': np.asarray(w),
                     f'{t}_q_int': q, f'{t}_q_node': qn})
    save('transform', **arrs)


def _tsx_consts():
    young, nu = 60000, 0.2                                                       # TSX:1663-1672
    G = young / (2 * (1 + nu)); Kb = young / (3 * (1 - 2 * nu))
    fr = 49 * np.pi / 180
    eta0 = 3 * np.tan(fr) / np.sqrt(9 + 12 * np.tan(fr) ** 2)
    c0 = 3 * 18.7 / np.sqrt(9 + 12 * np.tan(fr) ** 2)
    s0 = np.array([-45.0, -11.0, 0.0, -60.0]).reshape((-1, 1))                   # TSX:1675
    tr0 = s0[0] + s0[1] + s0[3]
    e_init = np.array([-nu * tr0 + (1 + nu) * s0[0], -nu * tr0 + (1 + nu) * s0[1], [0.0],
                       -nu * tr0 + (1 + nu) * s0[3]], dtype=float).reshape((-1, 1)) / young   # TSX:1677-1681
    return G, Kb, eta0, c0, s0, e_init


def _tsx_setup(m, et_name, co, el):
    G, Kb, *_ = _tsx_consts()
    et = m.LagrangeElementType[et_name]
    xi, wf = m.get_quadrature_volume(et)
    _, d1, d2 = m.get_local_basis_volume(et, xi)
    n_int = el.shape[1] * wf.size
    K, B, w, iD, jD, D = m.get_elastic_stiffness_matrix(el, co, G * np.ones(n_int), Kb * np.ones(n_int), d1, d2, wf)
    Q = np.ones(co.shape, dtype=bool)                                        # TSX:1695-1699
    Q[0, co[0] < -49.99] = 0; Q[0, co[0] > 49.99] = 0
    Q[1, co[1] < -49.99] = 0; Q[1, co[1] > 49.99] = 0
    return K, B, w.flatten(order='F'), iD, jD, D, Q, n_int


def _tsx_replay(m, K, B, w, iD, jD, D, Q, n_int, co, progress=None):
    """Replay of TSX:1729-1832 (the shipped driver breaks at TSX:1677 on NumPy>=1.24 and create_midpoints
    returns None for P1 - SURVEY C11) calling the reference's own functions, dense solves as the reference."""
    import scipy.sparse as ssp
    G, Kb, eta0, c0, s0, e_init = _tsx_consts()
    n_n = co.shape[1]
    sh = G * np.ones(n_int); bu = Kb * np.ones(n_int)
    eta = eta0 * np.ones(n_int); c = c0 * np.ones(n_int)
    d_zeta = 1 / 17; d_zeta_min = d_zeta / 10; d_zeta_old = d_zeta; zeta_old = 0
    F0 = (B.T @ np.reshape(np.tile(w, (3, 1)) * s0[0:3, :], (3 * n_int, 1), order='F')).reshape((2, -1), order='F')
    qf = Q.flatten(order='F')
    Kd = K.toarray()
    Kqq = Kd[np.ix_(qf, qf)]
    del Kd
    U_el = np.zeros((2, n_n))
    U_el.T[Q.T] = np.linalg.solve(Kqq, -F0.T[Q.T])
    del Kqq
    U_it = d_zeta * U_el
    dU = np.zeros((2, n_n)); U = np.zeros((2, n_n)); U_old = -U_it
    Ep_old = np.zeros((4, n_int))
    hist = []; nplast = []; Us = []; calls = 0
    while True:
        zeta = zeta_old + d_zeta
        E0 = zeta * e_init
        for it in range(25):
            E = (B @ U_it.reshape((-1, 1), order='F')).reshape((3, -1), order='F')
            cp = m.construct_constitutive_problem(E, E0, Ep_old, sh, bu, eta, c)
            calls += 1
            vD = np.tile(w, (9, 1)) * cp['ds']
            D_p = ssp.csr_matrix((m.flatten_row(vD)[0], (m.flatten_row(iD)[0] - 1, m.flatten_row(jD)[0] - 1)),
                                 shape=(3 * n_int, 3 * n_int))
            K_t = K + B.T * (D_p - D) * B
            F = (B.T @ (np.tile(w, (3, 1)) * cp['s'][0:3, :]).reshape((3 * n_int, 1), order='F')).reshape((2, n_n), order='F')
            Ktq = K_t.tocsr()[qf][:, qf].toarray()
            dU.T[Q.T] = np.linalg.solve(Ktq, -F.T[Q.T])
            del Ktq
            U_new = U_it + dU
            a, b, cc = dU.flatten(order='F'), U_it.flatten(order='F'), U_new.flatten(order='F')
            crit = np.sqrt(a @ K @ a) / (np.sqrt(b @ K @ b) + np.sqrt(cc @ K @ cc))
            if np.isnan(crit):
                break
            U_it = U_new
            if crit < 1e-12:
                break
        if crit < 1e-10:
            U_old = U; U = U_it
            E = (B @ U.flatten(order='F')).reshape((3, -1), order='F')
            cp = m.construct_constitutive_problem(E, E0, Ep_old, sh, bu, eta, c)   # accept stays False: C7
            calls += 1
            Ep_old = cp['ep']
            zeta_old = zeta; d_zeta_old = d_zeta
            hist.append(zeta); nplast.append(int(np.sum(cp['ind_p']))); Us.append(U.copy())
            if progress:
                progress(f'    step {len(hist)} zeta={zeta:.4f} n_plast={nplast[-1]} calls={calls}')
        else:
            d_zeta = d_zeta / 2
        U_it = d_zeta * (U - U_old) / d_zeta_old + U
        if zeta_old >= 1 or d_zeta < d_zeta_min:
            break
    return np.array(hist), np.array(nplast), np.array(Us), F0, calls


def gen_tsx(R):
    m = R['tsx']
    d = os.path.join(REF, 'tsx-tunnel')
    coord = np.genfromtxt(os.path.join(d, 'coord.csv'), delimiter=',')            # TSX:1687
    elem = np.genfromtxt(os.path.join(d, 'elem.csv'), delimiter=',', dtype=int) - 1   # TSX:1688
    kqq = np.genfromtxt(os.path.join(d, 'k_tangent_qq.csv'), delimiter=',')
    fq = np.genfromtxt(os.path.join(d, 'fq.csv'), delimiter=',')
    f0q = np.genfromtxt(os.path.join(d, 'f0q.csv'), delimiter=',')
    import scipy.sparse as ssp
    kqq_s = ssp.coo_matrix(kqq)
    arrs = dict(coord=coord, elem=elem.astype(np.int64), fq=fq, f0q=f0q,
                kqq_row=kqq_s.row.astype(np.int64), kqq_col=kqq_s.col.astype(np.int64), kqq_val=kqq_s.data,
                kqq_shape=np.array(kqq.shape))
    p2 = m.create_midpoints_P2(coord, elem)
    p4 = m.create_midpoints_P4(coord, elem)
    arrs.update(p2_coord=p2['coord_ext'], p2_elem=p2['elem_ext'].astype(np.int64),
                p4_coord=p4['coord_ext'], p4_elem=p4['elem_ext'].astype(np.int64))
    G, Kb, eta0, c0, s0, e_init = _tsx_consts()
    arrs['init_strain'] = e_init

    # P1: elastic tangent on the free DOFs (what k_tangent_qq.csv dumps) and the full replay
    K, B, w, iD, jD, D, Q, n_int = _tsx_setup(m, 'P1', coord, elem)
    qf = Q.flatten(order='F')
    Kqq = K.tocsr()[qf][:, qf].tocoo()
    arrs.update(p1_Kqq_row=Kqq.row.astype(np.int64), p1_Kqq_col=Kqq.col.astype(np.int64), p1_Kqq_val=Kqq.data)

    hist, nplast, Us, F0, _ = _tsx_replay(m, K, B, w, iD, jD, D, Q, n_int, coord)
    print('  tsx P1 replay: steps', len(hist), 'n_plast', nplast.tolist(), 'U[0,40]', Us[-1][0, 40])
    arrs.update(p1_zeta=hist, p1_nplast=nplast, p1_U_final=Us[-1], p1_U_step13=Us[12], p1_F0=F0)

    # P2: initial-stress load on the free DOFs (what f0q.csv dumps), TSX:1737
    K2, B2, w2, iD2, jD2, D2, Q2, n_int2 = _tsx_setup(m, 'P2', p2['coord_ext'], p2['elem_ext'])
    F0_2 = (B2.T @ np.reshape(np.tile(w2, (3, 1)) * s0[0:3, :], (3 * n_int2, 1), order='F')).reshape((2, -1), order='F')
    arrs['p2_F0'] = F0_2
    arrs['p2_Q'] = Q2
    K2c = K2.tocsr()
    arrs['p2_K_diag'] = K2c.diagonal()
    arrs['p2_K_frob'] = np.array(np.sqrt((K2c.data ** 2).sum()))
    # P4 elastic K scalars (30x30 element matrices, 12-pt rule with the C8 typo)
    K4, B4, w4, *_ = _tsx_setup(m, 'P4', p4['coord_ext'], p4['elem_ext'])
    K4c = K4.tocsr()
    arrs['p4_K_diag'] = K4c.diagonal()
    arrs['p4_K_frob'] = np.array(np.sqrt((K4c.data ** 2).sum()))
    arrs['p4_weight_sum'] = np.array(w4.sum())
    save('tsx', **arrs)


def gen_tsx_p2p4(R, types=('P2', 'P4')):
    """The TSX load-step sequence on the element types the reference's own demo runs (tsx-tunnel/sandbox.py:3-4:
    P4; the driver only works for P2 / P4, TSX:1629-1633): 17 steps, dense solves on the free DOFs as the
    reference does (P4: 14 288 free DOFs).  The meshes are the ones stored in tsx.npz (p2_*/p4_*)."""
    m = R['tsx']
    d = os.path.join(REF, 'tsx-tunnel')
    coord = np.genfromtxt(os.path.join(d, 'coord.csv'), delimiter=',')
    elem = np.genfromtxt(os.path.join(d, 'elem.csv'), delimiter=',', dtype=int) - 1
    for t in types:
        t0 = time.time()
        ext = m.create_midpoints_P2(coord, elem) if t == 'P2' else m.create_midpoints_P4(coord, elem)
        co, el = ext['coord_ext'], ext['elem_ext']
        K, B, w, iD, jD, D, Q, n_int = _tsx_setup(m, t, co, el)
        hist, nplast, Us, F0, calls = _tsx_replay(m, K, B, w, iD, jD, D, Q, n_int, co, progress=print)
        print(f'  tsx {t} replay: {time.time() - t0:.0f}s steps {len(hist)} calls {calls} n_plast {nplast.tolist()} '
              f'U[0,40] {Us[-1][0, 40]!r}')
        keep = sorted(set([0, 1, 2, len(Us) - 3, len(Us) - 2, len(Us) - 1]))
        save(f'tsx_{t.lower()}_trace', zeta=hist, nplast=nplast, n_calls=np.array(calls),
             U_steps=Us[keep], steps=np.array(keep), U_mon=Us[:, 0, 40], F0=F0)


def gen_el(R):
    """Config 1: Elasticity2D P1 elastic K assembly (EL:368-477) on the square
    with a corner cut-out (EL:935), levels 1 and 3."""
    m = R['el']
    arrs = {}
    for level in (1, 3):
        et = m.LagrangeElementType.P1
        mesh = m.assemble_mesh(level, et, 10, 5)
        xi, wf = m.get_quadrature_volume(et)
        _, d1, d2 = m.get_local_basis_volume(et, xi)
        elem1 = np.asarray(mesh['elements']).astype(np.int64).copy()      # 1-based (EL:389 shifts in place)
        coord = np.asarray(mesh['coordinates'])
        n_int = elem1.shape[1] * wf.size
        G = 206900 / (2 * (1 + 0.29)); Kb = 206900 / (3 * (1 - 2 * 0.29))
        K, w = m.get_elastic_stiffness_matrix(elem1.copy(), coord, G * np.ones(n_int), Kb * np.ones(n_int), d1, d2, wf)
        Kc = K.tocsr()
        tag = f'l{level}_'
        arrs.update({tag + 'elements_1based': elem1.astype(np.int32), tag + 'coordinates': coord,
                     tag + 'nnz': np.array(Kc.nnz), tag + 'trace': np.array(Kc.diagonal().sum()),
                     tag + 'frob': np.array(np.sqrt((Kc.data ** 2).sum())), tag + 'wsum': np.array(w.sum()),
                     tag + 'diag': Kc.diagonal(),
                     tag + 'Kx': Kc @ np.cos(np.arange(Kc.shape[0]) * 0.37)})
        print(f'  EL level {level}: nnz {Kc.nnz} trace {Kc.diagonal().sum():.12e}')
    save('el_p1', **arrs)


if __name__ == '__main__':
    R = _load_reference()
    which = sys.argv[1:] or ['tables', 'mesh_dp', 'setup', 'retmap', 'hotpath', 'dp_trace', 'tsx', 'el', 'transform',
                             'dp_trace_types', 'tsx_p2p4']
    for w in which:
        print('==', w)
        globals()['gen_' + w](R)
