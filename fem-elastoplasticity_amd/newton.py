"""
Callers of the hot path: the reference's load-stepping / semismooth-Newton drivers
re-stated around the GPU step.

  solve_strip_footing   Plasticity2D_DP/pythonFEM.py:986-1131  (adaptive load steps, footing pressure)
  solve_tsx_tunnel      tsx-tunnel/pythonFEM.py:1729-1832      (17 uniform steps of the initial-stress factor;
                        the accepting call leaves `apply_plastic_strain` False)

Per Newton iterate ONE call of the fused step replaces DP:1043-1058 (strain, return map, tangent, residual).
Step control, stopping norms and the extrapolation of the next iterate follow the reference line by line.
The linear solve (the reference's dense `np.linalg.solve` on a (2 n_n)^2 boolean-masked matrix) is
  linear_solver='direct'  SciPy SuperLU on the host CSR matrix (K travels to the host every iterate), or
  linear_solver='pcg'     conjugate gradients on the GPU (solver.py, block-Jacobi preconditioner); the iterate, K, F,
                          the plastic strain and the stopping norms then never leave the device, or
  linear_solver='amg'     the same with the smoothed-aggregation multigrid preconditioner built once from K_elast.
`pcg_forcing` (e.g. 1e-2) makes the Newton iteration inexact: the linear tolerance follows the previous iterate's
stopping quantity instead of being 1e-11 throughout (never looser than `pcg_forcing_cap`); `pcg_inexact_rtol` (e.g. 1e-2)
asks every linear solve for that relative residual only — Newton then converges linearly with about that factor instead
of quadratically, which on plastic tangents (hundreds of CG iterations per digit) is much the cheaper trade: BASELINE
configs[3] end to end 42 s -> 15 s with the same load history (tools/newton_bench.py --inexact 1e-2).  The converged
states are the same to the Newton tolerance either way.
`transform` (DP:760-816, nodal averaging used for the footing pressure that steers the step size) is re-stated
with `np.bincount` on the host and as `fep_transform_dev` on the device.
"""
import time

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


class _HostOps:
    """Vectors as ndarrays, K as csr_matrix, sparse direct solve."""

    def __init__(self, ctx, qf):
        self.ctx, self.qf = ctx, qf

    def vec(self, a):
        return np.array(a, dtype=np.float64).ravel()

    def zeros(self):
        return np.zeros(self.ctx.n_dof if hasattr(self.ctx, 'n_dof') else self.qf.size)

    def new_ep(self):
        return np.zeros((4, self.ctx.n_int))

    def step(self, U, Ep=None, accept=False, e0=None, want=('K', 'F'), keep_K=False):
        kw = {} if e0 is None else {'e0': e0}
        return self.ctx.step(U, Ep, apply_plastic_strain=accept, want=want, **kw)

    def solve(self, K, rhs, criterion=None):
        rhs = np.asarray(rhs).ravel()
        x = np.zeros(rhs.size)
        x[self.qf] = sspl.spsolve(K[self.qf][:, self.qf].tocsc(), rhs[self.qf])
        return x

    def matvec(self, K, v):
        return K @ v

    def energy(self, K, v):
        return float(np.sqrt(v @ (K @ v)))

    def host(self, v):
        return np.asarray(v)

    def nodal(self, q_int, elem, weight):
        return transform(self.host(q_int), elem, weight)

    def close(self):
        pass


class _DeviceOps:
    """Vectors, K data, plastic strain and stresses as device tensors; `MeshContext.step_dev` writes them and
    `KrylovSolver.pcg` solves on them.  A linear solve that breaks down or runs out of iterations yields NaNs,
    which the drivers treat like the reference treats a NaN criterion (DP:1076): the load step is halved."""

    def __init__(self, ctx, qf, rtol=1e-11, max_iter=200000, forcing=None, forcing_cap=1e-4, inexact_rtol=None):
        import torch
        from .solver import KrylovSolver
        self.torch, self.ctx, self.qf = torch, ctx, qf
        self.dev = torch.device('cuda', ctx.device)
        self.solver = KrylovSolver(ctx, qf)
        self.rtol, self.max_iter, self.forcing, self.forcing_cap = rtol, max_iter, forcing, forcing_cap
        self.inexact_rtol = inexact_rtol
        f64 = dict(dtype=torch.float64, device=self.dev)
        self.kd = torch.empty(ctx.nnz, **f64)
        self.F = torch.empty(ctx.n_dof, **f64)
        self.s = torch.empty((4, ctx.n_int), **f64)
        self.ind = torch.empty(ctx.n_int, dtype=torch.uint8, device=self.dev)
        self.counts = torch.zeros(2, dtype=torch.int64, device=self.dev)
        self.tmp = torch.empty(ctx.n_dof, **f64)
        self.pcg_iters = []

    def vec(self, a):
        return self.torch.from_numpy(np.array(a, dtype=np.float64).ravel()).to(self.dev)

    def zeros(self):
        return self.torch.zeros(self.ctx.n_dof, dtype=self.torch.float64, device=self.dev)

    def new_ep(self):
        return self.torch.zeros((4, self.ctx.n_int), dtype=self.torch.float64, device=self.dev)

    def step(self, U, Ep=None, accept=False, e0=None, want=('K', 'F'), keep_K=False):
        t = self.torch
        st = t.cuda.current_stream(self.dev).cuda_stream
        kd = None
        if 'K' in want:
            kd = t.empty(self.ctx.nnz, dtype=t.float64, device=self.dev) if keep_K else self.kd
        self.ctx.step_dev(st, U.data_ptr(), ep=0 if Ep is None else Ep.data_ptr(), accept=accept and Ep is not None,
                          e0=e0, s=self.s.data_ptr() if 's' in want else 0,
                          ind_p=self.ind.data_ptr() if 'ind_p' in want else 0,
                          k_data=0 if kd is None else kd.data_ptr(), f_out=self.F.data_ptr() if 'F' in want else 0,
                          counts=self.counts.data_ptr() if ('s' in want or 'ind_p' in want) else 0)   # logged on accepting calls only
        out = {'K': kd, 'F': self.F, 's': self.s, 'ind_p': self.ind}
        out = {k: v for k, v in out.items() if k in want}
        if 's' in want or 'ind_p' in want:                     # accepting calls: the counters are logged
            c = self.counts.cpu()
            out['n_smooth'], out['n_apex'] = int(c[0]), int(c[1])
        return out

    def setup_amg(self, K, coordinates):
        # built once from K_elast; rebuilding it from the current tangent when the plastic zone grows was measured
        # (1 M elements, 10 load steps: 6 rebuilds) and did not lower the iteration counts
        self.solver.setup_amg(self.host(K), coordinates, k_dev=K)

    def solve(self, K, rhs, criterion=None):
        # inexact Newton: while the iterate is far from converged the correction need not be solved to 11 digits.
        # `criterion` is the stopping quantity of the previous Newton iterate (DP:1075); the linear residual is asked
        # to be `forcing` times smaller than it, never looser than `forcing_cap` (1e-4) nor tighter than `rtol`.
        rtol = self.rtol
        if self.inexact_rtol and criterion is not None:        # (the elastic solves before the loop pass no criterion)
            rtol = max(self.rtol, self.inexact_rtol)
        elif self.forcing and criterion is not None and np.isfinite(criterion):
            rtol = min(self.forcing_cap, max(self.rtol, self.forcing * criterion))
        x = self.solver.pcg(K, rhs, rtol=rtol, max_iter=self.max_iter)
        self.pcg_iters.append(self.solver.last['iters'])
        if self.solver.last['state'] != 1:
            x.fill_(float('nan'))
        return x

    def matvec(self, K, v):
        return self.solver.spmv(K, v)

    def energy(self, K, v):
        self.solver.spmv(K, v, out=self.tmp)
        return float(self.torch.sqrt(self.torch.dot(v, self.tmp)))

    def host(self, v):
        return v.cpu().numpy()

    def nodal(self, q_int, elem, weight):
        t = self.torch
        q = q_int.contiguous()
        out = t.empty(self.ctx.n_n, dtype=t.float64, device=self.dev)
        self.ctx.transform_dev(t.cuda.current_stream(self.dev).cuda_stream, q.data_ptr(), out.data_ptr())
        return out.cpu().numpy()

    def close(self):
        self.solver.close()


def _make_ops(ctx, qf, linear_solver, pcg_rtol, pcg_forcing=None, pcg_forcing_cap=1e-4, pcg_inexact_rtol=None):
    if linear_solver == 'direct':
        return _HostOps(ctx, qf)
    if linear_solver in ('pcg', 'amg'):
        if not isinstance(ctx, MeshContext):
            raise ValueError(f"linear_solver='{linear_solver}' needs the GPU MeshContext")
        return _DeviceOps(ctx, qf, rtol=pcg_rtol, forcing=pcg_forcing, forcing_cap=pcg_forcing_cap, inexact_rtol=pcg_inexact_rtol)
    raise ValueError("linear_solver must be 'direct', 'pcg' or 'amg'")


def solve_strip_footing(element_type='P1', level=1, n_cells=None, size_xy=10, max_steps=None, zeta_max=1.0,
                        device=None, log=None, context_factory=None, linear_solver='direct', pcg_rtol=1e-11,
                        keep_U=True, pcg_forcing=None, pcg_forcing_cap=1e-4, pcg_inexact_rtol=None, _ops_factory=None):
    """Strip-footing benchmark of Plasticity2D_DP (DP:901-1131).  `level` as in the reference
    (N = size_xy * 2**level cells per side) or `n_cells` directly.  Returns a dict with the load history
    ('zeta', 'pressure'), the accepted displacements 'U' (list of (2,n_n)), final 'Ep', counters.
    `context_factory(elements, coordinates, dhatp1, dhatp2, wf)` may supply another object with MeshContext's
    `set_materials / step / geometry / close` (the tests drive the same loop with their CPU checker that way);
    `_ops_factory` supplies the vector / solve operations (dist_newton.py runs this loop on an element-sharded mesh)."""
    t = _coerce(element_type)
    young, poisson, c0, phi = 1e7, 0.48, 450, np.pi / 9                                   # DP:910-933
    shear0 = young / (2 * (1 + poisson))
    bulk0 = young / (3 * (1 - 2 * poisson))
    eta0 = 3 * np.tan(phi) / np.sqrt(9 + 12 * (np.tan(phi)) ** 2)
    c_0 = 3 * c0 / np.sqrt(9 + 12 * (np.tan(phi)) ** 2)
    t_setup = [time.perf_counter()]
    mesh = square_mesh(size_xy * 2 ** level if n_cells is None else n_cells, t, size_xy)  # DP:945
    t_setup.append(time.perf_counter())
    elem, coord, Q = mesh['elements'], mesh['coordinates'], mesh['Q']
    q_nd = mesh['dirichlet_nodes'][1, :] > 0
    n_n = coord.shape[1]
    d1, d2, wf = element_tables(t)
    ctx = (context_factory or (lambda *a: MeshContext(*a, device=device)))(elem, coord, d1, d2, wf)
    ctx.set_materials(shear0, bulk0, eta0, c_0)
    qf = Q.flatten(order='F')
    t_setup.append(time.perf_counter())
    ops = (_ops_factory or _make_ops)(ctx, qf, linear_solver, pcg_rtol, pcg_forcing, pcg_forcing_cap, pcg_inexact_rtol)
    K_elast = ops.step(ops.zeros(), want=('K',), keep_K=True)['K']                        # DP:977
    t_setup.append(time.perf_counter())
    if linear_solver == 'amg':
        ops.setup_amg(K_elast, coord)
    _, _, weight, _ = ctx.geometry()
    t_setup.append(time.perf_counter())
    if log:
        log('setup: mesh %.2f s, context %.2f s, solver + K_elast %.2f s, multigrid hierarchy %.2f s'
            % tuple(b - a for a, b in zip(t_setup[:-1], t_setup[1:])))

    d_zeta = 1 / 1000                                                                     # DP:989-994
    d_zeta_min = d_zeta / 1300
    d_zeta_old = d_zeta
    zeta_old = 0.0
    Ud = ops.vec((-d_zeta * mesh['dirichlet_nodes']).flatten(order='F'))                  # DP:997-1004
    U_it = Ud + ops.solve(K_elast, -ops.matvec(K_elast, Ud))
    U = ops.zeros()
    U_old = -U_it
    Ep_old = ops.new_ep()
    pressure_old = 0.0
    hist = {'zeta': [], 'pressure': [], 'U': [], 'counts': [], 'n_calls': 0, 'newton_its': []}
    criterion = None
    while True:
        zeta = zeta_old + d_zeta                                                          # DP:1031
        its = 0
        for _ in range(25):                                                               # DP:1040
            r = ops.step(U_it, Ep_old, accept=False, want=('K', 'F'))                     # DP:1043-1058
            hist['n_calls'] += 1
            its += 1
            dU = ops.solve(r['K'], -r['F'], criterion if its > 1 else 1.0)                # DP:1062-1066
            U_new = U_it + dU
            q1, q2, q3 = ops.energy(K_elast, dU), ops.energy(K_elast, U_it), ops.energy(K_elast, U_new)   # DP:1072-1074
            criterion = q1 / (q2 + q3)
            if np.isnan(criterion):                                                       # DP:1076
                break
            U_it = U_new
            if criterion < 1e-12:                                                         # DP:1086
                break
        if criterion < 1e-10:                                                             # DP:1091
            U_old = U
            U = U_it
            r = ops.step(U, Ep_old, accept=True, want=('s',))                             # DP:1095-1098
            hist['n_calls'] += 1
            zeta_old = zeta
            d_zeta_old = d_zeta
            pressure_arr = ops.nodal(r['s'][1, :], elem, weight)                          # DP:1105
            pressure = -np.mean(pressure_arr[q_nd]) / c0
            hist['zeta'].append(zeta)
            hist['pressure'].append(pressure)
            if keep_U:
                hist['U'].append(ops.host(U).reshape((2, -1), order='F').copy())
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
    hist['Ep'] = ops.host(Ep_old)
    hist['U_last'] = ops.host(U).reshape((2, -1), order='F').copy()
    hist['mesh'] = mesh
    hist['pcg_iters'] = getattr(ops, 'pcg_iters', None)
    ops.close()
    ctx.close()
    return hist


def solve_tsx_tunnel(coords=None, elem=None, element_type='P1', n_load_steps=17, monitor=(0, 40), device=None, log=None,
                     linear_solver='direct', pcg_rtol=1e-11, pcg_forcing=None, mesh_dir=None, pcg_inexact_rtol=None):
    """TSX tunnel excavation (TSX:1637-1832) on a given mesh (`coords` (2,n_n), `elem` (n_p,n_e) 0-based), or — as the
    reference does at TSX:1687-1690 — on the mesh read from `mesh_dir`/coord.csv, elem.csv with the midpoints of
    `element_type` added.  Returns the history of the monitored displacement, plastic-point counts and accepted
    displacements."""
    t = _coerce(element_type)
    if mesh_dir is not None:
        from .meshio import load_tsx_mesh
        coords, elem = load_tsx_mesh(mesh_dir, t)
    if coords is None or elem is None:
        raise ValueError('pass the mesh (coords, elem) or mesh_dir')
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
    d1, d2, wf = element_tables(t)
    ctx = MeshContext(elem, coords, d1, d2, wf, device=device)
    n_int = ctx.n_int
    assert n_int == elem.shape[1] * ELEMENT_SHAPE[t][1]
    ctx.set_materials(shear0, bulk0, eta0, c_0)
    ops = _make_ops(ctx, qf, linear_solver, pcg_rtol, pcg_forcing, pcg_inexact_rtol=pcg_inexact_rtol)
    K = ops.step(ops.zeros(), want=('K',), keep_K=True)['K']                              # TSX:1722
    if linear_solver == 'amg':
        ops.setup_amg(K, coords)
    _, F0 = ctx.assemble(None, s0 * np.ones((1, n_int)))                                   # TSX:1737

    d_zeta = 1 / n_load_steps                                                             # TSX:1730-1735
    d_zeta_min = d_zeta / 10
    d_zeta_old = d_zeta
    zeta_old = 0.0
    U_elast = ops.solve(K, ops.vec(-F0))                                                  # TSX:1748
    U_it = d_zeta * U_elast
    U = ops.zeros()
    U_old = -U_it
    Ep_old = ops.new_ep()
    hist = {'zeta': [], 'displ': [], 'n_plast': [], 'U': [], 'n_calls': 0}
    criterion = None
    while True:
        zeta = zeta_old + d_zeta
        e0 = zeta * init_strain                                                           # TSX:1765
        for it in range(25):
            r = ops.step(U_it, Ep_old, e0=e0, want=('K', 'F'))                            # TSX:1771-1778
            hist['n_calls'] += 1
            dU = ops.solve(r['K'], -r['F'], criterion if it > 0 else 1.0)                 # TSX:1781
            U_new = U_it + dU
            criterion = ops.energy(K, dU) / (ops.energy(K, U_it) + ops.energy(K, U_new))  # TSX:1788-1792
            if np.isnan(criterion):
                break
            U_it = U_new
            if criterion < 1e-12:
                break
        if criterion < 1e-10:                                                             # TSX:1804
            U_old = U
            U = U_it
            r = ops.step(U, Ep_old, e0=e0, want=('ind_p',))          # accept WITHOUT apply_plastic_strain (C7)
            hist['n_calls'] += 1
            Ep_old = ops.new_ep()                                    # 'ep' of a non-accepting call, TSX:1809
            zeta_old = zeta
            d_zeta_old = d_zeta
            Um = ops.host(U).reshape((2, -1), order='F')
            hist['zeta'].append(zeta)
            hist['displ'].append(Um[monitor])
            hist['n_plast'].append(int(ops.host(r['ind_p']).astype(bool).sum()))
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
    hist['pcg_iters'] = getattr(ops, 'pcg_iters', None)
    ops.close()
    ctx.close()
    return hist
