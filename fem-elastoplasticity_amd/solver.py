"""
Linear solve of one Newton iterate on the GPU ("later a GPU Krylov solver"):
    K[Q][:, Q] dU[Q] = -F[Q]          np.linalg.solve on a dense masked block in the reference, DP:1062-1066 / TSX:1781
Preconditioned conjugate gradients (2x2 node-block Jacobi) in HIP, `fep_solver_*` of include/fep.h.  K is used
where the assembly kernels left it: the `data` array on the context's CSR pattern, in device memory.  torch
tensors are only the holders of device memory here.

Two preconditioners: the 2x2 node-block Jacobi that is always there, and a smoothed-aggregation multigrid
(`setup_amg`): aggregates of the node graph, tentative prolongators from the rigid-body modes (two translations and
the rotation, 3 DOFs per aggregate), one damped-Jacobi smoothing step of the prolongator, Galerkin coarse operators.
The hierarchy is built once per mesh on the host with SciPy from a reference matrix (K_elast) and lives on the GPU;
every solve smooths with the CURRENT tangent on the mesh level and — `refresh`, the default — re-projects the coarse
operators from that tangent with the transfers of the reference matrix (numeric Galerkin products on the GPU; a third
fewer iterations on plastic tangents than with the reference matrix's coarse operators).
"""
import ctypes as C
import os
import sys
import time

import numpy as np
import scipy.sparse as ssp

from . import _lib


def _block_diag(A, bs):
    """(n, bs, bs) array of the diagonal blocks of the CSR matrix A."""
    n = A.shape[0] // bs
    D = np.zeros((n, bs, bs))
    for b in range(bs):
        for c in range(bs):
            k = c - b
            d = A.diagonal(k)
            D[:, b, c] = d[b::bs][:n] if k >= 0 else d[c::bs][:n]
    return D


def _block_diag_inverse(A, bs):
    """CSR matrix of the inverted diagonal blocks (blocks of one-node aggregates have an empty rotation row)."""
    D = _block_diag(A, bs)
    for b in range(bs):                                  # empty rows/columns (a rotation nobody interpolates from)
        z = D[:, b, b] == 0.0
        D[z, b, b] = 1.0
    tr = np.abs(np.einsum('nii->n', D)) / bs
    D = D + (1e-13 * tr)[:, None, None] * np.eye(bs)[None]
    Di = np.linalg.inv(D)
    n = D.shape[0]
    cols = (bs * np.arange(n)[:, None, None] + np.arange(bs)[None, None, :]) + np.zeros((1, bs, 1), dtype=np.int64)
    indptr = bs * np.arange(bs * n + 1, dtype=np.int64)
    return ssp.csr_matrix((Di.ravel(), cols.ravel(), indptr), shape=A.shape)


def _rho(A, Di, iters=15):
    """Largest eigenvalue of Di A by power iteration (Di A is similar to a symmetric positive semidefinite matrix)."""
    x = np.random.default_rng(1).normal(size=A.shape[0])
    lam = 1.0
    for _ in range(iters):
        y = Di @ (A @ x)
        ny = np.linalg.norm(y)
        if ny == 0.0:
            return 1.0
        lam = ny / np.linalg.norm(x)
        x = y / ny
    return float(lam)


def _aggregate(A, bs):
    # node graph = the bs consecutive CSR rows of a node read as ONE neighbour list (node ids repeat in it, unsorted across
    # the rows: fep_aggregate_host does not mind) — no COO round trip (1.5 of the hierarchy's 7 s at 1 M DOFs)
    n = A.shape[0] // bs
    ip = np.ascontiguousarray(A.indptr[::bs], dtype=np.int32)
    ix = np.ascontiguousarray(A.indices // bs, dtype=np.int32)
    agg = np.empty(n, dtype=np.int32)
    na = C.c_int64()
    _lib.check(_lib.lib().fep_aggregate_host(n, _lib.ptr(ip), _lib.ptr(ix), _lib.ptr(agg), C.byref(na)),
               'fep_aggregate_host')
    return agg.astype(np.int64), int(na.value)


def _tentative(agg, na, xy, bs):
    """Rigid-body modes of every aggregate about its centre; fine DOFs (u, v) for bs = 2, (u, v, theta) for bs = 3."""
    n = agg.size
    cnt = np.bincount(agg, minlength=na)
    cx = np.bincount(agg, weights=xy[0], minlength=na) / cnt
    cy = np.bincount(agg, weights=xy[1], minlength=na) / cnt
    dx, dy = xy[0] - cx[agg], xy[1] - cy[agg]
    i = np.arange(n)
    one = np.ones(n)
    rows = [bs * i, bs * i, bs * i + 1, bs * i + 1]
    cols = [3 * agg, 3 * agg + 2, 3 * agg + 1, 3 * agg + 2]
    vals = [one, -dy, one, dx]
    if bs == 3:
        rows.append(bs * i + 2); cols.append(3 * agg + 2); vals.append(one)
    P = ssp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(bs * n, 3 * na))
    return P, np.stack([cx, cy])


def _p(t):
    return C.c_void_p(t.data_ptr())


def _stream(torch, device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _spgemm(X, Y):
    """X @ Y for CSR factors through fep_spgemm_*_host (rows in parallel on the host's cores; SciPy's product is one thread),
    numerically-zero entries dropped as SciPy's product drops them; column ids ascending."""
    X, Y = ssp.csr_matrix(X), ssp.csr_matrix(Y)
    if X.shape[1] != Y.shape[0]:
        raise ValueError('shapes do not match')
    n, m, k = X.shape[0], X.shape[1], Y.shape[1]
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    xp, xi, xv, yp, yi, yv = i32(X.indptr), i32(X.indices), f64(X.data), i32(Y.indptr), i32(Y.indices), f64(Y.data)
    cp = np.empty(n + 1, dtype=np.int32)
    l = _lib.lib()
    _lib.check(l.fep_spgemm_count_host(n, m, k, _lib.ptr(xp), _lib.ptr(xi), _lib.ptr(yp), _lib.ptr(yi), _lib.ptr(cp)),
               'fep_spgemm_count_host')
    ci = np.empty(int(cp[n]), dtype=np.int32)
    cv = np.empty(int(cp[n]), dtype=np.float64)
    _lib.check(l.fep_spgemm_fill_host(n, m, k, _lib.ptr(xp), _lib.ptr(xi), _lib.ptr(xv), _lib.ptr(yp), _lib.ptr(yi), _lib.ptr(yv),
                                      _lib.ptr(cp), _lib.ptr(ci), _lib.ptr(cv)), 'fep_spgemm_fill_host')
    Cm = ssp.csr_matrix((cv, ci, cp), shape=(n, k))
    Cm.has_sorted_indices = True
    Cm.eliminate_zeros()
    return Cm


def _masked_operator(K, f):
    """Q K Q + (I - Q) without explicit zeros — what `(Dq @ K @ Dq + diags(1 - f)).tocsr()` returns, entry for entry, without
    the two sparse products and their temporaries (2.1 of the hierarchy's 7 s at 1 M DOFs): rows and columns of the few
    constrained DOFs zeroed in a copy of the values, their diagonal set to one."""
    K = ssp.csr_matrix(K)
    if not K.has_canonical_format:
        K = K.copy()
        K.sum_duplicates()
    ip, ix = K.indptr, K.indices
    data = K.data.copy()
    fixed_mask = np.asarray(f) == 0.0
    data[fixed_mask[ix]] = 0.0                           # columns
    missing = []
    for i in np.flatnonzero(fixed_mask):                 # rows, and the diagonal
        lo, hi = ip[i], ip[i + 1]
        data[lo:hi] = 0.0
        t = lo + np.searchsorted(ix[lo:hi], i)
        if t < hi and ix[t] == i:
            data[t] = 1.0
        else:
            missing.append(i)
    A = ssp.csr_matrix((data, ix.copy(), ip.copy()), shape=K.shape)
    A.eliminate_zeros()                                  # (compacts its own index arrays in place)
    if missing:                                          # a pattern without that diagonal entry
        m = np.asarray(missing)
        A = (A + ssp.csr_matrix((np.ones(m.size), (m, m)), shape=A.shape)).tocsr()
    return A


def build_amg_hierarchy(K, free_dof, coordinates, coarse_nodes=400, max_levels=8, rho0=None):
    """Smoothed-aggregation hierarchy of Q K Q + (I - Q) (host, SciPy).  Returns one dict per transfer k -> k+1:
    'P' (n_k x n_{k+1}), 'R' = P^T, 'A' = operator of level k+1 (its dense INVERSE in CSR form when 'last'),
    'D' = inverse of A's 3x3 block diagonal (None when 'last'), 'omega' = Jacobi damping on level k, 'size' =
    (DOFs, nnz) of the level-(k+1) operator — exactly what fep_solver_amg_push_level takes."""
    f = np.asarray(free_dof, dtype=bool).ravel().astype(np.float64)
    Dq = ssp.diags(f)
    A = _masked_operator(K, f)
    xy = np.asarray(coordinates, dtype=np.float64)
    bs, out = 2, []
    Di = _block_diag_inverse(A, bs)
    rho = rho0(A, Di) if callable(rho0) else _rho(A, Di)    # (setup_amg runs the mesh level's power iteration on the device)
    for level in range(max_levels):
        agg, na = _aggregate(A, bs)
        Pt, cxy = _tentative(agg, na, xy, bs)
        if bs == 2:
            Pt = (Dq @ Pt).tocsr()                       # constrained DOFs neither interpolate nor receive
        P = (Pt - (4.0 / (3.0 * rho)) * (Di @ _spgemm(A, Pt))).tocsr()
        R = P.T.tocsr()
        Ac = _spgemm(R, _spgemm(A, P))                   # Galerkin operator (threaded host products; SciPy: (P.T @ A @ P))
        last = na <= coarse_nodes or level == max_levels - 1 or 3 * na > 0.7 * A.shape[0]
        if last:
            dense = Ac.toarray()
            dense += 1e-10 * np.abs(dense).max() * np.eye(dense.shape[0])
            Aop, Dc = ssp.csr_matrix(np.linalg.inv(dense)), None
        else:
            Aop, Dc = Ac, _block_diag_inverse(Ac, 3)
        for M in (P, R, Aop, Dc):
            if M is not None:
                M.sort_indices()
        out.append({'P': P, 'R': R, 'A': Aop, 'D': Dc, 'omega': 4.0 / (3.0 * 1.05 * rho), 'last': last,
                    'size': (Ac.shape[0], Ac.nnz), 'Pt': Pt, 'rho': rho})      # (Pt, rho: offline studies, tools/transfer_study.py)
        if last:
            break
        A, xy, bs, Di = Ac, cxy, 3, Dc
        rho = _rho(A, Di)
    return out


class KrylovSolver:
    """PCG on the free DOFs `free_dof` (bool, DOF order = Q.flatten(order='F')) of the pattern of `ctx`
    (a MeshContext) or of an explicit `(indptr, indices)` pair with `device`."""

    def __init__(self, ctx, free_dof, device=None):
        import torch
        if hasattr(ctx, 'pattern'):
            ip, ix = ctx.pattern()
            device = ctx.device if device is None else device
        else:
            ip, ix = ctx
        ip = np.ascontiguousarray(ip, dtype=np.int32)
        ix = np.ascontiguousarray(ix, dtype=np.int32)
        fd = np.ascontiguousarray(np.asarray(free_dof).ravel() != 0, dtype=np.uint8)
        if fd.size != ip.size - 1 or fd.size % 2:
            raise ValueError('free_dof must hold one flag per DOF (2 per node)')
        self.device = 0 if device is None else device
        self._torch = torch
        self._dev = torch.device('cuda', self.device)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().fep_solver_create(C.byref(self._h), self.device, fd.size // 2, _lib.ptr(ip), _lib.ptr(ix),
                                                _lib.ptr(fd)), 'fep_solver_create')
        sz = (C.c_int64 * 4)()
        _lib.check(_lib.lib().fep_solver_sizes(self._h, sz), 'fep_solver_sizes')
        self.n_n, self.n_dof, self.nnz, self.n_free = [int(v) for v in sz]
        self.free_dof = fd.view(np.bool_)
        self._pattern = (ip, ix)
        self.amg_levels = None
        self.last = None

    def close(self):
        if getattr(self, '_h', None) is not None and self._h:
            _lib.lib().fep_solver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _vec(self, v, n):
        torch = self._torch
        if not isinstance(v, torch.Tensor):
            v = torch.from_numpy(np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel()))
        v = v.to(self._dev)
        if v.dtype != torch.float64 or v.numel() != n or not v.is_contiguous():
            raise ValueError(f'expected {n} contiguous float64 values')
        return v

    def _rho_dev(self, k_dev):
        """`_rho` for the mesh level with the operator passes on the device: A x = Q K Q x + (I - Q) x through spmv, the block
        diagonal's inverse as a batched 2x2 product; same start vector, same 15 steps (0.3 of the hierarchy's 1.6 s at 1 M DOFs)."""
        torch = self._torch
        kd = self._vec(k_dev, self.nnz)
        f = torch.from_numpy(self.free_dof.astype(np.float64)).to(self._dev)

        def rho(A, Di, iters=15):
            D = torch.from_numpy(np.ascontiguousarray(Di.data.reshape(-1, 2, 2))).to(self._dev)
            x = torch.from_numpy(np.random.default_rng(1).normal(size=A.shape[0])).to(self._dev)
            y = torch.empty_like(x)
            lam = 1.0
            for _ in range(iters):
                xq = x * f
                self.spmv(kd, xq, out=y, masked=True)
                y += x - xq
                y = torch.bmm(D, y.view(-1, 2, 1)).reshape(-1)
                ny = float(torch.linalg.vector_norm(y))
                if ny == 0.0:
                    return 1.0
                lam = ny / float(torch.linalg.vector_norm(x))
                x = y / ny
                y = torch.empty_like(x)
            return float(lam)
        return rho

    def setup_amg(self, K_ref, coordinates, coarse_nodes=400, max_levels=8, refresh=None, k_dev=None):
        """Builds the multigrid hierarchy from `K_ref` (csr_matrix on the pattern, or its data array; host) and the
        node coordinates (2, n_n), and loads it onto the device.  Returns [(DOFs, nnz)] per level.
        `refresh` (default: on unless FEP_AMG_REFRESH=0): every solve re-projects the coarse operators from its own
        tangent with the transfers built here (fep_solver_amg_enable_refresh); `self.amg_refresh` tells whether it is on.
        `k_dev`: the same values as a device tensor, when the caller has them there (the mesh level's eigenvalue estimate then
        runs on the device)."""
        if not hasattr(self, '_pattern'):
            raise ValueError('setup_amg needs the solver to have been created from a pattern')
        ip, ix = self._pattern
        data = K_ref.data if hasattr(K_ref, 'indptr') else np.asarray(K_ref, dtype=np.float64)
        K = ssp.csr_matrix((data, ix, ip), shape=(self.n_dof, self.n_dof))
        if refresh is None:
            refresh = os.environ.get('FEP_AMG_REFRESH', '1') != '0'
        if refresh:                                      # the coarsest operator is re-inverted by every solve: keep it small
            coarse_nodes = min(coarse_nodes, 64)
        t0 = time.perf_counter()
        levels = build_amg_hierarchy(K, self.free_dof, coordinates, coarse_nodes, max_levels,
                                     rho0=None if k_dev is None else self._rho_dev(k_dev))
        t1 = time.perf_counter()
        l = _lib.lib()

        def push_levels():
            _lib.check(l.fep_solver_amg_clear(self._h), 'fep_solver_amg_clear')
            for lv in levels:
                mats = []
                for M in (lv['P'], lv['R'], lv['A'], lv['D']):
                    if M is None:
                        mats += [None, None, None]
                    else:
                        mats += [np.ascontiguousarray(M.indptr, dtype=np.int32), np.ascontiguousarray(M.indices, dtype=np.int32),
                                 np.ascontiguousarray(M.data, dtype=np.float64)]
                _lib.check(l.fep_solver_amg_push_level(self._h, lv['P'].shape[0], lv['P'].shape[1], *[_lib.ptr(m) for m in mats],
                                                       float(lv['omega']), int(lv['last'])), 'fep_solver_amg_push_level')
        push_levels()
        self.amg_levels = [(K.shape[0], K.nnz)] + [lv['size'] for lv in levels]
        self.amg_refresh = False
        t2 = time.perf_counter()
        if refresh:
            rc = l.fep_solver_amg_enable_refresh(self._h)
            if rc == -5:
                # FEP_ERANGE: too large a coarsest level (hierarchy untouched) or a product plan beyond 32-bit counts, found
                # while the plans were built (hierarchy dropped, include/fep.h): the levels are pushed again and the solves
                # keep the coarse operators of K_ref
                push_levels()
            else:
                _lib.check(rc, 'fep_solver_amg_enable_refresh')
                self.amg_refresh = True
        self.amg_seconds = {'hierarchy': t1 - t0, 'upload': t2 - t1, 'refresh_plans': time.perf_counter() - t2}
        if os.environ.get('FEP_VERBOSE'):
            print('[fep] multigrid set-up: ' + ', '.join(f'{k} {v:.2f} s' for k, v in self.amg_seconds.items()), file=sys.stderr)
        return self.amg_levels

    def spmv(self, k_data, x, out=None, masked=False):
        """y = K x (masked: rows of constrained DOFs zeroed; x must be 0 there).  Device tensors in and out."""
        torch = self._torch
        k = self._vec(k_data, self.nnz)
        x = self._vec(x, self.n_dof)
        y = torch.empty(self.n_dof, dtype=torch.float64, device=self._dev) if out is None else out
        _lib.check(_lib.lib().fep_solver_spmv_dev(self._h, _stream(torch, self._dev), _p(k), _p(x), _p(y), int(masked)),
                   'fep_solver_spmv_dev')
        return y

    def pcg(self, k_data, b, out=None, rtol=1e-12, max_iter=100000, check_every=0, precond=None):
        """Solves K[Q][:,Q] x[Q] = b[Q]; returns the full-length device tensor x (0 on constrained DOFs).
        `precond`: 'jacobi', 'amg' (needs `setup_amg`), default = 'amg' when a hierarchy is loaded.
        `self.last` = {'iters', 'relres', 'state'} with state 1 = converged, 0 = max_iter, 2 = breakdown."""
        if precond is None:
            precond = 'amg' if self.amg_levels else 'jacobi'
        if precond not in ('jacobi', 'amg'):
            raise ValueError("precond must be 'jacobi' or 'amg'")
        fn = _lib.lib().fep_solver_amg_pcg_dev if precond == 'amg' else _lib.lib().fep_solver_pcg_dev
        torch = self._torch
        k = self._vec(k_data, self.nnz)
        b = self._vec(b, self.n_dof)
        x = torch.empty(self.n_dof, dtype=torch.float64, device=self._dev) if out is None else out
        it, st, rr = C.c_int(), C.c_int(), C.c_double()
        _lib.check(fn(self._h, _stream(torch, self._dev), _p(k), _p(b), _p(x), float(rtol), int(max_iter),
                      int(check_every), C.byref(it), C.byref(rr), C.byref(st)), 'fep_solver_%spcg_dev' % ('amg_' if precond == 'amg' else ''))
        self.last = {'iters': it.value, 'relres': rr.value, 'state': st.value, 'precond': precond}
        return x

    def solve_host(self, K, b, **kw):
        """Convenience for host data: `K` a csr_matrix on the pattern (or its data array), `b` (n_dof,) -> ndarray."""
        data = K.data if hasattr(K, 'data') and hasattr(K, 'indptr') else K
        return self.pcg(np.asarray(data, dtype=np.float64), b, **kw).cpu().numpy()
